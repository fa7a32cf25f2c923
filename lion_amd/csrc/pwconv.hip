// pwconv.hip -- G1: the 1x1 convolutions of SharedMLP (Conv1d / Conv2d with kernel 1,
// models/pvcnn2_ada.py:120-164) as an fp32-MFMA GEMM with the neighbouring AdaGN + Swish folded in.
//
//   y[b, co, l] = bias[co] + sum_ci W[co, ci] * act(x[b, ci, l]),   act(v) = swish(v * A[b,ci] + Bs[b,ci])  (PRO)
//
// The reference runs conv -> GroupNorm -> scale/shift -> sigmoid -> mul, five passes over an
// activation of up to 268 MB ([32,64,1024,32] in the first set-abstraction module); the GEMM itself is
// tiny (K <= 320).  These layers are bound by the bytes of the activations, so the kernel is organised
// around touching them once: the columns l (points, or (centre, neighbour) pairs) sit on the MFMA
// columns, which makes the B operand of v_mfma_f32_32x32x2_f32 a plain coalesced row read of x --
// lane (k-half, column) loads x[k][col] straight into the operand register, no LDS staging, no transpose --
// and every accumulator register a run of 32 consecutive columns of one output channel, i.e.
// 128-byte coalesced stores of the [B, Cout, L] layout.  The previous layer's AdaGN + Swish is
// applied to the operand in flight (PRO), this layer's GroupNorm sums leave with the epilogue
// (STATS: per (batch, channel, column tile) sum and sum of squares, folded by lion_groupnorm_fold).
// A wave owns VB x 32 columns x all Cout channels (CB = Cout / 32 row blocks); the weights (k-major
// packed copy [ceil2(Cin)][Cout]) are staged once per workgroup in LDS.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// WLDS = false (short activations: L = 16 .. a few thousand columns): the weights are NOT staged in LDS -- every wave
// reads its A operands (128-byte rows of the k-major packed copy) straight from L2.  A staged 64-137 KiB slice per
// workgroup is repaid only by many column tiles; with a handful of tiles it is pure latency, and a workgroup that needs
// most of a CU's LDS cannot start beside the voxelize / devoxelize / convolution workgroups of the main stream (the
// point branch of a PVConv runs on a side stream, in their shadow).
// OUT (round 4): 0 = store y; 1 = GroupNorm sums only, y is NOT stored; 2 = y is not stored either: this layer's own
// AdaGN + Swish (out_a / out_b f32[B, CoutY]) is applied to the accumulators and the maximum over each block of 32
// consecutive columns -- the 32 neighbours of a set-abstraction centre, reference pvcnn2_ada.py:375-377 -- goes to
// ymax f32[B, CoutY, L / 32] (passed as y).  The last layer of a set-abstraction MLP is evaluated twice this way (sums,
// fold, then activated maximum) and its [B, 64, 1024, 32] = 268 MB output never exists: 134 MB read twice instead of
// 134 MB read + 268 MB written + 268 MB read.
template <int CB, int VB, bool PRO, bool STATS, bool WLDS, int OUT = 0>
__global__ __launch_bounds__(256, 2) void pwconv_kernel(const float *__restrict__ x, const float *__restrict__ wp,
                                                        const float *__restrict__ bias, float *__restrict__ y,
                                                        int Cin, int Cout, int CoutY, int L,
                                                        const float *__restrict__ pro_a,
                                                        const float *__restrict__ pro_b, float *__restrict__ stats,
                                                        const float *__restrict__ out_a = nullptr,
                                                        const float *__restrict__ out_b = nullptr) {
  // Cout: channels of the packed weights (a multiple of 32, zero rows beyond CoutY); CoutY: channels of y / bias / stats
  constexpr int COUT = CB * 32; // output channels of this workgroup: [co0, co0 + COUT)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ksteps = (Cin + 1) >> 1;
  float *sw = smem;                       // [2 * ksteps][COUT] weight slice of this channel tile (WLDS)
  float *spa = sw + (WLDS ? 2 * ksteps * COUT : 0); // [Cin] prologue scale   (PRO)
  float *spb = spa + (PRO ? Cin : 0);     // [Cin] prologue shift   (PRO)
  float *sred = spb + (PRO ? Cin : 0);    // [4][COUT][2]           (STATS)
  float *sbias = sred + 4 * COUT * 2;     // [COUT] bias of this channel tile, zero beyond CoutY (read from global in the
                                          // epilogue, every load would sit in front of a store's vmcnt wait)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, co0 = blockIdx.y * COUT, cl = lane & 31, kh = lane >> 5;
  if (WLDS) {
    for (int e = tid; e < 2 * ksteps * COUT; e += 256) {
      const int k = e / COUT, c = e - k * COUT;
      sw[e] = wp[(size_t)k * Cout + co0 + c];
    }
  }
  if (PRO) {
    for (int c = tid; c < Cin; c += 256) { spa[c] = pro_a[(size_t)b * Cin + c]; spb[c] = pro_b[(size_t)b * Cin + c]; }
  }
  for (int c = tid; c < COUT; c += 256) sbias[c] = (bias && co0 + c < CoutY) ? bias[co0 + c] : 0.f;
  float *soa = sbias + COUT, *sob = soa + COUT; // [COUT] each: this layer's own AdaGN scalars (OUT == 2)
  if (OUT == 2) {
    for (int c = tid; c < COUT; c += 256) {
      const bool ok = co0 + c < CoutY;
      soa[c] = ok ? out_a[(size_t)b * CoutY + co0 + c] : 0.f;
      sob[c] = ok ? out_b[(size_t)b * CoutY + co0 + c] : 0.f;
    }
  }
  __syncthreads();
  const int col0 = (blockIdx.x * 4 + wave) * VB * 32;
  int col[VB];
  bool cok[VB];
#pragma unroll
  for (int vb = 0; vb < VB; ++vb) {
    col[vb] = col0 + vb * 32 + cl;
    cok[vb] = col[vb] < L;
    col[vb] = cok[vb] ? col[vb] : L - 1; // clamped load, select afterwards
  }
  f32x16 acc[CB][VB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int vb = 0; vb < VB; ++vb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][vb][i] = 0.f;

  const float *xb = x + (size_t)b * Cin * L;
  // The activation operand of a whole group of UN k-steps (all of K when Cin <= 2 UN) is requested
  // before the first MFMA: these layers are bound by the activation bytes, so what matters is bytes in
  // flight per CU (UN x VB x 256 B per wave), not MFMA issue.  Weights come from LDS just in time.
  constexpr int UN = 64 / VB;
  for (int s0 = 0; s0 < ksteps; s0 += UN) {
    float bv[UN][VB];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int kc = min(2 * (s0 + u) + kh, Cin - 1);
#pragma unroll
      for (int vb = 0; vb < VB; ++vb) bv[u][vb] = xb[(size_t)kc * L + col[vb]];
    }
    // Round 4: the prologue of the WHOLE group first, the MFMAs afterwards.  Interleaved per operand (rounds 1-3) every
    // pair of MFMAs sat behind its own dependent chain -- LDS read of the scalars, fma, v_exp, v_rcp, mul, select (the
    // ISA: `ds_read x2, wait, v, wait, v v exp v rcp v v, MFMA MFMA` per (k-step, column block)) -- and the matrix pipe
    // ran at ~40 % of its rate inside a wave: SA-0 layer 2 took 108 us without storing a byte, 4x its MFMA time.  As one
    // batch the 64 chains are independent and pipeline; the MFMA sweep that follows is back to back.
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = 2 * (s0 + u) + kh;
      const bool kok = k < Cin; // odd Cin: zero operand (the packed weights are zero there too)
      if (PRO) {
        const int kc = min(k, Cin - 1);
        const float pa_ = spa[kc], pb_ = spb[kc];
#pragma unroll
        for (int vb = 0; vb < VB; ++vb) bv[u][vb] = swish_fast(bv[u][vb] * pa_ + pb_);
      }
#pragma unroll
      for (int vb = 0; vb < VB; ++vb) {
        bv[u][vb] = (kok && cok[vb]) ? bv[u][vb] : 0.f;
        asm volatile("" : "+v"(bv[u][vb])); // materialised HERE: left alone the compiler sinks each chain back in front of its MFMAs
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (s0 + u < ksteps) { // uniform
        const int k = 2 * (s0 + u) + kh;
        float av[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
          av[cb] = WLDS ? sw[k * COUT + cb * 32 + cl] : wp[(size_t)k * Cout + co0 + cb * 32 + cl]; // rows exist to ceil2(Cin)
#pragma unroll
        for (int vb = 0; vb < VB; ++vb) {
#pragma unroll
          for (int cb = 0; cb < CB; ++cb)
            acc[cb][vb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb], bv[u][vb], acc[cb][vb], 0, 0, 0);
        }
      }
    }
  }

  // epilogue: + bias, [B, Cout, L] store.  acc register i of lane l: channel row (i&3) + 8*(i>>2) + 4*(l>>5),
  // column l&31 -> 32 consecutive columns per (register, half-wave).
  float *yb = y + ((size_t)b * CoutY + co0) * L;
  if (OUT == 2) {
    const int M = L >> 5; // centres: blocks of 32 columns (the launcher guarantees L % 32 == 0)
    float *ym = y + ((size_t)b * CoutY + co0) * M;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
        const bool rok = co0 + co < CoutY;
        const float bz = sbias[co], a2 = soa[co], b2 = sob[co];
#pragma unroll
        for (int vb = 0; vb < VB; ++vb) {
          float v = swish_fast((acc[cb][vb][i] + bz) * a2 + b2); // == affine_swish_max on the stored y, bit for bit
          v = row16_max(v);                                       // max over the 32 columns of this half-wave:
          v = fmaxf(v, LION_DPP_F32_ROWS(v, 0x142, 0xa));         // rows 1 / 3 take lane 15 of rows 0 / 2
          const int m = (col0 + vb * 32) >> 5;
          if (cl == 16 && rok && m < M) ym[(size_t)co * M + m] = v;
        }
      }
    return;
  }
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
      const bool rok = co0 + co < CoutY; // padded channel rows (CoutY % 32 != 0) are computed on zeros and dropped
      const float bz = sbias[co];
#pragma unroll
      for (int vb = 0; vb < VB; ++vb) {
        const float o = acc[cb][vb][i] + bz;
        acc[cb][vb][i] = cok[vb] ? o : 0.f;
        if (OUT == 0 && cok[vb] && rok) yb[(size_t)co * L + col[vb]] = o;
      }
    }
  if (STATS) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int vb = 0; vb < VB; ++vb) { s1 += acc[cb][vb][i]; s2 += acc[cb][vb][i] * acc[cb][vb][i]; }
        s1 = row16_sum_rn(s1); s2 = row16_sum_rn(s2);
        s1 = row_pair_sum_odd_rows(s1); s2 = row_pair_sum_odd_rows(s2);
        if (cl == 16) { // the row pair's sum lives in the odd rows
          const int co = cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
          sred[(wave * COUT + co) * 2] = s1;
          sred[(wave * COUT + co) * 2 + 1] = s2;
        }
      }
    __syncthreads();
    for (int c = tid; c < COUT && co0 + c < CoutY; c += 256) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { s1 += sred[(w * COUT + c) * 2]; s2 += sred[(w * COUT + c) * 2 + 1]; }
      float *o = stats + (((size_t)b * CoutY + co0 + c) * gridDim.x + blockIdx.x) * 2;
      o[0] = s1;
      o[1] = s2;
    }
  }
}


// ---- short activations (L <= PW_SMALL_L columns): latency, not bytes ------------------------------------------------
// The layers above with a few hundred columns (point branches of the r = 8 / 16 PVConvs, feature propagation, the
// last set-abstraction stages, the classifier, the attention projections) are GEMMs of a few MFLOP.  With the tiling
// of the large kernel they would run on 32 workgroups, each walking its k-steps behind one L2 round trip at a time
// (measured: 150 us at Cin = 256).  Here a workgroup owns 32 columns x 64 output channels and its four waves SPLIT K:
// a wave requests the operands of all its k-steps (<= 32 per round: 2 weight rows + 1 activation row of 128 bytes
// each) before the first MFMA -- one round trip -- and the four partial tiles are combined through LDS in a fixed
// order.  Same arithmetic (fmaf chains over k, then ((w0 + w1) + w2) + w3), prologue and GroupNorm sums as above.
constexpr int PW_SMALL_L = 4096;
template <bool PRO, bool STATS>
__global__ __launch_bounds__(256) void pwconv_small_kernel(const float *__restrict__ x, const float *__restrict__ wp,
                                                           const float *__restrict__ bias, float *__restrict__ y,
                                                           int Cin, int Cout, int CoutY, int L,
                                                           const float *__restrict__ pro_a,
                                                           const float *__restrict__ pro_b, float *__restrict__ stats) {
  __shared__ float part[4][2][1024];
  extern __shared__ __attribute__((aligned(16))) float smem[]; // [2][Cin] prologue scalars (PRO)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cl = lane & 31, kh = lane >> 5;
  const int b = blockIdx.z, co0 = blockIdx.y * 64, col = blockIdx.x * 32 + cl;
  const bool cok = col < L;
  const int colc = cok ? col : L - 1;
  const int ksteps = (Cin + 1) >> 1, per = (ksteps + 3) >> 2;
  const int s_lo = min(ksteps, wave * per), s_hi = min(ksteps, s_lo + per);
  float *spa = smem, *spb = smem + Cin;
  if (PRO) {
    for (int c = tid; c < Cin; c += 256) { spa[c] = pro_a[(size_t)b * Cin + c]; spb[c] = pro_b[(size_t)b * Cin + c]; }
    __syncthreads();
  }
  float brow[8]; // this thread's 8 output rows (tid / 32 + 8 j): fetched now, not in front of the stores
#pragma unroll
  for (int j = 0; j < 8; ++j) brow[j] = (bias && co0 + (tid >> 5) + 8 * j < CoutY) ? bias[co0 + (tid >> 5) + 8 * j] : 0.f;
  const float *xb = x + (size_t)b * Cin * L;
  f32x16 acc[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
  constexpr int UN = 32;
  for (int s0 = s_lo; s0 < s_hi; s0 += UN) {
    float av[UN][2], bv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = min(2 * min(s0 + u, ksteps - 1) + kh, 2 * ksteps - 1); // packed rows exist up to ceil2(Cin)
      av[u][0] = wp[(size_t)k * Cout + co0 + cl];
      av[u][1] = wp[(size_t)k * Cout + co0 + 32 + cl];
      bv[u] = xb[(size_t)min(k, Cin - 1) * L + colc];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = 2 * (s0 + u) + kh;
      float v = bv[u];
      if (PRO) {
        const int kc = min(k, Cin - 1);
        const float t = v * spa[kc] + spb[kc];
        v = swish_fast(t); // as in pwconv_kernel
      }
      v = (s0 + u < s_hi && k < Cin && cok) ? v : 0.f;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][0], v, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][1], v, acc[1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) part[wave][cb][((i & 3) + 8 * (i >> 2) + 4 * kh) * 32 + cl] = acc[cb][i];
  __syncthreads();
  // thread -> (channel row = tid / 32 + 8 j, column = tid % 32): 8 channels per thread, coalesced 128-byte row stores
  const int c = tid & 31, colo = blockIdx.x * 32 + c;
  const bool ok = colo < L;
  float *yb = y + ((size_t)b * CoutY + co0) * L;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = (tid >> 5) + 8 * j, cb = row >> 5, e = (row & 31) * 32 + c;
    float o = ((part[0][cb][e] + part[1][cb][e]) + part[2][cb][e]) + part[3][cb][e];
    const bool rok = co0 + row < CoutY;
    o += brow[j];
    if (ok && rok) yb[(size_t)row * L + colo] = o;
    if (STATS) {
      float s1 = ok ? o : 0.f, s2 = s1 * s1;
      s1 = row16_sum_rn(s1); s2 = row16_sum_rn(s2);
      s1 = row_pair_sum_odd_rows(s1); s2 = row_pair_sum_odd_rows(s2);
      if (c == 16 && rok) { // the row pair's sum lives in the odd rows
        float *so = stats + (((size_t)b * CoutY + co0 + row) * gridDim.x + blockIdx.x) * 2;
        so[0] = s1;
        so[1] = s2;
      }
    }
  }
}

// nn.Linear on a [B, K] activation (time embedding MLP latent_points_ada.py:47-48, the AdaGN style projections
// adagn.py:45-60 batched by models/adagn.py::StylePlan): y[b][o] = act(bias[o] + sum_k x[b][k] W[o][k]).
// The batch (<= 32 rows per slab) sits on the MFMA columns, 32 output features on the rows, the four waves of a
// workgroup split K and combine through LDS in a fixed order.  W comes k-major from lion_pwconv_pack_weights
// (coalesced 128-byte operand rows); x is a few KiB and stays in L1.  Latency bound by construction: one launch.
__global__ __launch_bounds__(256) void linear_rows_kernel(const float *__restrict__ x, const float *__restrict__ wp,
                                                          const float *__restrict__ bias, int Bn, int K, int O,
                                                          int Opad, int act, float slope, float *__restrict__ y) {
  __shared__ float part[4][1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cl = lane & 31, kh = lane >> 5;
  const int o0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  const int ksteps = (K + 1) >> 1, per = (ksteps + 3) >> 2;
  const int s_lo = min(ksteps, wave * per), s_hi = min(ksteps, s_lo + per);
  const int brow = min(b0 + cl, Bn - 1);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int s = s_lo; s < s_hi; ++s) {
    const int k = 2 * s + kh, kc = min(k, K - 1);
    const float a = wp[(size_t)k * Opad + o0 + cl];          // packed rows exist up to ceil2(K): zero beyond K
    const float v = (k < K && b0 + cl < Bn) ? x[(size_t)brow * K + kc] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v, acc, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) part[wave][((i & 3) + 8 * (i >> 2) + 4 * kh) * 32 + cl] = acc[i];
  __syncthreads();
#pragma unroll
  for (int e = tid; e < 1024; e += 256) {
    const int o = o0 + (e >> 5), b = b0 + (e & 31);
    if (o < O && b < Bn) {
      float v = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
      if (bias) v += bias[o];
      if (act == 1) v = v > 0.f ? v : 0.f;
      else if (act == 2) v = v > 0.f ? v : v * slope;
      y[(size_t)b * O + o] = v;
    }
  }
}

// [Cout][Cin] (nn.Conv1d / Conv2d weight with kernel 1) -> [ceil2(Cin)][ceil64(Cout)], zero padded
__global__ void pwconv_pack_kernel(const float *__restrict__ w, int Cout, int Cout_pad, int Cin, int Cin_pad,
                                   float *__restrict__ wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cin_pad * Cout_pad) return;
  const int co = i % Cout_pad, k = i / Cout_pad;
  wp[i] = (k < Cin && co < Cout) ? w[(size_t)co * Cin + k] : 0.f;
}

// Tile choice: a workgroup covers 4 waves x VB x 32 columns x CB x 32 output channels; CB x VB = 8
// accumulator tiles (128 VGPRs).  All output channels in one workgroup when Cout <= 256 (the activation
// is then read once); Cout = 32: 4 column blocks per wave.
struct PwPlan { int cb, vb; };
static int pw_pad(int Cout) { return (Cout + 31) / 32 * 32; }     // channel rows the tiles cover
static int pw_stride(int Cout) { return (Cout + 63) / 64 * 64; }  // row stride of the packed copy (zero columns beyond Cout)
static size_t pw_lds(int cb, int Cin, bool pro, bool wlds = true) {
  const int ksteps = (Cin + 1) / 2;
  return ((size_t)(wlds ? 2 * ksteps * cb * 32 : 0) + (pro ? 2 * Cin : 0) + 4 * cb * 32 * 2 + 3 * cb * 32) * 4;
}
constexpr size_t PW_LDS_MAX = 150 * 1024;
// Cout: any; the channel tile is the largest of 256 / 128 / 64 / 32 rows that divides ceil32(Cout) and whose weight slice
// (with the prologue scalars: the plan must not depend on the mode) fits LDS
static PwPlan pw_plan(int Cout, int Cin) {
  const int blocks = pw_pad(Cout) / 32;
  if (blocks <= 0 || Cin <= 0) return {0, 0};
  if (blocks % 8 == 0 && pw_lds(8, Cin, true) <= PW_LDS_MAX) return {8, 1};
  if (blocks % 4 == 0 && pw_lds(4, Cin, true) <= PW_LDS_MAX) return {4, 2};
  if (blocks % 2 == 0 && pw_lds(2, Cin, true) <= PW_LDS_MAX) return {2, 4};
  if (pw_lds(1, Cin, true) <= PW_LDS_MAX) return {1, 4};
  return {0, 0};
}

template <int CB, int VB>
static int launch_pw(const float *x, const float *wp, const float *bias, float *y, int B, int Cin, int Cout, int L,
                     const float *pa, const float *pb, float *stats, hipStream_t st) {
  const int Cpad = pw_pad(Cout);
  const dim3 grid(lion_cdiv(L, 4 * VB * 32), Cpad / (CB * 32), B);
  // weights through LDS only when a workgroup's column tile is one of many (the staging is then amortised by the L2)
  const bool wlds = (long)grid.x * B >= 2048;
  const size_t lds = pw_lds(CB, Cin, pa != nullptr, wlds);
#define LION_PW_GO(PRO_, ST_)                                                                              \
  {                                                                                                        \
    static LionLdsLimit cfg = {}, cfg_g = {};                                                              \
    if (wlds) {                                                                                            \
      if (int e = lion_dynamic_lds(&pwconv_kernel<CB, VB, PRO_, ST_, true>, lds, cfg)) return e;           \
      pwconv_kernel<CB, VB, PRO_, ST_, true><<<grid, 256, lds, st>>>(x, wp, bias, y, Cin, pw_stride(Cout), Cout, L, pa, pb, stats); \
    } else {                                                                                               \
      if (int e = lion_dynamic_lds(&pwconv_kernel<CB, VB, PRO_, ST_, false>, lds, cfg_g)) return e;        \
      pwconv_kernel<CB, VB, PRO_, ST_, false><<<grid, 256, lds, st>>>(x, wp, bias, y, Cin, pw_stride(Cout), Cout, L, pa, pb, stats); \
    }                                                                                                      \
  }
  if (pa && stats) LION_PW_GO(true, true)
  else if (pa) LION_PW_GO(true, false)
  else if (stats) LION_PW_GO(false, true)
  else LION_PW_GO(false, false)
#undef LION_PW_GO
  LION_LAUNCH_CHECK();
  return 0;
}

// the two passes of the "activated maximum without the layer's output" form (OUT = 1, OUT = 2): large activations only
// (weights through LDS), same tiling and the same column tiles of the statistics as launch_pw
template <int CB, int VB>
static int launch_pw_max(const float *x, const float *wp, const float *bias, float *ymax, int B, int Cin, int Cout, int L,
                         const float *pa, const float *pb, float *stats, const float *oa, const float *ob, hipStream_t st) {
  const int Cpad = pw_pad(Cout);
  const dim3 grid(lion_cdiv(L, 4 * VB * 32), Cpad / (CB * 32), B);
  if ((long)grid.x * B < 2048) return LION_EUNSUPPORTED;
  const size_t lds = pw_lds(CB, Cin, pa != nullptr, true);
#define LION_PWM_GO(PRO_, ST_, OUT_)                                                                      \
  {                                                                                                        \
    static LionLdsLimit cfg = {};                                                                          \
    if (int e = lion_dynamic_lds(&pwconv_kernel<CB, VB, PRO_, ST_, true, OUT_>, lds, cfg)) return e;       \
    pwconv_kernel<CB, VB, PRO_, ST_, true, OUT_><<<grid, 256, lds, st>>>(x, wp, bias, ymax, Cin, pw_stride(Cout), Cout, L, pa, pb, stats, oa, ob); \
  }
  if (oa) {
    if (pa) LION_PWM_GO(true, false, 2) else LION_PWM_GO(false, false, 2)
  } else {
    if (pa) LION_PWM_GO(true, true, 1) else LION_PWM_GO(false, true, 1)
  }
#undef LION_PWM_GO
  LION_LAUNCH_CHECK();
  return 0;
}

} // namespace

extern "C" {

// The last layer of a set-abstraction MLP without its output (pvcnn2_ada.py:120-164 + :375-377: conv -> AdaGN -> Swish ->
// max over the 32 neighbours): call once with out_a == NULL (GroupNorm sums into stats, nothing else written), fold
// (lion_groupnorm_fold), call again with out_a / out_b f32[B,Cout] = this layer's AdaGN scalars and ymax f32[B,Cout,L/32].
// Same arguments otherwise as lion_pwconv_forward; L % 32 == 0, L > 4096, large grids only (LION_EUNSUPPORTED otherwise:
// the caller then stores y and runs lion_affine_swish_max).  Bit-identical to that three-kernel path.
int lion_pwconv_forward_max(const float *x, const float *wp, const float *bias, int B, int Cin, int Cout, int L,
                            const float *pro_a, const float *pro_b, const float *out_a, const float *out_b, float *stats,
                            float *ymax, lionStream_t stream) {
  if (!x || !wp || B <= 0 || Cin <= 0 || Cout <= 0 || L <= 0) return LION_EINVAL;
  if ((pro_a == nullptr) != (pro_b == nullptr) || (out_a == nullptr) != (out_b == nullptr)) return LION_EINVAL;
  if (out_a ? !ymax : !stats) return LION_EINVAL;
  if (L % 32 != 0 || L <= PW_SMALL_L) return LION_EUNSUPPORTED;
  const PwPlan p = pw_plan(Cout, Cin);
  if (!p.cb) return LION_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (p.cb) {
  case 1: return launch_pw_max<1, 4>(x, wp, bias, ymax, B, Cin, Cout, L, pro_a, pro_b, stats, out_a, out_b, st);
  case 2: return launch_pw_max<2, 4>(x, wp, bias, ymax, B, Cin, Cout, L, pro_a, pro_b, stats, out_a, out_b, st);
  case 4: return launch_pw_max<4, 2>(x, wp, bias, ymax, B, Cin, Cout, L, pro_a, pro_b, stats, out_a, out_b, st);
  default: return launch_pw_max<8, 1>(x, wp, bias, ymax, B, Cin, Cout, L, pro_a, pro_b, stats, out_a, out_b, st);
  }
}

size_t lion_pwconv_packed_floats(int Cout, int Cin) { return (size_t)((Cin + 1) / 2 * 2) * pw_stride(Cout); }

int lion_pwconv_pack_weights(const float *w, int Cout, int Cin, float *wp, lionStream_t stream) {
  if (!w || !wp || Cout <= 0 || Cin <= 0) return LION_EINVAL;
  const int Cin_pad = (Cin + 1) / 2 * 2, Cout_pad = pw_stride(Cout);
  pwconv_pack_kernel<<<lion_cdiv(Cin_pad * Cout_pad, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(w, Cout, Cout_pad, Cin,
                                                                                                       Cin_pad, wp);
  LION_LAUNCH_CHECK();
  return 0;
}

// column tiles per batch element = rows of the stats tensor per channel (0: shape not supported)
int lion_pwconv_stat_tiles(int Cout, int Cin, int L) {
  if (L > 0 && L <= PW_SMALL_L && Cout > 0 && Cin > 0) return lion_cdiv(L, 32); // pwconv_small_kernel: 32-column tiles
  const PwPlan p = pw_plan(Cout, Cin);
  return (p.cb && L > 0) ? lion_cdiv(L, 4 * p.vb * 32) : 0;
}

// x f32[B,Cin,L], wp from lion_pwconv_pack_weights, bias f32[Cout] or NULL -> y f32[B,Cout,L];
// any Cout (channel tiles of 32..256, rows beyond Cout are zero weights and never stored), weight slice <= 150 KiB of LDS.  pro_a / pro_b f32[B,Cin] (both or neither):
// the input is swish(x*a+b).  stats f32[B,Cout,lion_pwconv_stat_tiles(Cout,Cin,L),2] or NULL.
// Built around the large activations (set-abstraction MLPs, L = M*U: one read + one write per layer); the short ones
// (L = 16 .. 256 points, classifier, attention projections) are latency bound whatever runs them and use it too.
int lion_pwconv_forward(const float *x, const float *wp, const float *bias, int B, int Cin, int Cout, int L,
                        const float *pro_a, const float *pro_b, float *y, float *stats, lionStream_t stream) {
  if (!x || !wp || !y || B <= 0 || Cin <= 0 || Cout <= 0 || L <= 0) return LION_EINVAL;
  if ((pro_a == nullptr) != (pro_b == nullptr)) return LION_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (L <= PW_SMALL_L) {
    const dim3 grid(lion_cdiv(L, 32), pw_stride(Cout) / 64, B);
    const size_t lds = pro_a ? (size_t)2 * Cin * 4 : 0;
    if (lds > 64 * 1024) return LION_EUNSUPPORTED;
    if (pro_a && stats) pwconv_small_kernel<true, true><<<grid, 256, lds, st>>>(x, wp, bias, y, Cin, pw_stride(Cout), Cout, L, pro_a, pro_b, stats);
    else if (pro_a) pwconv_small_kernel<true, false><<<grid, 256, lds, st>>>(x, wp, bias, y, Cin, pw_stride(Cout), Cout, L, pro_a, pro_b, stats);
    else if (stats) pwconv_small_kernel<false, true><<<grid, 256, lds, st>>>(x, wp, bias, y, Cin, pw_stride(Cout), Cout, L, pro_a, pro_b, stats);
    else pwconv_small_kernel<false, false><<<grid, 256, lds, st>>>(x, wp, bias, y, Cin, pw_stride(Cout), Cout, L, pro_a, pro_b, stats);
    LION_LAUNCH_CHECK();
    return 0;
  }
  const PwPlan p = pw_plan(Cout, Cin);
  if (!p.cb) return LION_EUNSUPPORTED;
  switch (p.cb) {
  case 1: return launch_pw<1, 4>(x, wp, bias, y, B, Cin, Cout, L, pro_a, pro_b, stats, st);
  case 2: return launch_pw<2, 4>(x, wp, bias, y, B, Cin, Cout, L, pro_a, pro_b, stats, st);
  case 4: return launch_pw<4, 2>(x, wp, bias, y, B, Cin, Cout, L, pro_a, pro_b, stats, st);
  default: return launch_pw<8, 1>(x, wp, bias, y, B, Cin, Cout, L, pro_a, pro_b, stats, st);
  }
}

// y f32[B,O] = act(x f32[B,K] W^T + bias), wp = lion_pwconv_pack_weights(W f32[O,K]) (k-major, O padded to 32);
// act 0 none / 1 relu / 2 leaky-relu(slope).  nn.Linear of the time-embedding MLP and the batched AdaGN projections.
int lion_linear_forward(const float *x, const float *wp, const float *bias, int B, int K, int O, int act, float slope,
                        float *y, lionStream_t stream) {
  if (!x || !wp || !y || B <= 0 || K <= 0 || O <= 0 || act < 0 || act > 2) return LION_EINVAL;
  const int Opad = pw_stride(O);
  linear_rows_kernel<<<dim3(pw_pad(O) / 32, lion_cdiv(B, 32)), 256, 0, static_cast<hipStream_t>(stream)>>>(
      x, wp, bias, B, K, O, Opad, act, slope, y);
  LION_LAUNCH_CHECK();
  return 0;
}

} // extern "C"
