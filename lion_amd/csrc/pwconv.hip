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

template <int CB, int VB, bool PRO, bool STATS>
__global__ __launch_bounds__(256, 2) void pwconv_kernel(const float *__restrict__ x, const float *__restrict__ wp,
                                                        const float *__restrict__ bias, float *__restrict__ y,
                                                        int Cin, int Cout, int L, const float *__restrict__ pro_a,
                                                        const float *__restrict__ pro_b, float *__restrict__ stats) {
  constexpr int COUT = CB * 32; // output channels of this workgroup: [co0, co0 + COUT)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ksteps = (Cin + 1) >> 1;
  float *sw = smem;                       // [2 * ksteps][COUT] weight slice of this channel tile
  float *spa = sw + 2 * ksteps * COUT;    // [Cin] prologue scale   (PRO)
  float *spb = spa + (PRO ? Cin : 0);     // [Cin] prologue shift   (PRO)
  float *sred = spb + (PRO ? Cin : 0);    // [4][COUT][2]           (STATS)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, co0 = blockIdx.y * COUT, cl = lane & 31, kh = lane >> 5;
  for (int e = tid; e < 2 * ksteps * COUT; e += 256) {
    const int k = e / COUT, c = e - k * COUT;
    sw[e] = wp[(size_t)k * Cout + co0 + c];
  }
  if (PRO) {
    for (int c = tid; c < Cin; c += 256) { spa[c] = pro_a[(size_t)b * Cin + c]; spb[c] = pro_b[(size_t)b * Cin + c]; }
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
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (s0 + u < ksteps) { // uniform
        const int k = 2 * (s0 + u) + kh;
        const bool kok = k < Cin; // odd Cin: zero operand (the packed weights are zero there too)
        float av[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) av[cb] = sw[k * COUT + cb * 32 + cl];
#pragma unroll
        for (int vb = 0; vb < VB; ++vb) {
          float v = bv[u][vb];
          if (PRO) {
            const int kc = min(k, Cin - 1);
            const float t = v * spa[kc] + spb[kc];
            v = t * __frcp_rn(1.0f + __expf(-t)); // swish
          }
          v = (kok && cok[vb]) ? v : 0.f;
#pragma unroll
          for (int cb = 0; cb < CB; ++cb)
            acc[cb][vb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb], v, acc[cb][vb], 0, 0, 0);
        }
      }
    }
  }

  // epilogue: + bias, [B, Cout, L] store.  acc register i of lane l: channel row (i&3) + 8*(i>>2) + 4*(l>>5),
  // column l&31 -> 32 consecutive columns per (register, half-wave).
  float *yb = y + ((size_t)b * Cout + co0) * L;
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
      const float bz = bias ? bias[co0 + co] : 0.f;
#pragma unroll
      for (int vb = 0; vb < VB; ++vb) {
        const float o = acc[cb][vb][i] + bz;
        acc[cb][vb][i] = cok[vb] ? o : 0.f;
        if (cok[vb]) yb[(size_t)co * L + col[vb]] = o;
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
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        if (cl == 0) {
          const int co = cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
          sred[(wave * COUT + co) * 2] = s1;
          sred[(wave * COUT + co) * 2 + 1] = s2;
        }
      }
    __syncthreads();
    for (int c = tid; c < COUT; c += 256) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { s1 += sred[(w * COUT + c) * 2]; s2 += sred[(w * COUT + c) * 2 + 1]; }
      float *o = stats + (((size_t)b * Cout + co0 + c) * gridDim.x + blockIdx.x) * 2;
      o[0] = s1;
      o[1] = s2;
    }
  }
}

// [Cout][Cin] (nn.Conv1d / Conv2d weight with kernel 1) -> [ceil2(Cin)][Cout], zero padded
__global__ void pwconv_pack_kernel(const float *__restrict__ w, int Cout, int Cin, int Cin_pad,
                                   float *__restrict__ wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cin_pad * Cout) return;
  const int co = i % Cout, k = i / Cout;
  wp[i] = k < Cin ? w[(size_t)co * Cin + k] : 0.f;
}

// Tile choice: a workgroup covers 4 waves x VB x 32 columns x CB x 32 output channels; CB x VB = 8
// accumulator tiles (128 VGPRs).  All output channels in one workgroup when Cout <= 256 (the activation
// is then read once); Cout = 32: 4 column blocks per wave.
struct PwPlan { int cb, vb; };
static PwPlan pw_plan(int Cout) {
  switch (Cout) {
  case 32: return {1, 4};
  case 64: return {2, 4};
  case 128: return {4, 2};
  case 256: return {8, 1};
  default: return {0, 0};
  }
}
static size_t pw_lds(int cb, int Cin, bool pro) {
  const int ksteps = (Cin + 1) / 2;
  return ((size_t)2 * ksteps * cb * 32 + (pro ? 2 * Cin : 0) + 4 * cb * 32 * 2) * 4;
}

template <int CB, int VB>
static int launch_pw(const float *x, const float *wp, const float *bias, float *y, int B, int Cin, int Cout, int L,
                     const float *pa, const float *pb, float *stats, hipStream_t st) {
  const dim3 grid(lion_cdiv(L, 4 * VB * 32), Cout / (CB * 32), B);
  const size_t lds = pw_lds(CB, Cin, pa != nullptr);
#define LION_PW_GO(PRO_, ST_)                                                                              \
  {                                                                                                        \
    static LionLdsLimit cfg = {};                                                                          \
    if (int e = lion_dynamic_lds(&pwconv_kernel<CB, VB, PRO_, ST_>, lds, cfg)) return e;                   \
    pwconv_kernel<CB, VB, PRO_, ST_><<<grid, 256, lds, st>>>(x, wp, bias, y, Cin, Cout, L, pa, pb, stats); \
  }
  if (pa && stats) LION_PW_GO(true, true)
  else if (pa) LION_PW_GO(true, false)
  else if (stats) LION_PW_GO(false, true)
  else LION_PW_GO(false, false)
#undef LION_PW_GO
  LION_LAUNCH_CHECK();
  return 0;
}

} // namespace

extern "C" {

size_t lion_pwconv_packed_floats(int Cout, int Cin) { return (size_t)((Cin + 1) / 2 * 2) * Cout; }

int lion_pwconv_pack_weights(const float *w, int Cout, int Cin, float *wp, lionStream_t stream) {
  if (!w || !wp || Cout <= 0 || Cin <= 0) return LION_EINVAL;
  const int Cin_pad = (Cin + 1) / 2 * 2;
  pwconv_pack_kernel<<<lion_cdiv(Cin_pad * Cout, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(w, Cout, Cin, Cin_pad, wp);
  LION_LAUNCH_CHECK();
  return 0;
}

// column tiles per batch element = rows of the stats tensor per channel (0: shape not supported)
int lion_pwconv_stat_tiles(int Cout, int L) {
  const PwPlan p = pw_plan(Cout);
  return (p.cb && L > 0) ? lion_cdiv(L, 4 * p.vb * 32) : 0;
}

// x f32[B,Cin,L], wp from lion_pwconv_pack_weights, bias f32[Cout] or NULL -> y f32[B,Cout,L];
// Cout in {32,64,128,256}, weight slice <= 150 KiB of LDS.  pro_a / pro_b f32[B,Cin] (both or neither):
// the input is swish(x*a+b).  stats f32[B,Cout,lion_pwconv_stat_tiles(Cout,L),2] or NULL.
// Meant for the large activations (set-abstraction MLPs, L = M*U); short ones are latency bound and
// better served by the library GEMM.
int lion_pwconv_forward(const float *x, const float *wp, const float *bias, int B, int Cin, int Cout, int L,
                        const float *pro_a, const float *pro_b, float *y, float *stats, lionStream_t stream) {
  if (!x || !wp || !y || B <= 0 || Cin <= 0 || Cout <= 0 || L <= 0) return LION_EINVAL;
  if ((pro_a == nullptr) != (pro_b == nullptr)) return LION_EINVAL;
  const PwPlan p = pw_plan(Cout);
  if (!p.cb || pw_lds(p.cb, Cin, pro_a != nullptr) > 150 * 1024) return LION_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (p.cb) {
  case 1: return launch_pw<1, 4>(x, wp, bias, y, B, Cin, Cout, L, pro_a, pro_b, stats, st);
  case 2: return launch_pw<2, 4>(x, wp, bias, y, B, Cin, Cout, L, pro_a, pro_b, stats, st);
  case 4: return launch_pw<4, 2>(x, wp, bias, y, B, Cin, Cout, L, pro_a, pro_b, stats, st);
  default: return launch_pw<8, 1>(x, wp, bias, y, B, Cin, Cout, L, pro_a, pro_b, stats, st);
  }
}

} // extern "C"
