// conv3d.hip -- C3: the 3x3x3 / pad 1 / stride 1 Conv3d of PVConv's voxel branch (97 % of the
// denoiser's FLOPs) as an implicit GEMM on the fp32-input matrix cores.
//
// Reference: nn.Conv3d in models/pvcnn2_ada.py:211-222 (cuDNN there, MIOpen's CK grouped-conv here:
// 49 TFLOP/s in the measured step, plus NCDHW<->NDHWC transposes around every call).
//
// D[co, voxel] = sum_{ci, tap} W[co, ci, tap] * X[ci, voxel + tap]      (per batch element)
// is computed with v_mfma_f32_32x32x2_f32: A = weights (32 output channels x 2 k), B = input
// (2 k x 32 voxels), exact fp32 (each MFMA is an fmaf chain, one rounding per product), so the
// result differs from any other fp32 convolution only by summation order.  Putting the voxels on the
// MFMA columns makes every accumulator register a run of 32 consecutive voxels of one channel, i.e.
// the NCDHW output is written with 128-byte coalesced stores and needs no layout change at all.
//
// Workgroup = TD x TH x TW output voxels (256) x COT output channels, 4 waves; wave w owns voxels
// [64w, 64w+64) x COT channels = 2x2 MFMA tiles (64 accumulator VGPRs).  K is walked in chunks of
// KC = 4 input channels (in-situ A/B over the whole denoiser step: KC 4 / 2 waves per SIMD 29.2 ms,
// KC 4 / 3 waves 30.2, KC 2 / 3 waves 31.4 -- although KC 2 wins the isolated micro-benchmark):
// the haloed input tile [KC][TD+2][TH+2][TW+2] (zero padded at the borders)
// and the weight slice [KC][27][COT] (pre-packed once per weight tensor into [Cin][27][Cout]) are
// staged in 41 KiB of LDS; the next chunk's global loads are in flight while the current chunk's
// 54 k-steps (216 MFMAs per wave) execute.  Each k-step is 4 conflict-free ds_read_b32 + 4 MFMAs.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef LION_CONV_KC
#define LION_CONV_KC 4
#endif
#ifndef LION_CONV_WAVES
#define LION_CONV_WAVES 2
#endif
constexpr int KC = LION_CONV_KC;

// PRO:   the staged input is swish(x * pro_a[b][ci] + pro_b[b][ci]) (AdaGN affine + Swish of the
//        previous layer, pvcnn2_ada.py:212-218, applied on the fly; zero padding stays zero).
// STATS: per (batch, output channel, spatial tile) sum and sum of squares of the output are written
//        to stats[b][co][tile][2] (GroupNorm statistics of the NEXT AdaGN without another pass).
// AdaGN affine + Swish of the previous layer, applied to staged inputs (and to the previous conv's bias for the
// constant-response decomposition below): ONE definition, so both see the same bits.
__device__ __forceinline__ float pro_act(float v, float pa, float pb) {
  const float t = v * pa + pb;
  return swish_fast(t);
}

template <int TD, int TH, int TW, int COT, int VB, bool PRO, bool STATS>
__global__ __launch_bounds__(256, LION_CONV_WAVES) void conv3d_k3_kernel(const float *__restrict__ x,
                                                                const float *__restrict__ wp,
                                                                const float *__restrict__ bias,
                                                                float *__restrict__ y, int Cin,
                                                                int Cout, int r,
                                                                const float *__restrict__ pro_a,
                                                                const float *__restrict__ pro_b,
                                                                const float *__restrict__ pro_bias,
                                                                const float *__restrict__ tconst,
                                                                float *__restrict__ stats,
                                                                int32_t *__restrict__ occ, int B,
                                                                int ntiles) {
  constexpr int TM = 256;                                // threads: 4 waves, each owning VB x 32 voxels
  static_assert(TD * TH * TW == 4 * VB * 32, "tile voxels = 4 waves x VB column blocks x 32");
  constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2, HALO = HD * HH * HW;
  constexpr int NJ = (HALO + TM - 1) / TM;               // staged input floats per thread and channel
  constexpr int WV4 = KC * 27 * COT / 4;                 // float4 of weights per chunk
  constexpr int NWV = (WV4 + TM - 1) / TM;               // staged weight float4 per thread
  constexpr int CB = COT / 32;                           // 32-channel MFMA row blocks
  constexpr int SWS = ((KC * 27 * COT + 255) / 256) * 256 + 256; // floats per weight buffer (+1 KiB: DMA lanes past the slice)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *sx = smem;                       // [KC][HALO]       input tile of the current chunk
  float *sw = sx + ((KC * HALO + 3) & ~3); // [2][SWS]         weight slices, double buffered (LDS-DMA)
  float *sbias = sw + 2 * SWS;            // [COT]
  const int npro = PRO ? ((Cin + 63) & ~63) : 0; // prologue scalars per input channel (Cin <= 256)
  float *spa = sbias + COT, *spb = spa + npro;
  float *spc = spb + npro;                // [npro]           activated constant of each input channel (delta mode)
  float *sred = spc + npro;               // [waves][COT][2]  per-wave channel sums (STATS)
  float *sT = sx;                         // [27][COT] constant response per border configuration (delta mode): loaded
                                          // after the K loop, into the then idle input tile
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Dense launch: grid = (B, tiles, Cout/COT); ids run batch-fastest, i.e. round-robin over the 8 XCDs, so a sample
  // (and its halo re-reads) stays on one XCD's L2.
  // Sparse launch (occ != NULL, the conv that reads the voxelised grid): persistent workgroups pull (sample, tile)
  // items from a list -- occupied tiles first, empty ones (K loop skipped, output = bias) last -- through one atomic
  // counter.  A static grid gains nothing from skipping: workgroup ids are bound to XCDs round-robin, the occupied
  // tiles of a cloud cluster in id space (measured: 74 % empty tiles, 0 % faster), and 400-us workgroups quantise
  // the tail; the queue balances over all CUs.  occ = [B*tiles flags][B*tiles list (per sample)][queue].
  __shared__ int s_work;
  const bool queued = occ != nullptr;
  const int ncz = Cout / COT;
  for (int iter = 0;; ++iter) {
  int b, tile, co0;
  if (queued) {
    __syncthreads(); // the previous item is done with LDS
    if (tid == 0) s_work = atomicAdd(occ + 2 * B * ntiles, 1);
    __syncthreads();
    const int work = s_work;
    if (work >= B * ntiles * ncz) break;
    const int item = work / ncz; // item-th entry: sample item % B, its (item / B)-th tile, occupied tiles first
    b = item % B;
    tile = occ[B * ntiles + b * ntiles + item / B];
    co0 = (work % ncz) * COT;
  } else {
    if (iter) break;
    b = blockIdx.x;
    tile = blockIdx.y;
    co0 = blockIdx.z * COT;
  }
  const int ntw = r / TW, nth = r / TH;
  const int tw_i = tile % ntw, th_i = (tile / ntw) % nth, td_i = tile / (ntw * nth);
  const int d0 = td_i * TD, h0 = th_i * TH, w0 = tw_i * TW;
  const int r2 = r * r, r3 = r2 * r;

  // Delta mode (tconst != NULL, PRO only).  The activated input of the second convolution of a PVConv is a per-channel
  // CONSTANT c = swish(A*bias1+Bs) wherever the first convolution saw no point (its output is exactly bias1 there), i.e.
  // almost everywhere: z = c*[inside the grid] + delta with delta sparse.  conv(c-field) is a per-(batch, border
  // configuration, output channel) constant tconst[b][27][Cout] (zero padding makes it differ on faces/edges/corners);
  // the MFMA loop convolves delta only, and tiles whose 2-voxel neighbourhood holds no point skip it (occ, margin 2).
  const bool delta = PRO && tconst != nullptr;
  if (PRO) {
    for (int c = tid; c < Cin; c += TM) {
      const float pa = pro_a[(size_t)b * Cin + c], pb = pro_b[(size_t)b * Cin + c];
      spa[c] = pa;
      spb[c] = pb;
      spc[c] = delta ? pro_act(pro_bias ? pro_bias[c] : 0.f, pa, pb) : 0.f;
    }
  }
  for (int c = tid; c < COT; c += TM) sbias[c] = bias ? bias[co0 + c] : 0.f;
  // spatial offsets of the halo positions this thread stages (the same for every input channel:
  // slot (c, j) is position p = tid + j*TM of channel c, so the channel -- and with it the prologue
  // scalars -- is uniform per slot and needs no per-element lookup).  Out-of-range positions (zero
  // padding) carry an offset past the buffer's num_records, for which buffer loads return 0: no
  // branch around any load, no select.
  int goff[NJ];
  bool gok[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int p = tid + j * TM;
    const int hd = p / (HH * HW), hh = (p / HW) % HH, hw = p % HW;
    const int gd = d0 - 1 + hd, gh = h0 - 1 + hh, gw = w0 - 1 + hw;
    gok[j] = p < HALO && gd >= 0 && gd < r && gh >= 0 && gh < r && gw >= 0 && gw < r;
    goff[j] = gok[j] ? ((gd * r + gh) * r + gw) * 4 : 0x7fffff00; // out of range: the buffer load returns 0
  }
  // LDS offsets of this lane's B operands (input): voxel (d,h,w) of each of the wave's 2 column blocks
  int boff[VB];
#pragma unroll
  for (int vb = 0; vb < VB; ++vb) {
    const int v = (wave * VB + vb) * 32 + (lane & 31);
    const int d = v / (TH * TW), h = (v / TW) % TH, w = v % TW;
    boff[vb] = (lane >> 5) * HALO + (d * HH + h) * HW + w;
  }
  const int aoff = (lane >> 5) * 27 * COT + (lane & 31);

  f32x16 acc[CB][VB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int vb = 0; vb < VB; ++vb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][vb][i] = 0.f;

  // Input: raw buffer loads -- one VGPR (the halo offset, shared by all channels) + an SGPR channel
  // offset per load instead of a 64-bit address pair each (the flat-address version of this kernel
  // spilled its staging registers to scratch and serialised every load behind a vmcnt(0)).
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(x + (size_t)b * Cin * r3), 0, Cin * r3 * 4, 0x00020000);
  float rx[KC][NJ];
  // Weights: LDS-DMA (global_load_lds_dwordx4) straight into the other weight buffer: no staging
  // registers, no ds_write pass.  Row (ci, tap) of the packed [Cin][27][Cout] tensor holds COT
  // contiguous floats of this workgroup's channel tile; float4 e of the slice lands at sw + 16 e.
  typedef __attribute__((address_space(3))) float lds_float;
  const uint32_t sw_lds = (uint32_t)(uintptr_t)(lds_float *)sw;
  auto load_chunk = [&](int q) {
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
      const int e0 = wave * 64 + i * TM; // wave-uniform
      if (e0 < WV4) {
        const int e = min(e0 + lane, WV4 - 1);
        const int row = e / (COT / 4), j4 = e - row * (COT / 4);
        const float *g = wp + ((size_t)q * KC * 27 + row) * Cout + co0 + j4 * 4;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(sw_lds + (uint32_t)(((q & 1) * SWS + e0 * 4) * 4));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
      }
    }
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        rx[c][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, goff[j], (q * KC + c) * r3 * 4, 0));
  };
  // occ (optional): 0 = every input voxel of this tile's halo is zero in ALL channels (conv1 of a PVConv reads the
  // voxelised grid, >= 94 % zeros, and whole tiles far from the cloud are empty).  The K loop is skipped and the
  // epilogue writes bias -- bit-identical to accumulating the zeros.
  const int wmask = queued ? (occ[b * ntiles + tile] & 0xf) : 0xf; // bit w: wave w's 64 voxels see a point (sparse launches)
  const bool empty = wmask == 0;
  const bool wave_on = (wmask >> wave) & 1;
  const int nchunks = empty ? 0 : Cin / KC;
  if (!empty) load_chunk(0);
  for (int q = 0; q < nchunks; ++q) {
    __syncthreads(); // everyone is done reading the previous chunk from LDS
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      float pa = 1.f, pb = 0.f, pc = 0.f;
      if (PRO) { pa = spa[q * KC + c]; pb = spb[q * KC + c]; pc = spc[q * KC + c]; } // uniform broadcast reads
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int p = tid + j * TM;
        float v = rx[c][j];
        if (PRO) v = pro_act(v, pa, pb) - pc; // pc = 0 unless delta mode
        if (p < HALO) sx[c * HALO + p] = (!PRO || gok[j]) ? v : 0.f; // zero padding stays zero
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's share of the weight slice q has landed
    __syncthreads();
    if (q + 1 < nchunks) load_chunk(q + 1); // in flight during the MFMAs below

    // 54 k-steps (2 channel pairs x 27 taps), software pipelined by hand: the LDS reads of step s+1
    // are issued before the MFMAs of step s (the compiler otherwise places each ds_read right in front
    // of its consumer and stalls ~100 cycles on lgkmcnt(0) every 256 MFMA cycles: 70 % -> MFMA-bound).
    constexpr int NS = (KC / 2) * 27;
    const float *swq = sw + (q & 1) * SWS;
    float av[2][CB], bv[2][VB];
    auto lds_step = [&](int st, int buf) {
      const int cp = st / 27, tap = st % 27;
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      const int toff = (kd * HH + kh) * HW + kw + cp * 2 * HALO;
#pragma unroll
      for (int vb = 0; vb < VB; ++vb) bv[buf][vb] = sx[boff[vb] + toff];
      const float *ap = swq + aoff + (cp * 2 * 27 + tap) * COT;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) av[buf][cb] = ap[cb * 32];
    };
    if (wave_on) { // a wave whose block is empty only takes part in the staging and the barriers
    lds_step(0, 0);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      if (st + 1 < NS) lds_step(st + 1, (st + 1) & 1);
      __builtin_amdgcn_sched_barrier(0); // keep the prefetch above this step's MFMAs
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int vb = 0; vb < VB; ++vb)
          acc[cb][vb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st & 1][cb], bv[st & 1][vb], acc[cb][vb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
  }

  if (delta) {
    __syncthreads(); // the last chunk's LDS reads are done: the input tile buffer becomes the response table
    for (int e = tid; e < 27 * COT; e += TM) sT[e] = tconst[((size_t)b * 27 + e / COT) * Cout + co0 + e % COT];
    __syncthreads();
  } else if (empty) {
    __syncthreads(); // sbias was written by other threads and no barrier of the K loop ran
  }
  // epilogue: + bias, NCDHW store.  acc register i of lane l: channel row (i&3) + 8*(i>>2) + 4*(l>>5),
  // voxel column l&31 -> 32 consecutive voxels per (register, half-wave).
  float *yb = y + ((size_t)b * Cout + co0) * r3;
#pragma unroll
  for (int vb = 0; vb < VB; ++vb) {
    const int v = (wave * VB + vb) * 32 + (lane & 31);
    const int d = v / (TH * TW), h = (v / TW) % TH, w = v % TW;
    const int gv = ((d0 + d) * r + (h0 + h)) * r + (w0 + w);
    // border configuration of this voxel: 0 = on the low face, 2 = on the high face, 1 = interior, per axis
    const int gd = d0 + d, gh = h0 + h, gw = w0 + w;
    const int cfg = (((gd == 0 ? 0 : gd == r - 1 ? 2 : 1) * 3 + (gh == 0 ? 0 : gh == r - 1 ? 2 : 1)) * 3 +
                     (gw == 0 ? 0 : gw == r - 1 ? 2 : 1));
    const float *addv = delta ? sT + cfg * COT : sbias;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        const float o = acc[cb][vb][i] + addv[co];
        acc[cb][vb][i] = o;
        yb[(size_t)co * r3 + gv] = o;
      }
  }
  if (STATS) {
    // channel sums over this tile: columns (voxels) live in the 32 lanes of a half-wave
    constexpr int NWAVE = 4;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float s1 = acc[cb][0][i], s2 = acc[cb][0][i] * acc[cb][0][i];
#pragma unroll
        for (int vb = 1; vb < VB; ++vb) { s1 += acc[cb][vb][i]; s2 += acc[cb][vb][i] * acc[cb][vb][i]; }
        s1 = row16_sum_rn(s1); s2 = row16_sum_rn(s2); // 16-lane rows on DPP, then the two rows of the half-wave
        s1 = row_pair_sum_odd_rows(s1); s2 = row_pair_sum_odd_rows(s2);
        if ((lane & 31) == 16) { // the row pair's sum lives in the odd rows
          const int co = cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
          sred[(wave * COT + co) * 2] = s1;
          sred[(wave * COT + co) * 2 + 1] = s2;
        }
      }
    __syncthreads();
    if (tid < COT) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVE; ++w) { s1 += sred[(w * COT + tid) * 2]; s2 += sred[(w * COT + tid) * 2 + 1]; }
      float *o = stats + (((size_t)b * Cout + co0 + tid) * ntiles + tile) * 2;
      o[0] = s1;
      o[1] = s2;
    }
  }
  } // work loop
  // Round 5: the queue re-arms itself -- the last workgroup to leave (every other one has popped its final, out-of-range
  // index) zeroes the queue and the exit counter, so ONE occupancy buffer serves every convolution that shares its
  // (cloud, resolution) pair without the per-use clone (7 copy launches per denoiser step).
  if (queued && tid == 0) {
    int32_t *q = occ + 2 * B * ntiles;
    if (atomicAdd(q + 1, 1) == (int)gridDim.x - 1) { q[0] = 0; q[1] = 0; }
  }
}

// GroupNorm(G groups) + adaptive affine folded into per-(batch, channel) scalars:
//   AdaGN(x)[b,c,:] = x * A[b,c] + Bs[b,c],  A = rstd_g * gamma_c * f_bc,
//   Bs = (beta_c - mean_g * rstd_g * gamma_c) * f_bc + g_bc            (models/adagn.py:61-64)
// from the conv epilogue's tile sums (fixed summation order, double accumulation).  Also emits the
// per-channel mean of the conv output (SE3d needs the mean of the normalised grid, which is affine in it).
__global__ void gn_fold_kernel(const float *__restrict__ stats, int C, int T, int G, float count,
                               const float *__restrict__ gamma, const float *__restrict__ beta,
                               const float *__restrict__ fac, const float *__restrict__ gbias, int ld_fg, float eps,
                               float *__restrict__ A, float *__restrict__ Bs, float *__restrict__ chmean) {
  // one 256-thread workgroup per (batch, group); C/G <= 64 channels per group.  LPC = 256 / pow2ceil(C/G) lanes share
  // a channel: lane `sub` accumulates tiles sub, sub + LPC, ... in double, the LPC partials are combined with an
  // xor-butterfly (consecutive lanes of one wave), the channel totals with a fixed-order loop: deterministic.
  __shared__ double cs[64][2];
  __shared__ double gs[2];
  const int b = blockIdx.y, g = blockIdx.x, cpg = C / G, tid = threadIdx.x;
  int cp2 = 1;
  while (cp2 < cpg) cp2 <<= 1;
  const int LPC = 256 / cp2, ch = tid / LPC, sub = tid % LPC; // LPC in {4, ..., 256}, a power of two
  double s1 = 0.0, s2 = 0.0;
  if (ch < cpg) {
    const float2 *p = reinterpret_cast<const float2 *>(stats + (((size_t)b * C + g * cpg + ch) * T) * 2);
    for (int t = sub; t < T; t += LPC) { const float2 v = p[t]; s1 += (double)v.x; s2 += (double)v.y; }
  }
  for (int m = 1; m < LPC && m < 64; m <<= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
  if (LPC > 64) { // a channel spans several waves (C/G <= 2): combine through LDS
    __shared__ double ws[4][2];
    if ((tid & 63) == 0) { ws[tid >> 6][0] = s1; ws[tid >> 6][1] = s2; }
    __syncthreads();
    if (sub == 0) {
      s1 = 0.0; s2 = 0.0;
      for (int w = 0; w < LPC / 64; ++w) { s1 += ws[(tid >> 6) + w][0]; s2 += ws[(tid >> 6) + w][1]; }
    }
  }
  if (ch < cpg && sub == 0) {
    cs[ch][0] = s1;
    cs[ch][1] = s2;
    chmean[(size_t)b * C + g * cpg + ch] = (float)(s1 / count);
  }
  __syncthreads();
  if (tid == 0) {
    double g1 = 0.0, g2 = 0.0;
    for (int c = 0; c < cpg; ++c) { g1 += cs[c][0]; g2 += cs[c][1]; }
    gs[0] = g1;
    gs[1] = g2;
  }
  __syncthreads();
  if (tid < cpg) {
    const double n = (double)count * cpg;
    const double mean = gs[0] / n;
    double var = gs[1] / n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const int c = g * cpg + tid;
    const float f = fac[(size_t)b * ld_fg + c], gb = gbias[(size_t)b * ld_fg + c];
    const float a0 = rstd * gamma[c];
    A[(size_t)b * C + c] = a0 * f;
    Bs[(size_t)b * C + c] = (beta[c] - (float)mean * a0) * f + gb;
  }
}

// gn_fold_kernel + the SE3d gate (csrc/pointwise.hip se_gate_kernel) in ONE launch, one workgroup per sample: the second
// AdaGN of a PVConv is always followed by the gate, and both are a few hundred flops behind a 5-us launch (round 3: 14
// launches per denoiser step less).  C <= 256, hidden <= 128.  Channel sums: LPC = 256 / pow2ceil(C) consecutive lanes per
// channel, each over tiles sub, sub + LPC, ... in double, combined with an xor-butterfly; group totals by a fixed-order loop.
__global__ __launch_bounds__(256) void gn_fold_se_kernel(const float *__restrict__ stats, int C, int T, int G, float count,
                                                         const float *__restrict__ gamma, const float *__restrict__ beta,
                                                         const float *__restrict__ fac, const float *__restrict__ gbias,
                                                         int ld_fg, float eps, const float *__restrict__ w1,
                                                         const float *__restrict__ w2, int H, float *__restrict__ A,
                                                         float *__restrict__ Bs) {
  __shared__ double cs[256][2];
  __shared__ float sA[256], sB[256], sm[256], sh[128];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cpg = C / G;
  int cp2 = 1;
  while (cp2 < C) cp2 <<= 1;
  const int LPC = 256 / cp2, ch = tid / LPC, sub = tid % LPC; // LPC in {1, ..., 64}
  double s1 = 0.0, s2 = 0.0;
  if (ch < C) {
    const float2 *p = reinterpret_cast<const float2 *>(stats + (((size_t)b * C + ch) * T) * 2);
    for (int t = sub; t < T; t += LPC) { const float2 v = p[t]; s1 += (double)v.x; s2 += (double)v.y; }
  }
  for (int m = 1; m < LPC; m <<= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
  if (ch < C && sub == 0) { cs[ch][0] = s1; cs[ch][1] = s2; }
  __syncthreads();
  if (tid < C) {
    const int g0 = (tid / cpg) * cpg;
    double g1 = 0.0, g2 = 0.0;
    for (int k = 0; k < cpg; ++k) { g1 += cs[g0 + k][0]; g2 += cs[g0 + k][1]; }
    const double n = (double)count * cpg, mean = g1 / n;
    double var = g2 / n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float f = fac[(size_t)b * ld_fg + tid], gb = gbias[(size_t)b * ld_fg + tid];
    const float a0 = rstd * gamma[tid];
    const float a = a0 * f, bb = (beta[tid] - (float)mean * a0) * f + gb;
    sA[tid] = a;
    sB[tid] = bb;
    sm[tid] = a * (float)(cs[tid][0] / count) + bb; // mean over the grid of AdaGN(y) (se_gate_kernel)
  }
  __syncthreads();
  for (int j = wave; j < H; j += 4) { // one wave per hidden unit: coalesced row of W1
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc += w1[(size_t)j * C + c] * sm[c];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) sh[j] = acc > 0.f ? acc : 0.f;
  }
  __syncthreads();
  if (tid < C) {
    float acc = 0.f;
    for (int j = 0; j < H; ++j) acc += w2[(size_t)tid * H + j] * sh[j];
    const float g = 1.0f / (1.0f + expf(-acc));
    A[(size_t)b * C + tid] = sA[tid] * g;
    Bs[(size_t)b * C + tid] = sB[tid] * g;
  }
}

// One workgroup per sample: (1) which voxels of each (d, h) row of the count grid hold a point (one bit per w: tiles span
// the whole w axis and r <= 32), (2) per spatial tile: any point within `margin` voxels of the tile, for margin 1 (the conv
// that reads the voxelised grid) and margin 2 (the delta of the second conv), (3) the sample's work list for each margin --
// occupied tiles first, empty ones after -- and the reset of the queue counter.  No global atomics, no host memset,
// deterministic order.  (Round 5's per-tile active-voxel bit maps and per-sample counts fed the voxel-compaction experiment,
// which was not adopted: tools/exp/conv3d_split_round5_experiments.hip; the shipped kernels never read them -- removed.)
// occ_m = [B*tiles flags][B*tiles list: tile ids, one segment per sample][queue counter, exit counter, mode, pad].
__global__ __launch_bounds__(1024) void conv_tile_occ_kernel(const int32_t *__restrict__ cnt, int r, int TD, int TH,
                                                             int32_t *__restrict__ occ1, int32_t *__restrict__ occ2,
                                                             int aware) {
  __shared__ unsigned col[32 * 32];   // bit w of col[d * r + h]: voxel (d, h, w) holds a point
  __shared__ int flag[2][256];
  const int b = blockIdx.x, B = gridDim.x, tid = threadIdx.x;
  const int nth = r / TH, ntiles = (r / TD) * nth, total = B * ntiles;
  for (int i = tid; i < r * r; i += 1024) col[i] = 0u;
  __syncthreads();
  const int4 *c4 = reinterpret_cast<const int4 *>(cnt + (size_t)b * r * r * r);
  for (int i = tid; i < (r * r * r) >> 2; i += 1024) {
    const int4 v = c4[i];
    if (v.x | v.y | v.z | v.w) { // r % 4 == 0: the four voxels lie in one row
      const unsigned m4 = (v.x ? 1u : 0u) | (v.y ? 2u : 0u) | (v.z ? 4u : 0u) | (v.w ? 8u : 0u);
      atomicOr(&col[(i << 2) / r], m4 << ((i << 2) % r)); // LDS integer atomic: the result does not depend on the order
    }
  }
  __syncthreads();
  // per tile and margin: a 4-bit mask, bit w = wave w's 64-voxel block (64 / r rows of one d-plane) has a point within
  // `margin`; 0 = the whole tile is empty
  const int RW = 64 / r; // rows per wave (tiles are r voxels wide)
  for (int e = tid; e < 2 * ntiles; e += 1024) {
    const int t = e % ntiles, margin = 1 + e / ntiles;
    const int dt = (t / nth) * TD, ht = (t % nth) * TH;
    int mask = 0;
    for (int w = 0; w < 4; ++w) {
      const int dl = dt + (w * RW) / TH, hl = ht + (w * RW) % TH;
      unsigned any = 0u;
      for (int d = max(dl - margin, 0); d <= min(dl + margin, r - 1); ++d)
        for (int h = max(hl - margin, 0); h <= min(hl + RW - 1 + margin, r - 1); ++h) any |= col[d * r + h];
      mask |= (any ? 1 : 0) << w;
    }
    flag[margin - 1][t] = mask;
  }
  __syncthreads();
  // the lists, occupied tiles first, both parts in ascending tile order: thread (margin, tile) finds its slot from the
  // ballots of the (<= 4) waves of its margin -- one serial loop over the tiles per margin took 10 of this kernel's 20 us
  // (round 3: with the step on one stream every microsecond of it is on the critical path, 7 launches per step)
  __shared__ unsigned long long bal[2][4];
  const int m = tid >> 8, t = tid & 255, lane = tid & 63, w = (tid >> 6) & 3; // threads 0-255: margin 1, 256-511: margin 2
  const bool live = tid < 512 && t < ntiles;
  const int f = live ? flag[m][t] : 0;
  const unsigned long long bl = __ballot(f != 0);
  if (tid < 512 && lane == 0) bal[m][w] = bl;
  __syncthreads();
  if (live) {
    int32_t *occ = m == 0 ? occ1 : occ2;
    if (occ) {
      int before = __popcll(bal[m][w] & ((1ull << lane) - 1ull)), occupied = 0;
      for (int ww = 0; ww < 4; ++ww) {
        const int c = __popcll(bal[m][ww]);
        if (ww < w) before += c;
        occupied += c;
      }
      int32_t *fl = occ + (size_t)b * ntiles, *list = occ + total + (size_t)b * ntiles;
      // (5) round 5 -- who READS a tile's output?  The convolution on the voxelised grid feeds the delta convolution, which
      // stages the halo of every tile with a point within 2 voxels: its output is read in those tiles and their 8
      // neighbours in the (d, h) tile grid (tiles span w), nowhere else.  The delta convolution feeds the devoxelisation,
      // which reads the 8 voxels around each point: inside the tiles with a point within 2 voxels (within 1, even), nowhere
      // else.  Bit 8 of a tile's flag = its output has a reader; a consumer-aware launch (word [2 total + 2] != 0) stores
      // nothing for an EMPTY tile without that bit -- it still contributes its GroupNorm sums.
      int need = flag[1][t] != 0;
      if (m == 0) {
        if (aware == 2) {
          // level 2: the delta convolution does not read the first one's output inside its EMPTY tiles either -- there it is
          // bias1 exactly, the activated input minus its constant exactly zero, and the split kernel stages zeros for such
          // rows without loading them (bit 9 of the margin-2 words below tells it): only occupied tiles are ever stored
          need = 0;
        } else {
          const int td = t / nth, th = t % nth;
          for (int dd = max(td - 1, 0); dd <= min(td + 1, r / TD - 1); ++dd)
            for (int hh = max(th - 1, 0); hh <= min(th + 1, nth - 1); ++hh) need |= flag[1][dd * nth + hh] != 0;
        }
      }
      // margin-2 words also carry bit 9 = the tile is occupied at margin 1 (see above)
      fl[t] = f | (need << 8) | ((m == 1 && flag[0][t] != 0) ? (1 << 9) : 0);
      list[f ? before : occupied + (t - before)] = t;
      if (b == 0 && t == 0) { occ[2 * total] = 0; occ[2 * total + 1] = 0; occ[2 * total + 2] = aware; } // queue, exit counter, mode
    }
  }
}

// tconst[b][cfg][co] = bias2[co] + sum_ci wsum[cfg][ci][co] * swish(A[b,ci] * bias1[ci] + Bs[b,ci]): the response of the
// second convolution to the constant field its input takes away from the points (double accumulation).
__global__ __launch_bounds__(256) void conv_tconst_kernel(const float *__restrict__ wsum, const float *__restrict__ bias2,
                                                          const float *__restrict__ bias1, const float *__restrict__ pro_a,
                                                          const float *__restrict__ pro_b, int Cin, int Cout,
                                                          float *__restrict__ tconst) {
  __shared__ float sc[256];
  const int cfg = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  for (int c = tid; c < Cin; c += 256)
    sc[c] = pro_act(bias1 ? bias1[c] : 0.f, pro_a[(size_t)b * Cin + c], pro_b[(size_t)b * Cin + c]);
  __syncthreads();
  for (int co = tid; co < Cout; co += 256) {
    double acc = bias2 ? (double)bias2[co] : 0.0;
    const float *w = wsum + (size_t)cfg * Cin * Cout + co;
    for (int c = 0; c < Cin; ++c) acc += (double)w[(size_t)c * Cout] * (double)sc[c];
    tconst[((size_t)b * 27 + cfg) * Cout + co] = (float)acc;
  }
}

// [Cout][Cin][27] (PyTorch) -> [Cin_pad][27][Cout], Cin padded with zeros to a multiple of KC
__global__ void conv3d_pack_kernel(const float *__restrict__ w, int Cout, int Cin, int Cin_pad,
                                   float *__restrict__ wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = Cin_pad * 27 * Cout;
  if (i >= total) return;
  const int co = i % Cout, t = (i / Cout) % 27, ci = i / (Cout * 27);
  wp[i] = ci < Cin ? w[((size_t)co * Cin + ci) * 27 + t] : 0.f;
}

template <int TD, int TH, int TW, int COT, int VB>
static int launch_conv_t(const float *x, const float *wp, const float *bias, float *y, int B, int Cin,
                         int Cout, int r, const float *pa, const float *pb, const float *pbias, const float *tconst,
                         float *stats, int32_t *occ, hipStream_t st) {
  const int tiles = (r / TD) * (r / TH) * (r / TW);
  static int cu_count[LION_MAX_DEVICES] = {0};
  int dev = 0;
  if (int e = lion_current_device(&dev)) return e;
  if (!cu_count[dev]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return LION_EINVAL;
    cu_count[dev] = prop.multiProcessorCount;
  }
  const int n_cu = cu_count[dev];
  const long items = (long)B * tiles * (Cout / COT);
  const dim3 grid = occ ? dim3((unsigned)(items < 2L * n_cu ? items : 2L * n_cu)) : dim3(B, tiles, Cout / COT);
  constexpr int NT = 256;
  constexpr int HALO = (TD + 2) * (TH + 2) * (TW + 2);
  constexpr int SWS = ((KC * 27 * COT + 255) / 256) * 256 + 256;
  static_assert(KC * HALO >= 27 * COT, "the response table must fit the input tile buffer");
  const size_t LDS = (size_t)(((KC * HALO + 3) & ~3) + 2 * SWS + COT + (pa ? 3 * ((Cin + 63) & ~63) : 0) +
                              (NT / 64) * COT * 2) * 4;
#define LION_CONV_GO(PRO_, ST_)                                                                           \
  {                                                                                                       \
    static LionLdsLimit cfg = {};                                                                         \
    if (int e = lion_dynamic_lds(&conv3d_k3_kernel<TD, TH, TW, COT, VB, PRO_, ST_>, LDS, cfg)) return e;  \
    conv3d_k3_kernel<TD, TH, TW, COT, VB, PRO_, ST_><<<grid, NT, LDS, st>>>(x, wp, bias, y, Cin, Cout, r, \
                                                                            pa, pb, pbias, tconst, stats, occ, B, \
                                                                            tiles);                      \
  }
  if (pa && stats) LION_CONV_GO(true, true)
  else if (pa) LION_CONV_GO(true, false)
  else if (stats) LION_CONV_GO(false, true)
  else LION_CONV_GO(false, false)
#undef LION_CONV_GO
  LION_LAUNCH_CHECK();
  return 0;
}

// Tile choice per (r, Cout): spatial tile = 4 waves x VB x 32 voxels, COT output channels per workgroup.
//   Cout % 64 == 0, >= 512 workgroups : VB 4 x COT 64  (4x2 MFMA tiles per wave: 6 LDS reads per 8 MFMAs, 2.4x halo
//                                                        instead of 3.2x -> less staging / prologue per MFMA)
//   Cout % 64 == 0 otherwise          : VB 2 x COT 64  (2x2 tiles, 4 LDS reads per 4 MFMAs)
//   Cout % 32 == 0 at r >= 16         : VB 4 x COT 32  (4x1 tiles: same MFMAs per step, less halo per voxel)
//   small grids (r = 8)               : VB 1 x COT 32  (twice the waves: 2 per SIMD instead of 1, so that one
//                                                        wave's staging / barriers hide under the other's MFMAs)
struct ConvPlan { int vb, cot, tiles; };
// sparse (work-queue) launches take the smaller 2x2 tiles: more of them are empty, the tail is finer grained, and the
// occupied ones spread over all CUs even when they are fewer than the resident workgroups.
static ConvPlan conv_plan(int r, int Cout, int B, bool sparse) {
  const int r3 = r * r * r;
  if (!sparse && Cout % 64 == 0 && r >= 16 && (long)(r3 / 512) * (Cout / 64) * B >= 512) return {4, 64, r3 / 512};
  if (Cout % 64 == 0 && (long)(r3 / 256) * (Cout / 64) * B >= 512) return {2, 64, r3 / 256};
  if (Cout % 32 == 0 && r >= 16 && !sparse) return {4, 32, r3 / 512};
  if (Cout % 32 == 0 && r >= 16) return {2, 32, r3 / 256};
  if (Cout % 32 == 0) return {1, 32, r3 / 128};
  return {0, 0, 0};
}

static void conv_tile_dims(int r, int vb, int *td, int *th, int *tw) {
  *tw = r;
  if (r == 32) { *td = vb == 2 ? 2 : 4; *th = 4; }
  else if (r == 16) { *td = vb == 2 ? 4 : 8; *th = 4; }
  else { *td = vb == 2 ? 4 : 2; *th = 8; }
}

static int launch_conv(const float *x, const float *wp, const float *bias, float *y, int B, int Cin, int Cout,
                       int r, const float *pa, const float *pb, const float *pbias, const float *tconst, float *stats,
                       int32_t *occ, hipStream_t st) {
  const ConvPlan p = conv_plan(r, Cout, B, occ != nullptr);
#define LION_CONV_TILE(R_, VB_, COT_, TD_, TH_, TW_)                                                       \
  if (r == R_ && p.vb == VB_ && p.cot == COT_)                                                             \
    return launch_conv_t<TD_, TH_, TW_, COT_, VB_>(x, wp, bias, y, B, Cin, Cout, r, pa, pb, pbias, tconst, stats, occ, st);
  LION_CONV_TILE(32, 2, 64, 2, 4, 32)
  LION_CONV_TILE(32, 4, 64, 4, 4, 32)
  LION_CONV_TILE(16, 4, 64, 8, 4, 16)
  LION_CONV_TILE(32, 4, 32, 4, 4, 32)
  LION_CONV_TILE(32, 2, 32, 2, 4, 32)
  LION_CONV_TILE(16, 2, 32, 4, 4, 16)
  LION_CONV_TILE(16, 2, 64, 4, 4, 16)
  LION_CONV_TILE(16, 4, 32, 8, 4, 16)
  LION_CONV_TILE(8, 2, 64, 4, 8, 8)
  LION_CONV_TILE(8, 1, 32, 2, 8, 8)
#undef LION_CONV_TILE
  return LION_EUNSUPPORTED;
}

} // namespace

// see common.h: the sparse plans of every channel class share one spatial tile per resolution (vb = 2); a plan change that
// breaks that must be met here, not by silently stale flags in another file
int lion_internal_sparse_tile_dims(int r, int *td, int *th) {
  if (r != 16 && r != 32) return LION_EUNSUPPORTED;
  const ConvPlan a = conv_plan(r, 64, 32, true), b = conv_plan(r, 32, 32, true), c = conv_plan(r, 64, 1, true);
  if (!a.vb || a.vb != b.vb || a.vb != c.vb) return LION_EUNSUPPORTED;
  int tw;
  conv_tile_dims(r, a.vb, td, th, &tw);
  return 0;
}

extern "C" {

// packed size in floats: ceil(Cin / 4) * 4 * 27 * Cout
size_t lion_conv3d_packed_floats(int Cout, int Cin) {
  return (size_t)((Cin + KC - 1) / KC * KC) * 27 * Cout;
}

int lion_conv3d_pack_weights(const float *w, int Cout, int Cin, float *wp, lionStream_t stream) {
  if (!w || !wp || Cout <= 0 || Cin <= 0) return LION_EINVAL;
  const int Cin_pad = (Cin + KC - 1) / KC * KC;
  const int total = Cin_pad * 27 * Cout;
  conv3d_pack_kernel<<<lion_cdiv(total, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(w, Cout, Cin, Cin_pad, wp);
  LION_LAUNCH_CHECK();
  return 0;
}

// x f32[B,Cin,r,r,r] with Cin % 4 == 0, wp from lion_conv3d_pack_weights, bias f32[Cout] or NULL
// -> y f32[B,Cout,r,r,r].   r in {8, 16, 32}.
// pro_a / pro_b f32[B,Cin] (both or neither): input is swish(x*a+b) (fused AdaGN + Swish prologue).
// stats f32[B,Cout,lion_conv3d_stat_tiles(r,Cout,B,sparse),2] or NULL: per-tile channel sums of the output.
// occ (lion_conv3d_tile_occupancy, consumed by this call; only without the prologue) or NULL: all-zero input tiles are
// skipped and the work is balanced through a queue.
int lion_conv3d_k3_fused_forward(const float *x, const float *wp, const float *bias, int B, int Cin,
                                 int Cout, int r, const float *pro_a, const float *pro_b, const float *pro_bias,
                                 const float *tconst, float *y, float *stats, int32_t *occ, lionStream_t stream) {
  if (!x || !wp || !y || B <= 0 || Cin <= 0 || Cout <= 0) return LION_EINVAL;
  if ((pro_a == nullptr) != (pro_b == nullptr)) return LION_EINVAL;
  if (tconst && !pro_a) return LION_EINVAL;
  if (occ && pro_a && !tconst) return LION_EINVAL; // an activated input is sparse only as constant + delta
  if (Cin % KC != 0 || (pro_a && Cin > 256)) return LION_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (r != 8 && r != 16 && r != 32) return LION_EUNSUPPORTED;
  return launch_conv(x, wp, bias, y, B, Cin, Cout, r, pro_a, pro_b, pro_bias, tconst, stats, occ, st);
}

int lion_conv3d_k3_forward(const float *x, const float *wp, const float *bias, int B, int Cin,
                           int Cout, int r, float *y, lionStream_t stream) {
  return lion_conv3d_k3_fused_forward(x, wp, bias, B, Cin, Cout, r, nullptr, nullptr, nullptr, nullptr, y, nullptr,
                                      nullptr, stream);
}

// occ i32[lion_conv3d_occupancy_ints(r,Cout,B)] = [B*tiles flags][B*tiles work list][queue counter, exit counter, mode, pad],
// tiles = lion_conv3d_stat_tiles(r,Cout,B,sparse): flag bits 0-3 = wave w's 64-voxel block of the tile has a point within the
// margin (cnt i32[B,r^3] from the voxelisation), bit 8 = the tile's output has a reader, bit 9 (margin-2 words) = occupied at
// margin 1.  Feed to lion_conv3d_k3_fused_forward (the convolution that reads the voxelised grid): the call pops the
// work queue and re-arms it on exit.
// wsum f32[27][Cin][Cout]: weights summed over the taps that stay inside the grid for each border configuration
// cfg = (cd*3 + ch)*3 + cw (0 low face, 1 interior, 2 high face per axis); bias2 f32[Cout] or NULL (this conv),
// bias1 f32[Cin] or NULL (the conv that produced the input), pro_a/pro_b f32[B,Cin] -> tconst f32[B][27][Cout].
int lion_conv3d_const_response(const float *wsum, const float *bias2, const float *bias1, const float *pro_a,
                               const float *pro_b, int B, int Cin, int Cout, float *tconst, lionStream_t stream) {
  if (!wsum || !pro_a || !pro_b || !tconst || B <= 0 || Cin <= 0 || Cout <= 0) return LION_EINVAL;
  if (Cin > 256) return LION_EUNSUPPORTED;
  conv_tconst_kernel<<<dim3(27, B), 256, 0, static_cast<hipStream_t>(stream)>>>(wsum, bias2, bias1, pro_a, pro_b, Cin,
                                                                                 Cout, tconst);
  LION_LAUNCH_CHECK();
  return 0;
}

size_t lion_conv3d_occupancy_ints(int r, int Cout, int B) {
  if (r != 16 && r != 32) return 0;
  // [flags][list][queue counter, exit counter, mode, pad]
  return (size_t)2 * B * conv_plan(r, Cout, B, true).tiles + 4;
}

// occ_m1 / occ_m2 (either may be NULL): the occupancy + work list for margin 1 (conv on the voxelised grid) and
// margin 2 (delta mode of the following conv); a fused forward re-arms the queue it popped (any number of launches in stream order).
int lion_conv3d_tile_occupancy(const int32_t *cnt, int B, int r, int Cout, int32_t *occ_m1, int32_t *occ_m2,
                               lionStream_t stream) {
  return lion_conv3d_tile_occupancy_aware(cnt, B, r, Cout, occ_m1, occ_m2, 0, stream);
}

// consumer_aware != 0: the convolutions that pop these buffers leave the output of EMPTY tiles that nobody reads unwritten
// (see conv_tile_occ_kernel (5)): only for the pair (conv on the voxelised grid -> delta conv -> devoxelisation) of a PVConv.
// consumer_aware == 2: additionally the delta convolution is the SPLIT kernel, which does not load the first convolution's
// output inside that convolution's empty tiles -- the first convolution then stores occupied tiles only.
int lion_conv3d_tile_occupancy_aware(const int32_t *cnt, int B, int r, int Cout, int32_t *occ_m1, int32_t *occ_m2,
                                     int consumer_aware, lionStream_t stream) {
  if (!cnt || (!occ_m1 && !occ_m2) || B <= 0 || Cout <= 0) return LION_EINVAL;
  if (r != 16 && r != 32) return LION_EUNSUPPORTED;
  const ConvPlan p = conv_plan(r, Cout, B, true);
  if (!p.vb || p.tiles > 256 || (((uintptr_t)cnt) & 15) != 0) return LION_EUNSUPPORTED;
  int td, th, tw;
  conv_tile_dims(r, p.vb, &td, &th, &tw);
  // the reader map of the first buffer is derived from the second margin's flags: both are computed in any case
  conv_tile_occ_kernel<<<B, 1024, 0, static_cast<hipStream_t>(stream)>>>(cnt, r, td, th, occ_m1, occ_m2,
                                                                         consumer_aware == 2 ? 2 : consumer_aware ? 1 : 0);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_conv3d_stat_tiles(int r, int Cout, int B, int sparse) {
  if (r != 8 && r != 16 && r != 32) return 0;
  return conv_plan(r, Cout, B, sparse != 0).tiles;
}

// lion_groupnorm_fold followed by lion_se_gate (w1 f32[H,C], w2 f32[C,H]) in one launch: A, Bs already carry the gate
int lion_groupnorm_fold_se(const float *stats, int B, int C, int T, int G, int voxels, const float *gamma,
                           const float *beta, const float *fac, const float *gbias, int ld_fg, float eps, const float *w1,
                           const float *w2, int H, float *A, float *Bs, lionStream_t stream) {
  if (!stats || !gamma || !beta || !fac || !gbias || !w1 || !w2 || !A || !Bs) return LION_EINVAL;
  if (B <= 0 || C <= 0 || T <= 0 || G <= 0 || C % G != 0 || ld_fg < C || H <= 0) return LION_EINVAL;
  if (C > 256 || C < 4 || H > 128) return LION_EUNSUPPORTED;
  gn_fold_se_kernel<<<B, 256, 0, static_cast<hipStream_t>(stream)>>>(stats, C, T, G, (float)voxels, gamma, beta, fac, gbias,
                                                                     ld_fg, eps, w1, w2, H, A, Bs);
  LION_LAUNCH_CHECK();
  return 0;
}

// stats f32[B,C,T,2] -> A, Bs, chmean f32[B,C]   (GroupNorm(G) folded with the AdaGN affine fac/gbias f32[B,C])
int lion_groupnorm_fold(const float *stats, int B, int C, int T, int G, int voxels, const float *gamma,
                        const float *beta, const float *fac, const float *gbias, int ld_fg, float eps,
                        float *A, float *Bs, float *chmean, lionStream_t stream) {
  if (!stats || !gamma || !beta || !fac || !gbias || !A || !Bs || !chmean) return LION_EINVAL;
  if (B <= 0 || C <= 0 || T <= 0 || G <= 0 || C % G != 0 || C / G > 64 || ld_fg < C) return LION_EINVAL;
  gn_fold_kernel<<<dim3(G, B), 256, 0, static_cast<hipStream_t>(stream)>>>(stats, C, T, G, (float)voxels, gamma,
                                                                         beta, fac, gbias, ld_fg, eps, A, Bs, chmean);
  LION_LAUNCH_CHECK();
  return 0;
}

} // extern "C"
