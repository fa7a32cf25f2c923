// conv3d_wgrad.hip -- weight gradient of the 3x3x3 / pad 1 / stride 1 Conv3d (training path of C3,
// models/pvcnn2_ada.py:211-222; the reference reaches cuDNN's backward-filter through autograd).
//
//   gw[co][ci][tap] = sum_{b, v} gy[b][co][v] * xpad[b][ci][v + tap]          (xpad: zero padded input)
//
// As a GEMM this is M = Cout, N = Cin * 27, K = B * r^3 (one million for 32 x 32^3): K is the long axis.  On
// v_mfma_f32_32x32x2_f32 the voxels therefore sit on the MFMA k axis: A = gy (32 output channels x 2 voxels, from an
// LDS tile [co][voxel] with an odd row stride -> conflict free), B = the haloed input tile shifted by the tap of the
// column (2 voxels x 32 (ci, tap) columns; the column order ci*27 + tap is exactly the memory order of a gw row, so
// the accumulator tile is written straight into the [Cout][Cin][27] layout).  A workgroup owns 32 output channels x
// CIT input channels (CIT*27 columns = 7 column blocks at CIT = 8) and walks the spatial tiles of ONE sample; its 4
// waves split each 256-voxel tile (combined through LDS in fixed order at the end), so there are B * splits partial
// results per (co, ci) tile, summed in a fixed order by a second kernel (deterministic, no atomics).  The next tile's
// loads travel through registers under the current tile's MFMAs.  Exact fp32 products, like the forward.
#include "split_ops.h"

namespace {

template <int TD, int TH, int TW, int CIT>
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                              int Cin, int Cout, int r, int TS,
                                                              float *__restrict__ partial) {
  constexpr int TV = TD * TH * TW; // 256 voxels per tile
  static_assert(TV == 256, "4 waves x 64 voxels");
  constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2, HALO = HD * HH * HW;
  constexpr int NCOL = CIT * 27, NB = (NCOL + 31) / 32;
  constexpr int GS = TV + 1; // row stride of the gy tile: odd -> the 32 channels of an A operand hit 32 banks
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *sgy = smem;            // [32][GS]
  float *sx = sgy + 32 * GS;    // [CIT][HALO]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / TS, ts = blockIdx.x % TS, cit = blockIdx.y, cot = blockIdx.z; // ts: every TS-th tile
  const int ci0 = cit * CIT, co0 = cot * 32;
  const int r2 = r * r, r3 = r2 * r;
  const int ntw = r / TW, nth = r / TH, ntiles = (r / TD) * nth * ntw;
  const int cl = lane & 31, kh = lane >> 5;

  // per-lane column constants: column c = nb*32 + cl -> (ci, tap) -> LDS offset of the shifted input
  int cofs[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int c = min(nb * 32 + cl, NCOL - 1); // padded columns recompute the last one; never stored
    const int ci = c / 27, tap = c - ci * 27;
    cofs[nb] = ci * HALO + ((tap / 9) * HH + (tap / 3) % 3) * HW + tap % 3;
  }
  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(x + ((size_t)b * Cin + ci0) * r3), 0, CIT * r3 * 4, 0x00020000);
  const float *gyb = gy + ((size_t)b * Cout + co0) * r3;

  // Staging through registers: the loads of tile t + TS are issued in front of the MFMAs of tile t (round 3; the kernel was
  // load -> barrier -> compute -> barrier with only the co-resident workgroup to hide the loads behind).
  constexpr int NXI = (HALO + 255) / 256; // halo positions per thread
  float rgy[32], rx[NXI][CIT];
  auto load_tile = [&](int t) {
    const int tw_i = t % ntw, th_i = (t / ntw) % nth, td_i = t / (ntw * nth);
    const int d0 = td_i * TD, h0 = th_i * TH, w0 = tw_i * TW;
    {
      const int v = tid, d = v / (TH * TW), h = (v / TW) % TH, w = v % TW;
      const int gv = ((d0 + d) * r + (h0 + h)) * r + (w0 + w);
#pragma unroll
      for (int c = 0; c < 32; ++c) rgy[c] = gyb[(size_t)c * r3 + gv];
    }
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int p = tid + 256 * i;
      const int hd = p / (HH * HW), hh = (p / HW) % HH, hw = p % HW;
      const int gd = d0 - 1 + hd, gh = h0 - 1 + hh, gw = w0 - 1 + hw;
      const bool ok = p < HALO && gd >= 0 && gd < r && gh >= 0 && gh < r && gw >= 0 && gw < r;
      const int off = ok ? ((gd * r + gh) * r + gw) * 4 : 0x7fffff00; // zero padding through out-of-range buffer offsets
#pragma unroll
      for (int c = 0; c < CIT; ++c)
        rx[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, off, c * r3 * 4, 0));
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int c = 0; c < 32; ++c) sgy[c * GS + tid] = rgy[c];
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int p = tid + 256 * i;
      if (p < HALO) {
#pragma unroll
        for (int c = 0; c < CIT; ++c) sx[c * HALO + p] = rx[i][c];
      }
    }
  };
  if (ts < ntiles) load_tile(ts);
  for (int t = ts; t < ntiles; t += TS) {
    __syncthreads(); // the previous tile's LDS reads are done
    store_tile();
    __syncthreads();
    if (t + TS < ntiles) load_tile(t + TS); // in flight under the MFMAs below
    // 32 k-steps of 2 voxels over this wave's 64 voxels
#pragma unroll 4
    for (int s = 0; s < 32; ++s) {
      const int v = wave * 64 + 2 * s + kh;
      const int d = v / (TH * TW), h = (v / TW) % TH, w = v % TW;
      const int vb = (d * HH + h) * HW + w;
      const float a = sgy[cl * GS + v];
      float bq[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bq[nb] = sx[cofs[nb] + vb];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[nb], acc[nb], 0, 0, 0);
    }
  }
  // The four waves hold partial sums over disjoint voxels of the same (co, column) tile: combine them through LDS in fixed
  // order (w0 + w1) + (w2 + w3), one column block at a time, and write ONE partial per workgroup (round 3: the reduce kernel
  // read 4x the bytes).  partial[blockIdx.x][co][ci*27 + tap]; acc register i of lane l: row (i&3) + 8*(i>>2) + 4*(l>>5),
  // column l&31
  float *pt = partial + (size_t)blockIdx.x * Cout * Cin * 27;
  float *red = smem; // [4 waves][16][64]
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[nb][i];
    __syncthreads();
    // 1024 values per column block, 4 per thread: value e = i * 64 + lane'
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = tid + 256 * k, i = e >> 6, ln = e & 63;
      const float v = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
      const int c = nb * 32 + (ln & 31);
      if (c < NCOL) {
        const int co = co0 + (i & 3) + 8 * (i >> 2) + 4 * (ln >> 5);
        pt[((size_t)co * Cin + ci0) * 27 + c] = v;
      }
    }
  }
}

// ---- round 4: the same weight gradient on the 16-bit matrix pipe at fp32 accuracy ---------------------------------------
// Forward and data gradient moved to fp16 x 2 split operands in round 2 (csrc/conv3d_split.hip); the weight gradient stayed
// on v_mfma_f32_32x32x2_f32 (0.67 of the 157 TF fp32 peak: 29 % of a VAE training step).  Here both operands are cut into
// fp16 pieces IN REGISTERS, from the same fp32 LDS tiles, right in front of v_mfma_f32_32x32x16_f16 (16 voxels per MFMA):
//   g = g_h + g_l,  x = x_h + x_l  (after a power-of-two block scale: per TENSOR until round 5, a running per-WORKGROUP scale
//   found inside the kernel since round 6 -- see the kernel),
//   acc += g_h x_h + g_h x_l + g_l x_h        (the dropped g_l x_l is 2^-22 relative)
// -- 3 MFMAs of 32 cycles per 16 voxels and column block instead of 8 of 64.  Differences to the forward's split: (i) ONE
// running scale per workgroup and operand (round 6; one per tensor before), not per tile and chunk -- the result is a sum over
// ~10^6 voxels, an element 2^-17 below the running maximum loses low bits that are 2^-39 of the largest term; (ii) the low pieces are NOT scaled up by 2048 (they are normal fp16
// down to 2^-3 of the scaled value, subnormal steps below are 2^-38 of the maximum), so main and correction products share
// one accumulator: 112 accumulator registers, as the fp32 kernel.  A fragment = 8 consecutive voxels of a row (TW % 8 == 0):
// gy rows at stride 260 floats (16-byte aligned, conflict-free b128 reads), x windows at the column's tap offset (dword reads).
constexpr int WG_GS = 260;
// Round 6: the cut moves from the fragment to the STAGING.  A staged element is cut once, when its tile is written to LDS,
// into one 32-bit word {low half: the fp16 main piece, high half: the fp16 remainder}; a fragment then gathers the main
// halves of its 8 words into 4 registers and the remainders into 4 more with v_perm_b32 -- 8 VALU per fragment instead of 24.
// An x element used to be cut once per TAP COLUMN that reads it (27 x per tile), 672 of the 825 vector instructions a thread
// spent per tile; now 57.5 staged elements x 4 + 7 x 4 x 8 permutes.  Same conversions (v_cvt_pk_f16_f32, round to nearest
// even), same pieces: the gradient is bit-identical to the round-4 kernel's.
__device__ __forceinline__ void cut2w(float a, float b, unsigned &wa, unsigned &wb) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  typedef float f2_t __attribute__((ext_vector_type(2)));
  const h2_t h = __builtin_convertvector(f2_t{a, b}, h2_t);
  const h2_t l = __builtin_convertvector(f2_t{a - (float)h[0], b - (float)h[1]}, h2_t);
  const unsigned hi2 = __builtin_bit_cast(unsigned, h), lo2 = __builtin_bit_cast(unsigned, l);
  wa = __builtin_amdgcn_perm(lo2, hi2, 0x05040100u);   // {h_a, l_a}
  wb = __builtin_amdgcn_perm(lo2, hi2, 0x07060302u);   // {h_b, l_b}
}
__device__ __forceinline__ unsigned cut1w(float a) {
  const _Float16 h = (_Float16)a;
  const _Float16 l = (_Float16)(a - (float)h);
  return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}
// main halves / remainders of two staged words -> one packed operand register each
__device__ __forceinline__ unsigned mains(unsigned w0, unsigned w1) { return __builtin_amdgcn_perm(w1, w0, 0x05040100u); }
__device__ __forceinline__ unsigned rests(unsigned w0, unsigned w1) { return __builtin_amdgcn_perm(w1, w0, 0x07060302u); }
__device__ __forceinline__ void cut2u(float a, float b, unsigned &hi2, unsigned &lo2) { // split_ops.h::cut2 without the 2048
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  typedef float f2_t __attribute__((ext_vector_type(2)));
  const h2_t h = __builtin_convertvector(f2_t{a, b}, h2_t);
  const h2_t l = __builtin_convertvector(f2_t{a - (float)h[0], b - (float)h[1]}, h2_t);
  hi2 = __builtin_bit_cast(unsigned, h);
  lo2 = __builtin_bit_cast(unsigned, l);
}

template <int TD, int TH, int TW, int CIT>
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_split_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                    int Cin, int Cout, int r, int TS, int units,
                                                                    const unsigned char *__restrict__ skip,
                                                                    float *__restrict__ partial) {
  constexpr int TV = TD * TH * TW; // 256 voxels per tile
  static_assert(TV == 256 && TW % 8 == 0, "4 waves x 64 voxels; fragments of 8 voxels stay inside a row");
  constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2, HALO = HD * HH * HW;
  constexpr int NCOL = CIT * 27, NB = (NCOL + 31) / 32;
  constexpr int GS = WG_GS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned *sgy = reinterpret_cast<unsigned *>(smem);   // [32][GS]  gy * 2^eg, cut: {main, remainder} halves per word
  unsigned *sx = sgy + 32 * GS;                         // [CIT][HALO]  x * 2^ex, cut
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Workgroup ids go round-robin over the 8 XCDs (private L2s).  A gy tile (32 KiB of the 59 staged per tile) is needed by
  // all Cin / CIT workgroups of a (sample split, output-channel tile) unit: those sit on ONE XCD, next to each other in
  // its launch order, and walk the tiles together -- the gy tile comes in from memory once per unit instead of once per
  // workgroup (64 -> 64 @ 32^3: 3.9 GB of staging reads, 3.3 TB/s, were the kernel's bound at 1160 us).
  const int NCI = Cin / CIT; // units = B * TS * (Cout / 32)
  const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
  const int cit = j % NCI, unit = (j / NCI) * 8 + xcd;
  if (unit >= units) return;
  const int nbts = units / (Cout / 32);
  const int bts = unit % nbts, cot = unit / nbts;
  const int b = bts / TS, ts = bts % TS, part = bts;
  const int ci0 = cit * CIT, co0 = cot * 32;
  const int r2 = r * r, r3 = r2 * r;
  const int ntw = r / TW, nth = r / TH, ntiles = (r / TD) * nth * ntw;
  const int cl = lane & 31, kh = lane >> 5;

  int cofs[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int c = min(nb * 32 + cl, NCOL - 1); // padded columns recompute the last one; never stored
    const int ci = c / 27, tap = c - ci * 27;
    cofs[nb] = ci * HALO + ((tap / 9) * HH + (tap / 3) % 3) * HW + tap % 3;
  }
  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(x + ((size_t)b * Cin + ci0) * r3), 0, CIT * r3 * 4, 0x00020000);
  const float *gyb = gy + ((size_t)b * Cout + co0) * r3;

  // Staging through registers, one dword per (thread, channel): 16-byte row loads (the forward kernel's staging) were
  // tried here and are SLOWER (1176 -> 1346 us at 64 -> 64 @ 32^3): this kernel is bound by the VALU of its in-register
  // cuts and by its dword LDS reads, not by the texture-address path.
  constexpr int NXI = (HALO + 255) / 256;
  float rgy[32], rx[NXI][CIT];
  auto load_tile = [&](int t) {
    const int tw_i = t % ntw, th_i = (t / ntw) % nth, td_i = t / (ntw * nth);
    const int d0 = td_i * TD, h0 = th_i * TH, w0 = tw_i * TW;
    {
      const int v = tid, d = v / (TH * TW), h = (v / TW) % TH, w = v % TW;
      const int gv = ((d0 + d) * r + (h0 + h)) * r + (w0 + w);
#pragma unroll
      for (int c = 0; c < 32; ++c) rgy[c] = gyb[(size_t)c * r3 + gv];
    }
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int p = tid + 256 * i;
      const int hd = p / (HH * HW), hh = (p / HW) % HH, hw = p % HW;
      const int gd = d0 - 1 + hd, gh = h0 - 1 + hh, gw = w0 - 1 + hw;
      const bool ok = p < HALO && gd >= 0 && gd < r && gh >= 0 && gh < r && gw >= 0 && gw < r;
      const int off = ok ? ((gd * r + gh) * r + gw) * 4 : 0x7fffff00;
#pragma unroll
      for (int c = 0; c < CIT; ++c)
        rx[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, off, c * r3 * 4, 0));
    }
  };
  // Round 6 -- the block scales are found IN the kernel.  Until now two extra passes over x and gy (absmax_kernel, 26 us each
  // at 268 MB, + a scales kernel and a memset: 3.1 ms of a VAE training step) fixed one power-of-two scale per TENSOR.  A
  // workgroup now keeps running scales 2^Ex, 2^Eg (max |tile| * 2^E in [2^13, 2^14) when chosen, only ever lowered -- the
  // forward kernel's monotone scheme): the maxima of the tile in registers are combined through two LDS words in front of the
  // barrier that was there anyway; when a later tile raises a maximum the accumulators are multiplied by the exact power-of-two
  // ratio first.  Finer than one scale per tensor (every product carries >= 22 bits relative to the largest operand the
  // WORKGROUP has seen), deterministic (no cross-workgroup state), and no launch besides the kernel and its reduction.
  __shared__ unsigned s_m[2][2];   // [parity][x, gy] bits of the tile's max |.| (finite values)
  int Ex = 127, Eg = 127;          // 127 = not chosen yet (scale 1)
  float sxs = 1.f, sgs = 1.f;
  if (tid < 4) s_m[tid >> 1][tid & 1] = 0u;
  auto tile_max = [&](int par) {   // this thread's staged registers -> s_m[par]
    // v_max_f32 with the |.| source modifier: one instruction per staged value (max ignores NaN operands; an infinite maximum
    // is not taken as a scale, below)
    float fg = 0.f, fx = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) fg = fmaxf(fg, fabsf(rgy[c]));
#pragma unroll
    for (int i = 0; i < NXI; ++i)
#pragma unroll
      for (int c = 0; c < CIT; ++c) fx = fmaxf(fx, fabsf(rx[i][c]));
    unsigned mg = __float_as_uint(fg), mx = __float_as_uint(fx);
    mg = mg <= 0x7f7fffffu ? mg : 0u;
    mx = mx <= 0x7f7fffffu ? mx : 0u;
    mg = wave_max_u32_lane63(mg);
    mx = wave_max_u32_lane63(mx);
    if (lane == 63) { if (mx) atomicMax(&s_m[par][0], mx); if (mg) atomicMax(&s_m[par][1], mg); }
  };
  auto store_tile = [&]() { // scaled and cut (two channels per conversion)
#pragma unroll
    for (int c = 0; c < 32; c += 2) {
      unsigned wa, wb;
      cut2w(rgy[c] * sgs, rgy[c + 1] * sgs, wa, wb);
      sgy[c * GS + tid] = wa;
      sgy[(c + 1) * GS + tid] = wb;
    }
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int p = tid + 256 * i;
      if (p < HALO) {
        static_assert(CIT % 2 == 0, "channel pairs");
#pragma unroll
        for (int c = 0; c < CIT; c += 2) {
          unsigned wa, wb;
          cut2w(rx[i][c] * sxs, rx[i][c + 1] * sxs, wa, wb);
          sx[c * HALO + p] = wa;
          sx[(c + 1) * HALO + p] = wb;
        }
      }
    }
  };
  // skip (or NULL): u8 [B][ntiles], 1 = no point within one voxel of the tile (wgrad_tile_skip_kernel, from the voxelisation's
  // counts): such a tile is not even loaded; without the map an all-zero window is still detected below, after its loads
  const unsigned char *skb = skip ? skip + (size_t)b * ntiles : nullptr;
  auto next_tile = [&](int t) {
    if (skb) while (t < ntiles && skb[t]) t += TS;
    return t;
  };
  int t = next_tile(ts);
  if (t < ntiles) load_tile(t);
  __syncthreads();                 // s_m zeroed
  int par = 0;
  for (int tn; t < ntiles; t = tn, par ^= 1) {
    tn = next_tile(t + TS);
    tile_max(par);
    __syncthreads();               // the tile's maxima are complete; the previous tile's LDS reads are done
    // A tile whose input window (halo included) is all zero adds nothing: no cut, no LDS stores, no MFMAs -- only its loads,
    // which were in flight already.  The first convolution of every PVConv reads a freshly voxelized grid (2048 points in 32^3
    // voxels: most 256-voxel tiles are empty); the tile's maximum is known here anyway.
    const bool empty = s_m[par][0] == 0u;   // uniform over the workgroup
    if (!empty) {
      const unsigned bx = s_m[par][0], bg = s_m[par][1];
      float f = 1.f;               // what the accumulated sums have to be multiplied by (<= 1, exact)
      if (bx) { const int e = scale_exp(__uint_as_float(bx)); if (e < Ex) { if (Ex != 127) f *= pow2f(max(e - Ex, -126)); Ex = e; sxs = pow2f(e); } }
      if (bg) { const int e = scale_exp(__uint_as_float(bg)); if (e < Eg) { if (Eg != 127) f *= pow2f(max(e - Eg, -126)); Eg = e; sgs = pow2f(e); } }
      if (f != 1.f) {              // (uniform over the workgroup; rare: a later tile raised a maximum)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[nb][i] *= f;
      }
    }
    if (!empty) store_tile();
    if (tid < 2) s_m[par ^ 1][tid] = 0u;   // the other parity: its last readers passed the barrier above
    __syncthreads();
    if (tn < ntiles) load_tile(tn);
    // 4 k-steps of 16 voxels over this wave's 64 voxels; a lane's fragment = voxels v0 .. v0 + 7 of one row
#pragma unroll 1
    for (int s = empty ? 4 : 0; s < 4; ++s) {
      const int v0 = wave * 64 + 16 * s + 8 * kh;
      const int d = v0 / (TH * TW), h = (v0 / TW) % TH, w = v0 % TW;
      const int vb = (d * HH + h) * HW + w;
      const u4 g0 = *reinterpret_cast<const u4 *>(sgy + cl * GS + v0);
      const u4 g1 = *reinterpret_cast<const u4 *>(sgy + cl * GS + v0 + 4);
      u4 ah, al;
      ah[0] = mains(g0[0], g0[1]); al[0] = rests(g0[0], g0[1]);
      ah[1] = mains(g0[2], g0[3]); al[1] = rests(g0[2], g0[3]);
      ah[2] = mains(g1[0], g1[1]); al[2] = rests(g1[0], g1[1]);
      ah[3] = mains(g1[2], g1[3]); al[3] = rests(g1[2], g1[3]);
      // column blocks in two groups (4 + the rest): the fragments of a group live in registers while its 3 x G MFMAs run,
      // and an accumulator is touched again G MFMAs later
      constexpr int G0 = NB < 4 ? NB : 4;
#pragma unroll
      for (int n0 = 0; n0 < NB; n0 += G0) {
        u4 bh[G0], bl[G0];
#pragma unroll
        for (int q = 0; q < G0; ++q) {
          const int nb = min(n0 + q, NB - 1);
          const unsigned *xp = sx + cofs[nb] + vb;
          unsigned xv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) xv[j] = xp[j];
#pragma unroll
          for (int m = 0; m < 4; ++m) { bh[q][m] = mains(xv[2 * m], xv[2 * m + 1]); bl[q][m] = rests(xv[2 * m], xv[2 * m + 1]); }
        }
#pragma unroll
        for (int q = 0; q < G0; ++q) if (n0 + q < NB) acc[n0 + q] = mma(ah, bh[q], acc[n0 + q]);
#pragma unroll
        for (int q = 0; q < G0; ++q) if (n0 + q < NB) acc[n0 + q] = mma(ah, bl[q], acc[n0 + q]);
#pragma unroll
        for (int q = 0; q < G0; ++q) if (n0 + q < NB) acc[n0 + q] = mma(al, bh[q], acc[n0 + q]);
      }
    }
  }
  // waves -> one partial per workgroup, unscaled (two exact power-of-two factors: their product may leave fp32's range)
  const float ux = Ex == 127 ? 1.f : pow2f(-Ex), ug = Eg == 127 ? 1.f : pow2f(-Eg);
  float *pt = partial + (size_t)part * Cout * Cin * 27;
  float *red = smem; // [4 waves][16][64]
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[nb][i];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = tid + 256 * k, i = e >> 6, ln = e & 63;
      const float v = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
      const int c = nb * 32 + (ln & 31);
      if (c < NCOL) {
        const int co = co0 + (i & 3) + 8 * (i >> 2) + 4 * (ln >> 5);
        pt[((size_t)co * Cin + ci0) * 27 + c] = (v * ux) * ug;
      }
    }
  }
}

// skip[b][t] = 1 when no voxel within one step of tile t (tiles of TD x TH x r voxels: they span the w axis) holds a point.
// cnt i32[B][r^3] from the voxelisation.  One workgroup per sample: row occupancy bits, then one thread per tile.
__global__ __launch_bounds__(256) void wgrad_tile_skip_kernel(const int32_t *__restrict__ cnt, int r, int TD, int TH,
                                                              unsigned char *__restrict__ skip) {
  __shared__ unsigned rowany[32 * 32];   // (d, h) row holds a point
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < r * r; i += 256) rowany[i] = 0u;
  __syncthreads();
  const int4 *c4 = reinterpret_cast<const int4 *>(cnt + (size_t)b * r * r * r);
  for (int i = tid; i < (r * r * r) >> 2; i += 256) {
    const int4 v = c4[i];
    if (v.x | v.y | v.z | v.w) rowany[(i << 2) / r] = 1u;   // (benign race: every writer stores 1)
  }
  __syncthreads();
  const int nth = r / TH, ntiles = (r / TD) * nth;
  for (int t = tid; t < ntiles; t += 256) {
    const int d0 = (t / nth) * TD, h0 = (t % nth) * TH;
    unsigned any = 0u;
    for (int d = max(d0 - 1, 0); d <= min(d0 + TD, r - 1); ++d)
      for (int h = max(h0 - 1, 0); h <= min(h0 + TH, r - 1); ++h) any |= rowany[d * r + h];
    skip[(size_t)b * ntiles + t] = any ? 0 : 1;
  }
}

__global__ __launch_bounds__(256) void conv3d_wgrad_reduce_kernel(const float *__restrict__ partial, int nparts,
                                                                  size_t n, float *__restrict__ gw) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0.0; // the partials are sums of ~8k products each; their sum (up to 1024 of them) is taken in double
  for (int p = 0; p < nparts; ++p) s += (double)partial[(size_t)p * n + i]; // fixed order
  gw[i] = (float)s;
}

template <int TD, int TH, int TW, int CIT>
static int launch_wgrad(const float *x, const float *gy, int B, int Cin, int Cout, int r, int TS, float *partial,
                        hipStream_t st) {
  constexpr int HALO = (TD + 2) * (TH + 2) * (TW + 2);
  const size_t lds = (size_t)(32 * 257 + CIT * HALO) * 4;
  static LionLdsLimit cfg = {};
  if (int e = lion_dynamic_lds(&conv3d_wgrad_kernel<TD, TH, TW, CIT>, lds, cfg)) return e;
  conv3d_wgrad_kernel<TD, TH, TW, CIT><<<dim3(B * TS, Cin / CIT, Cout / 32), 256, lds, st>>>(x, gy, Cin, Cout, r, TS,
                                                                                           partial);
  LION_LAUNCH_CHECK();
  return 0;
}

template <int TD, int TH, int TW, int CIT>
static int launch_wgrad_split(const float *x, const float *gy, int B, int Cin, int Cout, int r, int TS,
                              const unsigned char *skip, float *partial, hipStream_t st) {
  constexpr int HALO = (TD + 2) * (TH + 2) * (TW + 2);
  size_t lds = (size_t)(32 * WG_GS + CIT * HALO) * 4;
  if (lds < (size_t)4 * 16 * 64 * 4) lds = (size_t)4 * 16 * 64 * 4; // the epilogue's [4][16][64] reduction buffer
  static LionLdsLimit cfg = {};
  if (int e = lion_dynamic_lds(&conv3d_wgrad_split_kernel<TD, TH, TW, CIT>, lds, cfg)) return e;
  const int units = B * TS * (Cout / 32), nci = Cin / CIT;
  conv3d_wgrad_split_kernel<TD, TH, TW, CIT><<<dim3((unsigned)(((units + 7) / 8) * 8 * nci)), 256, lds, st>>>(
      x, gy, Cin, Cout, r, TS, units, skip, partial);
  LION_LAUNCH_CHECK();
  return 0;
}

} // namespace

extern "C" {

// spatial splits per sample: enough workgroups for ~2 per CU when the channel tiles alone are too few
static int wgrad_splits(int B, int Cin, int Cout, int r) {
  const int cit = Cin % 8 == 0 ? 8 : 4;
  const long wg = (long)B * (Cin / cit) * (Cout / 32);
  const int ntiles = r * r * r / 256;
  int ts = 1;
  while (ts < ntiles && ts < 8 && wg * ts < 512) ts *= 2;
  return ts;
}

// floats of scratch: B * splits partial copies of the [Cout,Cin,27] gradient (one per workgroup)
size_t lion_conv3d_wgrad_workspace_floats(int B, int Cin, int Cout, int r) {
  if (Cin % 4 != 0 || Cout % 32 != 0 || (r != 8 && r != 16 && r != 32)) return 0;
  return (size_t)B * wgrad_splits(B, Cin, Cout, r) * Cout * Cin * 27 + 64; // + the split kernel's maxima / scales
}

// x f32[B,Cin,r,r,r] (Cin % 4 == 0), gy f32[B,Cout,r,r,r] (Cout % 32 == 0), r in {8,16,32} -> gw f32[Cout,Cin,3,3,3]
int lion_conv3d_k3_wgrad(const float *x, const float *gy, int B, int Cin, int Cout, int r, float *gw, float *ws,
                         size_t ws_floats, lionStream_t stream) {
  if (!x || !gy || !gw || B <= 0 || Cin <= 0 || Cout <= 0) return LION_EINVAL;
  if (Cin % 4 != 0 || Cout % 32 != 0 || (r != 8 && r != 16 && r != 32)) return LION_EUNSUPPORTED;
  if (!ws || ws_floats < lion_conv3d_wgrad_workspace_floats(B, Cin, Cout, r)) return LION_EWORKSPACE;
  const int TS = wgrad_splits(B, Cin, Cout, r);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool c8 = Cin % 8 == 0;
  int rc;
  if (r == 32) rc = c8 ? launch_wgrad<2, 4, 32, 8>(x, gy, B, Cin, Cout, r, TS, ws, st) : launch_wgrad<2, 4, 32, 4>(x, gy, B, Cin, Cout, r, TS, ws, st);
  else if (r == 16) rc = c8 ? launch_wgrad<4, 4, 16, 8>(x, gy, B, Cin, Cout, r, TS, ws, st) : launch_wgrad<4, 4, 16, 4>(x, gy, B, Cin, Cout, r, TS, ws, st);
  else rc = c8 ? launch_wgrad<4, 8, 8, 8>(x, gy, B, Cin, Cout, r, TS, ws, st) : launch_wgrad<4, 8, 8, 4>(x, gy, B, Cin, Cout, r, TS, ws, st);
  if (rc) return rc;
  const size_t n = (size_t)Cout * Cin * 27;
  conv3d_wgrad_reduce_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(ws, B * TS, n, gw);
  LION_LAUNCH_CHECK();
  return 0;
}

// The same gradient on the 16-bit matrix pipe at fp32 accuracy (conv3d_wgrad_split_kernel above): same arguments and
// workspace; Cin % 8 == 0 (LION_EUNSUPPORTED otherwise: the caller uses lion_conv3d_k3_wgrad).
static int wgrad_split_impl(const float *x, const float *gy, const int32_t *cnt, int B, int Cin, int Cout, int r, float *gw,
                            float *ws, size_t ws_floats, lionStream_t stream) {
  if (!x || !gy || !gw || B <= 0 || Cin <= 0 || Cout <= 0) return LION_EINVAL;
  if (Cin % 8 != 0 || Cout % 32 != 0 || (r != 8 && r != 16 && r != 32)) return LION_EUNSUPPORTED;
  if (!ws || ws_floats < lion_conv3d_wgrad_workspace_floats(B, Cin, Cout, r)) return LION_EWORKSPACE;
  if (((((uintptr_t)x) | ((uintptr_t)gy) | ((uintptr_t)cnt)) & 15) != 0) return LION_EUNSUPPORTED;
  const int TS = wgrad_splits(B, Cin, Cout, r);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // (round 6: the block scales are found inside the kernel; no absmax passes)
  const size_t n = (size_t)Cout * Cin * 27;
  unsigned char *skip = nullptr;
  if (cnt) {   // the tile map lives behind the partials (the workspace's 64 spare floats are not enough: B * ntiles bytes)
    const int td = r == 32 ? 2 : 4, th = r == 8 ? 8 : 4, ntiles = (r / td) * (r / th);
    const size_t used = (size_t)B * TS * n;
    if (ws_floats < used + ((size_t)B * ntiles + 3) / 4) return LION_EWORKSPACE;
    skip = reinterpret_cast<unsigned char *>(ws + used);
    wgrad_tile_skip_kernel<<<B, 256, 0, st>>>(cnt, r, td, th, skip);
  }
  int rc;
  if (r == 32) rc = launch_wgrad_split<2, 4, 32, 8>(x, gy, B, Cin, Cout, r, TS, skip, ws, st);
  else if (r == 16) rc = launch_wgrad_split<4, 4, 16, 8>(x, gy, B, Cin, Cout, r, TS, skip, ws, st);
  else rc = launch_wgrad_split<4, 8, 8, 8>(x, gy, B, Cin, Cout, r, TS, skip, ws, st);
  if (rc) return rc;
  conv3d_wgrad_reduce_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(ws, B * TS, n, gw);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_conv3d_k3_wgrad_split(const float *x, const float *gy, int B, int Cin, int Cout, int r, float *gw, float *ws,
                               size_t ws_floats, lionStream_t stream) {
  return wgrad_split_impl(x, gy, nullptr, B, Cin, Cout, r, gw, ws, ws_floats, stream);
}

// x is a freshly voxelised grid and cnt i32[B, r^3] its per-voxel point counts (16-byte aligned): tiles without a point within one
// voxel add nothing to the gradient and are not loaded.  Workspace: lion_conv3d_wgrad_sparse_workspace_floats.
size_t lion_conv3d_wgrad_sparse_workspace_floats(int B, int Cin, int Cout, int r) {
  const size_t base = lion_conv3d_wgrad_workspace_floats(B, Cin, Cout, r);
  return base ? base + ((size_t)B * 256 + 3) / 4 : 0;   // <= 256 tiles per sample
}

int lion_conv3d_k3_wgrad_split_sparse(const float *x, const float *gy, const int32_t *cnt, int B, int Cin, int Cout, int r,
                                      float *gw, float *ws, size_t ws_floats, lionStream_t stream) {
  if (!cnt) return LION_EINVAL;
  return wgrad_split_impl(x, gy, cnt, B, Cin, Cout, r, gw, ws, ws_floats, stream);
}

} // extern "C"
