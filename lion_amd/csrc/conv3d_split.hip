// conv3d_split.hip -- C3 on the 16-bit matrix pipe at fp32 accuracy: the 3x3x3 / pad 1 Conv3d of PVConv's voxel
// branch (models/pvcnn2_ada.py:211-222) with every fp32 operand cut into two fp16 pieces.
//
// Why: conv3d.hip already runs at 0.86-0.89 of the fp32-input MFMA peak (157 TF), i.e. the fp32 pipe itself is the
// ceiling of shapes/s.  v_mfma_f32_32x32x16_f16 is 16x faster per FLOP, and
//   a = a_h + a_l / 2048,   a_h = fp16(a),  a_l = fp16((a - a_h) * 2048)       (22-23 significant bits)
//   main += W_h * X_h,   corr += W_h * X_l + W_l * X_h      (fp32 accumulation inside the MFMA)
//   D = main + corr / 2048                                   (the dropped W_l * X_l term is 2^-22 relative)
// costs 3 MFMAs of 32 cycles per K = 16 instead of 8 fp32 MFMAs of 64 cycles.  Measured error vs a float64
// convolution: 2.6e-7 rms of the output rms (the fp32 MFMA chain's own: 5e-7 -- it rounds the accumulator 8x more
// often); tests/test_conv_split_gpu.py holds it to the SAME bounds as the fp32 kernel.
//
// Range (fp16 has 5 exponent bits): both operands are block-scaled by exact powers of two.
//   weights: one scale per tensor, chosen at pack time so that max |w| * 2^ew lies in [2^13, 2^14);
//   activations: one scale per (workgroup tile, 16-channel chunk), kept MONOTONE along the K loop: the tile's
//   running maximum (after the fused AdaGN+Swish prologue) sets 2^E with max * 2^E in [2^13, 2^14); when a later
//   chunk raises the maximum the accumulators are multiplied by the (exact) power-of-two ratio first.  Every product
//   therefore carries >= 22 bits relative to the LARGEST operand the tile has seen -- block floating point with a
//   23-bit mantissa; there is no clamp: |x| > 65504, 1e-30 and mixed ranges are all representable; inf / nan are left
//   out of the maximum and propagate to exactly the outputs they reach, as in fp32 arithmetic.
//   The epilogue multiplies by 2^-(E + ew) (exact).
//
// Same contract and modes as conv3d.hip::conv3d_k3_kernel: AdaGN+Swish prologue (PRO), GroupNorm tile sums (STATS),
// persistent work queue + per-wave occupancy masks (occ), constant + delta decomposition (tconst).  K is walked in
// chunks of 16 input channels (Cin % 16 == 0; other layers stay on the fp32 kernel).  LDS operand planes
// [piece][k-half][HP halo positions][8 x fp16]: one ds_read_b128 per MFMA fragment, conflict free; the weight slice
// of a tap [piece][k-half][COT][8 x fp16] goes registers -> LDS in groups of 3 taps, double buffered (one barrier per
// group: 9 per chunk).
// History: tools/exp/split_*.hip (inner product 400 TF fp32-equivalent; whole layer 779 us vs 1973 us of the fp32
// kernel with statistics at B=32, 64->64, r=32).
#include "split_ops.h"

namespace {

// w f32[Cout][Cin][27] -> wp u16[Cin/16][27][piece][k-half][Cout][8]   (ci = chunk*16 + half*8 + j)
// pack: cuts w * 2^ew (split_ops.h: split_wmax_kernel left max |w| in the tail; split_tail_scale completes it).
__global__ void split_pack_kernel(const float *__restrict__ w, int Cout, int Cin, unsigned short *__restrict__ wp,
                                  unsigned *__restrict__ tail) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int ew = split_tail_scale(tail, i == 0);
  if (i >= Cout * Cin * 27) return;
  const int t = i % 27, c = (i / 27) % Cin, co = i / (27 * Cin), chunk = c / KS, g = (c % KS) / 8, j = c % 8;
  unsigned short hi, lo;
  cut(w[i] * pow2f(ew), hi, lo);
  const size_t base = ((size_t)chunk * 27 + t) * 4;
  wp[((base + 0 + g) * Cout + co) * 8 + j] = hi;
  wp[((base + 2 + g) * Cout + co) * 8 + j] = lo;
}

// "// @phase N" comments mark the phase boundaries that tools/build_timing_lib.sh turns into s_memtime counters in an
// instrumented COPY of this file (0 item prologue, 1 waits at the chunk's two plane barriers, 2 load issue + wait + activate +
// max, 3 max barrier, 4 cut + LDS write, 5 wait for the next weight group + group barrier, 6 taps, 7 epilogue); the product build carries no instrumentation and no switches.

// Round-5 experiments on the sparse plan that were built, bit-exact, measured on one box inside the sampling step and NOT
// adopted live in tools/exp/conv3d_split_round5_experiments.hip (this file with the switches LION_SPLIT_COMPACT /
// LION_SPLIT_FILL; build with tools/build_variant.sh NAME conv3d_split=tools/exp/conv3d_split_round5_experiments.hip:-D...):
//   * voxel compaction inside occupied tiles (active voxels packed into 32-column MFMA blocks, a one-block wave path):
//     sparse launches 8-11 % faster, dense ones 5-6 % slower (a third copy of the K walk costs the register allocation 40
//     bytes of scratch around the staging), step 6.84 -> 6.93 ms;
//   * a plane-fill kernel for empty tiles in front of the convolution: serialises 30-40 us per convolution, step 6.84 -> 7.05 ms
//     (as queue items inside this kernel it cost the dense layer 37 %).
// Evidence: profiles/r05a_conv_ab_variants_one_box.txt, r05a_conv_ab_r04_vs_fill_items_in_kernel.txt,
// r05b_conv_epilogue_phases.txt (where a launch's cycles go).  What was adopted instead is below: empty tiles nobody reads are
// not stored at all (aware levels), and the work queue re-arms itself.
template <int N> struct IntC { static constexpr int value = N; }; // compile-time count for generic lambdas

template <int TD, int TH, int TW, int CB, int VB, bool PRO, bool STATS, int OCC>
__global__ __launch_bounds__(256, OCC) void conv3d_split_kernel(const float *__restrict__ x, const u4 *__restrict__ wp,
                                                              const float *__restrict__ wtail,
                                                              const float *__restrict__ bias, float *__restrict__ y,
                                                              int Cin, int Cout, int r,
                                                              const float *__restrict__ pro_a,
                                                              const float *__restrict__ pro_b,
                                                              const float *__restrict__ pro_bias,
                                                              const float *__restrict__ tconst,
                                                              float *__restrict__ stats, int32_t *__restrict__ occ,
                                                              int B, int ntiles) {
  constexpr int TM = 256, COT = 32 * CB;
  static_assert(TD * TH * TW == 4 * VB * 32, "tile voxels = 4 waves x VB column blocks x 32");
  constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2, HALO = HD * HH * HW;
  constexpr int HP = (HALO + 63) / 64 * 64;   // plane stride: whole waves, so a staging wave never straddles two planes
  constexpr int WPL = 4 * COT;                // u4 per weight slice (one tap of one chunk, this channel tile)
  constexpr int TG = 3;                       // taps per barrier: the weight slices of a (kd, kh) row of taps travel together
  static_assert(WPL <= TM, "one u4 of a tap's weight slice per thread");
  static_assert(27 % TG == 0, "whole groups per chunk");
  static_assert(27 * COT * 4 <= 4 * HP * 16, "the response table must fit the operand planes");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u4 *sx = reinterpret_cast<u4 *>(smem);      // [piece][half][HP]
  u4 *sw = sx + 4 * HP;                       // [2][TG taps][piece][half][COT]
  float *sbias = reinterpret_cast<float *>(sw + 2 * TG * WPL); // [COT]
  const int npro = PRO ? ((Cin + 63) & ~63) : 0;
  float *spa = sbias + COT, *spb = spa + npro, *spc = spb + npro; // prologue scalars / activated constant per channel
  float *sred = spc + npro;                   // [4][COT][2]
  float *sT = reinterpret_cast<float *>(sx);  // [27][COT] constant response (delta mode), loaded after the K loop
  __shared__ int s_work;
  __shared__ unsigned s_max[2];               // bits of the chunk's max |activation| (double buffered over chunks)
  __shared__ unsigned char s_rowok[256];      // aware level 2, delta launches: this staging thread's halo row has been written
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, l32 = lane & 31;
  const float wscale_inv = wtail[2]; // 2^-ew of the packed weights (split_tail_scale)
  const bool queued = occ != nullptr;
  const int ncz = Cout / COT;
  const int n_tile_items = ntiles * B * ncz;
  // consumer-aware buffers (lion_conv3d_tile_occupancy_aware): empty tiles whose output nobody reads store nothing
  const int aware_level = queued ? occ[2 * B * ntiles + 2] : 0;
  const bool aware = aware_level != 0;
  // @phase-init
  for (int iter = 0;; ++iter) {
  int b, tile, co0;
  if (queued) { // see csrc/conv3d.hip: occ = [B*tiles wave masks][B*tiles list, occupied tiles first][queue counter]
    // (round 4: dealing the list round-robin to the resident workgroups instead -- no atomic, no dependent index load in
    // front of an item -- was measured and LOSES on every sparse launch: conv1 Gaussian clouds 395 -> 440 us, r = 16 flat
    // delta 171 -> 225 us, sampling step 6.82 -> 7.26 ms; dense 654 -> 648 us.  Items differ too much in cost -- wave
    // masks, empty tiles -- for a static deal; the queue's round trip is not what the in-step forms pay over the plain kernel.)
    __syncthreads();
    if (tid == 0) s_work = atomicAdd(occ + 2 * B * ntiles, 1);
    __syncthreads();
    const int work = s_work;
    if (work >= n_tile_items) break;
    const int item = work / ncz;
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
  const int d0 = (tile / (ntw * nth)) * TD, h0 = ((tile / ntw) % nth) * TH, w0 = (tile % ntw) * TW;
  const int r3 = r * r * r;
  // thread t owns voxel t of the tile ((d, h, w) order) for the bookkeeping of the sparse plan
  int n_act = 4 * VB * 32;
  int wmask = 0xf;
  if (queued) { // round-3 plan: bit w of the tile's flag = wave w's 64-voxel block sees a point; bit 8 = has a reader
    const int fw = occ[b * ntiles + tile];
    wmask = fw & 0xf;
    n_act = wmask ? 4 * VB * 32 : 0;
    if (aware && fw == 0) {
      // An empty tile WITHOUT A READER (conv_tile_occ_kernel (5)): no output is stored -- nobody stages or interpolates
      // from it -- only its GroupNorm sums are owed, in closed form: voxels per border configuration x the constant the
      // dense evaluation leaves there (bias, or the delta mode's constant response).  In the chain 73-80 % of the tiles of
      // an r = 32 launch are empty and most of them have no reader: 268 MB of constants per launch were written for nobody.
      if (STATS && tid < COT) {
        const bool dl = PRO && pro_a != nullptr && tconst != nullptr;
        const int nd[3] = {d0 == 0 ? 1 : 0, TD - (d0 == 0 ? 1 : 0) - (d0 + TD == r ? 1 : 0), d0 + TD == r ? 1 : 0};
        const int nh[3] = {h0 == 0 ? 1 : 0, TH - (h0 == 0 ? 1 : 0) - (h0 + TH == r ? 1 : 0), h0 + TH == r ? 1 : 0};
        const int nw[3] = {w0 == 0 ? 1 : 0, TW - (w0 == 0 ? 1 : 0) - (w0 + TW == r ? 1 : 0), w0 + TW == r ? 1 : 0};
        float s1 = 0.f, s2 = 0.f;
        if (dl) {
#pragma unroll
          for (int cfg = 0; cfg < 27; ++cfg) { // unrolled: constant indices keep the count arrays in registers
            const float n = (float)(nd[cfg / 9] * nh[(cfg / 3) % 3] * nw[cfg % 3]);
            const float tv = tconst[((size_t)b * 27 + cfg) * Cout + co0 + tid];
            s1 += n * tv;
            s2 += n * (tv * tv);
          }
        } else {
          const float tv = bias ? bias[co0 + tid] : 0.f;
          s1 = (float)(TD * TH * TW) * tv;
          s2 = (float)(TD * TH * TW) * (tv * tv);
        }
        float *o = stats + (((size_t)b * Cout + co0 + tid) * ntiles + tile) * 2;
        o[0] = s1;
        o[1] = s2;
      }
      continue;
    }
  }
  n_act = __builtin_amdgcn_readfirstlane(n_act);
  const int my_nvb = ((wmask >> wave) & 1) ? VB : 0; // column blocks this wave runs the taps on: all of its own, or none
  const bool pro_on = PRO && pro_a != nullptr; // the PRO instantiation also serves launches without a prologue (see
  const bool delta = pro_on && tconst != nullptr; // launch_split_t: its register allocation is the better one)
  if (pro_on) {
    for (int c = tid; c < Cin; c += TM) {
      const float pa = pro_a[(size_t)b * Cin + c], pb = pro_b[(size_t)b * Cin + c];
      spa[c] = pa;
      spb[c] = pb;
      spc[c] = delta ? pro_act(pro_bias ? pro_bias[c] : 0.f, pa, pb) : 0.f;
    }
  }
  for (int c = tid; c < COT; c += TM) sbias[c] = bias ? bias[co0 + c] : 0.f;
  if (tid < 2) s_max[tid] = 0u;
  // Aware level 2 (lion_conv3d_tile_occupancy_aware): the producer of x stored its occupied (margin-1) tiles only.  Inside
  // its empty tiles x is bias1 exactly, so this launch's staged value -- the activation minus its constant -- is exactly
  // zero there: such halo rows are not loaded (their quads take the out-of-range offset, for which buffer loads return 0,
  // and the prologue writes 0 for them).  Bit 9 of this buffer's flag words = the tile is occupied at margin 1.
  const bool rows_masked = delta && aware_level == 2;
  if (rows_masked) {
    constexpr int QR_ = (TW + 8) / 4, HH_ = TH + 2, HD_ = TD + 2;
    const int row = tid / QR_, hd = row / HH_, hh = row - hd * HH_;
    const int gd = d0 - 1 + hd, gh = h0 - 1 + hh;
    bool ok = tid < HD_ * HH_ * QR_ && gd >= 0 && gd < r && gh >= 0 && gh < r;
    if (ok) ok = (occ[b * ntiles + (gd / TD) * (r / TH) + gh / TH] >> 9) & 1;
    s_rowok[tid] = ok;
  }
  int E = 127; // exponent of the tile's activation scale 2^E; 127 = none yet (everything staged so far was zero)

  // Everything derived from the thread index (the staging offsets, the fragment bases of the tap loop) is RECOMPUTED per
  // chunk from an opaque copy of it: computed once here it stays alive across the tap loop, where the allocator -- at the
  // 256-register limit -- spills exactly such long-lived values, and the reloads (scratch loads wait with vmcnt, memory
  // operations retire in order) then drain the operand loads they sit between.  Quads outside the grid carry an offset
  // beyond num_records, for which buffer loads return 0.
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(x + (size_t)b * Cin * r3), 0, Cin * r3 * 4, 0x00020000);

  f32x16 acc[CB][VB], cor[CB][VB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int vb = 0; vb < VB; ++vb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][vb][i] = cor[cb][vb][i] = 0.f;

  const bool empty = n_act == 0;
  const int nchunks = empty ? 0 : Cin / KS;
  // this thread's u4 of a weight slice: element (pg, co) of the tile <- global [pg][Cout] at co0 + co
  const int we_g = (tid / COT) * Cout + co0 + (tid % COT);
  const bool w_thread = tid < WPL;
  // weight slices travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction lands at
  // M0 + lane * 16), one group of TG taps ahead of their use, into the buffer the group before last was read from.  No
  // registers and no VALU on the way: the register ring this replaces cost 12 VGPRs at the 256-register limit (87
  // spills), and the compiler was free to sink its loads next to their LDS writes (s_memtime phase counters: 29 % of a
  // wave's cycles went into waiting for them).  The DMA is issued right behind the group barrier and awaited (vmcnt 0)
  // in front of the next one.
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const uint32_t sw_lds0 = (uint32_t)(uintptr_t)(lds_byte *)reinterpret_cast<unsigned char *>(sw);
  const uint32_t sw_lds = sw_lds0 + (uint32_t)wave * 1024u;
  const uint32_t sx_lds = (uint32_t)(uintptr_t)(lds_byte *)reinterpret_cast<unsigned char *>(sx);
  auto weights_dma = [&](int sg) { // group sg of the K walk (chunk sg / 9, taps (sg % 9) * TG ..) -> buffer sg & 1
    if (w_thread) {                // wave uniform: WPL is a multiple of 64
#pragma unroll
      for (int t = 0; t < TG; ++t) {
        // the BUILTIN, not inline asm: the compiler must know that three more VM operations are in flight.  With an asm
        // DMA its wait for the scratch reloads of the tap loop's addresses (issued in front of the barrier, waited for
        // at first use) was vmcnt(0), which -- memory operations retire in order -- also waited for the DMA: 27 of the
        // 30 DMA instructions of this kernel were drained before the first MFMA of their group, every group began with
        // the round trip of the NEXT group's slices (tools/dma_drain_check.py; found statically at the end of round 2,
        // NOT yet measured on the GPU).  With the builtin the same wait is vmcnt(3) and the DMA flies under the taps.
        const u4 *gp = wp + ((size_t)sg * TG + t) * 4 * Cout + we_g;
        typedef __attribute__((address_space(3))) void lds_void;
        typedef __attribute__((address_space(1))) const void glb_void;
        lds_void *dstp = (lds_void *)(uintptr_t)__builtin_amdgcn_readfirstlane(sw_lds + (uint32_t)(((sg & 1) * TG + t) * WPL * 16));
        __builtin_amdgcn_global_load_lds((glb_void *)gp, dstp, 16, 0, 0);
      }
    }
  };
  if (nchunks) { weights_dma(0); weights_dma(1); } // nchunks >= 1 -> at least 9 groups
  // @phase 0
  // One chunk of the K walk, for a wave that got NVB column blocks of the tile's active voxels.  NVB = 0 is the copy run by
  // a wave without a block: it stages and takes part in every barrier and in the weight DMA, but owns no MFMA and no
  // accumulator; NVB = 1 runs the taps on one column block (CB x 1 accumulator tiles, half the MFMAs).  The copies are
  // separate LOOPS (the branch on the block count sits outside them): with the branch inside the chunk -- per tap or around
  // the 27 taps -- the register allocator split the accumulators' live ranges around the working path and parked five of
  // the eight tuples in scratch across the staging of every chunk (684-792 bytes, 64->64 at 1070 us instead of 705).
  auto chunk = [&](int q, auto nvb_c) {
    constexpr int NVB = decltype(nvb_c)::value;
    constexpr bool WORK = NVB > 0;
    __syncthreads(); // the previous chunk's planes are no longer read (and the prologue scalars are visible)
    // @phase 1
    {
    // Staging by aligned 16-byte row loads: thread rt owns one QUAD of a halo row -- 4 consecutive w of row (hd, hh),
    // starting at w0 - 4 + 4 qd (the rows are read from w0 - 4 to w0 + TW + 3: 6 / 10 quads, of which the first and the
    // last contribute one column each) -- for all 16 channels of the chunk: 16 dwordx4 loads per thread instead of 48
    // dword gathers (per-lane dword gathers are bound by the texture-address path: ~13 k cycles per chunk).  r % 4 == 0 and
    // w0 % 4 == 0: a quad lies entirely inside or entirely outside the grid.
    constexpr int QR = (TW + 8) / 4, IPH = HD * HH * QR;
    static_assert(IPH <= TM, "one quad per thread");
    int rt = tid;
    asm volatile("" : "+v"(rt));
    const int row = rt / QR, qd = rt - row * QR;
    const int hd = row / HH, hh = row - hd * HH;
    const int gd = d0 - 1 + hd, gh = h0 - 1 + hh, gw0 = w0 - 4 + 4 * qd;
    const bool gok = rt < IPH && gd >= 0 && gd < r && gh >= 0 && gh < r && gw0 >= 0 && gw0 < r && (!rows_masked || s_rowok[rt]);
    const int goff = gok ? ((gd * r + gh) * r + gw0) * 4 : 0x7fffff00;
    const int p0 = row * HW + 4 * qd - 3; // halo position of the quad's first column (column k is used iff 0 <= hw0 + k < HW)
    const int hw0 = 4 * qd - 3;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v[2][8];
#pragma unroll
    for (int ig = 0; ig < 2; ++ig)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[ig][j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(xrs, goff, (q * KS + ig * 8 + j) * r3 * 4, 0));
    unsigned mloc = 0u;
    if (pro_on) {
#pragma unroll
      for (int ig = 0; ig < 2; ++ig) {
        const int c0 = q * KS + ig * 8;
        const float4 a0 = *reinterpret_cast<const float4 *>(spa + c0), a1 = *reinterpret_cast<const float4 *>(spa + c0 + 4);
        const float4 b0 = *reinterpret_cast<const float4 *>(spb + c0), b1 = *reinterpret_cast<const float4 *>(spb + c0 + 4);
        const float4 c4 = *reinterpret_cast<const float4 *>(spc + c0), c5 = *reinterpret_cast<const float4 *>(spc + c0 + 4);
        const float pa8[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float pb8[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        const float pc8[8] = {c4.x, c4.y, c4.z, c4.w, c5.x, c5.y, c5.z, c5.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float act = pro_act(v[ig][j][k], pa8[j], pb8[j]) - pc8[j];
            v[ig][j][k] = gok ? act : 0.f;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool used = rt < IPH && hw0 + k >= 0 && hw0 + k < HW;
      unsigned mk = 0u;
#pragma unroll
      for (int ig = 0; ig < 2; ++ig)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned a = __float_as_uint(v[ig][j][k]) & 0x7fffffffu; // |t| as ordered bits; inf / nan do not set the scale
          mk = (a > mk && a <= 0x7f7fffffu) ? a : mk;
        }
      mloc = (used && mk > mloc) ? mk : mloc;
    }
    mloc = wave_max_u32_lane63(mloc);
    if (lane == 63 && mloc) atomicMax(&s_max[q & 1], mloc);
    // @phase 2
    __syncthreads(); // the chunk's maximum is complete
    // @phase 3
    const unsigned mbits = s_max[q & 1];
    if (tid == 0) s_max[(q + 1) & 1] = 0u; // its last readers passed the barrier at the top of this chunk
    if (mbits) {
      const int e = scale_exp(__uint_as_float(mbits));
      if (e < E) { // the tile's maximum grew: bring what has been accumulated onto the new (smaller) scale first
        if (WORK && E != 127) {
          const float f = pow2f(max(e - CONV_SPLIT_HEADROOM - E, -126));
#pragma unroll
          for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int vb = 0; vb < NVB; ++vb)
#pragma unroll
              for (int i = 0; i < 16; ++i) { acc[cb][vb][i] *= f; cor[cb][vb][i] *= f; }
        }
        E = e - CONV_SPLIT_HEADROOM;
      }
    }
    const float xs = E == 127 ? 1.0f : pow2f(E);
#pragma unroll
    for (int ig = 0; ig < 2; ++ig)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        u4 ph, pl;
#pragma unroll
        for (int m = 0; m < 4; ++m) { unsigned h2, l2; cut2(v[ig][2 * m][k] * xs, v[ig][2 * m + 1][k] * xs, h2, l2); ph[m] = h2; pl[m] = l2; }
        if (rt < IPH && hw0 + k >= 0 && hw0 + k < HW) {
          sx[(0 + ig) * HP + p0 + k] = ph;
          sx[(2 + ig) * HP + p0 + k] = pl;
        }
      }
    }
    // @phase 4
    // The 27 taps.  Weight slices travel in groups of TG taps through two buffers (group k of chunk q = walk index
    // sg = 9 q + k, buffer sg & 1).  Barrier k sits in front of the LAST tap of group k: by then every wave holds that
    // tap's fragments in registers, so buffer sg & 1 is free for the DMA of group sg + 2, and group sg + 1 (requested one
    // barrier earlier, awaited just before this one) is visible -- its first fragments are requested under the MFMAs of
    // this last tap.  Fragments are double buffered in registers: the reads of tap t + 1 are spread, one at a time,
    // between the MFMAs of tap t (a wave draws 1 KiB per 32 cycles from LDS at best; a burst of eight in front of a tap
    // takes 256 cycles to land), and ONE counted lgkmcnt wait in front of a tap finds them there.  Round 2 read just in
    // time -- five exposed LDS round trips per tap (`r6 wait M4 r wait M ...` in the ISA), as long as the MFMAs themselves.
    {
      const int par = q & 1;
      typedef __attribute__((address_space(3))) const u4 lds_u4;
      // opaque per-chunk base addresses: every fragment read = base + 16-bit immediate.  Left to itself the compiler
      // hoists 27 tap offsets x (VB + CB) addresses out of the chunk loop and spills them.
      uint32_t xq[NVB > 0 ? NVB : 1], wq2[2];
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int g_ = ln >> 5, l32_ = ln & 31;
#pragma unroll
      for (int vb = 0; vb < NVB; ++vb) { // halo position of this lane's voxel in the wave's column block vb
        const int v = (wave * VB + vb) * 32 + l32_;
        const int d = v / (TH * TW), h = (v / TW) % TH, w = v % TW;
        xq[vb] = sx_lds + (uint32_t)((g_ * HP + (d * HH + h) * HW + w) * 16);
        asm volatile("" : "+v"(xq[vb]));
      }
      wq2[0] = sw_lds0 + (uint32_t)((par * TG * WPL + g_ * COT + l32_) * 16);
      wq2[1] = sw_lds0 + (uint32_t)(((par ^ 1) * TG * WPL + g_ * COT + l32_) * 16);
      asm volatile("" : "+v"(wq2[0]));
      asm volatile("" : "+v"(wq2[1]));
      constexpr int NX = NVB > 0 ? NVB : 1;
      u4 wf[2][CB][2], xf[2][NX][2];
      constexpr int NR = 2 * NVB + 2 * CB; // fragment reads per tap
      // read r_ of a tap, in the order the tap's MFMAs need them: X_h (NVB), W_h (CB) -- the main sweep --, then X_l (NVB),
      // then W_l (CB); LDS returns in order, so the counted wait in front of the first MFMA covers only the first NVB + CB
      auto frag = [&](int tap, int s_, int r_) {
        const int pc = r_ >= NVB + CB, rr = pc ? r_ - (NVB + CB) : r_;
        if (rr < NVB) {
          const int vb = rr;
          const int toff = ((tap / 9) * HH + (tap / 3) % 3) * HW + tap % 3;
          xf[s_][vb][pc] = *(lds_u4 *)(uintptr_t)(xq[vb] + (uint32_t)((pc * 2 * HP + toff) * 16));
        } else {
          const int cb = rr - NVB, k = tap / TG, t = tap % TG;
          wf[s_][cb][pc] = *(lds_u4 *)(uintptr_t)(wq2[k & 1] + (uint32_t)((t * WPL + pc * 2 * COT + cb * 32) * 16));
        }
      };
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // group 9 q (requested two barriers ago / in the item prologue)
      __syncthreads(); // the chunk's operand planes and the first weight group are visible
      // @phase 1
      // a wave whose 64-voxel block sees no point (wave mask) takes part in the barriers and the weight DMA only: its own
      // copy of the walk, so that the working waves' 27 taps are straight-line code (one uniform branch per tap cost the
      // register allocator 350 bytes of scratch)
      auto group_barrier = [&](int k) {
        const int sg = q * (27 / TG) + k;
        // @phase 6
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // group sg + 1 has landed
        __syncthreads();
        // @phase 5
        if (sg + 2 < nchunks * (27 / TG)) weights_dma(sg + 2);
      };
      if constexpr (WORK) {
        // raised priority while the wave owns MFMAs: its issue wins the SIMD's arbitration against the co-resident
        // workgroup's staging VALU (64->64@32^3: 618-624 -> 603-614 us)
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int r_ = 0; r_ < NR; ++r_) frag(0, 0, r_);
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
          const int cur = tap & 1, nxt = cur ^ 1;
          if (tap % TG == TG - 1) group_barrier(tap / TG);
          // MFMA m of the tap: the CB VB main products, then the X_lo products, then the W_lo products (two MFMAs into one
          // accumulator are CB VB issues apart)
          auto mfma = [&](int m) {
            const int kind = m / (CB * NX), cb = (m / NX) % CB, vb = m % NX;
            if (kind == 0) acc[cb][vb] = mma(wf[cur][cb][0], xf[cur][vb][0], acc[cb][vb]);
            else if (kind == 1) cor[cb][vb] = mma(wf[cur][cb][0], xf[cur][vb][1], cor[cb][vb]);
            else cor[cb][vb] = mma(wf[cur][cb][1], xf[cur][vb][0], cor[cb][vb]);
          };
          constexpr int NM = 3 * CB * NX;
#pragma unroll
          for (int m = 0; m < NM; ++m) {
            // the reads of tap + 1: PER per MFMA slot from slot 1 on (1 at CB = 2: slots 1 .. 8 of 11; 2 at CB = 1), so that
            // the last of them has the rest of this tap to land
            constexpr int PER = (NR + NM - 2) / (NM - 1);
            if (m >= 1 && tap + 1 < 27) {
#pragma unroll
              for (int r_ = (m - 1) * PER; r_ < m * PER && r_ < NR; ++r_) frag(tap + 1, nxt, r_);
            }
            mfma(m);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_s_setprio(0);
      } else {
#pragma unroll
        for (int k = 0; k < 27 / TG; ++k) group_barrier(k);
      }
      // @phase 6
    }
  };
  if (my_nvb == VB) { for (int q = 0; q < nchunks; ++q) chunk(q, IntC<VB>{}); }
  else if (VB > 1 && my_nvb == 1) { for (int q = 0; q < nchunks; ++q) chunk(q, IntC<1>{}); }
  else { for (int q = 0; q < nchunks; ++q) chunk(q, IntC<0>{}); }

  if (delta) {
    __syncthreads(); // the last tap's LDS reads are done: the operand planes become the response table
    for (int e = tid; e < 27 * COT; e += TM) sT[e] = tconst[((size_t)b * 27 + e / COT) * Cout + co0 + e % COT];
    __syncthreads();
  } else if (empty) {
    __syncthreads(); // sbias was written by other threads and no barrier of the K loop ran
  }
  // epilogue: D = main + corr/2048 (+ bias | constant response), NCDHW store.  acc register i of lane l: channel row
  // (i&3) + 8*(i>>2) + 4*(l>>5), voxel column l&31.
  float *yb = y + ((size_t)b * Cout + co0) * r3;
  const float us_x = E == 127 ? 1.0f : pow2f(-E), us_w = wscale_inv; // exact powers of two
  // two passes: every output value first (the accumulators become the outputs), then NOTHING BUT stores.  In one loop
  // the compiler reloaded spilled values between the stores and waited for each reload with vmcnt(0|1) -- which also
  // waits for the stores issued before it: 18-23 store / wait / store sequences per epilogue (tools/store_wait_scan.py),
  // each a round trip to memory.
  int gvv[VB];
#pragma unroll
  for (int vb = 0; vb < VB; ++vb) {
    const int v = (wave * VB + vb) * 32 + l32;
    const int d = v / (TH * TW), h = (v / TW) % TH, w = v % TW;
    const int gd = d0 + d, gh = h0 + h, gw = w0 + w;
    gvv[vb] = (gd * r + gh) * r + gw;
    const int cfg = (((gd == 0 ? 0 : gd == r - 1 ? 2 : 1) * 3 + (gh == 0 ? 0 : gh == r - 1 ? 2 : 1)) * 3 +
                     (gw == 0 ? 0 : gw == r - 1 ? 2 : 1));
    const float *addv = delta ? sT + cfg * COT : sbias;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * g;
        acc[cb][vb][i] = ((acc[cb][vb][i] + cor[cb][vb][i] * (1.f / 2048.f)) * us_x) * us_w + addv[co];
      }
  }
#pragma unroll
  for (int vb = 0; vb < VB; ++vb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * g;
        const float o = acc[cb][vb][i];
        yb[(size_t)co * r3 + gvv[vb]] = o;
      }
  if (STATS) { // per-tile channel sums, as csrc/conv3d.hip
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int vb = 0; vb < VB; ++vb) {
          const float o = acc[cb][vb][i];
          s1 += o;
          s2 += o * o;
        }
        s1 = row16_sum_rn(s1); s2 = row16_sum_rn(s2);
        s1 = row_pair_sum_odd_rows(s1); s2 = row_pair_sum_odd_rows(s2);
        if (l32 == 16) { // the row pair's sum lives in the odd rows
          const int co = cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * g;
          sred[(wave * COT + co) * 2] = s1;
          sred[(wave * COT + co) * 2 + 1] = s2;
        }
      }
    __syncthreads();
    if (tid < COT) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { s1 += sred[(w * COT + tid) * 2]; s2 += sred[(w * COT + tid) * 2 + 1]; }
      float *o = stats + (((size_t)b * Cout + co0 + tid) * ntiles + tile) * 2;
      o[0] = s1;
      o[1] = s2;
    }
  }
  // @phase 7
  } // work loop
  // the queue re-arms itself (see csrc/conv3d.hip): the last workgroup to leave zeroes the queue and the exit counter
  if (queued && tid == 0) {
    int32_t *q = occ + 2 * B * ntiles;
    if (atomicAdd(q + 1, 1) == (int)gridDim.x - 1) { q[0] = 0; q[1] = 0; }
  }
  // @phase-flush
}

template <int TD, int TH, int TW, int CB, int VB, int OCC>
static int launch_split_t(const float *x, const u4 *wp, const float *wtail, const float *bias, float *y, int B, int Cin,
                          int Cout, int r,
                          const float *pa, const float *pb, const float *pbias, const float *tconst, float *stats,
                          int32_t *occ, hipStream_t st) {
  constexpr int COT = 32 * CB;
  constexpr int HALO = (TD + 2) * (TH + 2) * (TW + 2), HP = (HALO + 63) / 64 * 64;
  const int tiles = (r / TD) * (r / TH) * (r / TW);
  static int cu_count[LION_MAX_DEVICES] = {0};
  int dev = 0;
  if (int e = lion_current_device(&dev)) return e;
  if (!cu_count[dev]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return LION_EINVAL;
    cu_count[dev] = prop.multiProcessorCount;
  }
  const long items = (long)B * tiles * (Cout / COT);
  const long resident = (long)OCC * cu_count[dev];
  const dim3 grid = occ ? dim3((unsigned)(items < resident ? items : resident)) : dim3(B, tiles, Cout / COT);
  // (PRO = false, STATS = true) of the 64-channel tile gets 556-568 bytes of scratch from the register allocator where
  // (true, true) gets 304-360: launches with statistics and without a prologue run on the PRO instantiation with the
  // prologue switched off at run time (pro_a == nullptr)
  const bool pro_inst = pa != nullptr || (stats != nullptr && CB == 2 && Cin <= 256);
  const size_t LDS = (size_t)(4 * HP + 2 * 3 * 4 * COT) * 16 + // planes + two groups of 3 taps of weight slices
                     (size_t)(COT + (pro_inst ? 3 * ((Cin + 63) & ~63) : 0) + 4 * COT * 2) * 4;
#define LION_SPLIT_GO(PRO_, ST_)                                                                             \
  {                                                                                                          \
    static LionLdsLimit cfg = {};                                                                            \
    if (int e = lion_dynamic_lds(&conv3d_split_kernel<TD, TH, TW, CB, VB, PRO_, ST_, OCC>, LDS, cfg)) return e;   \
    conv3d_split_kernel<TD, TH, TW, CB, VB, PRO_, ST_, OCC><<<grid, 256, LDS, st>>>(x, wp, wtail, bias, y, Cin, Cout, r, pa, pb, \
                                                                              pbias, tconst, stats, occ, B, tiles); \
  }
  if (pro_inst && stats) LION_SPLIT_GO(true, true)
  else if (pa) LION_SPLIT_GO(true, false)
  else if (stats) LION_SPLIT_GO(false, true)
  else LION_SPLIT_GO(false, false)
#undef LION_SPLIT_GO
  LION_LAUNCH_CHECK();
  return 0;
}

// ---- r = 8: the pipelined form -------------------------------------------------------------------------------------
// A sample has only 512 voxels, so B * Cout / 32 half-sample tiles (256 voxels x 32 channels) are all the work there is:
// ONE workgroup per CU at B = 32, Cout = 128, nothing else resident to hide its latencies behind.  The tile therefore
// pipelines itself:
//   * operand planes double buffered: the global loads of chunk q+1 are issued in front of the 27 taps of chunk q and
//     land in registers while the MFMAs run; they are activated, scaled, cut and written to the other plane buffer
//     behind the taps (one barrier for the chunk maximum);
//   * weight slices by LDS-DMA in groups of 9 taps (one kd plane: 18 KiB), ring of three groups, issued TWO groups
//     (108 MFMAs per wave) ahead; waited for with a counted s_waitcnt (memory operations retire in order; the counts
//     below are the operations this wave is known to have issued behind the awaited DMA -- at least DMA_MIN weight
//     instructions per group and the NLOAD operand loads -- so they can only be too strict, never too lax);
//   * fragments of tap t+1 are read from LDS in front of the MFMAs of tap t.
// Two facts measured in round 2 shape the workgroup (tools/exp/lds_b128_probe.hip, s_memtime): ONE wave reads LDS at
// 32 B/clk and issues a 32x32x16 MFMA every ~64 cycles, whatever else the CU does -- four waves x (32 channels x 64
// voxels: 6 fragment reads per 6 MFMAs) sit on both limits (taps: 331 cycles per tap round for 192 of MFMA).  So the
// default is EIGHT waves (two per SIMD) x one column block: 4 reads per 3 MFMAs and wave, 255 B/clk of LDS with the
// conflict-free row order below, 54 -> 48 us at 128 -> 128, B = 32 (LION_CONV_R8_WAVES=4 selects the 4-wave form).
// Voxel -> lane: a column block of 32 voxels is 8 w x 4 halo rows chosen so that a 32-lane group of a ds_read_b128
// touches every 16-byte slot of the 512-byte LDS window once: rows {h, h+2, h+4, h+6} at row stride 12 (8 waves), rows
// {h, h+4, h+1, h+5} at stride 10 (4 waves: conflict free in 16-lane groups at the 128 B/clk four waves can draw).
// Dense only (the sparse plan starts at r = 16), prologue and statistics as conv3d_split_kernel, no delta mode.
template <bool PRO, bool STATS, int NW>
__global__ __launch_bounds__(64 * NW, 1) void conv3d_split_pipe_kernel(const float *__restrict__ x, const u4 *__restrict__ wp,
                                                                  const float *__restrict__ wtail,
                                                                  const float *__restrict__ bias, float *__restrict__ y,
                                                                  int Cin, int Cout, const float *__restrict__ pro_a,
                                                                  const float *__restrict__ pro_b,
                                                                  float *__restrict__ stats) {
  // NW = 4 waves x 2 column blocks or NW = 8 waves x 1 (two waves per SIMD: see the comment above).  Halo row stride HW:
  // 10 for NW = 4 (block rows {h, h+4, h+1, h+5}), 12 for NW = 8 (block rows {h, h+2, h+4, h+6}: 24 / 48 / 72 = 24, 16, 8
  // mod 32 -- the four rows of a 32-lane group fall into four different quarters of the 512-byte LDS window)
  constexpr int r = 8, r3 = 512, TD = 4, TH = 8, TW = 8, VB = 8 / NW, COT = 32, TM = 64 * NW;
  static_assert(NW == 4 || NW == 8, "4 waves x 2 column blocks or 8 waves x 1");
  constexpr int HD = TD + 2, HH = TH + 2, HW = NW == 8 ? TW + 4 : TW + 2, HALO = HD * HH * HW; // 600 / 720
  constexpr int HP = (HALO + 63) / 64 * 64;                                 // 640 / 768
  constexpr int WPL = 4 * COT, TG = 9, NG = 27 / TG;                        // 128 u4 per tap slice; groups of 9 taps
  constexpr int DMA_PER_GROUP = TG * WPL / 64, DMA_MIN = DMA_PER_GROUP / NW; // 18 wave instructions over the waves
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u4 *sx = reinterpret_cast<u4 *>(smem);      // [2][piece][half][HP]
  u4 *sw = sx + 2 * 4 * HP;                   // [3][TG][piece][half][COT]
  float *sbias = reinterpret_cast<float *>(sw + 3 * TG * WPL);
  const int npro = PRO ? ((Cin + 63) & ~63) : 0;
  float *spa = sbias + COT, *spb = spa + npro;
  float *sred = spb + npro;                   // [NW][COT][2]
  __shared__ unsigned s_max[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, l32 = lane & 31;
  const int b = blockIdx.x, tile = blockIdx.y, co0 = blockIdx.z * COT, d0 = tile * TD;
  const float wscale_inv = wtail[2];
  if (PRO)
    for (int c = tid; c < Cin; c += TM) { spa[c] = pro_a[(size_t)b * Cin + c]; spb[c] = pro_b[(size_t)b * Cin + c]; }
  if (tid < COT) sbias[tid] = bias ? bias[co0 + tid] : 0.f;
  if (tid < 2) s_max[tid] = 0u;
  int E = 127;
  // @phase-init

  // Staging by aligned 16-byte row loads (round 3, as conv3d_split_kernel).  At r = 8 the tile spans the whole (h, w)
  // plane: its halo in h and w lies outside the grid -- always zero -- so only the 6 x 8 rows x 2 quads of real voxels
  // move at all (the planes' halo slots are zeroed once, below).  Thread t < 384 owns (k-half ig, 4 of its 8 channels,
  // row, quad): 4 dwordx4 loads, 16 values, 4 positions x 2 pieces x 8 bytes of LDS; threads 384 .. 511 issue the same
  // number of (out-of-range, zero) loads so that the counted vmcnt waits below stay uniform.
  static_assert(NW == 8, "the quad staging is laid out for 512 threads");
  constexpr int NLOAD = 4;
  const int st_ig = tid / 192, st_u = tid % 192, st_ch4 = st_u / 96, st_rq = st_u % 96;
  const int st_row = st_rq >> 1, st_quad = st_rq & 1, st_hd = st_row >> 3, st_gh = st_row & 7;
  const int st_gd = d0 - 1 + st_hd;
  const bool gok = tid < 384 && st_gd >= 0 && st_gd < r;
  const int goff = gok ? ((st_gd * r + st_gh) * r + st_quad * 4) * 4 + (st_ig * 8 + st_ch4 * 4) * r3 * 4 : 0x7fffff00;
  // byte offset of (position of voxel 0 of the quad, this thread's 8-byte channel half) inside a [piece][half] plane pair
  const int st_pos = (st_hd * HH + st_gh + 1) * HW + st_quad * 4 + 1;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(x + (size_t)b * Cin * r3), 0, Cin * r3 * 4, 0x00020000);
  // this lane's voxel in column block vbk = wave * VB + vb: d = vbk / 2, w = l % 8,
  // h = 2 (vbk % 2) + {0, 4, 1, 5}[l / 8] (NW = 4) or (vbk % 2) + {0, 2, 4, 6}[l / 8] (NW = 8)
  int xbase[VB], vox[VB];
#pragma unroll
  for (int vb = 0; vb < VB; ++vb) {
    const int vbk = wave * VB + vb;
    const int d = vbk >> 1, w = l32 & 7;
    const int h = NW == 8 ? (vbk & 1) + 2 * (l32 >> 3) : (vbk & 1) * 2 + ((l32 >> 4) & 1) + 4 * ((l32 >> 3) & 1);
    xbase[vb] = (d * HH + h) * HW + w;
    vox[vb] = ((d0 + d) * r + h) * r + w;
  }
  f32x16 acc[VB], cor[VB], cor2[VB]; // cor += W_h X_l, cor2 += W_l X_h: no two consecutive MFMAs share an accumulator
#pragma unroll
  for (int vb = 0; vb < VB; ++vb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[vb][i] = cor[vb][i] = cor2[vb][i] = 0.f;

  const int nchunks = Cin / KS, ngroups = nchunks * NG;
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const uint32_t sw_lds = (uint32_t)(uintptr_t)(lds_byte *)reinterpret_cast<unsigned char *>(sw);
  auto weights_dma = [&](int sg) { // group sg (chunk sg / 3, taps 9 (sg % 3) ..) -> ring slot sg % 3
    const u4 *src = wp + (size_t)sg * TG * 4 * Cout + co0;
    const uint32_t dst0 = sw_lds + (uint32_t)((sg % 3) * TG * WPL * 16);
    for (int i = wave; i < DMA_PER_GROUP; i += NW) { // instruction i: tap i / 2, planes 2 (i % 2) + {0, 1}, 32 channels each
      const u4 *gp = src + (size_t)((i >> 1) * 4 + (i & 1) * 2 + g) * Cout + l32;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(dst0 + (uint32_t)(i * 1024));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gp), "s"(dst) : "memory");
    }
  };
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 v[NLOAD];
  auto issue_loads = [&](int q) { // the operand loads of a chunk, unconditionally (outside the grid: offset past the end -> 0)
#pragma unroll
    for (int j = 0; j < NLOAD; ++j)
      v[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(xrs, goff, (q * KS + j) * r3 * 4, 0));
  };
  auto stage = [&](int q) { // registers -> activated, scaled, cut -> plane buffer q & 1 (contains the chunk-maximum barrier)
    unsigned mloc = 0u;
    if (PRO) {
      const int c0 = q * KS + st_ig * 8 + st_ch4 * 4;
      const float4 a4 = *reinterpret_cast<const float4 *>(spa + (tid < 384 ? c0 : 0));
      const float4 b4 = *reinterpret_cast<const float4 *>(spb + (tid < 384 ? c0 : 0));
      const float pa[4] = {a4.x, a4.y, a4.z, a4.w}, pb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int j = 0; j < NLOAD; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float act = pro_act(v[j][k], pa[j], pb[j]);
          v[j][k] = gok ? act : 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < NLOAD; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned a = __float_as_uint(v[j][k]) & 0x7fffffffu;
        mloc = (a > mloc && a <= 0x7f7fffffu) ? a : mloc;
      }
    mloc = wave_max_u32_lane63(mloc);
    if (lane == 63 && mloc) atomicMax(&s_max[q & 1], mloc);
    __syncthreads(); // the chunk's maximum is complete
    const unsigned mbits = s_max[q & 1];
    if (tid == 0) s_max[(q + 1) & 1] = 0u; // last read behind the previous chunk's maximum barrier
    if (mbits) {
      const int e = scale_exp(__uint_as_float(mbits));
      if (e < E) {
        if (E != 127) {
          const float f = pow2f(max(e - CONV_SPLIT_HEADROOM - E, -126));
#pragma unroll
          for (int vb = 0; vb < VB; ++vb)
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[vb][i] *= f; cor[vb][i] *= f; cor2[vb][i] *= f; }
        }
        E = e - CONV_SPLIT_HEADROOM;
      }
    }
    const float xs = E == 127 ? 1.0f : pow2f(E);
    if (tid < 384) {
      uint2 *dst2 = reinterpret_cast<uint2 *>(sx + (q & 1) * 4 * HP);
#pragma unroll
      for (int k = 0; k < 4; ++k) { // voxel k of the quad: this thread's 4 channels = 8 bytes of the position's u4
        unsigned h0, l0, h1, l1;
        cut2(v[0][k] * xs, v[1][k] * xs, h0, l0);
        cut2(v[2][k] * xs, v[3][k] * xs, h1, l1);
        dst2[((0 + st_ig) * HP + st_pos + k) * 2 + st_ch4] = make_uint2(h0, h1);
        dst2[((2 + st_ig) * HP + st_pos + k) * 2 + st_ch4] = make_uint2(l0, l1);
      }
    }
  };

  // the halo slots (and the row-stride padding) of both plane buffers are never written again: zero everything once
  for (int e = tid; e < 2 * 4 * HP; e += TM) sx[e] = u4{0u, 0u, 0u, 0u};
  weights_dma(0);
  if (ngroups > 1) weights_dma(1);
  issue_loads(0);
  __syncthreads(); // prologue scalars, s_max = 0
  stage(0);
  // @phase 0
  for (int q = 0; q < nchunks; ++q) {
    const u4 *sxq = sx + (q & 1) * 4 * HP;
    const bool more = q + 1 < nchunks;
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
      const int sg = q * NG + grp;
      // group sg's slices must have landed.  Issued behind them by this wave, in order: [grp 0] the DMA of group sg + 1;
      // [grp 1, 2] the DMA of group sg + 1 and this chunk's operand prefetch (when there is a next chunk)
      const bool dma_behind = sg + 1 < ngroups;
      if (grp == 0 || !more) { if (dma_behind) wait_vm<DMA_MIN>(); else wait_vm<0>(); }
      else { if (dma_behind) wait_vm<DMA_MIN + NLOAD>(); else wait_vm<NLOAD>(); }
      // @phase 1
      __syncthreads(); // slices of group sg and (grp 0) the planes of chunk q visible; ring slot of group sg - 1 free
      // @phase 5
      if (sg + 2 < ngroups) weights_dma(sg + 2);
      if (grp == 0 && more) issue_loads(q + 1);
      const u4 *swg = sw + (sg % 3) * TG * WPL;
      u4 wf[2][2], xf[2][VB][2];
      auto frags = [&](int t, int s_) {
        const int tap = grp * TG + t;
        const int toff = ((tap / 9) * HH + (tap / 3) % 3) * HW + tap % 3;
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
          wf[s_][pc] = swg[t * WPL + (pc * 2 + g) * COT + l32];
#pragma unroll
          for (int vb = 0; vb < VB; ++vb) xf[s_][vb][pc] = sxq[(pc * 2 + g) * HP + xbase[vb] + toff];
        }
      };
      frags(0, 0);
#pragma unroll
      for (int t = 0; t < TG; ++t) {
        if (t + 1 < TG) frags(t + 1, (t + 1) & 1);
        __builtin_amdgcn_sched_barrier(0); // keep the reads of tap t + 1 in front of the MFMAs of tap t (one wave per SIMD:
#pragma unroll                             // nothing else hides the LDS latency)
        for (int vb = 0; vb < VB; ++vb) {
          acc[vb] = mma(wf[t & 1][0], xf[t & 1][vb][0], acc[vb]);
          cor[vb] = mma(wf[t & 1][0], xf[t & 1][vb][1], cor[vb]);
          cor2[vb] = mma(wf[t & 1][1], xf[t & 1][vb][0], cor2[vb]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // @phase 6
    }
    if (more) stage(q + 1); // plane buffer (q + 1) & 1: last read by the taps of chunk q - 1, two barriers ago
    // @phase 4
  }

  float *yb = y + ((size_t)b * Cout + co0) * r3;
  const float us_x = E == 127 ? 1.0f : pow2f(-E), us_w = wscale_inv;
#pragma unroll
  for (int vb = 0; vb < VB; ++vb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = (i & 3) + 8 * (i >> 2) + 4 * g;
      const float o = ((acc[vb][i] + (cor[vb][i] + cor2[vb][i]) * (1.f / 2048.f)) * us_x) * us_w + sbias[co];
      acc[vb][i] = o;
      yb[(size_t)co * r3 + vox[vb]] = o;
    }
  if (STATS) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float s1 = acc[0][i], s2 = acc[0][i] * acc[0][i];
#pragma unroll
      for (int vb = 1; vb < VB; ++vb) { s1 += acc[vb][i]; s2 += acc[vb][i] * acc[vb][i]; }
      s1 = row16_sum_rn(s1); s2 = row16_sum_rn(s2);
      s1 = row_pair_sum_odd_rows(s1); s2 = row_pair_sum_odd_rows(s2);
      if (l32 == 16) { // the row pair's sum lives in the odd rows
        const int co = (i & 3) + 8 * (i >> 2) + 4 * g;
        sred[(wave * COT + co) * 2] = s1;
        sred[(wave * COT + co) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    if (tid < COT) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { s1 += sred[(w * COT + tid) * 2]; s2 += sred[(w * COT + tid) * 2 + 1]; }
      float *o = stats + (((size_t)b * Cout + co0 + tid) * (r / TD) + tile) * 2;
      o[0] = s1;
      o[1] = s2;
    }
  }
  // @phase 7
  // @phase-flush
}

template <int NW>
static int launch_split_pipe(const float *x, const u4 *wp, const float *wtail, const float *bias, float *y, int B, int Cin,
                             int Cout, const float *pa, const float *pb, float *stats, hipStream_t st) {
  constexpr int HP = NW == 8 ? 768 : 640, COT = 32;
  const dim3 grid(B, 2, Cout / COT);
  const size_t LDS = (size_t)(2 * 4 * HP + 3 * 9 * 4 * COT) * 16 +
                     (size_t)(COT + (pa ? 2 * ((Cin + 63) & ~63) : 0) + NW * COT * 2) * 4;
#define LION_PIPE_GO(PRO_, ST_)                                                                              \
  {                                                                                                          \
    static LionLdsLimit cfg = {};                                                                            \
    if (int e = lion_dynamic_lds(&conv3d_split_pipe_kernel<PRO_, ST_, NW>, LDS, cfg)) return e;              \
    conv3d_split_pipe_kernel<PRO_, ST_, NW><<<grid, 64 * NW, LDS, st>>>(x, wp, wtail, bias, y, Cin, Cout, pa, pb, stats); \
  }
  if (pa && stats) LION_PIPE_GO(true, true)
  else if (pa) LION_PIPE_GO(true, false)
  else if (stats) LION_PIPE_GO(false, true)
  else LION_PIPE_GO(false, false)
#undef LION_PIPE_GO
  LION_LAUNCH_CHECK();
  return 0;
}

// tiles: always the 4 waves x 2 column blocks geometry of the product's sparse plan (so that the occupancy lists of
// lion_conv3d_tile_occupancy apply unchanged), one column block per wave at r = 8
struct SplitPlan { int vb, cb, tiles; };
static SplitPlan split_plan(int r, int Cout) {
  const int r3 = r * r * r;
  const int cb = Cout % 64 == 0 ? 2 : Cout % 32 == 0 ? 1 : 0;
  // r = 8: the pipelined half-sample kernel (conv3d_split_pipe_kernel): 2 tiles of 256 voxels, 32 channels per workgroup
  // (history: 128-voxel tiles 144 us, whole-sample tiles x 32 channels on B * Cout/32 = 128 workgroups 104-108 us at
  // 128->128, B=32, against 114 us of the fp32 kernel)
  if (r == 8) return {2, Cout % 32 == 0 ? 1 : 0, 2};
  return {2, cb, r3 / 256};
}

} // namespace

extern "C" {

// number of uint16 in the packed weights: (Cin/16) * 27 * [2 pieces][2 halves] * Cout * 8 pieces + an 8-halfword tail
// {max |w| bits, ew, 2^-ew, 0} (the tensor's power-of-two scale)
static size_t split_piece_halfs(int Cout, int Cin) { return (size_t)(Cin / KS) * 27 * 4 * Cout * 8; }
size_t lion_conv3d_split_packed_halfs(int Cout, int Cin) { return split_piece_halfs(Cout, Cin) + 8; }

int lion_conv3d_split_pack_weights(const float *w, int Cout, int Cin, uint16_t *wp, lionStream_t stream) {
  if (!w || !wp || Cout <= 0 || Cin <= 0) return LION_EINVAL;
  if (Cin % KS != 0 || (((uintptr_t)wp) & 15) != 0) return LION_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned *tail = reinterpret_cast<unsigned *>(wp + split_piece_halfs(Cout, Cin));
  const int n = Cout * Cin * 27;
  if (hipMemsetAsync(tail, 0, 16, st) != hipSuccess) return LION_EINVAL;
  split_wmax_kernel<<<min(lion_cdiv(n, 2048), 128), 256, 0, st>>>(w, n, tail);
  split_pack_kernel<<<lion_cdiv(n, 256), 256, 0, st>>>(w, Cout, Cin, wp, tail);
  LION_LAUNCH_CHECK();
  return 0;
}

// @phase-reader

int lion_conv3d_split_stat_tiles(int r, int Cout) {
  if (r != 8 && r != 16 && r != 32) return 0;
  return split_plan(r, Cout).tiles;
}

// Arguments exactly as lion_conv3d_k3_fused_forward (include/lion_hip.h), wp from lion_conv3d_split_pack_weights;
// stats has lion_conv3d_split_stat_tiles(r, Cout) tiles; occ from lion_conv3d_tile_occupancy (same tile geometry:
// 4 waves x 2 column blocks of 32 voxels, the fp32 kernel's sparse plan).
int lion_conv3d_k3_split_forward(const float *x, const uint16_t *wp, const float *bias, int B, int Cin, int Cout,
                                 int r, const float *pro_a, const float *pro_b, const float *pro_bias,
                                 const float *tconst, float *y, float *stats, int32_t *occ, lionStream_t stream) {
  if (!x || !wp || !y || B <= 0 || Cin <= 0 || Cout <= 0) return LION_EINVAL;
  if ((pro_a == nullptr) != (pro_b == nullptr)) return LION_EINVAL;
  if (tconst && !pro_a) return LION_EINVAL;
  if (occ && pro_a && !tconst) return LION_EINVAL;
  if (Cin % KS != 0 || (pro_a && Cin > 256)) return LION_EUNSUPPORTED;
  if (r != 8 && r != 16 && r != 32) return LION_EUNSUPPORTED;
  if (occ && r == 8) return LION_EUNSUPPORTED;
  const SplitPlan p = split_plan(r, Cout);
  if (!p.cb) return LION_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const u4 *w4 = reinterpret_cast<const u4 *>(wp);
  const float *wtail = reinterpret_cast<const float *>(wp + split_piece_halfs(Cout, Cin));
  if (r == 8) {
    if (tconst) return LION_EUNSUPPORTED;
    return launch_split_pipe<8>(x, w4, wtail, bias, y, B, Cin, Cout, pro_a, pro_b, stats, st);
  }
#define LION_SPLIT_TILE(R_, VB_, CB_, TD_, TH_, TW_, OCC_)                                                  \
  if (r == R_ && p.vb == VB_ && p.cb == CB_)                                                                \
    return launch_split_t<TD_, TH_, TW_, CB_, VB_, OCC_>(x, w4, wtail, bias, y, B, Cin, Cout, r, pro_a, pro_b, pro_bias, tconst, \
                                                   stats, occ, st);
  LION_SPLIT_TILE(32, 2, 2, 2, 4, 32, 2)
  LION_SPLIT_TILE(32, 2, 1, 2, 4, 32, 2)
  LION_SPLIT_TILE(16, 2, 2, 4, 4, 16, 2)
  LION_SPLIT_TILE(16, 2, 1, 4, 4, 16, 2)
#undef LION_SPLIT_TILE
  return LION_EUNSUPPORTED;
}

} // extern "C"
