// conv3d_split_pc.hip -- the r >= 16 form of the split-operand 3x3x3 convolution (csrc/conv3d_split.hip holds the
// arithmetic, the weight packing, the r = 8 kernel and the C entry point): ONE persistent workgroup per CU, eight waves in
// two roles.
//
//   consumers (waves 0-3, one per SIMD): nothing but the matrix pipe.  Each owns 64 voxels x COT output channels of the
//     256-voxel tile (2 x CB accumulator pairs) and walks the 27 taps of a 16-channel chunk with double-buffered
//     fragments: the eight fragments of tap t + 1 (1 KiB each) are requested half way through tap t and land under its
//     second half.  They also bring the weight slices in (LDS-DMA, groups of 3 taps, two buffers) and run the
//     tile's epilogue (scale back, bias | constant response, NCDHW stores, GroupNorm sums) straight from the accumulators.
//   producers (waves 4-7, the second wave of each SIMD): everything else.  They stage the NEXT chunk while the consumers
//     multiply the current one -- global loads of the haloed tile, fused AdaGN + Swish (+ the delta form), the tile's
//     running maximum, the cut into fp16 hi / lo pieces, the LDS planes -- into the other of two plane buffers, pop the
//     work queue, and write the tiles that hold no point (bias | constant response only) on the side.
//
// Why (round 2 measurements): with every wave doing both jobs the 256-register budget of two waves
// per SIMD held 128 accumulators + 56 staged values + fragments + addresses, the compiler read fragments just in time
// (four exposed LDS round trips per tap: 65-70 cycles per MFMA of a wave, the pipe issues one per 32), two co-resident
// workgroups overlapped staging and taps only by accident, and the pipe was busy 43 % of the time.  Here the consumer
// stream holds 128 + 64 registers and no VALU work at all, the producers hold 56 + temporaries.
//
// Synchronisation: gfx950 has one workgroup barrier.  Both roles run the same STEP loop with exactly NB = 9 barriers per
// step; in step s the producers stage job s and the consumers multiply job s - 1 (job = one 16-channel chunk of one
// non-empty tile; plane buffer, scale slot and weight-buffer parity are all derived from the job number).  Barrier k of
// a step sits in front of the LAST tap of weight group k (3 taps): behind it group k + 1 (DMA issued a barrier ago,
// awaited by every consumer wave just before) is visible, group k's buffer is free (its last fragments are in
// registers) and takes the DMA of group k + 2, and after barrier 8 the planes of the next job are complete.  The
// producers cut their work into the same nine slices: 0 prologue + issue of all 56 loads per thread, 1 queue / empty
// tiles (under the loads' latency), 2-5 activation and maximum, [barrier 5: the maximum is complete] 6-8 cut and planes.
// Everything the roles tell each other travels through a small control block and an item ring in LDS, written in front
// of one barrier and read behind it; a step without work still runs its nine barriers, so the counts cannot diverge.
#include "split_ops.h"

namespace {

// LDS-only release / acquire around the barrier: the global loads a producer has in flight across its slices must not be
// drained by a barrier (the general __syncthreads() fences every address space)
__device__ __forceinline__ void pc_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int P> struct PcPar { static constexpr int value = P; };

constexpr int PC_RING = 64; // item ring (power of two)
constexpr int PC_POP = 4;   // items per queue pop (one returning atomic + two dependent lookups, one lane per item)
struct PcItem { int b, tile, co0, wmask; }; // wmask 0: no point within the margin (constant tile); < 0: end of the stream
// every lane reads the same entry: hand the fields to the scalar unit (addresses derived from them stay uniform)
__device__ __forceinline__ PcItem pc_ring_read(const PcItem *ring, int i) {
  const PcItem v = ring[i & (PC_RING - 1)];
  return PcItem{__builtin_amdgcn_readfirstlane(v.b), __builtin_amdgcn_readfirstlane(v.tile),
                __builtin_amdgcn_readfirstlane(v.co0), __builtin_amdgcn_readfirstlane(v.wmask)};
}
struct PcCtl {
  int planned;   // jobs whose staging has begun  (leader, slice 0)
  int staged;    // jobs whose planes are complete (leader, slice 8)
  int exit_step; // first step nobody runs any more (INT_MAX until the leader has seen the end of the stream)
  int tail;      // ring entries published
  int chead;     // ring entries the consumers are done with
  int E[2];      // scale exponent the planes of job j were cut with: E[j & 1]
  unsigned smax[2];
};

// Epilogue of one wave quarter (64 voxels x COT channels): D = main + corr / 2048 scaled back, + addend, NCDHW stores,
// and the quarter's GroupNorm sums.  acc register i of lane l: channel row (i & 3) + 8 (i >> 2) + 4 (l >> 5), voxel
// column l & 31.  ZERO: a tile without points -- the accumulators are zero by definition, the output is the addend.
// addend(cfg, co): bias[co] or the constant response of border configuration cfg (delta mode).
template <int TD, int TH, int TW, int CB, bool STATS, bool ZERO, bool STORE = true, typename Addend>
__device__ __forceinline__ void pc_epilogue(f32x16 (&acc)[CB][2], f32x16 (&cor)[CB][2], float us_x, float us_w, Addend addend,
                                            int wq, int lane, float *__restrict__ yb, int r, int d0, int h0, int w0,
                                            float *__restrict__ st, int st_cstride) {
  const int g = lane >> 5, l32 = lane & 31;
  // The per-channel offsets (channel * r^3, channel * statistics stride) are wave uniform: scalar base + per-lane 32-bit
  // offset.  r^3 and the stride pass through an empty asm so that the 32 products per pointer are formed HERE -- as loop
  // invariants the compiler formed them as 64-bit per-lane values at kernel entry, spilled all 88 and reloaded one in
  // front of every store (a scratch reload between stores makes each store wait for the ones before it).
  int r3 = r * r * r;
  asm volatile("" : "+s"(r3));
  asm volatile("" : "+s"(st_cstride));
  unsigned voff[2];
  int cfg[2];
#pragma unroll
  for (int vb = 0; vb < 2; ++vb) {
    const int v = (wq * 2 + vb) * 32 + l32;
    const int d = v / (TH * TW), h = (v / TW) % TH, w = v % TW;
    const int gd = d0 + d, gh = h0 + h, gw = w0 + w;
    voff[vb] = (unsigned)(4 * g * r3 + (gd * r + gh) * r + gw);
    cfg[vb] = (((gd == 0 ? 0 : gd == r - 1 ? 2 : 1) * 3 + (gh == 0 ? 0 : gh == r - 1 ? 2 : 1)) * 3 +
               (gw == 0 ? 0 : gw == r - 1 ? 2 : 1));
  }
  const unsigned soff = (unsigned)(4 * g * st_cstride);
  if (!ZERO) { // main + corr / 2048 at the operands' scales, in place, before anything else is loaded: kills the 64 (32)
    // correction registers while the epilogue's own temporaries are not alive yet
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int vb = 0; vb < 2; ++vb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[cb][vb][i] = ((acc[cb][vb][i] + cor[cb][vb][i] * (1.f / 2048.f)) * us_x) * us_w;
    __builtin_amdgcn_sched_barrier(0);
  }
  // One channel block (32 channels x the wave's 64 voxels = 2 x 16 registers) at a time: values, sums, stores, fenced
  // off from the next block.  Left to itself the scheduler forms all 64 outputs in NEW registers beside the 128
  // accumulators and the prefetched fragments of the next job, and the allocator pays for that peak by spilling
  // accumulators INSIDE the tap loop (seen: 880 bytes of scratch, 12 spills + 13 reloads per tap).  No global load and no
  // scratch access sits between the stores, so none of them waits for the ones before it (tools/store_wait_scan.py).
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
    for (int vb = 0; vb < 2; ++vb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * g;
        const float a = addend(cfg[vb], co);
        acc[cb][vb][i] = ZERO ? a : acc[cb][vb][i] + a;
      }
    __builtin_amdgcn_sched_barrier(0);
    if (STATS) { // per (tile, wave quarter) channel sums: sum over the quarter's 64 voxels, fixed tree
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float s1 = acc[cb][0][i] + acc[cb][1][i];
        float s2 = acc[cb][0][i] * acc[cb][0][i] + acc[cb][1][i] * acc[cb][1][i];
        s1 = row16_sum_rn(s1); s2 = row16_sum_rn(s2);
        s1 = row_pair_sum_odd_rows(s1); s2 = row_pair_sum_odd_rows(s2);
        if (l32 == 16) { // the row pair's sum lives in the odd rows
          float *o = st + (size_t)(cb * 32 + (i & 3) + 8 * (i >> 2)) * st_cstride; // uniform
          o[soff] = s1;
          o[soff + 1] = s2;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (STORE) {
#pragma unroll
      for (int vb = 0; vb < 2; ++vb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float *yc = yb + (size_t)(cb * 32 + (i & 3) + 8 * (i >> 2)) * r3; // uniform
          yc[voff[vb]] = acc[cb][vb][i];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int TD, int TH, int TW, int CB, bool PRO, bool STATS>
__global__ __launch_bounds__(512, 2) void conv3d_split_pc_kernel(const float *__restrict__ x, const u4 *__restrict__ wp,
                                                                const float *__restrict__ wtail,
                                                                const float *__restrict__ bias, float *__restrict__ y,
                                                                int Cin, int Cout, int r,
                                                                const float *__restrict__ pro_a,
                                                                const float *__restrict__ pro_b,
                                                                const float *__restrict__ pro_bias,
                                                                const float *__restrict__ tconst,
                                                                float *__restrict__ stats, int32_t *__restrict__ occ,
                                                                int B, int ntiles) {
  constexpr int TM = 256, COT = 32 * CB, VB = 2, NB = 9, TG = 3;
  static_assert(TD * TH * TW == 256, "tile = 4 wave quarters x 2 column blocks x 32 voxels");
  constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2, HALO = HD * HH * HW;
  constexpr int HP = (HALO + 63) / 64 * 64;   // plane stride: whole waves, so a staging wave never straddles two planes
  // staging items of the producers: (k-half, halo row, group of 4 positions along w).  A thread loads its item's 8 channels
  // with ONE 16-byte load each (4 consecutive w): the texture-address unit handles a wave instruction one 16-byte request
  // at a time, and the dword-per-lane gather of round 2 (56 loads per thread, misaligned quads) cost ~13 000 cycles per
  // chunk -- more than the chunk's 324 MFMAs (tools/r3: producers alone 5.6 us per step without any arithmetic).
  // Tiles span the grid's full width (TW == r, checked by the launcher): the two halo columns of a row lie outside the
  // grid -- zeros, written once at kernel start -- and a row's interior is TW / 4 aligned, always-in-range quads.
  constexpr int ROWS = HD * HH, GR = TW / 4;        // halo rows, 4-voxel groups per row
  constexpr int IH = (ROWS * GR + 63) / 64 * 64;    // items per k-half, whole waves (the k-half of an item is wave uniform)
  constexpr int NI = (2 * IH + TM - 1) / TM;        // items per producer thread
  static_assert(NI == 2, "slice plan of the producers: two items per thread");
  constexpr int WPL = 4 * COT;                // u4 per weight slice (one tap of one chunk, this channel tile)
  static_assert(WPL <= TM && WPL % 64 == 0, "one u4 of a tap's weight slice per consumer thread, whole waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u4 *sx = reinterpret_cast<u4 *>(smem);                 // [2 buffers][piece][half][HP]
  u4 *sw = sx + 2 * 4 * HP;                              // [2 buffers][TG taps][piece][half][COT]
  float *sadd = reinterpret_cast<float *>(sw + 2 * TG * WPL); // [3 slots][27 | 1][COT]: constant response | bias of an item
  const bool delta_launch = PRO && pro_a != nullptr && tconst != nullptr;
  const int add_rows = delta_launch ? 27 : 1;
  const int npro = PRO ? ((Cin + 63) & ~63) : 0;
  float *spa = sadd + 3 * add_rows * COT, *spb = spa + npro, *spc = spb + npro;
  PcItem *ring = reinterpret_cast<PcItem *>(spc + npro);
  PcCtl *ctl = reinterpret_cast<PcCtl *>(ring + PC_RING);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  const int wq = wave & 3;                   // consumer: the wave's quarter of the tile; producer: its part of the staging
  const int rt = tid & 255;                  // thread index inside the role
  const bool queued = occ != nullptr;
  const int ncz = Cout / COT, nchunks = Cin / KS;
  const long total_work = (long)B * ntiles * ncz;
  const int r3 = r * r * r, ntw = r / TW, nth = r / TH;
  const bool pro_on = PRO && pro_a != nullptr; // the PRO instantiation also serves launches without a prologue
  const bool delta = delta_launch;
  const float wscale_inv = wtail[2];          // 2^-ew of the packed weights (split_wscale_kernel)
  const int st_tiles = ntiles * 4;            // statistics entries per (sample, channel): (tile, wave quarter)

  for (int e = tid; e < 2 * 4 * HP; e += 512) { // the halo columns outside the grid (and the plane padding) stay zero
    const int p = e % HP, hw = p % HW;
    if (p >= HALO || hw == 0 || hw == HW - 1) sx[e] = u4{0u, 0u, 0u, 0u};
  }
  if (tid == 0) {
    ctl->planned = 0; ctl->staged = 0; ctl->exit_step = 0x7fffffff; ctl->tail = 0; ctl->chead = 0;
    ctl->E[0] = ctl->E[1] = 127; ctl->smax[0] = ctl->smax[1] = 0u;
  }
  __syncthreads();

  // Nothing derived from the lane index is shared between the roles: each takes an OPAQUE copy.  One register allocation
  // serves both code paths; a common subexpression formed in front of the role branch (an LDS fragment address, say) is
  // alive through the producers' high-pressure staging too, gets spilled there, and is reloaded -- with a vmcnt(0) that
  // drains the weight DMA -- inside the consumers' tap loop.
  if (producer) {
    // ================================================================================================ producers
    int lane_p = lane;
    asm volatile("" : "+v"(lane_p));
    int ph = 0;            // ring walk: entries examined for jobs
    int pf = 0;            // ring entries whose tiles without points have been written (pf <= ph)
    int pj = 0;            // jobs staged so far == number of the next job
    int nitem = 0;         // non-empty items begun (slot of the addend table: nitem % 3)
    bool ended = false;    // the ring walk has reached the end-of-stream entry
    int npop = 0;          // pops done (static work assignment)
    bool pop_done = false; // the leader has published the end of the stream
    bool pending = false;  // the loads of job pj are in flight (requested at the end of the previous step)
    // current item of the staging (valid while q < nchunks)
    int ib = 0, itile = 0, ico0 = 0, q = nchunks, d0 = 0, h0 = 0, w0 = 0, E = 127;
    float v[NI][8][4]; // [item][channel][voxel of the group]
    const int rt_entry = rt;
    for (int s = 0;; ++s) {
      if (s >= ctl->exit_step) break;
      // everything derived from the thread index is recomputed per step from an opaque copy: hoisted to the kernel entry
      // it would stay alive through the consumers' code as well (one register allocation for both roles)
      int rt = rt_entry;
      asm volatile("" : "+v"(rt));
      // item = rt + 256 i -> k-half item / IH (wave uniform), row (item % IH) / GR, group (item % IH) % GR.  Rows outside
      // the grid carry an offset beyond num_records: the loads return 0.
      struct Geo { int ig, p0, goff; bool live, rowok; };
      auto geo = [&](int i) {
        const int item = rt + TM * i;
        Geo gq;
        gq.ig = __builtin_amdgcn_readfirstlane(min(item / IH, 1));
        const int rg = item - gq.ig * IH;
        const int row = rg / GR, grp = rg - row * GR, hd = row / HH, hh = row - hd * HH;
        gq.live = item < 2 * IH && rg < ROWS * GR;
        const int gd = d0 - 1 + hd, gh = h0 - 1 + hh;
        gq.rowok = gq.live && gd >= 0 && gd < r && gh >= 0 && gh < r;
        gq.p0 = (hd * HH + hh) * HW + 1 + grp * 4; // halo position of the group's first voxel (column 0 is padding)
        gq.goff = gq.rowok ? ((gd * r + gh) * r + grp * 4) * 4 : 0x7fffff00;
        return gq;
      };
      const bool have = pending;
      u4 *dst = sx + (pj & 1) * 4 * HP;
      // ---- slices 0-5: activation (AdaGN + Swish of the previous convolution, delta form) and the chunk maximum, the 16
      // (item, channel) rows of 4 voxels dealt 3 3 3 3 2 2 over the slices -- each lighter than the consumers' three taps
      // between two barriers, so that the matrix pipe does not wait for a slice (the loads were requested a step ago)
      unsigned mloc = 0u;
      auto activate = [&](int row0, int row1) {
#pragma unroll
        for (int rw = row0; rw < row1; ++rw) {
          const int i = rw / 8, j = rw % 8;
          const Geo gq = geo(i);
          float pa = 0.f, pb = 0.f, pc = 0.f;
          if (pro_on) { // broadcast reads of the channel's prologue scalars (wave-uniform address)
            const int c = q * KS + gq.ig * 8 + j;
            pa = spa[c]; pb = spb[c]; pc = spc[c];
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            // zero padding stays zero; delta mode stages the deviation from the per-channel constant; the activation
            // unconditionally with a select behind it (a branch per value otherwise)
            const float t = pro_on ? pro_act(v[i][j][k], pa, pb) - pc : v[i][j][k];
            const float u = gq.rowok ? t : 0.f;
            v[i][j][k] = u;
            const unsigned a = __float_as_uint(u) & 0x7fffffffu; // |u| as ordered bits; inf / nan do not set the scale:
            mloc = (a > mloc && a <= 0x7f7fffffu) ? a : mloc;    // they pass through the cut as inf / nan
          }
        }
      };
      if (have) activate(0, 3);
      pc_barrier(); // 0
      if (have) activate(3, 6);
      pc_barrier(); // 1
      if (have) activate(6, 9);
      pc_barrier(); // 2
      if (have) activate(9, 12);
      pc_barrier(); // 3
      if (have) activate(12, 14);
      pc_barrier(); // 4
      if (have) {
        activate(14, 16);
        mloc = wave_max_u32_lane63(mloc);
        if (lane_p == 63 && mloc) atomicMax(&ctl->smax[pj & 1], mloc);
      }
      pc_barrier(); // 5: the chunk's maximum is complete
      float xs = 1.f;
      if (have) {
        const unsigned mbits = (unsigned)__builtin_amdgcn_readfirstlane((int)ctl->smax[pj & 1]);
        if (mbits) {
          const int e = scale_exp(__uint_as_float(mbits));
          if (e < E) E = e - CONV_SPLIT_HEADROOM;
        }
        if (rt == 0) { ctl->E[pj & 1] = E; ctl->smax[(pj + 1) & 1] = 0u; } // the other slot: last read in step s - 1
        xs = E == 127 ? 1.0f : pow2f(E);
      }
      // ---- slices 6-8: cut into hi / lo pieces at the tile's scale, planes of buffer pj & 1 (3 + 3 + 2 voxels); the queue
      // and the tiles without points ride in slices 6 and 7
      auto cutwrite = [&](int i, int k0, int k1) { // voxels k0 .. k1 - 1 of item i
        const Geo gq = geo(i);
#pragma unroll
        for (int k = k0; k < k1; ++k) {
          unsigned short hi[8], lo[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) cut(v[i][j][k] * xs, hi[j], lo[j]);
          u4 ph4, pl4;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            ph4[m] = (unsigned)hi[2 * m] | ((unsigned)hi[2 * m + 1] << 16);
            pl4[m] = (unsigned)lo[2 * m] | ((unsigned)lo[2 * m + 1] << 16);
          }
          if (gq.live) {
            dst[(0 + gq.ig) * HP + gq.p0 + k] = ph4;
            dst[(2 + gq.ig) * HP + gq.p0 + k] = pl4;
          }
        }
      };
      if (have) cutwrite(0, 0, 3);
      {
        // one tile without points per step: every producer wave writes its quarter (64 voxels x COT channels of bias |
        // constant response).  Statistics: the consumers' epilogue with zero accumulators and no stores -- the same tree,
        // so the sums are bit-identical to the dense evaluation of such a tile; stores: 16 bytes per lane.
        while (pf < ph && pc_ring_read(ring, pf).wmask != 0) ++pf;
        if (pf < ph) {
          const PcItem it = pc_ring_read(ring, pf);
          ++pf;
          const int fd0 = (it.tile / (ntw * nth)) * TD, fh0 = ((it.tile / ntw) % nth) * TH, fw0 = (it.tile % ntw) * TW;
          const float *ga = delta ? tconst + (size_t)it.b * 27 * Cout + it.co0 : (bias ? bias + it.co0 : nullptr);
          const int gs = delta ? Cout : 0;
          float *yb = y + ((size_t)it.b * Cout + it.co0) * r3;
          if (STATS) {
            f32x16 za[CB][2], zc[CB][2];
            pc_epilogue<TD, TH, TW, CB, true, true, false>(
                za, zc, 1.f, 1.f, [&](int cfg, int co) { return ga ? ga[cfg * gs + co] : 0.f; }, wq, lane_p, yb, r, fd0, fh0, fw0,
                stats + (((size_t)it.b * Cout + it.co0) * st_tiles + it.tile * 4 + wq) * 2, st_tiles * 2);
          }
          // lane -> (channel of a group of four, quad of voxels): quad qd = lane % 16 of the quarter's 16, channel lane / 16
          const int qd = lane_p & 15, v0 = wq * 64 + qd * 4;
          const int vd = v0 / (TH * TW), vh = (v0 / TW) % TH, vw = v0 % TW; // TW % 4 == 0: a quad stays inside a row
          const int gd = fd0 + vd, gh = fh0 + vh, gw = fw0 + vw;
          const int crow = ((gd == 0 ? 0 : gd == r - 1 ? 2 : 1) * 3 + (gh == 0 ? 0 : gh == r - 1 ? 2 : 1)) * 3;
          const int cfg0 = crow + (gw == 0 ? 0 : 1), cfg3 = crow + (gw + 3 == r - 1 ? 2 : 1), cfgm = crow + 1;
          float *yq = yb + (size_t)((gd * r + gh) * r + gw);
#pragma unroll 4
          for (int cc = 0; cc < COT / 4; ++cc) {
            const int co = cc * 4 + (lane_p >> 4);
            const float a0 = ga ? ga[cfg0 * gs + co] : 0.f, am = ga ? ga[cfgm * gs + co] : 0.f, a3 = ga ? ga[cfg3 * gs + co] : 0.f;
            *reinterpret_cast<float4 *>(yq + (size_t)co * r3) = make_float4(a0, am, am, a3);
          }
        }
      }
      pc_barrier(); // 6
      if (have) { cutwrite(0, 3, 4); cutwrite(1, 0, 2); }
      const int tail0 = ctl->tail;
      if (wave == 4 && !pop_done) { // the leader refills the ring (visible to the walk below, behind barrier 7)
        const int chead = ctl->chead;
        const int low = min(chead, pf);
        // pop when the walk is about to run dry, in small lots: what a workgroup has popped it must finish, and the tail
        // of the launch is only as balanced as the lots are small
        if (tail0 - ph < PC_POP && tail0 - low <= PC_RING - PC_POP) {
          long work;
          if (queued) {
            int base = 0;
            if (lane_p == 0) base = atomicAdd(occ + 2 * B * ntiles, PC_POP);
            work = (long)__builtin_amdgcn_readfirstlane(base) + lane_p;
          } else {
            work = (long)blockIdx.x + (long)(npop * PC_POP + lane_p) * gridDim.x;
          }
          ++npop;
          if (lane_p < PC_POP) {
            PcItem it;
            if (work >= total_work) { it.b = 0; it.tile = 0; it.co0 = 0; it.wmask = -1; }
            else {
              const int item = (int)(work / ncz);
              it.b = item % B;
              it.tile = queued ? occ[B * ntiles + it.b * ntiles + item / B] : item / B;
              it.co0 = (int)(work % ncz) * COT;
              it.wmask = queued ? occ[it.b * ntiles + it.tile] : 0xf;
            }
            ring[(tail0 + lane_p) & (PC_RING - 1)] = it;
          }
          const long last = queued ? (long)__builtin_amdgcn_readfirstlane((int)work) + PC_POP - 1
                                   : (long)blockIdx.x + (long)((npop - 1) * PC_POP + PC_POP - 1) * gridDim.x;
          if (last >= total_work) pop_done = true;
          if (lane_p == 0) ctl->tail = tail0 + PC_POP;
        }
      }
      pc_barrier(); // 7
      if (have) {
        cutwrite(1, 2, 4);
        if (rt == 0) ctl->staged = pj + 1;
        ++pj;
        ++q;
      }
      // ---- the next job: the tile's next chunk, or the next non-empty ring entry (tiles without points on the way are left
      // to the fill head); its loads are requested NOW and land under slices 0-5 of the step that activates them
      pending = false;
      if (q < nchunks) pending = true;
      else if (!ended) {
        const int tail = ctl->tail;
        while (ph < tail && pc_ring_read(ring, ph).wmask == 0) ++ph;
        if (ph < tail) {
          const PcItem it = pc_ring_read(ring, ph);
          if (it.wmask < 0) ended = true;
          else {
            ib = it.b; itile = it.tile; ico0 = it.co0; q = 0; E = 127; ++ph;
            d0 = (itile / (ntw * nth)) * TD; h0 = ((itile / ntw) % nth) * TH; w0 = (itile % ntw) * TW;
            pending = true;
            if (pro_on) // (their last readers passed barrier 5 of this step)
              for (int c = rt; c < Cin; c += TM) {
                const float pa = pro_a[(size_t)ib * Cin + c], pb = pro_b[(size_t)ib * Cin + c];
                spa[c] = pa;
                spb[c] = pb;
                spc[c] = delta ? pro_act(pro_bias ? pro_bias[c] : 0.f, pa, pb) : 0.f;
              }
            float *sa = sadd + (nitem % 3) * add_rows * COT; // read by the consumers' epilogue >= 2 steps later
            for (int e = rt; e < add_rows * COT; e += TM)
              sa[e] = delta ? tconst[((size_t)ib * 27 + e / COT) * Cout + ico0 + e % COT] : (bias ? bias[ico0 + e] : 0.f);
            ++nitem;
          }
        }
      }
      if (pending) {
        if (rt == 0) ctl->planned = pj + 1;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x + (size_t)ib * Cin * r3), 0, Cin * r3 * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < NI; ++i) { // the 8 channels x 4 voxels of each item: 16 loads of 16 bytes per thread
          const Geo gq = geo(i);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const u4 t4 = __builtin_amdgcn_raw_buffer_load_b128(xrs, gq.goff, (q * KS + gq.ig * 8 + j) * r3 * 4, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[i][j][k] = __uint_as_float(t4[k]);
          }
        }
      }
      // end of the stream: everything popped has been walked, every tile without points written, the last job staged (it
      // is multiplied in step s + 1)
      if (ended && pf == ph && rt == 0 && ctl->exit_step == 0x7fffffff) ctl->exit_step = s + 2;
      pc_barrier(); // 8: the planes of job pj - 1 are complete
    }
  } else {
    // ================================================================================================ consumers
    int lane_c = lane;
    asm volatile("" : "+v"(lane_c));
    const int g = lane_c >> 5, l32 = lane_c & 31;
    int ch = 0;        // ring entries this role is done with
    int cj = 0;        // jobs multiplied so far == number of the next job
    int nitem = 0;     // non-empty items begun
    u4 wf[2][CB][2], xf[2][VB][2];
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    typedef __attribute__((address_space(3))) const u4 lds_u4;
    // LDS byte addresses of this lane's fragments in buffer 0 (the per-job bases are derived from them below)
    uint32_t xa0[VB], wa0; // plane buffer 0 / weight buffer 0; the other buffer is a constant number of bytes further
    {
      const uint32_t sx_lds = (uint32_t)(uintptr_t)(lds_byte *)reinterpret_cast<unsigned char *>(sx);
      const uint32_t sw0_lds = (uint32_t)(uintptr_t)(lds_byte *)reinterpret_cast<unsigned char *>(sw);
#pragma unroll
      for (int vb = 0; vb < VB; ++vb) { // halo position of the lane's voxel v = (wq * VB + vb) * 32 + lane % 32, plane half g
        const int v = (wq * VB + vb) * 32 + l32;
        const int d = v / (TH * TW), h = (v / TW) % TH, w = v % TW;
        xa0[vb] = sx_lds + (uint32_t)((g * HP + (d * HH + h) * HW + w) * 16);
      }
      wa0 = sw0_lds + (uint32_t)((g * COT + l32) * 16);
    }
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void glb_void;
    const uint32_t sw_lds = (uint32_t)(uintptr_t)(lds_byte *)reinterpret_cast<unsigned char *>(sw) + (uint32_t)wq * 1024u;
    const bool w_thread = rt < WPL; // wave uniform
    // weight group grp (3 taps) of chunk qq of channel tile co0 -> buffer `buf`: one u4 per thread and tap by LDS-DMA
    // address = scalar base (weights + the slice's offset, SALU) + ONE per-lane byte offset.  (Formed as per-lane 64-bit
    // pointers the 27 slice addresses of a chunk were strength-reduced into 27 live pointer pairs, spilled at the tile's
    // start and reloaded in front of every DMA.)
    const uint32_t w_lane = (uint32_t)(((rt / COT) * Cout + (rt % COT)) * 16);
    auto weights_dma = [&](int co0, int qq, int grp, int buf) {
      if (w_thread) {
#pragma unroll
        for (int t = 0; t < TG; ++t) {
          // scalar base + the lane's 32-bit byte offset (saddr form: no VALU, no address registers); inline asm, not the
          // builtin: every wait for these DMAs is written out below (vmcnt(0) in front of each barrier), and a DMA the
          // compiler knows about makes it wait vmcnt(0) in front of the first DS read behind it -- it cannot tell that the
          // reads go to the OTHER weight buffer -- which stalled the first taps of every step on the DMA just issued.
          // (There is no scratch access in the consumer loop whose vmcnt wait could drain an invisible DMA --
          // tests/test_isa_cpu.py keeps it that way.)
          const char *ub = reinterpret_cast<const char *>(wp) + ((size_t)(qq * 27 + grp * TG + t) * 4 * Cout + co0) * 16; // uniform
          const uint32_t dst = __builtin_amdgcn_readfirstlane(sw_lds + (uint32_t)((buf * TG + t) * WPL * 16));
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(w_lane), "s"(ub), "s"(dst) : "memory");
        }
      }
    };
    // the job staged during the current step, known behind barrier 0: does it exist, and which weights does it need
    struct Next { bool have; int co0, q; };
    auto peek_next = [&](int nj, bool same_item, int co0_cur, int q_next) {
      Next n = {ctl->planned > nj, co0_cur, q_next};
      if (n.have && !same_item) { // it opens a new item: the next non-empty ring entry
        int c2 = ch;
        PcItem it = pc_ring_read(ring, c2);
        while (it.wmask == 0) { ++c2; it = pc_ring_read(ring, c2); }
        n.co0 = it.co0;
        n.q = 0;
      }
      return n;
    };
    int s = 0;
    for (;;) { // ---- one tile per round
      // steps without a staged job: the nine barriers, and the first two weight groups of the job being staged (if any)
      bool finished = false;
      for (;;) {
        if (s >= ctl->exit_step) { finished = true; break; }
        if (cj < ctl->staged) break;
        // tiles without points are the producers' business: step over them and SAY so -- the ring is refilled only as far
        // as the slower of the two heads allows, and a head that waits for a tile of its own behind a run of empty ones
        // longer than the ring would stall the refill for good
        {
          const int tail = ctl->tail;
          while (ch < tail && pc_ring_read(ring, ch).wmask == 0) ++ch;
          if (rt == 0) ctl->chead = ch;
        }
        Next nx = {false, 0, 0};
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          pc_barrier();
          if (k == 0) nx = peek_next(cj, false, 0, 0);
          if (k + 2 >= NB && nx.have) weights_dma(nx.co0, nx.q, k + 2 - NB, (k + cj + 1) & 1); // groups 0, 1 of job cj
        }
        ++s;
      }
      if (finished) break;
      // the tile: the next non-empty ring entry (published at least a step ago)
      PcItem it = pc_ring_read(ring, ch);
      while (it.wmask == 0) { ++ch; it = pc_ring_read(ring, ch); }
      ++ch;
      const int ib = it.b, itile = it.tile, ico0 = it.co0;
      const int d0 = (itile / (ntw * nth)) * TD, h0 = ((itile / ntw) % nth) * TH, w0 = (itile % ntw) * TW;
      const bool wave_on = (it.wmask >> wq) & 1; // a wave whose 64 voxels see no point skips its MFMAs, not its barriers
      ++nitem;
      int E = 127;
      f32x16 acc[CB][VB], cor[CB][VB];
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int vb = 0; vb < VB; ++vb)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[cb][vb][i] = cor[cb][vb][i] = 0.f;
      for (int q = 0; q < nchunks; ++q) { // one step per chunk: the producers stage a tile's chunks in consecutive steps
        // the scale the planes were cut with; when the tile's maximum grew, bring what has been accumulated onto it
        const int e = __builtin_amdgcn_readfirstlane(ctl->E[cj & 1]);
        if (e != E) {
          if (E != 127 && wave_on) {
            const float f = pow2f(max(e - E, -126));
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
              for (int vb = 0; vb < VB; ++vb)
#pragma unroll
                for (int i = 0; i < 16; ++i) { acc[cb][vb][i] *= f; cor[cb][vb][i] *= f; }
          }
          E = e;
        }
        // The 27 taps (+ the nine barriers) of the step.  Which plane buffer (job parity) and which weight buffer (group +
        // job parity) a read goes to is a matter of BASE ADDRESSES selected here; the fragment registers are walked the
        // same way in every job (tap t on halves t & 1), so there is one copy of this code.  The price: the fragments of a
        // job's first tap are requested at its start, not under the last tap of the job before (~300 of ~11 000 cycles) --
        // with 27 taps per job that hand-over needs the register halves of odd and even jobs swapped, i.e. two copies of
        // the walk whose register assignments the compiler reconciled through scratch at every step boundary.
        const int par = cj & 1;
        uint32_t xq[VB], wq2[2]; // wq2[p]: buffer of the groups with (k & 1) == p
#pragma unroll
        for (int vb = 0; vb < VB; ++vb) {
          xq[vb] = xa0[vb] + (uint32_t)(par * 4 * HP * 16);
          asm volatile("" : "+v"(xq[vb])); // opaque: every fragment read = this base + a 16-bit immediate (the region lies
        }                                   // beyond 64 KiB: from one base the compiler built 18 address registers)
        wq2[0] = wa0 + (uint32_t)(par * TG * WPL * 16);
        wq2[1] = wa0 + (uint32_t)((par ^ 1) * TG * WPL * 16);
        asm volatile("" : "+v"(wq2[0]));
        asm volatile("" : "+v"(wq2[1]));
        constexpr int NR = 2 * VB + 2 * CB; // fragment reads per tap: 2 pieces x (VB operand + CB weight) fragments
        auto frag = [&](int tap, int s_, int r_) { // read r_ of tap `tap` into halves s_
          if (r_ < 2 * VB) {
            const int pc = r_ / VB, vb = r_ % VB;
            const int toff = ((tap / 9) * HH + (tap / 3) % 3) * HW + tap % 3;
            xf[s_][vb][pc] = *(lds_u4 *)(uintptr_t)(xq[vb] + (uint32_t)((pc * 2 * HP + toff) * 16));
          } else {
            const int pc = (r_ - 2 * VB) / CB, cb = (r_ - 2 * VB) % CB, k = tap / TG, t = tap % TG;
            wf[s_][cb][pc] = *(lds_u4 *)(uintptr_t)(wq2[k & 1] + (uint32_t)((t * WPL + pc * 2 * COT + cb * 32) * 16));
          }
        };
        Next nx = {false, 0, 0};
        if (wave_on) {
#pragma unroll
          for (int r_ = 0; r_ < NR; ++r_) frag(0, 0, r_);
        }
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
          const int k = tap / TG, cur = tap & 1, nxt = cur ^ 1;
          if (tap % TG == TG - 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's share of the weight group requested a barrier ago
            pc_barrier(); // k
            if (k == 0) nx = peek_next(cj + 1, q + 1 < nchunks, ico0, q + 1);
            // group k + 2 of the walk: into the buffer of group k (its last fragments are in registers)
            if (k + 2 < NB) weights_dma(ico0, q, k + 2, (k + par) & 1);
            else if (nx.have) weights_dma(nx.co0, nx.q, k + 2 - NB, (k + par) & 1);
          }
          if (wave_on) {
            // MFMA m of the tap: the 2 CB main products, the 2 CB X_lo products, the 2 CB W_lo products (two MFMAs on the
            // same accumulator are 2 CB issues apart)
            auto mfma = [&](int m) {
              const int kind = m / (2 * CB), cb = (m / 2) % CB, vb = m % 2;
              if (kind == 0) acc[cb][vb] = mma(wf[cur][cb][0], xf[cur][vb][0], acc[cb][vb]);
              else if (kind == 1) cor[cb][vb] = mma(wf[cur][cb][0], xf[cur][vb][1], cor[cb][vb]);
              else cor[cb][vb] = mma(wf[cur][cb][1], xf[cur][vb][0], cor[cb][vb]);
            };
            // The NR fragments of tap + 1 are requested one at a time between this tap's MFMAs (slots 1 .. NM - 1), into the
            // other halves: a wave draws 1 KiB per 32 cycles from LDS at best, so a burst of eight takes 256 cycles to land
            // -- spread out they are under way while the matrix pipe works, and the one lgkmcnt wait in front of the next
            // tap's first MFMA finds them there.  Round 2 read just in time: four exposed LDS round trips per tap, 65-70
            // cycles per MFMA of a wave against the pipe's 32.
            constexpr int NM = 6 * CB;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
              if (m >= 1 && tap + 1 < 27) {
#pragma unroll
                for (int r_ = (m - 1) * NR / (NM - 1); r_ < m * NR / (NM - 1); ++r_) frag(tap + 1, nxt, r_);
              }
              mfma(m);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        ++cj;
        ++s;
      }
      // the tile is complete: epilogue straight from the accumulators, no barrier
      {
        const float us_x = E == 127 ? 1.0f : pow2f(-E);
        const float *sa = sadd + ((nitem - 1) % 3) * add_rows * COT;
        const int cs = delta ? COT : 0;
        float *yb = y + ((size_t)ib * Cout + ico0) * r3;
        float *stp = STATS ? stats + (((size_t)ib * Cout + ico0) * st_tiles + itile * 4 + wq) * 2 : nullptr;
        if (wave_on) {
          pc_epilogue<TD, TH, TW, CB, STATS, false>(acc, cor, us_x, wscale_inv, [&](int cfg, int co) { return sa[cfg * cs + co]; },
                                                    wq, lane_c, yb, r, d0, h0, w0, stp, st_tiles * 2);
        } else {
          f32x16 za[CB][2], zc[CB][2];
          pc_epilogue<TD, TH, TW, CB, STATS, true>(za, zc, 1.f, 1.f, [&](int cfg, int co) { return sa[cfg * cs + co]; },
                                                   wq, lane_c, yb, r, d0, h0, w0, stp, st_tiles * 2);
        }
        if (rt == 0) ctl->chead = ch;
      }
    }
  }
}

template <int TD, int TH, int TW, int CB>
static int launch_pc_t(const float *x, const u4 *wp, const float *wtail, const float *bias, float *y, int B, int Cin, int Cout,
                       int r, const float *pa, const float *pb, const float *pbias, const float *tconst, float *stats,
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
  const unsigned grid = (unsigned)(items < cu_count[dev] ? items : cu_count[dev]);
  const bool pro_inst = pa != nullptr; // PRO also carries the delta form
  const int add_rows = (pa && tconst) ? 27 : 1;
  const size_t LDS = (size_t)(2 * 4 * HP + 2 * 3 * 4 * COT) * 16 +
                     (size_t)(3 * add_rows * COT + (pro_inst ? 3 * ((Cin + 63) & ~63) : 0)) * 4 +
                     PC_RING * sizeof(PcItem) + sizeof(PcCtl) + 16;
#define LION_PC_GO(PRO_, ST_)                                                                                \
  {                                                                                                          \
    static LionLdsLimit cfg = {};                                                                            \
    if (int e = lion_dynamic_lds(&conv3d_split_pc_kernel<TD, TH, TW, CB, PRO_, ST_>, LDS, cfg)) return e;      \
    conv3d_split_pc_kernel<TD, TH, TW, CB, PRO_, ST_><<<grid, 512, LDS, st>>>(x, wp, wtail, bias, y, Cin, Cout, r, pa, pb, \
                                                                          pbias, tconst, stats, occ, B, tiles); \
  }
  if (pro_inst && stats) LION_PC_GO(true, true)
  else if (pro_inst) LION_PC_GO(true, false)
  else if (stats) LION_PC_GO(false, true)
  else LION_PC_GO(false, false)
#undef LION_PC_GO
  LION_LAUNCH_CHECK();
  return 0;
}

} // namespace

// r in {16, 32}, Cout % 32 == 0, Cin % 16 == 0; tiles 2x4x32 (r = 32) and 4x4x16 (r = 16) -- the geometry of
// lion_conv3d_tile_occupancy's lists.  Statistics: 4 * tiles entries per (sample, channel).
__attribute__((visibility("hidden"))) int lion_split_pc_launch(const float *x, const void *wp, const float *wtail,
                                                               const float *bias, float *y, int B, int Cin, int Cout, int r,
                                                               const float *pa, const float *pb, const float *pbias,
                                                               const float *tconst, float *stats, int32_t *occ,
                                                               hipStream_t st) {
  const u4 *w4 = static_cast<const u4 *>(wp);
  const int cb = Cout % 64 == 0 ? 2 : 1; // (both tile shapes span the grid's width: TW == r)
  if (r == 32 && cb == 2) return launch_pc_t<2, 4, 32, 2>(x, w4, wtail, bias, y, B, Cin, Cout, r, pa, pb, pbias, tconst, stats, occ, st);
  if (r == 32 && cb == 1) return launch_pc_t<2, 4, 32, 1>(x, w4, wtail, bias, y, B, Cin, Cout, r, pa, pb, pbias, tconst, stats, occ, st);
  if (r == 16 && cb == 2) return launch_pc_t<4, 4, 16, 2>(x, w4, wtail, bias, y, B, Cin, Cout, r, pa, pb, pbias, tconst, stats, occ, st);
  if (r == 16 && cb == 1) return launch_pc_t<4, 4, 16, 1>(x, w4, wtail, bias, y, B, Cin, Cout, r, pa, pb, pbias, tconst, stats, occ, st);
  return LION_EUNSUPPORTED;
}
