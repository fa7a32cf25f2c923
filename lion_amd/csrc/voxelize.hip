// voxelize.hip -- average voxelization (K1, K2, K3) and the fused Voxelization.forward (P1).
//
// Reference: third_party/pvcnn/functional/src/voxelization/vox.cu:18-110, vox.cpp:17-79,
//            models/pvcnn2_ada.py:173-188.
//
// MI355X design (not the reference's "one block per cloud + global float atomics + memset").
// What was measured on the chip (tools/exp/*.hip) shaped it: a plain zero fill of the [B,C,r^3] grid
// with one 16-byte store per lane per channel reaches 7.5-8.0 TB/s; every global load that a store
// has to wait for costs dearly once the memory pipe is saturated with stores (vmcnt counts stores
// too: two dependent round trips in front of the stores -> 2.3 TB/s); random 4-byte gathers cost one
// TA cycle per lane and a 128-byte line per miss, so they must be spread over many CUs and kept on
// one XCD's L2 per cloud.  Hence ONE kernel whose store phase depends on LDS only:
//   vox_fused_kernel  grid = (cloud, slab of r^3/S voxels), 1024 threads, the S slabs of a cloud on
//   the same XCD.  A) voxel ids of all N points (optionally the fused P1 normalisation with the
//   oracle's fixed summation tree -> bit-exact indices), LDS histogram of the slab, count slab out;
//   scan + counting sort of the slab's points by (voxel, point index) in LDS.  B) per-voxel feature
//   means for a chunk of channels gathered from the L2-resident rows into LDS (ascending point
//   index -> bit-exact vs the sequential oracle; 8 independent gathers in flight per lane).
//   C) the slab of the dense grid is written exactly once: per channel one int4 slot read + 4 LDS
//   lookups + one 16-byte store per lane, no global load anywhere near the stores.
// No memset, no global atomics, no workspace.
//   fallback (r^3 or N too large for LDS, misaligned output): memset + integer/float atomics, like
//            the reference but with a (points x batch) grid.  Within 1e-6 of the oracle, not bit-exact.
#include "common.h"

namespace {

constexpr int VT = 1024;    // threads of the per-cloud index kernel
constexpr int MAXP = 8;     // points per thread kept in registers  (N <= 8192)
constexpr int LDS_LIMIT = 160 * 1024;
constexpr int PREFETCH_PAD = 2 * 16 * 4;   // bytes: 2 * PW words behind the arena (vox_means_and_store, phase B2)

__device__ __forceinline__ int padv(int v) { return v + (v >> 5); }
__host__ __device__ inline int align4i(int x) { return (x + 3) & ~3; }

// LDS counter increment of every valid lane's slot, aggregated per wave: the lanes that share a slot send ONE atomic (their
// number), each takes base + its rank among them.  The latents of the sampling chain put 100+ of a cloud's 2048 points into
// one voxel (round 5, tools/tile_shape_estimate.py --clouds: 2048 points in ~160 voxels): 64 lanes on one LDS address
// serialise, and the index kernel of such clouds took 73 us against 10.6 us on Gaussian clouds.  At most 8 groups are
// peeled (the loop stops after two singleton groups: a cloud without crowding pays two iterations); whatever is left issues
// plain atomics.  The arrival order only has to be SOME permutation -- the rank inside a bucket is recomputed from the
// point indices afterwards -- so the result is unchanged, bit for bit.
__device__ __forceinline__ int lds_slot_add_aggregated(int32_t *arr, int idx, bool valid, int lane) {
  int ret = 0;
  unsigned long long todo = __ballot(valid);
  int singles = 0;
#pragma unroll 1
  for (int it = 0; it < 8 && todo != 0ull && singles < 2; ++it) {
    const int leader = __ffsll((long long)todo) - 1;
    const int k = __shfl(idx, leader, 64);
    const unsigned long long m = __ballot(valid && idx == k) & todo;
    const bool mine = (m >> lane) & 1ull;
    int base = 0;
    if (lane == leader) base = atomicAdd(&arr[k], (int)__popcll(m));
    base = __shfl(base, leader, 64);
    if (mine) ret = base + (int)__popcll(m & ((1ull << lane) - 1ull));
    singles += __popcll(m) == 1 ? 1 : 0;
    todo &= ~m;
  }
  if ((todo >> lane) & 1ull) ret = atomicAdd(&arr[idx], 1);
  return ret;
}

// ---------------------------------------------------------------------------------------------
// vox_fused_kernel: see the header comment.  NP = points per thread (N <= NP * 1024).
// LDS (dynamic): slot[SVP] | ust[nw] | fscr[64] | iscr[32] | arena[arena_words]; during phase A the
// arena holds hist[SVP] | tmp[nw], afterwards prod[ch * my_pts] | vm[ch * my_occ].
// ---------------------------------------------------------------------------------------------
// Phases B + C of one (cloud, slab, channel range) workgroup, shared by the fused kernel and the plan-driven scatter
// kernel.  In LDS: slot[SV] dense (voxel -> rank among the slab's occupied voxels, -1 = empty), ust[my_occ]
// ((start << 16) | count per occupied voxel); per thread: the sorted position (or -1) and 1 / count of its NP points.
template <int NP, bool READ = false>
__device__ __forceinline__ void vox_means_and_store(
    const float *__restrict__ feat, float *__restrict__ out, int b, int C, int N, int r3, int lo, int SV, int cs, int CS,
    int my_pts, int my_occ, const int (&posv)[NP], const float (&invp)[NP], const int32_t *slot, const int32_t *ust,
    float *arena, int arena_words, int ch_cap, int c_begin = 0, int c_end = -1) {
  const int tid = threadIdx.x;
  const int q4 = SV >> 2; // int4 groups in the slab
  // rows of prod are ps = my_pts | 1 words apart: B2 reads them with one lane per CHANNEL (stride ps), an odd stride
  // touches every LDS bank once
  const int ps = my_pts | 1;
  const int ch_max = min(ch_cap, my_pts > 0 ? min(C, arena_words / (ps + my_occ)) : C);
  const int step_c = VT / q4, step_g = VT - step_c * q4;
  const int c_per = (C + CS - 1) / CS;
  // channels [c_lo, c_hi): the (cs, CS) range, or -- c_end >= 0 -- the caller's explicit range (one chunk of the scatter)
  const int c_lo = c_end >= 0 ? c_begin : cs * c_per, c_hi = c_end >= 0 ? c_end : min(C, cs * c_per + c_per);
  for (int c0 = c_lo; c0 < c_hi; c0 += ch_max) {
    const int ch = min(ch_max, c_hi - c0);
    float *prod = arena, *vm = arena + ch * ps;
    __syncthreads(); // hist/tmp (or the previous chunk's vm) are dead from here on
    if (my_pts > 0) {
      // B1: every lane reads ITS points' feature rows fully coalesced (the 8 slabs of a cloud share
      // the rows through one L2); only lanes whose point lies in the slab keep feat * (1 / count),
      // at the point's sorted position.  No gathers, no dependent loads.
      const float *fb = feat + ((size_t)b * C + c0) * N;
      constexpr int CU = 16 / NP; // 16 independent row loads in flight per lane
      for (int cl0 = 0; cl0 < ch; cl0 += CU) {
        float f[CU][NP];
#pragma unroll
        for (int k = 0; k < CU; ++k)
#pragma unroll
          for (int p = 0; p < NP; ++p)
            f[k][p] = fb[(size_t)min(cl0 + k, ch - 1) * N + min(tid + p * VT, N - 1)];
#pragma unroll
        for (int k = 0; k < CU; ++k)
#pragma unroll
          for (int p = 0; p < NP; ++p)
            if (posv[p] >= 0 && cl0 + k < ch) prod[(cl0 + k) * ps + posv[p]] = mul_rn(f[k][p], invp[p]);
      }
      __syncthreads();
      // B2: per-voxel means from LDS, summed in ascending point index (vox.cu:59-71).  Round 5: consecutive lanes take the
      // CHANNELS of one voxel (item = voxel * ch + channel), not consecutive voxels of one channel: the lanes of a wave then
      // walk runs of the same length.  With a voxel per lane, the one lane that held a crowded voxel (the chain's latents
      // put ~900 of a cloud's 2048 points into one) kept its wave for 900 dependent adds while 63 lanes waited, once per
      // channel: 22 of a workgroup's 38 us (profiles/archive/r05b_scatter_units_wallclock_log.txt).  The order of every sum is
      // unchanged -- bit-identical.
      const int items = my_occ * ch;
      for (int it = tid; it < items; it += VT) {
        const int u = it / ch, cl = it - u * ch;
        const int info = ust[u], st = info >> 16, n = info & 0xffff;
        const float *pr = prod + cl * ps + st;
        // the sum's ORDER is fixed, its operands do not depend on it: two buffers of PW, the next one is fetched while the
        // current one is added (reads past the run fetch words that are never added; PREFETCH_PAD keeps them inside the allocation)
        float acc = add_rn(0.f, pr[0]);
        if (n <= 8) {   // (nearly every voxel of a Gaussian cloud)
          for (int k = 1; k < n; ++k) acc = add_rn(acc, pr[k]);
          vm[cl * my_occ + u] = acc;
          continue;
        }
        // (8 deep, the 1900-point voxel of a chain cloud costs 21 cycles per add: LDS latency shows; the 1024-point
        // instantiation has no registers for more -- 16 deep it loses its second workgroup per CU)
        constexpr int PW = (NP == 2 || NP == 4) ? 16 : 8;   // (NP = 8, 1024 threads x <= 128 registers: 16 deep spills)
        float ta[PW], tb[PW];
#pragma unroll
        for (int j = 0; j < PW; ++j) ta[j] = pr[1 + j];
        int k = 1;
        for (;;) {
          if (k + PW > n) {
#pragma unroll
            for (int j = 0; j < PW; ++j)
              if (k + j < n) acc = add_rn(acc, ta[j]);
            break;
          }
#pragma unroll
          for (int j = 0; j < PW; ++j) tb[j] = pr[k + PW + j];
#pragma unroll
          for (int j = 0; j < PW; ++j) acc = add_rn(acc, ta[j]);
          k += PW;
          if (k + PW > n) {
#pragma unroll
            for (int j = 0; j < PW; ++j)
              if (k + j < n) acc = add_rn(acc, tb[j]);
            break;
          }
#pragma unroll
          for (int j = 0; j < PW; ++j) ta[j] = pr[k + PW + j];
#pragma unroll
          for (int j = 0; j < PW; ++j) acc = add_rn(acc, tb[j]);
          k += PW;
        }
        vm[cl * my_occ + u] = acc;
      }
    }
    __syncthreads();
    // C: the slab of the dense grid for channels [c0, c0 + ch): LDS -> 16-byte stores only
    int cl = tid / q4, g = tid - cl * q4;
    float *obase = out + ((size_t)b * C + c0) * r3 + lo;
    while (cl < ch) {
      const int4 sl = *reinterpret_cast<const int4 *>(slot + 4 * g);
      const float *vmc = vm + cl * my_occ;
      const float x = vmc[max(sl.x, 0)], y = vmc[max(sl.y, 0)], z = vmc[max(sl.z, 0)], w = vmc[max(sl.w, 0)];
      float4 o;
      o.x = sl.x >= 0 ? x : 0.f;
      o.y = sl.y >= 0 ? y : 0.f;
      o.z = sl.z >= 0 ? z : 0.f;
      o.w = sl.w >= 0 ? w : 0.f;
      // slot -2 (lion_voxel_scatter_read): a z-row no convolution tile stages -- its zeros have no reader, nothing is stored
      if (!READ || sl.x != -2) *reinterpret_cast<float4 *>(obase + (size_t)cl * r3 + 4 * g) = o;
      cl += step_c; g += step_g;
      if (g >= q4) { g -= q4; ++cl; }
    }
  }
}

// The index plan of a (coordinates, resolution) pair -- what phase A computes, kept in memory so that every voxelisation
// of the same cloud at the same resolution (a forward of the denoiser voxelises 4 distinct (cloud, r) pairs 14 times)
// starts at phase B.  int32 words, per cloud b and slab s (Ncap = align4(N)):
//   pos [B][N]          (slab << 16) | position of the point in its slab's (voxel, point index) order
//   inv [B][N]          float 1 / count of the point's voxel
//   hdr [B][S][4]       points in the slab, occupied voxels in the slab
//   ust [B][S][Ncap]    (start << 16) | count per occupied voxel, ascending voxel id
//   uvl [B][S][Ncap]    the occupied voxel's id relative to the slab
struct VoxPlanPtrs {
  int32_t *pos;
  float *inv;
  int32_t *hdr, *ust, *uvl;
};
__host__ __device__ inline size_t vox_plan_words(int B, int N, int S) {
  return (size_t)B * (2 * (size_t)N + (size_t)S * (4 + 2 * (size_t)align4i(N)));
}
__host__ __device__ inline VoxPlanPtrs vox_plan_ptrs(void *base, int B, int N, int S) {
  VoxPlanPtrs q;
  int32_t *w = static_cast<int32_t *>(base);
  q.pos = w; w += (size_t)B * N;
  q.inv = reinterpret_cast<float *>(w); w += (size_t)B * N;
  q.hdr = w; w += (size_t)B * S * 4;
  q.ust = w; w += (size_t)B * S * align4i(N);
  q.uvl = w;
  return q;
}

template <bool FUSE_P1, int NP>
__global__ __launch_bounds__(VT) void vox_fused_kernel(
    const float *__restrict__ feat, const int32_t *__restrict__ coords_i,
    const float *__restrict__ coords_f, int B, int C, int N, int r, int S, int CS, int SV, int n_words,
    int arena_words, int ch_cap, int normalize, float eps, float *__restrict__ out,
    float *__restrict__ norm_coords, int32_t *__restrict__ ind, int32_t *__restrict__ cnt, void *plan) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int SVP = ((SV + (SV >> 5) + 3) & ~3) + 4;         // padded (1 word per 32) + 16-byte aligned
  int32_t *slot = reinterpret_cast<int32_t *>(smem);       // [SVP] arrival counters, later dense slots
  int32_t *ust = slot + SVP;                               // [nw] (start << 16) | count per occupied voxel
  float *fscr = reinterpret_cast<float *>(ust + n_words);  // 64
  int *iscr = reinterpret_cast<int *>(fscr + 64);          // 32
  float *arena = reinterpret_cast<float *>(iscr + 32);     // [arena_words]
  uint32_t *hist = reinterpret_cast<uint32_t *>(arena);    // [SVP] count, later (urank << 16) | count
  int32_t *tmp = reinterpret_cast<int32_t *>(hist + SVP);  // [nw] bucket contents in arrival order

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // workgroups go round-robin over the 8 XCDs: keep the S slabs of a cloud on one XCD so that
  // its feature rows are fetched into one L2 only
  const int W = S * CS; // workgroups per cloud: S voxel slabs x CS channel ranges
  const int L = blockIdx.x, grp = L / (8 * W), j8 = L - grp * 8 * W;
  const int b = grp * 8 + (j8 & 7), wq = j8 >> 3, slab = wq % S, cs = wq / S;
  if (b >= B) return;
  const int r2 = r * r, r3 = r2 * r;
  const int lo = slab * SV;

  for (int v = tid; v < SVP; v += VT) { hist[v] = 0u; slot[v] = 0; }

  int myv[NP];
  if (FUSE_P1) {
    const float *co = coords_f + (size_t)b * 3 * N;
    float px[NP], py[NP], pz[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int i = tid + p * VT;
      px[p] = py[p] = pz[p] = 0.0f;
      if (i < N) { px[p] = co[i]; py[p] = co[i + N]; pz[p] = co[i + 2 * N]; }
    }
    // mean over the points with the oracle's fixed tree: 1024 strided partials (ascending j),
    // then p[t] += p[t+s], s = 1..512.  An xor butterfly evaluates the same tree bit for bit.
    float part[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int i = tid + p * VT;
      if (i < N) {
        part[0] = add_rn(part[0], px[p]);
        part[1] = add_rn(part[1], py[p]);
        part[2] = add_rn(part[2], pz[p]);
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      part[a] = row16_sum_rn(part[a]); // s = 1, 2, 4, 8 (same tree, DPP); then across the 4 rows
#pragma unroll
      for (int s = 16; s < 64; s <<= 1) part[a] = add_rn(part[a], __shfl_xor(part[a], s, 64));
      if (lane == 0) fscr[a * 16 + wave] = part[a];
    }
    __syncthreads();
    float mean[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float v = fscr[a * 16 + (lane & 15)];
      v = row16_sum_rn(v);
      mean[a] = div_rn(v, (float)N);
    }
    float denom = 1.0f;
    if (normalize) {
      float mx = 0.0f;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int i = tid + p * VT;
        if (i < N) {
          const float x = sub_rn(px[p], mean[0]), y = sub_rn(py[p], mean[1]),
                      z = sub_rn(pz[p], mean[2]);
          const float nr = sqrt_rn(add_rn(add_rn(mul_rn(x, x), mul_rn(y, y)), mul_rn(z, z)));
          mx = nr > mx ? nr : mx;
        }
      }
      mx = row16_max(mx);
#pragma unroll
      for (int s = 16; s < 64; s <<= 1) { const float o = __shfl_xor(mx, s, 64); mx = o > mx ? o : mx; }
      if (lane == 0) fscr[48 + wave] = mx;
      __syncthreads();
      float m2 = fscr[48 + (lane & 15)];
      m2 = row16_max(m2);
      denom = add_rn(mul_rn(m2, 2.0f), eps);
    }
    float *nc = norm_coords + (size_t)b * 3 * N;
    const float rf = (float)r, hi = (float)(r - 1);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int i = tid + p * VT;
      myv[p] = 0;
      if (i < N) {
        float c3[3] = {px[p], py[p], pz[p]};
        int q[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          float v = sub_rn(c3[a], mean[a]);
          v = normalize ? add_rn(div_rn(v, denom), 0.5f) : div_rn(add_rn(v, 1.0f), 2.0f);
          v = mul_rn(v, rf);
          v = v < 0.0f ? 0.0f : v;
          v = v > hi ? hi : v;
          if (wq == 0) nc[i + a * N] = v;
          q[a] = (int)rintf(v);
        }
        myv[p] = q[0] * r2 + q[1] * r + q[2];
      }
    }
  } else {
    const int32_t *co = coords_i + (size_t)b * 3 * N;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int i = tid + p * VT;
      myv[p] = 0;
      if (i < N) myv[p] = co[i] * r2 + co[i + N] * r + co[i + 2 * N]; // vox.cu:31
    }
  }
  __syncthreads(); // hist zeroed
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int i = tid + p * VT;
    if (i < N) {
      if (wq == 0) ind[(size_t)b * N + i] = myv[p];
      // memory safety only: coordinates outside [0,r) are a caller error (as in the reference)
      myv[p] = min(max(myv[p], 0), r3 - 1) - lo;
    }
    const bool in_slab = i < N && myv[p] >= 0 && myv[p] < SV;
    lds_slot_add_aggregated(reinterpret_cast<int32_t *>(hist), in_slab ? padv(myv[p]) : 0, in_slab, lane);
  }
  __syncthreads();
  // dense count slab (16 bytes per lane)
  for (int v = tid * 4; v < SV && cs == 0; v += VT * 4)
    *reinterpret_cast<int4 *>(cnt + (size_t)b * r3 + lo + v) =
        make_int4((int)hist[padv(v)], (int)hist[padv(v + 1)], (int)hist[padv(v + 2)], (int)hist[padv(v + 3)]);
  if (feat == nullptr && plan == nullptr) return;
  VoxPlanPtrs pl = {};
  if (plan) pl = vox_plan_ptrs(plan, B, N, S);
  int32_t *uvl_g = plan ? pl.uvl + ((size_t)b * S + slab) * n_words : nullptr;

  // ---- scan: rank among the slab's occupied voxels + start in the sorted list -------------------
  const int VPT = (SV + VT - 1) / VT;
  const int v0 = min(tid * VPT, SV), v1 = min(v0 + VPT, SV);
  int lp = 0, lq = 0;
  for (int v = v0; v < v1; ++v) { const int c = (int)hist[padv(v)]; lp += c; lq += (c > 0); }
  const int packed = (lp << 16) | lq;
  const int incl = wave_incl_scan(packed, lane);
  if (lane == 63) iscr[wave] = incl;
  __syncthreads();
  int excl = incl - packed, total = 0;
  for (int w = 0; w < VT / 64; ++w) { const int t = iscr[w]; if (w < wave) excl += t; total += t; }
  const int my_pts = total >> 16, my_occ = total & 0xffff;
  {
    int run = excl >> 16, urun = excl & 0xffff;
    for (int v = v0; v < v1; ++v) {
      const uint32_t c = hist[padv(v)];
      if (c) {
        hist[padv(v)] = ((uint32_t)urun << 16) | c;
        ust[urun] = (run << 16) | (int)c;
        if (uvl_g) uvl_g[urun] = v;
        ++urun;
        run += (int)c;
      }
    }
  }
  __syncthreads();
  // ---- counting sort by (voxel, point index): arrival order, then the rank inside the bucket ----
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int i = tid + p * VT;
    const bool in_slab = i < N && myv[p] >= 0 && myv[p] < SV;
    const int a = lds_slot_add_aggregated(slot, in_slab ? padv(myv[p]) : 0, in_slab, lane);
    if (in_slab) tmp[(ust[hist[padv(myv[p])] >> 16] >> 16) + a] = i;
  }
  __syncthreads();
  // rank inside the bucket = number of smaller point indices -> position in the (voxel, index) order
  int posv[NP];
  float invp[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int i = tid + p * VT;
    posv[p] = -1;
    invp[p] = 0.f;
    if (i < N && myv[p] >= 0 && myv[p] < SV) {
      const int h = ust[hist[padv(myv[p])] >> 16];
      const int s = h >> 16, c = h & 0xffff;
      // crowded voxels (real latents put 100+ points into one voxel): 8 independent LDS reads per round trip instead
      // of one dependent read per step
      int rank = 0, q = 0;
      for (; q + 8 <= c; q += 8) {
        int t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = tmp[s + q + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) rank += (t[j] < i) ? 1 : 0;
      }
      for (; q < c; ++q) rank += (tmp[s + q] < i) ? 1 : 0;
      posv[p] = s + rank;
      invp[p] = div_rn(1.0f, (float)c); // vox.cu:66 (== float(1.0 / cnt))
    }
  }
  if (plan) {   // index mode: everything the scatter kernel needs to start at phase B
    if (tid == 0) {
      int32_t *h = pl.hdr + ((size_t)b * S + slab) * 4;
      h[0] = my_pts; h[1] = my_occ; h[2] = 0; h[3] = 0;
    }
    int32_t *ust_g = pl.ust + ((size_t)b * S + slab) * n_words;
    for (int u = tid; u < my_occ; u += VT) ust_g[u] = ust[u];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int i = tid + p * VT;
      if (i < N && posv[p] >= 0) {
        pl.pos[(size_t)b * N + i] = (slab << 16) | posv[p];
        pl.inv[(size_t)b * N + i] = invp[p];
      }
    }
    if (feat == nullptr) return;
  }
  // dense slots, unpadded so that phase C reads them 16 bytes at a time
  for (int v = tid; v < SV; v += VT) {
    const uint32_t h = hist[padv(v)];
    slot[v] = (h & 0xffff) ? (int)(h >> 16) : -1;
  }
  vox_means_and_store<NP>(feat, out, b, C, N, r3, lo, SV, cs, CS, my_pts, my_occ, posv, invp, slot, ust, arena,
                          arena_words, ch_cap);
}

// vox_scatter_kernel: phases B + C from a stored plan.  LDS (dynamic): slot[SV] | ust[n_words] | arena.
template <int NP, bool READ>
__global__ __launch_bounds__(VT) void vox_scatter_kernel(
    const float *__restrict__ feat, const void *__restrict__ plan, int B, int C, int N, int r3, int S, int CS, int SV,
    int n_words, int arena_words, int ch_cap, float *__restrict__ out, const int32_t *__restrict__ occ_flags, int r, int TD,
    int TH) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int32_t *slot = reinterpret_cast<int32_t *>(smem);
  int32_t *ust = slot + SV;
  int32_t *hdr_s = ust + n_words;   // [16][4]: points, occupied voxels, channels per chunk, chunks left with the owner
  float *arena = reinterpret_cast<float *>(hdr_s + 64);
  const int tid = threadIdx.x;
  const int W = S * CS;
  const int L = blockIdx.x, grp = L / (8 * W), j8 = L - grp * 8 * W;
  const int b = grp * 8 + (j8 & 7), wq = j8 >> 3, slab = wq % S, cs = wq / S;
  if (b >= B) return;
  const VoxPlanPtrs pl = vox_plan_ptrs(const_cast<void *>(plan), B, N, S);
  // Round 5 -- empty slabs adopt the extra chunks of crowded ones.  A slab whose points x channels do not fit the LDS arena
  // walks its channel range in chunks, one after the other in ONE workgroup; the latents of the sampling chain sit in one or
  // two slabs of a cloud (up to ~1900 of 2048 points in a single voxel) and leave most of the others without a point, so
  // that workgroup ran four chunks -- each with a ~1900-long chain of ordered adds -- while its neighbours were done after
  // 10 us of zeros: (64, 2048, 32) took 110 us on the chain's clouds against 61 us on Gaussian ones
  // (profiles/archive/r05b_scatter_units_wallclock_log.txt).  Now the cloud's workgroups whose own slab is EMPTY take over chunks
  // of the crowded slabs (and run them first, then their zeros); a cloud without an empty slab -- nearly every Gaussian
  // one -- keeps the old assignment.  Same arithmetic per (voxel, channel): bit-identical.
  const int c_per = (C + CS - 1) / CS, cg_lo = cs * c_per, cg_hi = min(C, cg_lo + c_per);
  if (cg_lo >= cg_hi) return;
  // one round trip for the cloud's S headers: lane s of every wave holds slab s's (points, occupied voxels, channels per
  // chunk, chunks); the scans below read them with lane broadcasts
  int l_pts = 0, l_occ = 0, l_chm = 1, l_nc = 0;
  {
    const int ls = tid & 63;
    if (ls < S) {
      const int32_t *hs = pl.hdr + ((size_t)b * S + ls) * 4;
      l_pts = hs[0];
      l_occ = hs[1];
      l_chm = min(ch_cap, l_pts > 0 ? min(C, arena_words / ((l_pts | 1) + l_occ)) : C);
      l_nc = l_pts > 0 ? (cg_hi - cg_lo + l_chm - 1) / l_chm : 1;
    }
  }
  // every point's (slab << 16 | sorted position) and 1 / count: in flight together with the headers, filtered per chunk
  int posw[NP];
  float invw[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int i = min(tid + p * VT, N - 1);
    posw[p] = (tid + p * VT < N) ? pl.pos[(size_t)b * N + i] : -1;   // (slab field 0xffff: nobody's)
    invw[p] = pl.inv[(size_t)b * N + i];
  }
  const unsigned long long empt = __ballot((tid & 63) < S && l_pts == 0);   // bit s: slab s has no point
  // Who runs what, computed alike by every workgroup of the cloud (lane s = slab s: chunks still with their owner / load of
  // an adopter in half chunks, its zeros counting one).  Greedy: the last chunk of the slab with the most chunks left goes
  // to the least loaded adopter as long as the move lowers the larger of the two; an adopter takes at most five.
  // Clouds of more than 1024 points only (NP >= 2): the 1024-point instantiation runs two workgroups per CU and the
  // registers of this block cost it that -- (128, 1024, 16) on Gaussian clouds 31 -> 38 us with it, for 54 -> 41 us on the
  // chain's clouds and nothing on the chain's step (profiles/archive/r05b_scatter_adoption_ab.txt).
  int rem = l_nc, ld2 = (((empt >> (tid & 63)) & 1) != 0) ? 1 : 0x7fffff, taken = 0;
  unsigned long long adopted = 0;   // the chunks this workgroup adopted: 12 bits each, (slab << 8) | chunk
  int n_adopted = 0;
#ifndef LION_VOXS_ADOPT     // A/B builds: tools/build_variant.sh noadopt 'voxelize:-DLION_VOXS_ADOPT=0'
#define LION_VOXS_ADOPT 1
#endif
  if (LION_VOXS_ADOPT && NP >= 2 && empt != 0) {
    for (int it = 0; it < 5 * 16; ++it) {
      int km = (rem << 8) | (63 - (tid & 63)), ka = (ld2 << 8) | (tid & 63);
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) {
        km = max(km, __shfl_xor(km, m));
        ka = min(ka, __shfl_xor(ka, m));
      }
      km = __builtin_amdgcn_readfirstlane(km);
      ka = __builtin_amdgcn_readfirstlane(ka);
      const int smax = 63 - (km & 0xff), nmax = km >> 8, a_ = ka & 0xff, la = ka >> 8;
      if (nmax < 2 || nmax > 256 || la + 2 >= 2 * nmax) break;   // (a chunk index has 8 bits in `adopted`)
      if ((tid & 63) == smax) --rem;
      if ((tid & 63) == a_) {
        ld2 += 2;
        if (++taken == 5) ld2 = 0x7fffff;
      }
      if (a_ == slab) adopted |= (unsigned long long)((smax << 8) | (nmax - 1)) << (12 * n_adopted++);
    }
  }
  // the headers move to LDS: held in lanes across the chunk loop they cost the 1024-point instantiation 14 registers and
  // (128, 1024, 16) went from 31 to 37 us
  if (tid < S) *reinterpret_cast<int4 *>(hdr_s + 4 * tid) = make_int4(l_pts, l_occ, l_chm, rem);
  __syncthreads();
  const int own_nc = __builtin_amdgcn_readfirstlane(hdr_s[4 * slab + 3]);
  n_adopted = __builtin_amdgcn_readfirstlane(n_adopted);   // (uniform by construction; said so for the register allocator)
  for (int t = 0; t <= n_adopted; ++t) {
  // adopted chunks first: the zeros of the own slab then drain behind nobody
  const int slab_u = __builtin_amdgcn_readfirstlane(t < n_adopted ? (int)((adopted >> (12 * t + 8)) & 0xf) : slab);
  const int k_u = __builtin_amdgcn_readfirstlane(t < n_adopted ? (int)((adopted >> (12 * t)) & 0xff) : 0);
  // (wave-uniform values: kept in scalar registers -- as vector registers they cost the 1024-point instantiation its
  // second workgroup per CU)
  const int my_pts = __builtin_amdgcn_readfirstlane(hdr_s[4 * slab_u]), my_occ = __builtin_amdgcn_readfirstlane(hdr_s[4 * slab_u + 1]),
            chm_u = __builtin_amdgcn_readfirstlane(hdr_s[4 * slab_u + 2]);
  // the own slab: the chunks that were not given away (vox_means_and_store walks the range)
  const int c_begin = cg_lo + k_u * chm_u,
            c_end = t < n_adopted ? min(cg_hi, c_begin + chm_u) : my_pts > 0 ? min(cg_hi, cg_lo + own_nc * chm_u) : cg_hi;
  if (t > 0) __syncthreads(); // the previous chunk's store phase is done with slot / the arena
  int posv[NP];
  float invp[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const bool mine = (posw[p] >> 16) == slab_u;
    posv[p] = mine ? (posw[p] & 0xffff) : -1;
    invp[p] = mine ? invw[p] : 0.f;
  }
  if constexpr (READ) {
    // Round 5: the grid's only reader is the sparse convolution that pops occ_flags (lion_conv3d_tile_occupancy, margin 1):
    // it stages the halo -- one voxel in d and h, all of w -- of the tiles with a point within one voxel, nothing else.  A
    // z-row (d, h) outside every such halo is never read: its voxels (all empty -- a point's own row lies inside an occupied
    // tile) get slot -2 and phase C stores nothing there.  The chain's clouds leave ~70 % of an r = 32 grid unread.
    __shared__ unsigned char rowneed[1024];   // rows of this slab (SV / r <= 1024)
    const int nth = r / TH, ntiles = (r / TD) * nth, row0 = (slab_u * SV) / r, nrows = SV / r;
    const int32_t *fl = occ_flags + (size_t)b * ntiles;
    for (int q = tid; q < nrows; q += VT) {
      const int d = (row0 + q) / r, h = (row0 + q) % r;
      int need = 0;
      for (int dd = max(d - 1, 0); dd <= min(d + 1, r - 1); ++dd)
        for (int hh = max(h - 1, 0); hh <= min(h + 1, r - 1); ++hh) need |= fl[(dd / TD) * nth + hh / TH] & 0xf;
      rowneed[q] = need != 0;
    }
    __syncthreads();
    for (int v = tid; v < SV; v += VT) slot[v] = rowneed[v / r] ? -1 : -2;
  } else {
    for (int v = tid; v < SV; v += VT) slot[v] = -1;
  }
  __syncthreads();
  const int32_t *ust_g = pl.ust + ((size_t)b * S + slab_u) * n_words;
  const int32_t *uvl_g = pl.uvl + ((size_t)b * S + slab_u) * n_words;
  for (int u = tid; u < my_occ; u += VT) {
    ust[u] = ust_g[u];
    slot[uvl_g[u]] = u;
  }
  vox_means_and_store<NP, READ>(feat, out, b, C, N, r3, slab_u * SV, SV, cs, CS, my_pts, my_occ, posv, invp, slot, ust, arena,
                                arena_words, ch_cap, c_begin, c_end);
  } // chunks of this workgroup
}

// ---------------------------------------------------------------------------------------------
// Fallback path (any r, any N): integer atomics for the counts, float atomics for the scatter.
// ---------------------------------------------------------------------------------------------
__global__ void vox_index_atomic_kernel(const int32_t *__restrict__ coords, int N, int r,
                                        int32_t *__restrict__ ind, int32_t *__restrict__ cnt) {
  const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int r2 = r * r, r3 = r2 * r;
  const int32_t *co = coords + (size_t)b * 3 * N;
  const int v = co[i] * r2 + co[i + N] * r + co[i + 2 * N];
  ind[(size_t)b * N + i] = v;
  atomicAdd(cnt + (size_t)b * r3 + min(max(v, 0), r3 - 1), 1);
}

__global__ void vox_scatter_atomic_kernel(const float *__restrict__ feat,
                                          const int32_t *__restrict__ ind,
                                          const int32_t *__restrict__ cnt, int C, int N, int r3,
                                          int CT, float *__restrict__ out) {
  const int b = blockIdx.z, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int pos = min(max(ind[(size_t)b * N + i], 0), r3 - 1);
  const int cur = cnt[(size_t)b * r3 + pos];
  if (cur <= 0) return;
  const float inv = div_rn(1.0f, (float)cur);
  const int c0 = blockIdx.y * CT, c1 = min(C, c0 + CT);
  for (int c = c0; c < c1; ++c)
    atomicAdd(out + ((size_t)b * C + c) * r3 + pos, mul_rn(feat[((size_t)b * C + c) * N + i], inv));
}

// Stand-alone P1 for the fallback path: one 1024-thread workgroup per cloud, strided loops.
__global__ __launch_bounds__(VT) void p1_kernel(const float *__restrict__ coords, int N, int r,
                                                int normalize, float eps,
                                                float *__restrict__ norm_coords,
                                                int32_t *__restrict__ vox) {
  __shared__ float fscr[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const float *co = coords + (size_t)b * 3 * N;
  float mean[3];
  for (int a = 0; a < 3; ++a) {
    float part = 0.f;
    for (int k = tid; k < N; k += VT) part = add_rn(part, co[k + (size_t)a * N]);
    for (int s = 1; s < 64; s <<= 1) part = add_rn(part, __shfl_xor(part, s, 64));
    __syncthreads();
    if (lane == 0) fscr[wave] = part;
    __syncthreads();
    float v = fscr[lane & 15];
    for (int s = 1; s < 16; s <<= 1) v = add_rn(v, __shfl_xor(v, s, 64));
    mean[a] = div_rn(v, (float)N);
  }
  float denom = 1.0f;
  if (normalize) {
    float mx = 0.f;
    for (int k = tid; k < N; k += VT) {
      const float x = sub_rn(co[k], mean[0]), y = sub_rn(co[k + N], mean[1]),
                  z = sub_rn(co[k + 2 * (size_t)N], mean[2]);
      const float nr = sqrt_rn(add_rn(add_rn(mul_rn(x, x), mul_rn(y, y)), mul_rn(z, z)));
      mx = nr > mx ? nr : mx;
    }
    for (int s = 1; s < 64; s <<= 1) { const float o = __shfl_xor(mx, s, 64); mx = o > mx ? o : mx; }
    __syncthreads();
    if (lane == 0) fscr[wave] = mx;
    __syncthreads();
    float m2 = fscr[lane & 15];
    for (int s = 1; s < 16; s <<= 1) { const float o = __shfl_xor(m2, s, 64); m2 = o > m2 ? o : m2; }
    denom = add_rn(mul_rn(m2, 2.0f), eps);
  }
  const float rf = (float)r, hi = (float)(r - 1);
  for (int a = 0; a < 3; ++a)
    for (int k = tid; k < N; k += VT) {
      float v = sub_rn(co[k + (size_t)a * N], mean[a]);
      v = normalize ? add_rn(div_rn(v, denom), 0.5f) : div_rn(add_rn(v, 1.0f), 2.0f);
      v = mul_rn(v, rf);
      v = v < 0.0f ? 0.0f : v;
      v = v > hi ? hi : v;
      norm_coords[(size_t)b * 3 * N + k + (size_t)a * N] = v;
      vox[(size_t)b * 3 * N + k + (size_t)a * N] = (int)rintf(v);
    }
}

// K3: gx[b,c,i] = gy[b,c,ind[i]] * (1/cnt)   (single writer, no atomics needed)
__global__ __launch_bounds__(256) void vox_grad_kernel(const float *__restrict__ gy,
                                                       const int32_t *__restrict__ ind,
                                                       const int32_t *__restrict__ cnt, int C,
                                                       int N, int r3, int CT,
                                                       float *__restrict__ gx) {
  const int b = blockIdx.z, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int pos = min(max(ind[(size_t)b * N + i], 0), r3 - 1);
  const int cur = cnt[(size_t)b * r3 + pos];
  const float inv = cur > 0 ? div_rn(1.0f, (float)cur) : 0.0f;
  const int c0 = blockIdx.y * CT, c1 = min(C, c0 + CT);
  for (int c = c0; c < c1; ++c) {
    const float g = cur > 0 ? mul_rn(gy[((size_t)b * C + c) * r3 + pos], inv) : 0.0f;
    gx[((size_t)b * C + c) * N + i] = g;
  }
}

struct VoxPlan {
  bool fast;
  int S, CS, SV, n_words, arena_words, NP;
  size_t lds;
  size_t off_vox, total;
};

static inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

static VoxPlan make_plan(int B, int C, int N, int r) {
  VoxPlan p;
  const long r3 = (long)r * r * r;
  // Clouds of <= 1024 points (one point per thread: 42 registers, 8 waves per SIMD) run TWO workgroups per CU on half the
  // LDS each: one's index / mean phases overlap the other's store phase ((128,1024,16), B = 32: 41 -> 35 us).  The
  // 2048-point kernel needs 74 registers (one workgroup per CU); forced to 62 and split into 16 slabs it is slower
  // (70 -> 74 us), so r = 32 keeps one workgroup per CU.
  const bool two_per_cu = N <= VT;
  const long wgs = two_per_cu ? 512 : 256;
  // PREFETCH_PAD: phase B2's double-buffered operand fetch reads up to 2 * PW (= 32) words past a voxel's run; for the last
  // row of the arena that is past the arena itself -- the words are never added, but they are allocated
  const size_t lds_limit = (two_per_cu ? (size_t)LDS_LIMIT / 2 : (size_t)LDS_LIMIT) - PREFETCH_PAD;
  // slabs per cloud: enough workgroups to touch every CU, slabs of >= 512 voxels (int4 groups)
  int S = 1;
  while (S < 16 && (long)B * S < wgs && r3 / (S * 2) >= 512 && (r3 % (S * 2 * 4)) == 0) S *= 2;
  p.S = S;
  // ... and, for small grids, ranges of channels (each workgroup redoes the cheap index phase)
  int CS = 1;
  while ((long)B * S * CS < wgs && C / (CS * 2) >= 8) CS *= 2;
  p.CS = CS;
  p.SV = (int)(r3 / S);
  p.n_words = align4i(N);
  p.NP = N <= VT ? 1 : N <= 2 * VT ? 2 : N <= 4 * VT ? 4 : 8;
  const size_t svp = (((size_t)p.SV + (p.SV >> 5) + 3) & ~(size_t)3) + 4;
  const size_t fixed = (svp + (size_t)p.n_words + 64 + 32) * 4;
  // arena: phase A needs hist + tmp; afterwards (points + occupied voxels of the slab) floats per
  // channel, for as many channels as fit
  const size_t need_a = (svp + (size_t)p.n_words) * 4;
  const size_t nocc = (size_t)(N < p.SV ? N : p.SV);
  size_t want = ((size_t)N + 1 + nocc) * (C > 0 ? C : 1) * 4;   // (rows of N | 1 words: vox_means_and_store)
  if (want < need_a) want = need_a;
  const size_t avail = fixed < lds_limit ? lds_limit - fixed : 0;
  const size_t arena = want < avail ? want : avail;
  p.arena_words = (int)(arena / 4);
  p.lds = fixed + (size_t)p.arena_words * 4 + PREFETCH_PAD;
  p.fast = r3 <= (1L << 17) && (r3 % 4 == 0) && (p.SV % 4 == 0) && N <= MAXP * VT && N >= 1 &&
           arena >= need_a && (size_t)p.arena_words >= (size_t)N + 1 + nocc;
  size_t o = 0;
  p.off_vox = o; o += up256((size_t)B * 3 * N * 4); // int coords (fallback P1)
  p.total = o;
  return p;
}

static int voxelize_impl(const float *feat, const int32_t *coords_i, const float *coords_f, int B,
                         int C, int N, int r, int normalize, float eps, float *out,
                         float *norm_coords, int32_t *ind, int32_t *cnt, void *ws, size_t ws_bytes,
                         hipStream_t st, void *plan = nullptr) {
  if (B <= 0 || N <= 0 || r <= 0 || !ind || !cnt) return LION_EINVAL;
  if (feat && (C <= 0 || !out)) return LION_EINVAL;
  if (!coords_i && !coords_f) return LION_EINVAL;
  if (coords_f && !norm_coords) return LION_EINVAL;
  if ((long)r * r * r > (1L << 30)) return LION_EUNSUPPORTED;
  const VoxPlan p = make_plan(B, feat ? C : 0, N, r);
  if (plan) {   // index mode (lion_voxel_index): fast path only, no workspace
    if (!p.fast || ((((uintptr_t)cnt) & 15) != 0)) return LION_EUNSUPPORTED;
  } else if (!ws || ws_bytes < p.total) return LION_EWORKSPACE;
  const int r3 = r * r * r;
  char *w = static_cast<char *>(ws);
  const bool aligned = (((uintptr_t)out | (uintptr_t)cnt) & 15) == 0;   // (out == nullptr in index mode)
  if (p.fast && aligned) {
    const int grid = ((B + 7) / 8) * 8 * p.S * p.CS;
    const int ch_cap = 64; // channel chunk (as many as the arena holds).  Round 3 sweep at (64,2048,32), us with / without P1: 8: 77.5 / 75.3, 16: 72.9 / 67.6, 24: 65.9 / 63.4, 32: 66.7 / 64.0, 64: 64.3 / 62.1
#define LION_VOX_LAUNCH(P1, NPV)                                                                       \
  {                                                                                                    \
    static LionLdsLimit cfg = {};                                                                      \
    if (int e = lion_dynamic_lds(&vox_fused_kernel<P1, NPV>, p.lds, cfg)) return e;                    \
    vox_fused_kernel<P1, NPV><<<grid, VT, p.lds, st>>>(feat, coords_i, coords_f, B, C, N, r, p.S, p.CS, p.SV, \
                                                       p.n_words, p.arena_words, ch_cap, normalize, eps, out, \
                                                       norm_coords, ind, cnt, plan);                   \
  }
#define LION_VOX_NP(P1)                                                                                \
  switch (p.NP) {                                                                                      \
  case 1: LION_VOX_LAUNCH(P1, 1) break;                                                                \
  case 2: LION_VOX_LAUNCH(P1, 2) break;                                                                \
  case 4: LION_VOX_LAUNCH(P1, 4) break;                                                                \
  default: LION_VOX_LAUNCH(P1, 8) break;                                                               \
  }
    if (coords_f) { LION_VOX_NP(true) } else { LION_VOX_NP(false) }
#undef LION_VOX_NP
#undef LION_VOX_LAUNCH
    LION_LAUNCH_CHECK();
    return 0;
  }
  // fallback
  const int32_t *ci = coords_i;
  if (coords_f) {
    int32_t *vox = reinterpret_cast<int32_t *>(w + p.off_vox);
    p1_kernel<<<B, VT, 0, st>>>(coords_f, N, r, normalize, eps, norm_coords, vox);
    LION_LAUNCH_CHECK();
    ci = vox;
  }
  hipError_t e = hipMemsetAsync(cnt, 0, (size_t)B * r3 * 4, st);
  if (e != hipSuccess) return (int)e;
  vox_index_atomic_kernel<<<dim3(lion_cdiv(N, 256), B), 256, 0, st>>>(ci, N, r, ind, cnt);
  LION_LAUNCH_CHECK();
  if (feat) {
    e = hipMemsetAsync(out, 0, (size_t)B * C * r3 * 4, st);
    if (e != hipSuccess) return (int)e;
    const int CT = 16;
    vox_scatter_atomic_kernel<<<dim3(lion_cdiv(N, 256), lion_cdiv(C, CT), B), 256, 0, st>>>(
        feat, ind, cnt, C, N, r3, CT, out);
    LION_LAUNCH_CHECK();
  }
  return 0;
}

} // namespace

extern "C" {

int lion_abi_version(void) { return 1; }

size_t lion_avg_voxelize_workspace_bytes(int B, int C, int N, int r) {
  (void)C;
  if (B <= 0 || N <= 0 || r <= 0) return 0;
  return make_plan(B, C, N, r).total;
}

int lion_avg_voxelize_forward(const float *feat, const int32_t *coords, int B, int C, int N, int r,
                              float *out, int32_t *ind, int32_t *cnt, void *ws, size_t ws_bytes,
                              lionStream_t stream) {
  if (!feat || !coords) return LION_EINVAL;
  return voxelize_impl(feat, coords, nullptr, B, C, N, r, 0, 0.f, out, nullptr, ind, cnt, ws,
                       ws_bytes, static_cast<hipStream_t>(stream));
}

int lion_voxelize_points_forward(const float *feat, const float *coords, int B, int C, int N,
                                 int r, int normalize, float eps, float *out, float *norm_coords,
                                 int32_t *ind, int32_t *cnt, void *ws, size_t ws_bytes,
                                 lionStream_t stream) {
  if (!coords) return LION_EINVAL;
  return voxelize_impl(feat, nullptr, coords, B, C, N, r, normalize, eps, out, norm_coords, ind,
                       cnt, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

size_t lion_voxel_plan_bytes(int B, int N, int r) {
  if (B <= 0 || N <= 0 || r <= 0 || (long)r * r * r > (1L << 30)) return 0;
  const VoxPlan p = make_plan(B, 0, N, r);
  if (!p.fast) return 0;
  return vox_plan_words(B, N, p.S) * 4;
}

int lion_voxel_index(const float *coords, int B, int N, int r, int normalize, float eps, float *norm_coords,
                     int32_t *ind, int32_t *cnt, void *plan, size_t plan_bytes, lionStream_t stream) {
  if (!coords || !plan || !norm_coords) return LION_EINVAL;
  const size_t need = lion_voxel_plan_bytes(B, N, r);
  if (need == 0) return LION_EUNSUPPORTED;
  if (plan_bytes < need) return LION_EWORKSPACE;
  return voxelize_impl(nullptr, nullptr, coords, B, 0, N, r, normalize, eps, nullptr, norm_coords, ind, cnt, nullptr, 0,
                       static_cast<hipStream_t>(stream), plan);
}

static int voxel_scatter_impl(const float *feat, const void *plan, size_t plan_bytes, int B, int C, int N, int r, float *out,
                              const int32_t *occ_flags, lionStream_t stream);

int lion_voxel_scatter(const float *feat, const void *plan, size_t plan_bytes, int B, int C, int N, int r, float *out,
                       lionStream_t stream) {
  return voxel_scatter_impl(feat, plan, plan_bytes, B, C, N, r, out, nullptr, stream);
}

// The same scatter for a grid whose ONLY reader is the sparse convolution popping occ_m1 (lion_conv3d_tile_occupancy[_aware],
// margin 1; r in {16, 32}): z-rows outside the halo of every occupied tile are not written (left as allocated).
int lion_voxel_scatter_read(const float *feat, const void *plan, size_t plan_bytes, int B, int C, int N, int r,
                            const int32_t *occ_m1, float *out, lionStream_t stream) {
  if (!occ_m1) return LION_EINVAL;
  if (r != 16 && r != 32) return LION_EUNSUPPORTED;
  return voxel_scatter_impl(feat, plan, plan_bytes, B, C, N, r, out, occ_m1, stream);
}

static int voxel_scatter_impl(const float *feat, const void *plan, size_t plan_bytes, int B, int C, int N, int r, float *out,
                              const int32_t *occ_flags, lionStream_t stream) {
  if (!feat || !plan || !out || B <= 0 || C <= 0 || N <= 0 || r <= 0) return LION_EINVAL;
  const size_t need = lion_voxel_plan_bytes(B, N, r);
  if (need == 0 || (((uintptr_t)out) & 15) != 0) return LION_EUNSUPPORTED;
  if (plan_bytes < need) return LION_EWORKSPACE;
  const VoxPlan p0 = make_plan(B, C, N, r);
  if (!p0.fast) return LION_EUNSUPPORTED;
  VoxPlan p = p0;
  // Option: TWO workgroups per CU for every cloud size (the scatter kernel needs 40 registers): one's mean phase (B) under
  // the other's store phase (C).  Half the LDS each -> more channel chunks; 512 workgroups -> channel ranges (the plan
  // makes the per-workgroup prologue cheap).  Measured (round 4, (64,2048,32)): 66 us against 58 with one workgroup per CU -- off; -DLION_VOXS_TWO=1 builds it.
#ifndef LION_VOXS_TWO
#define LION_VOXS_TWO 0
#endif
  const bool two_per_cu = LION_VOXS_TWO || N <= VT;
  if (two_per_cu) {
    int CS = 1;
    while ((long)B * p.S * CS < 512 && C / (CS * 2) >= 8) CS *= 2;
    p.CS = CS;
  }
  // (the static LDS of the reader-aware instantiation -- its 1-KiB row map -- comes out of the same budget)
  const size_t lds_limit = (two_per_cu ? (size_t)LDS_LIMIT / 2 : (size_t)LDS_LIMIT) - PREFETCH_PAD - (occ_flags ? 1024 + 64 : 0);
  const size_t fixed = ((size_t)p.SV + (size_t)p.n_words + 64) * 4;   // slot | ust | the cloud's slab headers
  const size_t nocc = (size_t)(N < p.SV ? N : p.SV);
  const size_t want = ((size_t)N + 1 + nocc) * C * 4;
  const size_t avail = fixed < lds_limit ? lds_limit - fixed : 0;
  const size_t arena = want < avail ? want : avail;
  if (arena / 4 < (size_t)N + 1 + nocc) return LION_EUNSUPPORTED;
  const size_t lds = fixed + arena + PREFETCH_PAD;
  const int grid = ((B + 7) / 8) * 8 * p.S * p.CS;
  const int r3 = r * r * r;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int occ_td = 0, occ_th = 0;   // the tile geometry of the flags: the sparse convolution's own choice (csrc/conv3d.hip)
  if (occ_flags)
    if (int e = lion_internal_sparse_tile_dims(r, &occ_td, &occ_th)) return e;
#define LION_VOXS_LAUNCH(NPV)                                                                          \
  if (occ_flags) {                                                                                     \
    static LionLdsLimit cfg = {};                                                                      \
    if (int e = lion_dynamic_lds(&vox_scatter_kernel<NPV, true>, lds, cfg)) return e;                  \
    vox_scatter_kernel<NPV, true><<<grid, VT, lds, st>>>(feat, plan, B, C, N, r3, p.S, p.CS, p.SV, p.n_words, \
                                                         (int)(arena / 4), 64, out, occ_flags, r, occ_td, occ_th); \
  } else {                                                                                             \
    static LionLdsLimit cfg = {};                                                                      \
    if (int e = lion_dynamic_lds(&vox_scatter_kernel<NPV, false>, lds, cfg)) return e;                 \
    vox_scatter_kernel<NPV, false><<<grid, VT, lds, st>>>(feat, plan, B, C, N, r3, p.S, p.CS, p.SV, p.n_words, \
                                                          (int)(arena / 4), 64, out, nullptr, r, 0, 0); \
  }
  switch (p.NP) {
  case 1: LION_VOXS_LAUNCH(1) break;
  case 2: LION_VOXS_LAUNCH(2) break;
  case 4: LION_VOXS_LAUNCH(4) break;
  default: LION_VOXS_LAUNCH(8) break;
  }
#undef LION_VOXS_LAUNCH
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_avg_voxelize_backward(const float *gy, const int32_t *ind, const int32_t *cnt, int B,
                               int C, int N, int r3, float *gx, lionStream_t stream) {
  if (!gy || !ind || !cnt || !gx || B <= 0 || C <= 0 || N <= 0 || r3 <= 0) return LION_EINVAL;
  const int CT = 8;
  vox_grad_kernel<<<dim3(lion_cdiv(N, 256), lion_cdiv(C, CT), B), 256, 0,
                    static_cast<hipStream_t>(stream)>>>(gy, ind, cnt, C, N, r3, CT, gx);
  LION_LAUNCH_CHECK();
  return 0;
}

} // extern "C"
