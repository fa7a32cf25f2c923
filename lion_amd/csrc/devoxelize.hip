// devoxelize.hip -- trilinear devoxelization (K4 forward, K5 backward).
//
// Reference: third_party/pvcnn/functional/src/interpolate/trilinear_devox.cu:21-162,
//            trilinear_devox.cpp:18-95.
//
// Forward: r = 32: only the 32-byte pieces of z-rows that some point reads are moved, compacted by LDS-DMA into a ring of
// LDS buffers (devox_ring_kernel); r <= 16: the [r^3] grid of a (batch, channel) is streamed through LDS
// (devox_slab_kernel: LDS-DMA, double buffered, several whole channel grids per round) and the 8
// corners are gathered from LDS -- global 4-byte gathers cost one TA cycle per lane, which bounds
// the plain gather kernel (devox_fwd_kernel, kept as the fallback for N > 2048 / odd r) at ~2x the
// time.  A lane owns up to 8 points; the corner indices / weights use the same expressions and the same left-to-right
// evaluation as the reference: bit-exact vs the oracle.
//
// Backward: the reference issues 8*C global float atomics per point into a memset grid.  Here one
// workgroup owns one (batch, channel) slab, accumulates it in LDS with ds_add_f32 (16 KiB at r=16,
// 128 KiB at r=32) and writes the dense slab exactly once with coalesced 16-byte stores: no memset,
// no global atomics.  Summation order inside a voxel is not fixed (LDS atomics), so backward is
// compared with tolerance, like the reference itself (atomicAdd order is undefined there too).
#include "common.h"

namespace {

template <bool V> struct BoolT { static constexpr bool value = V; };

struct Corners {
  int ix[8];
  float w[8];
};

__device__ __forceinline__ Corners corners_of(float x, float y, float z, int r, int r2) {
  Corners k;
  const float xl = floorf(x), yl = floorf(y), zl = floorf(z);
  const float xd1 = sub_rn(x, xl), yd1 = sub_rn(y, yl), zd1 = sub_rn(z, zl);
  const float xd0 = sub_rn(1.0f, xd1), yd0 = sub_rn(1.0f, yd1), zd0 = sub_rn(1.0f, zd1);
  // trilinear_devox.cu:52-59 -- (a*b)*c, left to right
  k.w[0] = mul_rn(mul_rn(xd0, yd0), zd0);
  k.w[1] = mul_rn(mul_rn(xd0, yd0), zd1);
  k.w[2] = mul_rn(mul_rn(xd0, yd1), zd0);
  k.w[3] = mul_rn(mul_rn(xd0, yd1), zd1);
  k.w[4] = mul_rn(mul_rn(xd1, yd0), zd0);
  k.w[5] = mul_rn(mul_rn(xd1, yd0), zd1);
  k.w[6] = mul_rn(mul_rn(xd1, yd1), zd0);
  k.w[7] = mul_rn(mul_rn(xd1, yd1), zd1);
  const int xlo = (int)xl, ylo = (int)yl, zlo = (int)zl;
  const int xs = (xd1 > 0.0f) ? r2 : 0; // :64-66, (x_hi & r2)
  const int ys = (yd1 > 0.0f) ? r : 0;
  const int zs = (zd1 > 0.0f) ? 1 : 0;
  k.ix[0] = xlo * r2 + ylo * r + zlo;
  k.ix[1] = k.ix[0] + zs;
  k.ix[2] = k.ix[0] + ys;
  k.ix[3] = k.ix[2] + zs;
  k.ix[4] = k.ix[0] + xs;
  k.ix[5] = k.ix[4] + zs;
  k.ix[6] = k.ix[4] + ys;
  k.ix[7] = k.ix[6] + zs;
  return k;
}

// AFF: out = scale[b,c] * interp(feat) + shift[b,c] * sum(w)  ==  interp(scale*feat + shift): the
// per-(batch, channel) affine that AdaGN + the SE gate apply to the whole grid (pvcnn2_ada.py:219-226)
// commutes with the interpolation, so the two full-grid passes are replaced by two scalars here.
template <int CT, bool AFF>
__global__ __launch_bounds__(256) void devox_fwd_kernel(const float *__restrict__ coords,
                                                        const float *__restrict__ feat, int C,
                                                        int N, int r, int training,
                                                        float *__restrict__ out,
                                                        int32_t *__restrict__ inds,
                                                        float *__restrict__ wgts,
                                                        const float *__restrict__ scale,
                                                        const float *__restrict__ shift) {
  const int b = blockIdx.z, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int r2 = r * r, r3 = r2 * r;
  const float *co = coords + (size_t)b * 3 * N;
  Corners k = corners_of(co[i], co[i + N], co[i + 2 * N], r, r2);
  if (training && blockIdx.y == 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      wgts[((size_t)b * 8 + q) * N + i] = k.w[q];
      inds[((size_t)b * 8 + q) * N + i] = k.ix[q];
    }
  }
  // memory safety for out-of-contract coordinates (the reference would read out of bounds)
#pragma unroll
  for (int q = 0; q < 8; ++q) k.ix[q] = min(max(k.ix[q], 0), r3 - 1);
  const int c0 = blockIdx.y * CT, c1 = min(C, c0 + CT);
  const float *f = feat + ((size_t)b * C + c0) * r3;
  float *o = out + ((size_t)b * C + c0) * N + i;
  float wsum = 0.f;
  if (AFF) {
    wsum = k.w[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) wsum += k.w[q];
  }
#pragma unroll 4
  for (int c = c0; c < c1; ++c, f += r3, o += N) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = f[k.ix[q]];
    float acc = mul_rn(k.w[0], v[0]); // :96-103, left to right
#pragma unroll
    for (int q = 1; q < 8; ++q) acc = add_rn(acc, mul_rn(k.w[q], v[q]));
    if (AFF) acc = acc * scale[(size_t)b * C + c] + shift[(size_t)b * C + c] * wsum;
    *o = acc;
  }
}


// ---------------------------------------------------------------------------------------------
// devox_slab_kernel: the forward through LDS.  Global 4-byte gathers cost one TA cycle per lane
// (8 corners x C channels x N points = 55 us of TA time at (64,2048,32), the floor of the kernel
// above); the grid of one (batch, channel) is at most 128 KiB and every 128-byte line of it is
// touched by some point anyway.  So: stream the grid through LDS in slabs of <= 9 x-planes
// (PX planes + 1 halo plane for the x+1 corners, 36 KiB at r=32; the whole 16 KiB / 2 KiB grid at
// r=16 / r=8) with coalesced 16-byte loads, double buffered -- the next slab's loads are in flight
// in registers while the lanes gather the 8 corners of their points from the current one -- and
// write the [C,N] rows coalesced.  A workgroup walks CT channels x XS slabs; a lane owns <= 8 points
// whose (base index, fractions) stay in registers.  Same expressions as corners_of()/the gather
// kernel: bit-exact vs the oracle.
// ---------------------------------------------------------------------------------------------
template <int LD, bool AFF>
__global__ __launch_bounds__(256) void devox_slab_kernel(
    const float *__restrict__ coords, const float *__restrict__ feat, int C, int N, int r, int PX,
    int XS, int CT, int CI, int slab_floats, int training, float *__restrict__ out,
    int32_t *__restrict__ inds, float *__restrict__ wgts, const float *__restrict__ scale,
    const float *__restrict__ shift) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *lds = reinterpret_cast<float *>(smem);
  constexpr int PP = 8;
  const int tid = threadIdx.x, b = blockIdx.y, c0 = blockIdx.x * CT;
  const int nch = min(CT, C - c0);
  const int r2 = r * r, r3 = r2 * r;
  const float *co = coords + (size_t)b * 3 * N;
  uint16_t *sorted = reinterpret_cast<uint16_t *>(lds + 2 * slab_floats); // [N] point ids grouped by slab
  int *scnt = reinterpret_cast<int *>(sorted + ((N + 1) & ~1));            // [XS] counts, then cursors
  unsigned *need = reinterpret_cast<unsigned *>(scnt + XS);                // [ceil(r^2 / 32)] bit (x*r + y): some
                                                                           // point reads the z-row (x, y, :)

  const int ngrp = (nch + CI - 1) / CI; // groups of CI channels staged together (CI > 1 only when XS == 1)
  const int nit = ngrp * XS;
  // float4 count of iteration (group g, slab xs): CI whole grids when XS == 1, else PX(+1 halo) planes
  auto slab_count4 = [&](int g, int xs) {
    return XS == 1 ? (min(CI, nch - g * CI) * r3) >> 2 : ((min(r, xs * PX + PX + 1) - xs * PX) * r2) >> 2;
  };
  // LDS-DMA (global_load_lds_dwordx4): 1 KiB per wave instruction lands at M0 + lane * 16, no staging
  // registers and no ds_write pass.  hipcc does not count asm memory operations: the matching
  // "s_waitcnt vmcnt(0)" sits in front of the barrier at the top of the slab loop.
  typedef __attribute__((address_space(3))) float lds_float;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_float *)lds;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto issue = [&](int it) {
    const int g = it / XS, xs = it - g * XS;
    const float4 *src = reinterpret_cast<const float4 *>(feat + ((size_t)b * C + c0 + g * CI) * r3 + (size_t)xs * PX * r2);
    const int n4 = slab_count4(g, xs);
    const uint32_t dst0 = lds_base + (uint32_t)((it & 1) * slab_floats * 4 + wave * 1024);
    const int row0 = XS == 1 ? 0 : xs * PX * r;                // first z-row of the slab (row id = x * r + y)
#pragma unroll
    for (int j = 0; j < LD; ++j) {
      const int f = tid + j * 256;                             // float4 of the slab this lane moves
      const float4 *gp = src + f;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(dst0 + j * 4096);
      // only the z-rows some point of the cloud interpolates from are fetched (43 % of them at N = 2048, r = 32;
      // a row is r floats = r/4 lanes of the 1 KiB wave instruction): inactive lanes move nothing, their LDS
      // bytes stay stale and are never read; a wave instruction without any needed row is skipped altogether
      int row = (4 * f) / r;
      if (XS == 1) row %= r2;
      else row += row0;
      const bool want = f < n4 && ((need[row >> 5] >> (row & 31)) & 1u);
      if (want) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gp), "s"(dst) : "memory");
      }
    }
  };
  // ---- which z-rows are read at all: the four (x, y) pairs of every point, with the reference's "hi = lo when the
  // fraction is 0" rule (trilinear_devox.cu:64-66) ----
  const int nwords = (r2 + 31) >> 5;
  for (int w = tid; w < nwords; w += 256) need[w] = 0u;
  __syncthreads();
  for (int i = tid; i < N; i += 256) {
    const float x = co[i], y = co[i + N];
    const float xl = floorf(x), yl = floorf(y);
    if (xl >= 0.f && yl >= 0.f && xl < (float)r && yl < (float)r) {
      const int x0 = (int)xl, y0 = (int)yl;
      const int x1 = min(x0 + (sub_rn(x, xl) > 0.0f ? 1 : 0), r - 1), y1 = min(y0 + (sub_rn(y, yl) > 0.0f ? 1 : 0), r - 1);
      const int ra = x0 * r + y0, rb = x0 * r + y1, rc = x1 * r + y0, rd = x1 * r + y1;
      atomicOr(&need[ra >> 5], 1u << (ra & 31));
      atomicOr(&need[rb >> 5], 1u << (rb & 31));
      atomicOr(&need[rc >> 5], 1u << (rc & 31));
      atomicOr(&need[rd >> 5], 1u << (rd & 31));
    }
  }
  __syncthreads();
  issue(0); // in flight during the rest of the per-point setup

  // ---- group the points by x-slab so that a wave's lanes are active together (a lane's p-th point is
  // sorted[tid + 256 p]); the order inside a slab is irrelevant (points are independent) ----
  for (int s2 = tid; s2 < XS; s2 += 256) scnt[s2] = 0;
  __syncthreads();
  int sl0[PP];
#pragma unroll
  for (int p = 0; p < PP; ++p) {
    const int i = tid + p * 256;
    sl0[p] = -1;
    if (i < N) {
      sl0[p] = min(max((int)floorf(co[i]) / PX, 0), XS - 1);
      if (XS > 1) atomicAdd(&scnt[sl0[p]], 1);
    }
  }
  __syncthreads();
  if (tid == 0) { // exclusive scan of <= 64 counters
    int run = 0;
    for (int s2 = 0; s2 < XS; ++s2) { const int c = scnt[s2]; scnt[s2] = run; run += c; }
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < PP; ++p) {
    const int i = tid + p * 256;
    if (i < N) sorted[XS > 1 ? atomicAdd(&scnt[sl0[p]], 1) : i] = (uint16_t)i;
  }
  __syncthreads();

  int ix0[PP], xsl[PP], pid[PP];
  float xd1[PP], yd1[PP], zd1[PP];
#pragma unroll
  for (int p = 0; p < PP; ++p) {
    const int k = tid + p * 256;
    xsl[p] = -1;
    ix0[p] = 0;
    pid[p] = 0;
    xd1[p] = yd1[p] = zd1[p] = 0.f;
    if (k < N) {
      const int i = sorted[k];
      pid[p] = i;
      const float x = co[i], y = co[i + N], z = co[i + 2 * N];
      const Corners kk = corners_of(x, y, z, r, r2);
      if (training && blockIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          wgts[((size_t)b * 8 + q) * N + i] = kk.w[q];
          inds[((size_t)b * 8 + q) * N + i] = kk.ix[q];
        }
      }
      const float xl = floorf(x), yl = floorf(y), zl = floorf(z);
      xd1[p] = sub_rn(x, xl);
      yd1[p] = sub_rn(y, yl);
      zd1[p] = sub_rn(z, zl);
      ix0[p] = kk.ix[0];
      xsl[p] = min(max((int)xl / PX, 0), XS - 1);
      // memory safety for out-of-contract coordinates (the reference would read out of bounds):
      // such a point gets 0 instead of garbage
      const int xin = (int)xl - xsl[p] * PX;
      const bool ok = xl >= 0.f && yl >= 0.f && zl >= 0.f && xl < (float)r && yl < (float)r && zl < (float)r &&
                      kk.ix[7] < r3 && xin >= 0 && xin < PX;
      if (!ok) xsl[p] = -2;
    }
  }

  for (int it = 0; it < nit; ++it) {
    const int g = it / XS, xs = it - g * XS;
    const int cg = c0 + g * CI, cin = min(CI, nch - g * CI);
    float *buf = lds + (it & 1) * slab_floats;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's share of slab `it` has landed
    __syncthreads();                                  // ... everybody's; and slab it-1 is no longer read
    if (it + 1 < nit) issue(it + 1);
    const int base = xs * PX * r2;
    float *o = out + ((size_t)b * C + cg) * N;
#pragma unroll
    for (int p = 0; p < PP; ++p) {
      if (xsl[p] == xs) {
        const float xd0 = sub_rn(1.0f, xd1[p]), yd0 = sub_rn(1.0f, yd1[p]), zd0 = sub_rn(1.0f, zd1[p]);
        const int xo = (xd1[p] > 0.0f) ? r2 : 0, yo = (yd1[p] > 0.0f) ? r : 0, zo = (zd1[p] > 0.0f) ? 1 : 0;
        const float w0 = mul_rn(mul_rn(xd0, yd0), zd0), w1 = mul_rn(mul_rn(xd0, yd0), zd1[p]);
        const float w2 = mul_rn(mul_rn(xd0, yd1[p]), zd0), w3 = mul_rn(mul_rn(xd0, yd1[p]), zd1[p]);
        const float w4 = mul_rn(mul_rn(xd1[p], yd0), zd0), w5 = mul_rn(mul_rn(xd1[p], yd0), zd1[p]);
        const float w6 = mul_rn(mul_rn(xd1[p], yd1[p]), zd0), w7 = mul_rn(mul_rn(xd1[p], yd1[p]), zd1[p]);
        float wsum = w0;
        if (AFF) { wsum += w1; wsum += w2; wsum += w3; wsum += w4; wsum += w5; wsum += w6; wsum += w7; }
        const float *f0 = buf + (ix0[p] - base);
        for (int ci = 0; ci < cin; ++ci, f0 += r3) {
          const float *f1 = f0 + xo;
          const float v0 = f0[0], v1 = f0[zo], v2 = f0[yo], v3 = f0[yo + zo];
          const float v4 = f1[0], v5 = f1[zo], v6 = f1[yo], v7 = f1[yo + zo];
          float a = mul_rn(w0, v0); // trilinear_devox.cu:96-103, left to right
          a = add_rn(a, mul_rn(w1, v1));
          a = add_rn(a, mul_rn(w2, v2));
          a = add_rn(a, mul_rn(w3, v3));
          a = add_rn(a, mul_rn(w4, v4));
          a = add_rn(a, mul_rn(w5, v5));
          a = add_rn(a, mul_rn(w6, v6));
          a = add_rn(a, mul_rn(w7, v7));
          if (AFF) a = a * scale[(size_t)b * C + cg + ci] + shift[(size_t)b * C + cg + ci] * wsum;
          o[(size_t)ci * N + pid[p]] = a;
        }
      } else if (xsl[p] == -2 && xs == 0) {
        for (int ci = 0; ci < cin; ++ci) o[(size_t)ci * N + pid[p]] = 0.f;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// devox_ring_kernel (r = 32): only the PIECES of z-rows that some point interpolates from are moved at all, through a
// ring of D LDS buffers with D - 1 channels in flight.
// A cloud of N = 2048 points touches <= 4 (x, y) rows per point: 37-52 % of the 1024 z-rows (128 B each) of a channel
// grid for Gaussian latents, 15 % for surface-like clouds -- and of a row only the 32-byte piece(s) around z_lo, z_lo+1
// (Gaussian: 1025 of 4096 pieces = 32 KiB per channel against 56 KiB of whole rows).  Round 2-3 moved whole rows through
// two 72-KiB buffers: one channel in flight, 57 KB per CU, one DMA round trip (~3 us) per channel -- latency x
// concurrency, not bytes, set the 34 us.  Here (round 4):
//   * per cloud: piece bitmap (LDS atomicOr), ranks by popcount prefix -> slot per needed piece, piece list;
//   * a channel's needed pieces are compacted by LDS-DMA (global_load_lds_dwordx4 ... nt, per-lane addresses computed
//     once) into one of D = min(4, 144 KiB / channel bytes) buffers; channels ci+1 .. ci+D-1 are in flight while channel
//     ci is gathered (8 ds_read_b32 per point from precomputed 16-bit offsets) and written with coalesced [C, N] row
//     stores.  The wait for channel ci is COUNTED: every wave issues the same number of DMA instructions per channel
//     (lanes beyond the list read element 0 into a slot nobody reads) and the same PP stores per channel (buffer stores:
//     lanes beyond N carry an out-of-range offset and are dropped), vector memory operations retire in order, so
//     "channel ci has landed" = at most (operations issued behind it) outstanding.  The barrier is the bare instruction:
//     __syncthreads() carries a fence that the compiler lowers to vmcnt(0), which would drain the ring.
// Same corner expressions / evaluation order as corners_of(): bit-exact vs the oracle.  A cloud whose pieces exceed one
// 72-KiB buffer and points with z beyond the clamp range take the in-kernel global-gather path.
// ---------------------------------------------------------------------------------------------
constexpr int DVR_R = 32, DVR_NT = 512, DVR_PP = 2048 / DVR_NT; // r, threads, points per lane (N <= 2048)
constexpr int DVR_RING_BYTES = 144 * 1024;                      // the ring; one workgroup per CU
constexpr int DVR_NJ = 9;                                       // DMA instructions per thread and channel, at most (72 KiB)

__device__ __forceinline__ void wait_vm_uniform(int n) { // n: wave-uniform number of operations that may stay outstanding
  n = __builtin_amdgcn_readfirstlane(n);
#define LION_WVM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n < 0 ? 0 : n) {
    LION_WVM(0) LION_WVM(1) LION_WVM(2) LION_WVM(3) LION_WVM(4) LION_WVM(5) LION_WVM(6) LION_WVM(7) LION_WVM(8) LION_WVM(9)
    LION_WVM(10) LION_WVM(11) LION_WVM(12) LION_WVM(13) LION_WVM(14) LION_WVM(15) LION_WVM(16) LION_WVM(17) LION_WVM(18)
    LION_WVM(19) LION_WVM(20) LION_WVM(21) LION_WVM(22) LION_WVM(23) LION_WVM(24) LION_WVM(25) LION_WVM(26) LION_WVM(27)
    LION_WVM(28) LION_WVM(29) LION_WVM(30) LION_WVM(31) LION_WVM(32) LION_WVM(33) LION_WVM(34) LION_WVM(35) LION_WVM(36)
    LION_WVM(37) LION_WVM(38) LION_WVM(39)
  default: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break; // stricter than asked for: never wrong
  }
#undef LION_WVM
}

// G = floats per piece (8: 32-byte pieces; 32: whole z-rows)
// MODE 0: self-contained (the C-ABI drop-in entry points); MODE 2: one workgroup per cloud writes the cloud's PLAN -- piece
// list, per-point corner offsets and status, everything the coordinates alone decide -- and ends; MODE 1: starts from
// a plan: the first channels' DMA leaves one global round trip after the kernel begins (the plan's piece list) instead of
// behind coordinates -> bitmap -> ranks -> list (three barriers), and the per-point work runs under it.  The models
// devoxelise the same (cloud, r = 32) four times per forward (lion_amd/models/pvcnn2_ada.py).
// plan of a cloud (devox_plan_stride() bytes): int np, int fits, 8 bytes of padding, u16 plist[NPMAX], u16 off[N_max = 2048][8] (float offsets
// of the 8 corners inside a ring buffer), i8 st[2048].
template <int G> constexpr int devox_plan_stride() { return 16 + (DVR_NJ * DVR_NT / (G / 4)) * 2 + 2048 * 16 + 2048; } // 16-byte multiples

template <bool AFF, int G, int MODE>
__global__ __launch_bounds__(DVR_NT) void devox_ring_kernel(const float *__restrict__ coords,
                                                            const float *__restrict__ feat, int C, int N, int CT,
                                                            int training, float *__restrict__ out,
                                                            int32_t *__restrict__ inds, float *__restrict__ wgts,
                                                            const float *__restrict__ scale,
                                                            const float *__restrict__ shift, int dmax,
                                                            unsigned char *__restrict__ plan) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PP = DVR_PP, r = DVR_R, r2 = r * r, r3 = r2 * r, NT = DVR_NT;
  constexpr int PPR = r / G, NP = r2 * PPR, nwords = NP / 32, LP = G / 4; // pieces per row / per grid, DMA lanes per piece
  constexpr int NPMAX = DVR_NJ * NT / LP;                                 // pieces one buffer can hold
  static_assert(nwords <= 128, "two bitmap words per lane of the ranking wave");
  float *lds = reinterpret_cast<float *>(smem);
  const int tid = threadIdx.x, b = blockIdx.y, c0 = blockIdx.x * CT;
  const int nch = min(CT, C - c0);
  unsigned *need = reinterpret_cast<unsigned *>(smem + DVR_RING_BYTES); // [nwords] bit = piece id (x*r + y) * PPR + z / G
  unsigned *base = need + nwords;                                      // [nwords] rank of the word's first piece
  uint16_t *plist = reinterpret_cast<uint16_t *>(base + nwords);       // [NPMAX] piece id of each slot
  float *s_sc = reinterpret_cast<float *>(plist + NPMAX);              // [16] scale, [16] shift of this workgroup's channels
  int *s_np = reinterpret_cast<int *>(s_sc + 32);
  const float *co = coords + (size_t)b * 3 * N;
  unsigned char *pl_b = (MODE != 0) ? plan + (size_t)b * devox_plan_stride<G>() : nullptr;
  const uint16_t *pl_list = reinterpret_cast<const uint16_t *>(pl_b + 16);
  uint16_t *pl_off = reinterpret_cast<uint16_t *>(pl_b + 16 + NPMAX * 2);
  signed char *pl_st = reinterpret_cast<signed char *>(pl_b + 16 + NPMAX * 2 + 2048 * 16);

  if (tid < nwords) need[tid] = 0u;
  if (AFF && tid < 2 * 16) {
    const int ci = tid & 15;
    s_sc[tid] = ci < nch ? (tid < 16 ? scale : shift)[(size_t)b * C + c0 + ci] : 0.f;
  }
  __syncthreads();
  // ---- the ring (declared here: MODE 1 starts it before the per-point work) ----
  int np = 0, nj = 0, buf_bytes = 0, D = 2;
  bool fits = false;
  int goff[DVR_NJ];
  typedef __attribute__((address_space(3))) float lds_float;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_float *)lds;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int ops = 0;        // vector memory operations this wave has issued inside the ring (DMA instructions and stores)
  int mark[4] = {0, 0, 0, 0}; // ops right behind the DMA of the channel that lives in ring slot k
  auto ring_shape = [&]() {
    nj = (np * LP + NT - 1) / NT;         // DMA instructions per thread and channel (wave-uniform)
    buf_bytes = nj * NT * 16;             // every lane of every instruction lands inside the buffer
    D = nj == 0 ? 2 : min(min(dmax, 4), DVR_RING_BYTES / (nj ? buf_bytes : 1));
    fits = nj <= DVR_NJ && D >= 2;
  };
  auto issue = [&](int ci) {
    const float *src = feat + ((size_t)b * C + c0 + ci) * r3;
    const uint32_t dst0 = lds_base + (uint32_t)((ci % D) * buf_bytes + wave * 1024);
#pragma unroll
    for (int j = 0; j < DVR_NJ; ++j) {
      if (j < nj) { // wave-uniform: every wave issues nj instructions per channel
        const uint32_t dst = __builtin_amdgcn_readfirstlane(dst0 + j * (NT * 16));
        const float *gp = src + goff[j];
        unsigned keep; // nt: every piece is read exactly once per call
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gp), "s"(dst) : "memory");
      }
    }
    ops += nj;
    mark[ci % D] = ops;
  };
  if (MODE == 1) {
    const int *hdr = reinterpret_cast<const int *>(pl_b);
    unsigned pl[DVR_NJ];
#pragma unroll
    for (int j = 0; j < DVR_NJ; ++j) pl[j] = pl_list[(tid + j * NT) / LP]; // always inside the list's allocation
    np = __builtin_amdgcn_readfirstlane(hdr[0]);
    ring_shape();
    fits = fits && hdr[1] != 0;
    const int n4 = fits ? np * LP : 0;
#pragma unroll
    for (int j = 0; j < DVR_NJ; ++j) {
      const int f = tid + j * NT;
      goff[j] = f < n4 ? (int)pl[j] * G + (f % LP) * 4 : 0;
    }
    if (fits)
      for (int k = 0; k < D - 1 && k < nch; ++k) issue(k);
  }
  float xd1[PP], yd1[PP], zd1[PP];
  int row4[PP][4], zl_[PP], st[PP]; // st: 1 = regular point, 0 = none, -2 = out of the grid's memory,
                                    // 2 = z_lo + 1 runs into the next row (flat indexing, global path)
#pragma unroll
  for (int p = 0; p < PP; ++p) {
    const int i = tid + p * NT;
    st[p] = 0;
    xd1[p] = yd1[p] = zd1[p] = 0.f;
    row4[p][0] = row4[p][1] = row4[p][2] = row4[p][3] = zl_[p] = 0;
    if (i < N) {
      const float x = co[i], y = co[i + N], z = co[i + 2 * N];
      const Corners kk = corners_of(x, y, z, r, r2);
      if (training && blockIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          wgts[((size_t)b * 8 + q) * N + i] = kk.w[q];
          inds[((size_t)b * 8 + q) * N + i] = kk.ix[q];
        }
      }
      const float xl = floorf(x), yl = floorf(y), zl = floorf(z);
      xd1[p] = sub_rn(x, xl);
      yd1[p] = sub_rn(y, yl);
      zd1[p] = sub_rn(z, zl);
      const bool ok = xl >= 0.f && yl >= 0.f && zl >= 0.f && xl < (float)r && yl < (float)r && zl < (float)r &&
                      kk.ix[7] < r3;
      st[p] = ok ? 1 : -2;
      if (ok) {
        const int x0 = (int)xl, y0 = (int)yl;
        const int x1 = x0 + (xd1[p] > 0.0f ? 1 : 0), y1 = y0 + (yd1[p] > 0.0f ? 1 : 0); // ix[7] < r3 keeps them in range
        zl_[p] = (int)zl;
        row4[p][0] = x0 * r + y0; row4[p][1] = x0 * r + y1; row4[p][2] = x1 * r + y0; row4[p][3] = x1 * r + y1;
        // Voxelization clamps coordinates to [0, r-1], so z_lo = r-1 comes with a zero fraction.  A caller that hands
        // over z in (r-1, r) gets the reference's flat indexing (the "+1" element is the first of the next row):
        // such points read the grid directly instead of the compacted pieces.
        if (zl_[p] == r - 1 && zd1[p] > 0.0f) st[p] = 2;
      }
      if (MODE != 1 && st[p] == 1) {
        const int zh = zl_[p] + (zd1[p] > 0.0f ? 1 : 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int pl = row4[p][k] * PPR + zl_[p] / G, ph = row4[p][k] * PPR + zh / G;
          atomicOr(&need[pl >> 5], 1u << (pl & 31));
          if (ph != pl) atomicOr(&need[ph >> 5], 1u << (ph & 31));
        }
      }
    }
  }
  auto slot_of = [&](int pc) { return (int)(base[pc >> 5] + __popc(need[pc >> 5] & ((1u << (pc & 31)) - 1u))); };
  if (MODE != 1) {
    __syncthreads();
    if (tid < 64) { // ranks: exclusive prefix of the words' popcounts, two words per lane
      const int w0 = 2 * tid, w1 = 2 * tid + 1;
      const int c0_ = w0 < nwords ? __popc(need[w0]) : 0, c1_ = w1 < nwords ? __popc(need[w1]) : 0;
      const int inc = wave_incl_scan(c0_ + c1_, tid);
      if (w0 < nwords) base[w0] = (unsigned)(inc - c0_ - c1_);
      if (w1 < nwords) base[w1] = (unsigned)(inc - c1_);
      if (tid == 63) *s_np = inc;
    }
    __syncthreads();
    np = *s_np;
    ring_shape();
    if (fits)
      for (int pc = tid; pc < NP; pc += NT)
        if ((need[pc >> 5] >> (pc & 31)) & 1u) plist[slot_of(pc)] = (uint16_t)pc;
    __syncthreads();
    // the float4 this lane moves in DMA instruction j is the same for every channel: (piece, part) -> float offset inside
    // a channel grid, computed once (no index arithmetic, no LDS lookup in the per-channel issue loop)
    const int n4 = fits ? np * LP : 0;
#pragma unroll
    for (int j = 0; j < DVR_NJ; ++j) {
      const int f = tid + j * NT;
      goff[j] = f < n4 ? (int)plist[f / LP] * G + (f % LP) * 4 : 0;
    }
    if (MODE == 0 && fits)
      for (int k = 0; k < D - 1 && k < nch; ++k) issue(k); // the first channels are in flight during the rest of the setup
  }
  // pieces -> LDS float offsets (inside a ring buffer) of the 8 corners: (row k, z_lo) in the low half, (row k, z_hi)
  // in the high half of off[p][k]; flat offsets inside a channel grid for the global path
  // ... and the 8 corner weights (+ their sum for the affine form) of each point: computed once, not per channel
  unsigned boff[PP][8]; // BYTE offsets inside a ring buffer
  float wq[PP][8], wsum4[PP];
#pragma unroll
  for (int p = 0; p < PP; ++p) {
    const int zh = zl_[p] + (zd1[p] > 0.0f ? 1 : 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      boff[p][2 * k] = boff[p][2 * k + 1] = 0u;
      if (MODE != 1 && fits && st[p] == 1) {
        boff[p][2 * k] = (unsigned)(slot_of(row4[p][k] * PPR + zl_[p] / G) * G + zl_[p] % G) * 4u;
        boff[p][2 * k + 1] = (unsigned)(slot_of(row4[p][k] * PPR + zh / G) * G + zh % G) * 4u;
      }
      row4[p][k] = row4[p][k] * r + zl_[p]; // flat offset of (row k, z_lo)
    }
    const float xd0 = sub_rn(1.0f, xd1[p]), yd0 = sub_rn(1.0f, yd1[p]), zd0 = sub_rn(1.0f, zd1[p]);
    wq[p][0] = mul_rn(mul_rn(xd0, yd0), zd0); wq[p][1] = mul_rn(mul_rn(xd0, yd0), zd1[p]);
    wq[p][2] = mul_rn(mul_rn(xd0, yd1[p]), zd0); wq[p][3] = mul_rn(mul_rn(xd0, yd1[p]), zd1[p]);
    wq[p][4] = mul_rn(mul_rn(xd1[p], yd0), zd0); wq[p][5] = mul_rn(mul_rn(xd1[p], yd0), zd1[p]);
    wq[p][6] = mul_rn(mul_rn(xd1[p], yd1[p]), zd0); wq[p][7] = mul_rn(mul_rn(xd1[p], yd1[p]), zd1[p]);
    float ws = wq[p][0];
#pragma unroll
    for (int q = 1; q < 8; ++q) ws += wq[p][q];
    wsum4[p] = ws;
  }
  if (MODE == 1) { // corner offsets from the plan (st is recomputed above from the same coordinates)
#pragma unroll
    for (int p = 0; p < PP; ++p) {
      const int i = tid + p * NT;
      const uint4 o4 = *reinterpret_cast<const uint4 *>(pl_off + (size_t)(i < 2048 ? i : 0) * 8);
      boff[p][0] = (o4.x & 0xffffu) * 4u; boff[p][1] = (o4.x >> 16) * 4u; boff[p][2] = (o4.y & 0xffffu) * 4u; boff[p][3] = (o4.y >> 16) * 4u;
      boff[p][4] = (o4.z & 0xffffu) * 4u; boff[p][5] = (o4.z >> 16) * 4u; boff[p][6] = (o4.w & 0xffffu) * 4u; boff[p][7] = (o4.w >> 16) * 4u;
    }
  }
  if (MODE == 2) { // write the plan and end
    if (tid == 0) { int *hdr = reinterpret_cast<int *>(pl_b); hdr[0] = np; hdr[1] = fits ? 1 : 0; }
    uint16_t *wl = reinterpret_cast<uint16_t *>(pl_b + 16);
    for (int s_ = tid; s_ < NPMAX; s_ += NT) wl[s_] = (fits && s_ < np) ? plist[s_] : (uint16_t)0;
#pragma unroll
    for (int p = 0; p < PP; ++p) {
      const int i = tid + p * NT;
      if (i < 2048) {
        uint4 o4;
        o4.x = (boff[p][0] >> 2) | ((boff[p][1] >> 2) << 16); o4.y = (boff[p][2] >> 2) | ((boff[p][3] >> 2) << 16);
        o4.z = (boff[p][4] >> 2) | ((boff[p][5] >> 2) << 16); o4.w = (boff[p][6] >> 2) | ((boff[p][7] >> 2) << 16);
        *reinterpret_cast<uint4 *>(pl_off + (size_t)i * 8) = o4;
        pl_st[i] = (signed char)st[p];
      }
    }
    return;
  }
  bool any2 = false;
#pragma unroll
  for (int p = 0; p < PP; ++p) any2 |= st[p] == 2;
  const bool wave_generic = !fits || __ballot(any2) != 0ull; // wave-uniform
  float *ob = out + ((size_t)b * C + c0) * N;
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(ob, 0, nch * N * 4, 0x00020000);
  for (int ci = 0; ci < nch; ++ci) {
    const float *gsrc = feat + ((size_t)b * C + c0 + ci) * r3;
    // an LDS-address-space pointer: with a generic one the compiler merges the two gather paths below into flat loads,
    // whose vmcnt(0) drains the ring
    typedef __attribute__((address_space(3))) const float lds_cfloat;
    const uint32_t lbase = lds_base + (uint32_t)((ci % D) * buf_bytes);
    if (fits) {
      wait_vm_uniform(ops - mark[ci % D]);                         // this wave's share of channel ci has landed
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // ... everybody's; channel ci-1's buffer is free again
      if (ci + D - 1 < nch) issue(ci + D - 1);
    }
    float sc = 1.f, sh = 0.f;
    if (AFF) { sc = s_sc[ci]; sh = s_sc[16 + ci]; }
    float a4[PP];
    // Two copies of the gather, chosen per WAVE: the one every regular wave runs holds no global load at all -- with the
    // two paths in one body the compiler waits (vmcnt) for the possibly-pending global loads where the paths meet,
    // and any vmcnt wait it places drains the ring.
    auto gather = [&](auto generic_c) {
      constexpr bool GENERIC = decltype(generic_c)::value;
#pragma unroll
      for (int p = 0; p < PP; ++p) {
        float a = 0.f; // st == -2: out-of-contract coordinates (the reference would read out of bounds)
        if (st[p] > 0) {
          float v[8];
          if (!GENERIC || (fits && st[p] == 1)) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = *(lds_cfloat *)(uintptr_t)(lbase + boff[p][q]);
          } else { // rare: dense clouds (more pieces than a buffer holds) or z beyond the clamp range
            const int zo = (zd1[p] > 0.0f) ? 1 : 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[2 * k] = gsrc[row4[p][k]]; v[2 * k + 1] = gsrc[row4[p][k] + zo]; }
          }
          a = mul_rn(wq[p][0], v[0]); // trilinear_devox.cu:96-103, left to right
#pragma unroll
          for (int q = 1; q < 8; ++q) a = add_rn(a, mul_rn(wq[p][q], v[q]));
          if (AFF) a = a * sc + sh * wsum4[p];
        }
        a4[p] = a;
      }
    };
    if (wave_generic) gather(BoolT<true>{}); else gather(BoolT<false>{});
    // one store per (point slot, channel) from every wave: the counted waits above rely on it
#pragma unroll
    for (int p = 0; p < PP; ++p) {
      const int i = tid + p * NT;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a4[p]), ors, i < N ? (ci * N + i) * 4 : 0x7fffff00, 0, 0);
    }
    ops += PP;
  }
}

// One workgroup per (b, c, part of the grid); its part accumulated in LDS.  gridDim.z parts (round 6: two at r = 32 -- a
// 128-KiB slab allowed ONE workgroup per CU, so its zero / atomic / store phases ran back to back with nothing beside them;
// two 64-KiB halves let a second workgroup's atomics run under the first one's stores; every workgroup scans all 8 N
// entries and keeps those that fall into its part).
__global__ __launch_bounds__(1024) void devox_bwd_lds_kernel(const float *__restrict__ gy,
                                                             const int32_t *__restrict__ inds,
                                                             const float *__restrict__ wgts, int C,
                                                             int N, int r3,
                                                             float *__restrict__ gx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *slab = reinterpret_cast<float *>(smem);
  const int tid = threadIdx.x, nt = blockDim.x;
  const int c = blockIdx.x, b = blockIdx.y;
  const int part = r3 / gridDim.z, lo = blockIdx.z * part;    // (r3 % (4 gridDim.z) == 0: checked by the launcher)
  for (int v = tid * 4; v < part; v += nt * 4)
    *reinterpret_cast<float4 *>(slab + v) = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const float *g = gy + ((size_t)b * C + c) * N;
  const int32_t *id = inds + (size_t)b * 8 * N;
  const float *wg = wgts + (size_t)b * 8 * N;
  for (int i = tid; i < N; i += nt) {
    const float gv = g[i];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int ix = min(max(id[(size_t)q * N + i], 0), r3 - 1) - lo;
      if (ix >= 0 && ix < part) atomicAdd(slab + ix, mul_rn(wg[(size_t)q * N + i], gv)); // ds_add_f32
    }
  }
  __syncthreads();
  float *o = gx + ((size_t)b * C + c) * r3 + lo;
  for (int v = tid * 4; v < part; v += nt * 4)
    *reinterpret_cast<float4 *>(o + v) = *reinterpret_cast<const float4 *>(slab + v);
}

// The same scatter with the AdaGN(+SE) backward's elementwise pass folded in (training, round 6): the voxel branch of a PVConv ends
// AdaGN -> SE3d -> devoxelize, all linear, so the gradient of the AdaGN's INPUT x is
//     dx = A' scatter(gy) + Q + R x        (A', Q, R f32[B, C] from lion_gn_train_bwd_fold with the gate)
// and the dense gradient of the devoxelisation never has to exist: the slab starts as Q + R x (one read of x) instead of zeros,
// the corner contributions are added scaled by A', the slab is stored (one write of dx).  Was: store the scatter, read it twice
// with x (reduction + apply), store dx.
__global__ __launch_bounds__(1024) void devox_bwd_affine_kernel(const float *__restrict__ gy, const int32_t *__restrict__ inds,
                                                                const float *__restrict__ wgts, const float *__restrict__ x,
                                                                const float *__restrict__ Ap, const float *__restrict__ Q,
                                                                const float *__restrict__ R, int C, int N, int r3,
                                                                float *__restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *slab = reinterpret_cast<float *>(smem);
  const int tid = threadIdx.x, nt = blockDim.x;
  const int c = blockIdx.x, b = blockIdx.y;
  const int part = r3 / gridDim.z, lo = blockIdx.z * part;
  const size_t row = (size_t)b * C + c;
  const float a = Ap[row], q = Q[row], rr = R[row];
  const float *xr = x + row * r3 + lo;
  for (int v = tid * 4; v < part; v += nt * 4) {
    const float4 t = *reinterpret_cast<const float4 *>(xr + v);
    *reinterpret_cast<float4 *>(slab + v) = make_float4(q + rr * t.x, q + rr * t.y, q + rr * t.z, q + rr * t.w);
  }
  __syncthreads();
  const float *g = gy + row * N;
  const int32_t *id = inds + (size_t)b * 8 * N;
  const float *wg = wgts + (size_t)b * 8 * N;
  for (int i = tid; i < N; i += nt) {
    const float gv = a * g[i];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ix = min(max(id[(size_t)k * N + i], 0), r3 - 1) - lo;
      if (ix >= 0 && ix < part) atomicAdd(slab + ix, mul_rn(wg[(size_t)k * N + i], gv)); // ds_add_f32
    }
  }
  __syncthreads();
  float *o = dx + row * r3 + lo;
  for (int v = tid * 4; v < part; v += nt * 4)
    *reinterpret_cast<float4 *>(o + v) = *reinterpret_cast<const float4 *>(slab + v);
}

// Fallback for slabs that do not fit LDS (r^3 * 4 > 128 KiB or r^3 % 4 != 0).
__global__ void devox_bwd_atomic_kernel(const float *__restrict__ gy,
                                        const int32_t *__restrict__ inds,
                                        const float *__restrict__ wgts, int C, int N, int r3,
                                        int CT, float *__restrict__ gx) {
  const int b = blockIdx.z, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int ix[8];
  float w[8];
  for (int q = 0; q < 8; ++q) {
    ix[q] = min(max(inds[((size_t)b * 8 + q) * N + i], 0), r3 - 1);
    w[q] = wgts[((size_t)b * 8 + q) * N + i];
  }
  const int c0 = blockIdx.y * CT, c1 = min(C, c0 + CT);
  for (int c = c0; c < c1; ++c) {
    const float gv = gy[((size_t)b * C + c) * N + i];
    for (int q = 0; q < 8; ++q) atomicAdd(gx + ((size_t)b * C + c) * r3 + ix[q], mul_rn(w[q], gv));
  }
}

} // namespace

constexpr int DVR_G = 8; // floats per piece: 32-byte pieces
constexpr size_t DVR_LDS = (size_t)DVR_RING_BYTES + (size_t)2 * (DVR_R * DVR_R * (DVR_R / DVR_G) / 32) * 4 +
                           (size_t)(DVR_NJ * DVR_NT / (DVR_G / 4)) * 2 + 32 * 4 + 16;
static int devox_ring_depth() { // LION_DEVOX_RING: A/B switch (2 = one channel in flight, the round-3 schedule)
  static const int d = getenv("LION_DEVOX_RING") ? atoi(getenv("LION_DEVOX_RING")) : 4;
  return d < 2 ? 2 : d;
}

extern "C" {

static int devox_launch(const float *coords, const float *feat, int B, int C, int N, int r,
                        int training, float *out, int32_t *inds, float *wgts, const float *scale,
                        const float *shift, hipStream_t st) {
  const int r2 = r * r;
  // r = 32 (the large calls): compacted needed pieces of rows through a ring of LDS buffers
  if (r == DVR_R && N <= 2048 && (((uintptr_t)feat) & 15) == 0) {
    int CT = 8; // one workgroup per CU: the per-cloud setup is amortised over CT channels
    while (CT > 1 && (long)B * lion_cdiv(C, CT) < 256) CT >>= 1;
    dim3 grid(lion_cdiv(C, CT), B);
    static LionLdsLimit cfgr0 = {}, cfgr1 = {};
    if (scale) {
      if (int e = lion_dynamic_lds(&devox_ring_kernel<true, DVR_G, 0>, DVR_LDS, cfgr1)) return e;
      devox_ring_kernel<true, DVR_G, 0><<<grid, DVR_NT, DVR_LDS, st>>>(coords, feat, C, N, CT, training, out, inds, wgts, scale, shift, devox_ring_depth(), nullptr);
    } else {
      if (int e = lion_dynamic_lds(&devox_ring_kernel<false, DVR_G, 0>, DVR_LDS, cfgr0)) return e;
      devox_ring_kernel<false, DVR_G, 0><<<grid, DVR_NT, DVR_LDS, st>>>(coords, feat, C, N, CT, training, out, inds, wgts, scale, shift, devox_ring_depth(), nullptr);
    }
    LION_LAUNCH_CHECK();
    return 0;
  }
  // LDS-staged path: slabs of <= 9216 floats (36 KiB), two buffers
  if ((r2 % 4) == 0 && N <= 2048 && r2 <= 4608 && (((uintptr_t)feat) & 15) == 0) {
    const int planes_max = 9216 / r2;
    const int PX = planes_max >= r ? r : planes_max - 1;
    const int XS = lion_cdiv(r, PX);
    // small grids: CI whole channel grids per slab (<= 36 KiB), so that an iteration is worth its barrier
    const int CI = XS == 1 ? (9216 / (r2 * r) < 16 ? 9216 / (r2 * r) : 16) : 1;
    const int ld0 = lion_cdiv((XS == 1 ? CI * r2 * r : (PX + 1) * r2) / 4, 256);
    const int ld = ld0 <= 1 ? 1 : ld0 <= 2 ? 2 : ld0 <= 4 ? 4 : 9;
    const int slab_floats = ld * 1024; // buffer stride: every lane of every DMA instruction lands inside it
    const size_t lds = (size_t)2 * slab_floats * 4 + (size_t)((N + 1) & ~1) * 2 + (size_t)XS * 4 +
                       (size_t)((r2 + 31) / 32) * 4;
    // channels per workgroup: amortise the per-point setup over >= 4 slabs of channels, keep >= 512 workgroups
    int CT = 4 * CI;
    while (CT > CI && (long)B * lion_cdiv(C, CT) < 512) CT >>= 1;
    dim3 grid(lion_cdiv(C, CT), B);
#define DEVOX_SLAB(LD_)                                                                                    \
  {                                                                                                        \
    static LionLdsLimit cfg0 = {}, cfg1 = {};                                                              \
    if (scale) {                                                                                           \
      if (int e = lion_dynamic_lds(&devox_slab_kernel<LD_, true>, lds, cfg1)) return e;                    \
      devox_slab_kernel<LD_, true><<<grid, 256, lds, st>>>(coords, feat, C, N, r, PX, XS, CT, CI, slab_floats, \
                                                           training, out, inds, wgts, scale, shift);       \
    } else {                                                                                               \
      if (int e = lion_dynamic_lds(&devox_slab_kernel<LD_, false>, lds, cfg0)) return e;                   \
      devox_slab_kernel<LD_, false><<<grid, 256, lds, st>>>(coords, feat, C, N, r, PX, XS, CT, CI, slab_floats, \
                                                            training, out, inds, wgts, scale, shift);      \
    }                                                                                                      \
  }
    if (ld <= 1) DEVOX_SLAB(1)
    else if (ld <= 2) DEVOX_SLAB(2)
    else if (ld <= 4) DEVOX_SLAB(4)
    else DEVOX_SLAB(9)
#undef DEVOX_SLAB
    LION_LAUNCH_CHECK();
    return 0;
  }
  const int pt = lion_cdiv(N, 256);
  // channel tile: keep >= ~2048 workgroups in flight, amortise the corner computation
  int ct = 16;
  while (ct > 2 && (long)B * pt * lion_cdiv(C, ct) < 2048) ct >>= 1;
  dim3 grid(pt, lion_cdiv(C, ct), B);
#define DEVOX_CASE(CT_)                                                                                   \
  if (scale) devox_fwd_kernel<CT_, true><<<grid, 256, 0, st>>>(coords, feat, C, N, r, training, out, inds, wgts, scale, shift); \
  else devox_fwd_kernel<CT_, false><<<grid, 256, 0, st>>>(coords, feat, C, N, r, training, out, inds, wgts, scale, shift)
  switch (ct) {
  case 16: DEVOX_CASE(16); break;
  case 8:  DEVOX_CASE(8); break;
  case 4:  DEVOX_CASE(4); break;
  default: DEVOX_CASE(2); break;
  }
#undef DEVOX_CASE
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_trilinear_devoxelize_forward(const float *coords, const float *feat, int B, int C, int N,
                                      int r, int training, float *out, int32_t *inds, float *wgts,
                                      lionStream_t stream) {
  if (!coords || !feat || !out || B <= 0 || C <= 0 || N <= 0 || r <= 0) return LION_EINVAL;
  if (training && (!inds || !wgts)) return LION_EINVAL;
  return devox_launch(coords, feat, B, C, N, r, training, out, inds, wgts, nullptr, nullptr,
                      static_cast<hipStream_t>(stream));
}

// devoxelize(scale[b,c] * feat + shift[b,c]) without materialising the scaled grid (inference path).
int lion_trilinear_devoxelize_affine_forward(const float *coords, const float *feat, const float *scale,
                                             const float *shift, int B, int C, int N, int r, float *out,
                                             lionStream_t stream) {
  if (!coords || !feat || !out || !scale || !shift || B <= 0 || C <= 0 || N <= 0 || r <= 0) return LION_EINVAL;
  return devox_launch(coords, feat, B, C, N, r, 0, out, nullptr, nullptr, scale, shift,
                      static_cast<hipStream_t>(stream));
}

// ---- the same in two steps (r = 32, N <= 2048): what the coordinates alone decide is computed once per cloud ----
size_t lion_devoxelize_plan_bytes(int B, int N, int r) {
  if (B <= 0 || N <= 0 || N > 2048 || r != DVR_R) return 0;
  return (size_t)B * devox_plan_stride<DVR_G>();
}

// coords f32[B,3,N] (voxel coordinates, as lion_trilinear_devoxelize_forward) -> plan (lion_devoxelize_plan_bytes bytes,
// 16-byte aligned)
int lion_trilinear_devoxelize_plan(const float *coords, int B, int N, int r, void *plan, size_t plan_bytes,
                                   lionStream_t stream) {
  if (!coords || !plan || B <= 0 || N <= 0) return LION_EINVAL;
  const size_t need = lion_devoxelize_plan_bytes(B, N, r);
  if (!need || (((uintptr_t)plan) & 15) != 0) return LION_EUNSUPPORTED;
  if (plan_bytes < need) return LION_EWORKSPACE;
  static LionLdsLimit cfg = {};
  if (int e = lion_dynamic_lds(&devox_ring_kernel<false, DVR_G, 2>, DVR_LDS, cfg)) return e;
  devox_ring_kernel<false, DVR_G, 2><<<dim3(1, B), DVR_NT, DVR_LDS, static_cast<hipStream_t>(stream)>>>(
      coords, nullptr, 1, N, 1, 0, nullptr, nullptr, nullptr, nullptr, nullptr, devox_ring_depth(),
      static_cast<unsigned char *>(plan));
  LION_LAUNCH_CHECK();
  return 0;
}

// trilinear_devoxelize (eval) of feat f32[B,C,32^3] at the plan's coordinates (the SAME coords the plan was made from);
// scale / shift f32[B,C] (both or neither): the affine form.  Bit-identical to the one-step entry points.
int lion_trilinear_devoxelize_planned_forward(const void *plan, size_t plan_bytes, const float *coords, const float *feat,
                                              const float *scale, const float *shift, int B, int C, int N, int r,
                                              float *out, lionStream_t stream) {
  if (!plan || !coords || !feat || !out || B <= 0 || C <= 0 || N <= 0) return LION_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return LION_EINVAL;
  const size_t need = lion_devoxelize_plan_bytes(B, N, r);
  if (!need || (((uintptr_t)plan) & 15) != 0 || (((uintptr_t)feat) & 15) != 0) return LION_EUNSUPPORTED;
  if (plan_bytes < need) return LION_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int CT = 8;
  while (CT > 1 && (long)B * lion_cdiv(C, CT) < 256) CT >>= 1;
  dim3 grid(lion_cdiv(C, CT), B);
  unsigned char *pl = const_cast<unsigned char *>(static_cast<const unsigned char *>(plan));
  static LionLdsLimit cfg0 = {}, cfg1 = {};
  if (scale) {
    if (int e = lion_dynamic_lds(&devox_ring_kernel<true, DVR_G, 1>, DVR_LDS, cfg1)) return e;
    devox_ring_kernel<true, DVR_G, 1><<<grid, DVR_NT, DVR_LDS, st>>>(coords, feat, C, N, CT, 0, out, nullptr, nullptr, scale, shift, devox_ring_depth(), pl);
  } else {
    if (int e = lion_dynamic_lds(&devox_ring_kernel<false, DVR_G, 1>, DVR_LDS, cfg0)) return e;
    devox_ring_kernel<false, DVR_G, 1><<<grid, DVR_NT, DVR_LDS, st>>>(coords, feat, C, N, CT, 0, out, nullptr, nullptr, nullptr, nullptr, devox_ring_depth(), pl);
  }
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_trilinear_devoxelize_backward(const float *gy, const int32_t *inds, const float *wgts,
                                       int B, int C, int N, int r3, float *gx,
                                       lionStream_t stream) {
  if (!gy || !inds || !wgts || !gx || B <= 0 || C <= 0 || N <= 0 || r3 <= 0) return LION_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if ((size_t)r3 * 4 <= 128 * 1024 && (r3 % 4) == 0) {
    // (64, 2048, 32), B = 32, one box: 1 part 203 us, 2 parts 189 us, 4 parts 254 us, 8 parts 349 us -- every workgroup scans
    // all 8 N (index, weight) pairs, so more parts cost more than their overlap returns
    int parts = (size_t)r3 * 4 > 64 * 1024 ? 2 : 1;
    if (r3 % (4 * parts) != 0) parts = 1;
    const size_t lds = (size_t)(r3 / parts) * 4;
    static LionLdsLimit configured = {};
    if (int e = lion_dynamic_lds(&devox_bwd_lds_kernel, lds, configured)) return e;
    const int nt = r3 >= 16384 ? 1024 : 256;
    devox_bwd_lds_kernel<<<dim3(C, B, parts), nt, lds, st>>>(gy, inds, wgts, C, N, r3, gx);
    LION_LAUNCH_CHECK();
    return 0;
  }
  hipError_t e = hipMemsetAsync(gx, 0, (size_t)B * C * r3 * 4, st);
  if (e != hipSuccess) return (int)e;
  const int CT = 8;
  devox_bwd_atomic_kernel<<<dim3(lion_cdiv(N, 256), lion_cdiv(C, CT), B), 256, 0, st>>>(
      gy, inds, wgts, C, N, r3, CT, gx);
  LION_LAUNCH_CHECK();
  return 0;
}

// dx f32[B,C,r3] = A' scatter(gy) + Q + R x  (see devox_bwd_affine_kernel); x f32[B,C,r3] 16-byte aligned, r3 * 4 <= 128 KiB, r3 % 8 == 0
int lion_trilinear_devoxelize_backward_affine(const float *gy, const int32_t *inds, const float *wgts, const float *x,
                                              const float *Ap, const float *Q, const float *R, int B, int C, int N, int r3,
                                              float *dx, lionStream_t stream) {
  if (!gy || !inds || !wgts || !x || !Ap || !Q || !R || !dx || B <= 0 || C <= 0 || N <= 0 || r3 <= 0) return LION_EINVAL;
  if ((size_t)r3 * 4 > 128 * 1024 || (r3 % 8) != 0 || (((uintptr_t)x | (uintptr_t)dx) & 15) != 0) return LION_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int parts = (size_t)r3 * 4 > 64 * 1024 ? 2 : 1;
  const size_t lds = (size_t)(r3 / parts) * 4;
  static LionLdsLimit configured = {};
  if (int e = lion_dynamic_lds(&devox_bwd_affine_kernel, lds, configured)) return e;
  const int nt = r3 >= 16384 ? 1024 : 256;
  devox_bwd_affine_kernel<<<dim3(C, B, parts), nt, lds, st>>>(gy, inds, wgts, x, Ap, Q, R, C, N, r3, dx);
  LION_LAUNCH_CHECK();
  return 0;
}

} // extern "C"
