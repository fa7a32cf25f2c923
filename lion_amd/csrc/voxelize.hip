// voxelize.hip -- average voxelization (K1, K2, K3) and the fused Voxelization.forward (P1).
//
// Reference: third_party/pvcnn/functional/src/voxelization/vox.cu:18-110, vox.cpp:17-79,
//            models/pvcnn2_ada.py:173-188.
//
// MI355X design (not the reference's "one block per cloud + global float atomics + memset").
// What was measured on the chip (tools/exp/*.hip) shaped it: a plain zero fill of the [B,C,r^3] grid
// with one 16-byte store per lane per channel reaches 7.5-8.0 TB/s; every load that a store has to
// wait for costs dearly once the memory pipe is saturated with stores (two dependent round trips
// + a block barrier in front of the stores: 2.3 TB/s; one int4 load + one batch of independent
// gathers: 5.8 TB/s).  Hence three kernels:
//   1  vox_index_kernel  one 1024-thread workgroup per cloud.  The r^3 occupancy histogram lives in
//      LDS (128 KiB at r=32, padded so the segmented scan is bank-conflict free); points are
//      counting-sorted by (voxel, point index); outputs: ind, the dense count grid, a dense "slot"
//      grid (rank of the voxel among the occupied ones, -1 if empty), the occupied-voxel list
//      (start,count) and the sorted point list.  No memset, no global atomics.  With FUSE_P1 the
//      kernel first normalises the raw float coordinates (mean / max-norm / round-half-even) with
//      the fixed summation tree the oracle documents, so voxel indices are bit-exact.
//   2  vox_mean_kernel   one lane per (occupied voxel, channel tile): sums the voxel's points in
//      ascending point index (bit-exact vs the sequential oracle; 1-2 gathers from the L2-resident
//      feature row on average) into a compact table vmean[B,C,U].
//   3  vox_dense_kernel  writes the dense [C, r^3] grid exactly once: int4 slot load, (rare)
//      independent table gathers for a 16-channel tile, then 16 back-to-back 16-byte stores per lane.
//      >= 94 % of the lanes store zeros and never wait for anything but the slot load.
// HBM traffic ~= the algorithmic 4*B*(3N + C*N + C*r^3 + N + r^3) bytes + the slot grid (4*B*r^3 x2).
//   fallback (r^3 or N too large for LDS): memset + integer/float atomics, like the reference but
//            with a (points x batch) grid.  Within 1e-6 of the oracle, not bit-exact.
#include "common.h"

namespace {

constexpr int VT = 1024;    // threads of the per-cloud index kernel
constexpr int MAXP = 8;     // points per thread kept in registers  (N <= 8192)
constexpr int LDS_LIMIT = 160 * 1024;

__device__ __forceinline__ int padv(int v) { return v + (v >> 5); }
__host__ __device__ inline int align4i(int x) { return (x + 3) & ~3; }

struct IndexLds {
  int hist_words, n_words;
  size_t bytes;
};
static inline IndexLds index_lds(int N, long r3) {
  IndexLds l;
  l.hist_words = align4i((int)(r3 + (r3 >> 5) + 1));
  l.n_words = align4i(N);
  // hist | tmp[N] | ust[N] | 64 floats | 64 ints
  l.bytes = ((size_t)l.hist_words + 2 * (size_t)l.n_words + 128) * 4;
  return l;
}

template <bool FUSE_P1>
__global__ __launch_bounds__(VT) void vox_index_kernel(
    const int32_t *__restrict__ coords_i, const float *__restrict__ coords_f, int N, int r,
    int normalize, float eps, float *__restrict__ norm_coords, int32_t *__restrict__ ind,
    int32_t *__restrict__ cnt, int32_t *__restrict__ vslot, int32_t *__restrict__ sorted,
    int32_t *__restrict__ uinfo, int32_t *__restrict__ ucount, int hist_words, int n_words) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t *hist = reinterpret_cast<uint32_t *>(smem); // count, later (urank << 16) | count
  int32_t *tmp = reinterpret_cast<int32_t *>(hist + hist_words);
  int32_t *ust = tmp + n_words;                         // (start << 16) | count per occupied voxel
  float *fscr = reinterpret_cast<float *>(ust + n_words); // 64 floats
  int *iscr = reinterpret_cast<int *>(fscr + 64);         // 64 ints

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int r2 = r * r, r3 = r2 * r;

  for (int v = tid; v < hist_words; v += VT) hist[v] = 0u;

  int myv[MAXP];
  if (FUSE_P1) {
    const float *co = coords_f + (size_t)b * 3 * N;
    float px[MAXP], py[MAXP], pz[MAXP];
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int i = tid + p * VT;
      px[p] = py[p] = pz[p] = 0.0f;
      if (i < N) { px[p] = co[i]; py[p] = co[i + N]; pz[p] = co[i + 2 * N]; }
    }
    // mean over the points with the oracle's fixed tree: 1024 strided partials (ascending j),
    // then p[t] += p[t+s], s = 1..512.  An xor butterfly evaluates the same tree bit for bit.
    float part[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int i = tid + p * VT;
      if (i < N) {
        part[0] = add_rn(part[0], px[p]);
        part[1] = add_rn(part[1], py[p]);
        part[2] = add_rn(part[2], pz[p]);
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int s = 1; s < 64; s <<= 1) part[a] = add_rn(part[a], __shfl_xor(part[a], s, 64));
      if (lane == 0) fscr[a * 16 + wave] = part[a];
    }
    __syncthreads();
    float mean[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float v = fscr[a * 16 + (lane & 15)];
#pragma unroll
      for (int s = 1; s < 16; s <<= 1) v = add_rn(v, __shfl_xor(v, s, 64));
      mean[a] = div_rn(v, (float)N);
    }
    float denom = 1.0f;
    if (normalize) {
      float mx = 0.0f;
#pragma unroll
      for (int p = 0; p < MAXP; ++p) {
        const int i = tid + p * VT;
        if (i < N) {
          const float x = sub_rn(px[p], mean[0]), y = sub_rn(py[p], mean[1]),
                      z = sub_rn(pz[p], mean[2]);
          const float nr = sqrt_rn(add_rn(add_rn(mul_rn(x, x), mul_rn(y, y)), mul_rn(z, z)));
          mx = nr > mx ? nr : mx;
        }
      }
#pragma unroll
      for (int s = 1; s < 64; s <<= 1) { const float o = __shfl_xor(mx, s, 64); mx = o > mx ? o : mx; }
      if (lane == 0) fscr[48 + wave] = mx;
      __syncthreads();
      float m2 = fscr[48 + (lane & 15)];
#pragma unroll
      for (int s = 1; s < 16; s <<= 1) { const float o = __shfl_xor(m2, s, 64); m2 = o > m2 ? o : m2; }
      denom = add_rn(mul_rn(m2, 2.0f), eps);
    }
    float *nc = norm_coords + (size_t)b * 3 * N;
    const float rf = (float)r, hi = (float)(r - 1);
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
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
          nc[i + a * N] = v;
          q[a] = (int)rintf(v);
        }
        myv[p] = q[0] * r2 + q[1] * r + q[2];
      }
    }
  } else {
    const int32_t *co = coords_i + (size_t)b * 3 * N;
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int i = tid + p * VT;
      myv[p] = 0;
      if (i < N) myv[p] = co[i] * r2 + co[i + N] * r + co[i + 2 * N]; // vox.cu:31
    }
  }

  int myarr[MAXP];
  __syncthreads(); // hist zeroed
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int i = tid + p * VT;
    myarr[p] = 0;
    if (i < N) {
      ind[(size_t)b * N + i] = myv[p];
      // memory safety only: coordinates outside [0,r) are a caller error (as in the reference)
      myv[p] = min(max(myv[p], 0), r3 - 1);
      myarr[p] = (int)atomicAdd(&hist[padv(myv[p])], 1u);
    }
  }
  __syncthreads();

  // exclusive scan of the histogram: thread t owns voxels [t*VPT, (t+1)*VPT) (conflict free
  // thanks to the +v/32 padding); one packed block scan gives both the point offset (start) and
  // the rank among occupied voxels; each entry becomes (urank << 16) | count.
  const int VPT = (r3 + VT - 1) / VT;
  const int v0 = min(tid * VPT, r3), v1 = min(v0 + VPT, r3);
  int lp = 0, lo = 0;
  for (int v = v0; v < v1; ++v) { const int c = (int)hist[padv(v)]; lp += c; lo += (c > 0); }
  const int packed = (lp << 16) | lo; // points < 2^14, occupied <= N < 2^14
  const int incl = wave_incl_scan(packed, lane);
  if (lane == 63) iscr[wave] = incl;
  __syncthreads();
  int excl = incl - packed;
  for (int w = 0; w < wave; ++w) excl += iscr[w];
  int run = excl >> 16, urun = excl & 0xffff;
  for (int v = v0; v < v1; ++v) {
    const uint32_t c = hist[padv(v)];
    if (c) {
      hist[padv(v)] = ((uint32_t)urun << 16) | c;
      ust[urun] = (run << 16) | (int)c;
      ++urun;
      run += (int)c;
    }
  }
  if (tid == VT - 1) ucount[b] = urun;
  __syncthreads();

  // bucket placement in arrival order, then the deterministic rank inside the bucket
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int i = tid + p * VT;
    if (i < N) tmp[(ust[hist[padv(myv[p])] >> 16] >> 16) + myarr[p]] = i;
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int i = tid + p * VT;
    if (i < N) {
      const int h = ust[hist[padv(myv[p])] >> 16];
      const int s = h >> 16, c = h & 0xffff;
      int rank = 0;
      for (int k = 0; k < c; ++k) rank += (tmp[s + k] < i) ? 1 : 0;
      sorted[(size_t)b * N + s + rank] = i;
    }
  }
  // dense outputs, 16 bytes per lane: counts and slots (r3 % 4 == 0)
  for (int v = tid * 4; v < r3; v += VT * 4) {
    int4 c, sl;
    const uint32_t h0 = hist[padv(v)], h1 = hist[padv(v + 1)], h2 = hist[padv(v + 2)], h3 = hist[padv(v + 3)];
    c.x = h0 & 0xffff; c.y = h1 & 0xffff; c.z = h2 & 0xffff; c.w = h3 & 0xffff;
    sl.x = c.x ? (int)(h0 >> 16) : -1; sl.y = c.y ? (int)(h1 >> 16) : -1;
    sl.z = c.z ? (int)(h2 >> 16) : -1; sl.w = c.w ? (int)(h3 >> 16) : -1;
    *reinterpret_cast<int4 *>(cnt + (size_t)b * r3 + v) = c;
    *reinterpret_cast<int4 *>(vslot + (size_t)b * r3 + v) = sl;
  }
  for (int u = tid; u < N; u += VT) uinfo[(size_t)b * N + u] = ust[u]; // entries >= U are unused
}

// vmean[b][c][u] = sum_k feat[b][c][p_k] * (1/n), p_k ascending (vox.cu:59-71 with the order fixed
// to ascending point index, as in the oracle).  One workgroup per (batch, tile of CT channels): the
// CT feature rows are staged in LDS with coalesced 16-byte loads (random 4-byte gathers straight
// from global memory cost one TA cycle per lane: 53 us for 3.2 M gathers; from LDS they are free),
// then lanes walk the occupied voxels and write the table coalesced.
__global__ __launch_bounds__(256) void vox_mean_kernel(const float *__restrict__ feat,
                                                       const int32_t *__restrict__ sorted,
                                                       const int32_t *__restrict__ uinfo,
                                                       const int32_t *__restrict__ ucount, int C,
                                                       int N, int CT, float *__restrict__ vmean) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *rows = reinterpret_cast<float *>(smem);            // [CT][N]
  int32_t *ssrt = reinterpret_cast<int32_t *>(rows + (size_t)CT * N); // [N] sorted point list
  int32_t *sinf = ssrt + N;                                  // [N] (start << 16) | count
  const int tid = threadIdx.x, b = blockIdx.y;
  const int c0 = blockIdx.x * CT, nc = min(CT, C - c0);
  const float *src = feat + ((size_t)b * C + c0) * N;
  const int tot = nc * N;
  if ((N & 3) == 0) {
    for (int i = tid * 4; i < tot; i += 1024)
      *reinterpret_cast<float4 *>(rows + i) = *reinterpret_cast<const float4 *>(src + i);
  } else {
    for (int i = tid; i < tot; i += 256) rows[i] = src[i];
  }
  // everything a lane will need comes in with this one batch of coalesced loads (a per-voxel
  // uinfo -> sorted -> feature chain of dependent global loads was 7x slower)
  const int U = ucount[b];
  for (int i = tid; i < N; i += 256) ssrt[i] = sorted[(size_t)b * N + i];
  for (int i = tid; i < U; i += 256) sinf[i] = uinfo[(size_t)b * N + i];
  __syncthreads();
  float *dst = vmean + ((size_t)b * C + c0) * N;
  for (int u = tid; u < U; u += 256) {
    const int info = sinf[u], st = info >> 16, n = info & 0xffff;
    const float inv = div_rn(1.0f, (float)n); // vox.cu:66 (== float(1.0 / cnt))
    const int32_t *srt = ssrt + st;
    if (n <= 4) {
      int p[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) p[k] = srt[k < n ? k : 0];
      for (int c = 0; c < nc; ++c) {
        const float *rw = rows + c * N;
        float acc = mul_rn(rw[p[0]], inv); // 0 + x == x
        const float a1 = add_rn(acc, mul_rn(rw[p[1]], inv));
        acc = n > 1 ? a1 : acc;
        const float a2 = add_rn(acc, mul_rn(rw[p[2]], inv));
        acc = n > 2 ? a2 : acc;
        const float a3 = add_rn(acc, mul_rn(rw[p[3]], inv));
        acc = n > 3 ? a3 : acc;
        dst[(size_t)c * N + u] = acc;
      }
    } else {
      for (int c = 0; c < nc; ++c) {
        const float *rw = rows + c * N;
        float acc = 0.f;
        for (int k = 0; k < n; ++k) acc = add_rn(acc, mul_rn(rw[srt[k]], inv));
        dst[(size_t)c * N + u] = acc;
      }
    }
  }
}

// Dense write: 128-thread workgroups, 512 voxels x CT channels each.
template <int CT>
__global__ __launch_bounds__(128) void vox_dense_kernel(const int32_t *__restrict__ vslot,
                                                        const float *__restrict__ vmean, int C,
                                                        int N, int r3, float *__restrict__ out) {
  const int b = blockIdx.z, c0 = blockIdx.y * CT;
  const int v0 = (blockIdx.x * 128 + threadIdx.x) * 4;
  if (v0 >= r3) return;
  float *ob = out + ((size_t)b * C + c0) * r3 + v0;
  const int4 s = *reinterpret_cast<const int4 *>(vslot + (size_t)b * r3 + v0);
  const int nc = min(CT, C - c0);
  if ((s.x & s.y & s.z & s.w) < 0) { // all four slots are -1: empty
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < nc) *reinterpret_cast<float4 *>(ob + (size_t)c * r3) = z;
    return;
  }
  const float *tb = vmean + ((size_t)b * C + c0) * N;
  float4 v[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) { // all gathers first ...
    v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < nc) {
      if (s.x >= 0) v[c].x = tb[(size_t)c * N + s.x];
      if (s.y >= 0) v[c].y = tb[(size_t)c * N + s.y];
      if (s.z >= 0) v[c].z = tb[(size_t)c * N + s.z];
      if (s.w >= 0) v[c].w = tb[(size_t)c * N + s.w];
    }
  }
#pragma unroll
  for (int c = 0; c < CT; ++c) // ... then nothing but stores
    if (c < nc) *reinterpret_cast<float4 *>(ob + (size_t)c * r3) = v[c];
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
  IndexLds lds;
  size_t off_sorted, off_uinfo, off_ucount, off_vslot, off_vmean, off_vox, total;
};

static inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

static VoxPlan make_plan(int B, int C, int N, int r) {
  VoxPlan p;
  const long r3 = (long)r * r * r;
  p.lds = index_lds(N, r3 < (1L << 20) ? r3 : 0);
  p.fast = r3 <= (1L << 17) && (r3 % 4 == 0) && N <= MAXP * VT && N < 16384 && N >= 1 &&
           p.lds.bytes <= (size_t)LDS_LIMIT;
  size_t o = 0;
  p.off_sorted = o; o += up256((size_t)B * N * 4);
  p.off_uinfo = o;  o += up256((size_t)B * N * 4);
  p.off_ucount = o; o += up256((size_t)B * 4);
  p.off_vslot = o;  o += up256((size_t)B * r3 * 4);
  p.off_vmean = o;  o += up256((size_t)B * (C > 0 ? C : 1) * N * 4);
  p.off_vox = o;    o += up256((size_t)B * 3 * N * 4); // int coords (fallback P1)
  p.total = o;
  return p;
}

template <bool FUSE_P1>
static int set_index_lds(size_t bytes) {
  static size_t configured = 0;
  if (bytes > configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&vox_index_kernel<FUSE_P1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    configured = bytes;
  }
  return 0;
}

static int launch_mean_dense(const float *feat, const int32_t *sorted, const int32_t *uinfo,
                             const int32_t *ucount, const int32_t *vslot, float *vmean, int B, int C,
                             int N, int r3, float *out, hipStream_t st) {
  {
    int ct = (int)((32 * 1024) / ((size_t)N * 4)); // rows per workgroup: <= 32 KiB of LDS (+ 2 index rows)
    if (ct < 1) ct = 1;                           // N <= 8192 on this path -> one row is <= 32 KiB
    if (ct > 8) ct = 8;
    while (ct > 1 && (long)B * lion_cdiv(C, ct) < 512) ct >>= 1;
    vox_mean_kernel<<<dim3(lion_cdiv(C, ct), B), 256, ((size_t)ct + 2) * N * 4, st>>>(
        feat, sorted, uinfo, ucount, C, N, ct, vmean);
    LION_LAUNCH_CHECK();
  }
  const int vt = lion_cdiv(r3, 512);
  int ct = 16;
  while (ct > 4 && (long)B * vt * lion_cdiv(C, ct) < 2048) ct >>= 1;
  dim3 grid(vt, lion_cdiv(C, ct), B);
  switch (ct) {
  case 16: vox_dense_kernel<16><<<grid, 128, 0, st>>>(vslot, vmean, C, N, r3, out); break;
  case 8:  vox_dense_kernel<8><<<grid, 128, 0, st>>>(vslot, vmean, C, N, r3, out); break;
  default: vox_dense_kernel<4><<<grid, 128, 0, st>>>(vslot, vmean, C, N, r3, out); break;
  }
  LION_LAUNCH_CHECK();
  return 0;
}

static int voxelize_impl(const float *feat, const int32_t *coords_i, const float *coords_f, int B,
                         int C, int N, int r, int normalize, float eps, float *out,
                         float *norm_coords, int32_t *ind, int32_t *cnt, void *ws, size_t ws_bytes,
                         hipStream_t st) {
  if (B <= 0 || N <= 0 || r <= 0 || !ind || !cnt) return LION_EINVAL;
  if (feat && (C <= 0 || !out)) return LION_EINVAL;
  if (!coords_i && !coords_f) return LION_EINVAL;
  if (coords_f && !norm_coords) return LION_EINVAL;
  if ((long)r * r * r > (1L << 30)) return LION_EUNSUPPORTED;
  const VoxPlan p = make_plan(B, feat ? C : 0, N, r);
  if (!ws || ws_bytes < p.total) return LION_EWORKSPACE;
  const int r3 = r * r * r;
  char *w = static_cast<char *>(ws);
  int32_t *sorted = reinterpret_cast<int32_t *>(w + p.off_sorted);
  int32_t *uinfo = reinterpret_cast<int32_t *>(w + p.off_uinfo);
  int32_t *ucount = reinterpret_cast<int32_t *>(w + p.off_ucount);
  int32_t *vslot = reinterpret_cast<int32_t *>(w + p.off_vslot);
  float *vmean = reinterpret_cast<float *>(w + p.off_vmean);
  if (p.fast) {
    if (coords_f) {
      int e = set_index_lds<true>(p.lds.bytes);
      if (e) return e;
      vox_index_kernel<true><<<B, VT, p.lds.bytes, st>>>(nullptr, coords_f, N, r, normalize, eps,
                                                         norm_coords, ind, cnt, vslot, sorted, uinfo,
                                                         ucount, p.lds.hist_words, p.lds.n_words);
    } else {
      int e = set_index_lds<false>(p.lds.bytes);
      if (e) return e;
      vox_index_kernel<false><<<B, VT, p.lds.bytes, st>>>(coords_i, nullptr, N, r, 0, 0.f, nullptr,
                                                          ind, cnt, vslot, sorted, uinfo, ucount,
                                                          p.lds.hist_words, p.lds.n_words);
    }
    LION_LAUNCH_CHECK();
    if (feat)
      return launch_mean_dense(feat, sorted, uinfo, ucount, vslot, vmean, B, C, N, r3, out, st);
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
