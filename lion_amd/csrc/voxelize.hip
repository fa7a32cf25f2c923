// voxelize.hip -- average voxelization (K1, K2, K3) and the fused Voxelization.forward (P1).
//
// Reference: third_party/pvcnn/functional/src/voxelization/vox.cu:18-110, vox.cpp:17-79,
//            models/pvcnn2_ada.py:173-188.
//
// MI355X design (not the reference's "one block per cloud + global float atomics + memset"):
//   phase 1  vox_index_kernel   one 1024-thread workgroup per cloud.  The r^3 occupancy histogram
//            lives in LDS (128 KiB at r=32, padded so the segmented scan is bank-conflict free),
//            points are counting-sorted by (voxel, point index) and the sorted list, the dense
//            count grid and the per-1024-voxel chunk offsets go to HBM with coalesced stores.
//            No memset, no global atomics.  With FUSE_P1 the same kernel first normalises the raw
//            float coordinates (mean / max-norm / round-half-even) using the fixed summation tree
//            the oracle documents, so voxel indices are bit-exact.
//   phase 2  vox_dense_kernel   writes the dense [C, r^3] grid exactly once with 16-byte coalesced
//            stores straight from registers (>= 94 % of it is zeros); an occupied voxel gathers its
//            points from the L2-resident feature rows in ascending point index, so every output
//            float is bit-identical to the sequential oracle.  HBM traffic ~= the algorithmic
//            4*B*(3N + C*N + C*r^3 + N + r^3) bytes.
//   fallback (r^3 or N too large for LDS): memset + integer/float atomics, like the reference but
//            with a (points x batch) grid.  Within 1e-6 of the oracle, not bit-exact.
#include "common.h"

namespace {

constexpr int VT = 1024;    // threads of the per-cloud index kernel
constexpr int MAXP = 8;     // points per thread kept in registers  (N <= 8192)
constexpr int CHUNK = 1024; // voxels per dense-write workgroup (256 threads x 4)
constexpr int LDS_LIMIT = 160 * 1024;

__device__ __forceinline__ int padv(int v) { return v + (v >> 5); }
__host__ __device__ inline int align4i(int x) { return (x + 3) & ~3; }

struct IndexLds {
  int hist_words, tmp_words;
  size_t bytes;
};
static inline IndexLds index_lds(int N, int r3) {
  IndexLds l;
  l.hist_words = align4i(r3 + (r3 >> 5) + 1);
  l.tmp_words = align4i(N);
  l.bytes = (size_t)(l.hist_words + l.tmp_words + 128) * 4;
  return l;
}

template <bool FUSE_P1>
__global__ __launch_bounds__(VT) void vox_index_kernel(
    const int32_t *__restrict__ coords_i, const float *__restrict__ coords_f, int N, int r,
    int normalize, float eps, float *__restrict__ norm_coords, int32_t *__restrict__ ind,
    int32_t *__restrict__ cnt, int32_t *__restrict__ sorted, int32_t *__restrict__ chunk_start,
    int nchunks, int hist_words, int tmp_words) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t *hist = reinterpret_cast<uint32_t *>(smem);
  int32_t *tmp = reinterpret_cast<int32_t *>(hist + hist_words);
  float *fscr = reinterpret_cast<float *>(tmp + tmp_words); // 64 floats
  int *iscr = reinterpret_cast<int *>(fscr + 64);           // 64 ints

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
    __syncthreads();
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
  // thanks to the +v/32 padding), block scan of the per-thread totals, then rewrite each
  // entry as (start << 16) | count.
  const int VPT = (r3 + VT - 1) / VT;
  const int v0 = min(tid * VPT, r3), v1 = min(v0 + VPT, r3);
  int local = 0;
  for (int v = v0; v < v1; ++v) local += (int)hist[padv(v)];
  const int incl = wave_incl_scan(local, lane);
  if (lane == 63) iscr[wave] = incl;
  __syncthreads();
  int run = incl - local;
  for (int w = 0; w < wave; ++w) run += iscr[w];
  for (int v = v0; v < v1; ++v) {
    const uint32_t c = hist[padv(v)];
    hist[padv(v)] = ((uint32_t)run << 16) | c;
    run += (int)c;
  }
  __syncthreads();

  // bucket placement in arrival order, then the deterministic rank inside the bucket
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int i = tid + p * VT;
    if (i < N) tmp[(hist[padv(myv[p])] >> 16) + myarr[p]] = i;
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int i = tid + p * VT;
    if (i < N) {
      const uint32_t h = hist[padv(myv[p])];
      const int s = (int)(h >> 16), c = (int)(h & 0xffffu);
      int rank = 0;
      for (int k = 0; k < c; ++k) rank += (tmp[s + k] < i) ? 1 : 0;
      sorted[(size_t)b * N + s + rank] = i;
    }
  }
  for (int v = tid; v < r3; v += VT) cnt[(size_t)b * r3 + v] = (int32_t)(hist[padv(v)] & 0xffffu);
  for (int q = tid; q <= nchunks; q += VT)
    chunk_start[(size_t)b * (nchunks + 1) + q] =
        (q * CHUNK < r3) ? (int32_t)(hist[padv(q * CHUNK)] >> 16) : N;
}

// Dense write of one (batch, channel tile, 1024-voxel chunk).
template <int CT>
__global__ __launch_bounds__(256) void vox_dense_kernel(
    const float *__restrict__ feat, const int32_t *__restrict__ cnt,
    const int32_t *__restrict__ sorted, const int32_t *__restrict__ chunk_start, int C, int N,
    int r3, int nchunks, float *__restrict__ out) {
  __shared__ int wt[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = blockIdx.x, b = blockIdx.z;
  const int c_begin = blockIdx.y * CT, c_end = min(C, c_begin + CT);
  const int v0 = chunk * CHUNK + tid * 4;
  const bool inb = v0 < r3; // r3 % 4 == 0 is guaranteed by the launcher
  const int cs0 = chunk_start[(size_t)b * (nchunks + 1) + chunk];
  const int cs1 = chunk_start[(size_t)b * (nchunks + 1) + chunk + 1];
  float *obase = out + ((size_t)b * C) * r3 + v0;

  if (cs0 == cs1) { // whole chunk empty (wave-uniform): pure zero streaming
    if (inb) {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int c = c_begin; c < c_end; ++c) *reinterpret_cast<float4 *>(obase + (size_t)c * r3) = z;
    }
    return;
  }

  int c4[4] = {0, 0, 0, 0};
  if (inb) {
    const int4 t = *reinterpret_cast<const int4 *>(cnt + (size_t)b * r3 + v0);
    c4[0] = t.x; c4[1] = t.y; c4[2] = t.z; c4[3] = t.w;
  }
  const int total = c4[0] + c4[1] + c4[2] + c4[3];
  const int incl = wave_incl_scan(total, lane);
  if (lane == 63) wt[wave] = incl;
  __syncthreads();
  int start = cs0 + incl - total;
  for (int w = 0; w < wave; ++w) start += wt[w];

  const int32_t *srt = sorted + (size_t)b * N;
  float inv[4];
  int off[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    inv[q] = c4[q] > 0 ? div_rn(1.0f, (float)c4[q]) : 0.0f; // vox.cu:66 (== float(1.0/cnt))
    off[q] = (q == 0) ? 0 : off[q - 1] + c4[q - 1];
  }
  // common case: <= 4 points in these 4 voxels -> member indices in registers
  int mi[4] = {0, 0, 0, 0};
  if (total > 0 && total <= 4) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < total) mi[k] = srt[start + k];
  }

  for (int c = c_begin; c < c_end; ++c) {
    float val[4] = {0.f, 0.f, 0.f, 0.f};
    if (total > 0) {
      const float *frow = feat + ((size_t)b * C + c) * N;
      if (total <= 4) {
        float f[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) f[k] = (k < total) ? frow[mi[k]] : 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (k >= off[q] && k < off[q] + c4[q]) val[q] = add_rn(val[q], mul_rn(f[k], inv[q]));
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float acc = 0.f;
          for (int k = 0; k < c4[q]; ++k)
            acc = add_rn(acc, mul_rn(frow[srt[start + off[q] + k]], inv[q]));
          val[q] = acc;
        }
      }
    }
    if (inb)
      *reinterpret_cast<float4 *>(obase + (size_t)c * r3) = make_float4(val[0], val[1], val[2], val[3]);
  }
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
  int nchunks;
  IndexLds lds;
  size_t off_sorted, off_chunk, off_vox, total;
};

static VoxPlan make_plan(int B, int N, int r) {
  VoxPlan p;
  const long r3 = (long)r * r * r;
  p.nchunks = (int)((r3 + CHUNK - 1) / CHUNK);
  p.lds = index_lds(N, (int)(r3 < (1 << 24) ? r3 : 0));
  p.fast = r3 <= 65536 * 2 && (r3 % 4 == 0) && N <= MAXP * VT && N <= 65535 && N >= 1 &&
           p.lds.bytes <= (size_t)LDS_LIMIT;
  size_t o = 0;
  p.off_sorted = o; o += ((size_t)B * N * 4 + 255) & ~(size_t)255;
  p.off_chunk = o;  o += ((size_t)B * (p.nchunks + 1) * 4 + 255) & ~(size_t)255;
  p.off_vox = o;    o += ((size_t)B * 3 * N * 4 + 255) & ~(size_t)255; // int coords (fallback P1)
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

static int pick_ct(int B, int C, int nchunks) {
  // enough workgroups to fill 256 CUs several times over, as few channel tiles as possible so the
  // count grid is re-read as little as possible.
  int ct = 64;
  while (ct > 4 && (long)B * nchunks * ((C + ct - 1) / ct) < 2048) ct >>= 1;
  return ct;
}

static int launch_dense(const float *feat, const int32_t *cnt, const int32_t *sorted,
                        const int32_t *chunk_start, int B, int C, int N, int r3, int nchunks,
                        float *out, hipStream_t st) {
  const int ct = pick_ct(B, C, nchunks);
  dim3 grid(nchunks, (C + ct - 1) / ct, B);
  switch (ct) {
  case 64: vox_dense_kernel<64><<<grid, 256, 0, st>>>(feat, cnt, sorted, chunk_start, C, N, r3, nchunks, out); break;
  case 32: vox_dense_kernel<32><<<grid, 256, 0, st>>>(feat, cnt, sorted, chunk_start, C, N, r3, nchunks, out); break;
  case 16: vox_dense_kernel<16><<<grid, 256, 0, st>>>(feat, cnt, sorted, chunk_start, C, N, r3, nchunks, out); break;
  case 8:  vox_dense_kernel<8><<<grid, 256, 0, st>>>(feat, cnt, sorted, chunk_start, C, N, r3, nchunks, out); break;
  default: vox_dense_kernel<4><<<grid, 256, 0, st>>>(feat, cnt, sorted, chunk_start, C, N, r3, nchunks, out); break;
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
  const VoxPlan p = make_plan(B, N, r);
  if (!ws || ws_bytes < p.total) return LION_EWORKSPACE;
  const int r3 = r * r * r;
  char *w = static_cast<char *>(ws);
  int32_t *sorted = reinterpret_cast<int32_t *>(w + p.off_sorted);
  int32_t *chunk_start = reinterpret_cast<int32_t *>(w + p.off_chunk);
  if (p.fast) {
    if (coords_f) {
      int e = set_index_lds<true>(p.lds.bytes);
      if (e) return e;
      vox_index_kernel<true><<<B, VT, p.lds.bytes, st>>>(nullptr, coords_f, N, r, normalize, eps,
                                                         norm_coords, ind, cnt, sorted, chunk_start,
                                                         p.nchunks, p.lds.hist_words, p.lds.tmp_words);
    } else {
      int e = set_index_lds<false>(p.lds.bytes);
      if (e) return e;
      vox_index_kernel<false><<<B, VT, p.lds.bytes, st>>>(coords_i, nullptr, N, r, 0, 0.f, nullptr,
                                                          ind, cnt, sorted, chunk_start, p.nchunks,
                                                          p.lds.hist_words, p.lds.tmp_words);
    }
    LION_LAUNCH_CHECK();
    if (feat) return launch_dense(feat, cnt, sorted, chunk_start, B, C, N, r3, p.nchunks, out, st);
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
  return make_plan(B, N, r).total;
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
