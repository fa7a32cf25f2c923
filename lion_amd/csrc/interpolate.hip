// interpolate.hip -- K11 three nearest neighbours + K12 inverse-distance interpolation.
//
// Reference: third_party/pvcnn/functional/src/interpolate/neighbor_interpolate.cu:20-75 (3-NN,
// one thread per point scanning the centres from global memory), :90-116 (interpolate),
// :145-170 (gradient, 3 global float atomics per element), neighbor_interpolate.cpp.
//
// 3-NN: a lane owns a point, the centres are staged in LDS as SoA tiles (every lane reads the same
// centre -> LDS broadcast), grid = (point tiles, batch).  The reference keeps its running bests
// in double but only ever stores float distances in them, so float compares are equivalent; the
// initial 1e40 is represented by +inf (1e40 > FLT_MAX) and clamps to 1e10 exactly as 1e40 does.
// Weight arithmetic follows :58-73 literally (products of two floats formed in double and
// rounded to float == the correctly rounded float product).  Indices and weights are bit-exact.
#include "common.h"
#include <math.h>

namespace {

constexpr int NN_TILE = 2048;
constexpr int NN_SPLIT = 4; // lanes per point

// insert candidate (d, i) into the running three: strictly smaller distance, or -- LEX only -- equal distance and lower index
template <bool LEX>
__device__ __forceinline__ void nn_insert(float d, int i, float &b0, float &b1, float &b2, int &i0, int &i1, int &i2) {
  const bool lt2 = d < b2 || (LEX && d == b2 && i < i2);
  if (lt2) {
    const bool lt1 = d < b1 || (LEX && d == b1 && i < i1);
    const bool lt0 = d < b0 || (LEX && d == b0 && i < i0);
    // selects, not nested branches: the three lists shift by at most one place
    b2 = lt1 ? b1 : d;  i2 = lt1 ? i1 : i;
    b1 = lt0 ? b0 : (lt1 ? d : b1);  i1 = lt0 ? i0 : (lt1 ? i : i1);
    b0 = lt0 ? d : b0;  i0 = lt0 ? i : i0;
  }
}

// Round 5: FOUR lanes per point, lane q scans the centres q, q + 4, q + 8, ... of the LDS tile (a quad reads four
// consecutive words: conflict free, the 16 quads of a wave read the same four), four distances are formed before the
// running three are touched, and the quad's lists are merged by (distance, index) -- which is the order the reference's
// one-thread scan with strict '<' produces: its result is the three smallest centres by (distance, index), whatever the
// order they are met in.  The round-1 kernel (one lane per point, 256 workgroups for (2048, 1024) at B = 32: one wave per
// SIMD walking 1024 dependent compare chains) took ~100 us of the step's 153 us of 3-NN; bit-exact as before.
__global__ __launch_bounds__(256) void three_nn_kernel(const float *__restrict__ points,
                                                       const float *__restrict__ centers, int N,
                                                       int M, int32_t *__restrict__ idx,
                                                       float *__restrict__ wgt) {
  __shared__ float sx[NN_TILE], sy[NN_TILE], sz[NN_TILE];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int q = tid & (NN_SPLIT - 1);
  const int j = blockIdx.x * (256 / NN_SPLIT) + (tid >> 2);
  const float *pc = points + (size_t)b * 3 * N;
  const float *cc = centers + (size_t)b * 3 * M;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (j < N) { ux = pc[j]; uy = pc[j + N]; uz = pc[j + 2 * N]; }
  float best0 = INFINITY, best1 = INFINITY, best2 = INFINITY;
  int i0 = 0, i1 = 0, i2 = 0;
  for (int t0 = 0; t0 < M; t0 += NN_TILE) {
    const int tn = min(NN_TILE, M - t0);
    __syncthreads();
    for (int k = tid; k < tn; k += 256) {
      sx[k] = cc[t0 + k]; sy[k] = cc[t0 + k + M]; sz[k] = cc[t0 + k + 2 * M];
    }
    __syncthreads();
    int k = q;
    for (; k + 3 * NN_SPLIT < tn; k += 4 * NN_SPLIT) { // this lane's next four centres: distances first, then the inserts
      float d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) d[u] = sqdist3(ux, uy, uz, sx[k + u * NN_SPLIT], sy[k + u * NN_SPLIT], sz[k + u * NN_SPLIT]); // :44
#pragma unroll
      for (int u = 0; u < 4; ++u) nn_insert<false>(d[u], t0 + k + u * NN_SPLIT, best0, best1, best2, i0, i1, i2); // :45-59
    }
    for (; k < tn; k += NN_SPLIT)
      nn_insert<false>(sqdist3(ux, uy, uz, sx[k], sy[k], sz[k]), t0 + k, best0, best1, best2, i0, i1, i2);
  }
  // merge the quad's four lists into lane 0's: candidates of lanes 1, 2, 3 by (distance, index)
  {
    const float c0 = best0, c1 = best1, c2 = best2;
    const int k0 = i0, k1 = i1, k2 = i2;
    const int base = (tid & 63) & ~(NN_SPLIT - 1);
#pragma unroll
    for (int o = 1; o < NN_SPLIT; ++o) {
      const float e0 = __shfl(c0, base + o, 64), e1 = __shfl(c1, base + o, 64), e2 = __shfl(c2, base + o, 64);
      const int m0 = __shfl(k0, base + o, 64), m1 = __shfl(k1, base + o, 64), m2 = __shfl(k2, base + o, 64);
      // a list that never filled a place still carries (+inf, 0): +inf is never smaller than anything, it stays out
      if (e0 < INFINITY) nn_insert<true>(e0, m0, best0, best1, best2, i0, i1, i2);
      if (e1 < INFINITY) nn_insert<true>(e1, m1, best0, best1, best2, i0, i1, i2);
      if (e2 < INFINITY) nn_insert<true>(e2, m2, best0, best1, best2, i0, i1, i2);
    }
  }
  if (j >= N || q != 0) return;
  best0 = fmaxf(fminf(1e10f, best0), 1e-10f); // :61-63
  best1 = fmaxf(fminf(1e10f, best1), 1e-10f);
  best2 = fmaxf(fminf(1e10f, best2), 1e-10f);
  const float d0d1 = mul_rn(best0, best1), d0d2 = mul_rn(best0, best2), d1d2 = mul_rn(best1, best2);
  const float inv = div_rn(1.0f, add_rn(add_rn(d0d1, d0d2), d1d2));
  int32_t *id = idx + (size_t)b * 3 * N;
  float *w = wgt + (size_t)b * 3 * N;
  w[j] = mul_rn(d1d2, inv);         id[j] = i0;
  w[j + N] = mul_rn(d0d2, inv);     id[j + N] = i1;
  w[j + 2 * N] = mul_rn(d0d1, inv); id[j + 2 * N] = i2;
}

template <int CT>
__global__ __launch_bounds__(256) void three_nn_interp_kernel(const float *__restrict__ cfeat,
                                                              const int32_t *__restrict__ idx,
                                                              const float *__restrict__ wgt, int C,
                                                              int N, int M,
                                                              float *__restrict__ out) {
  const int b = blockIdx.z, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  const int32_t *id = idx + (size_t)b * 3 * N;
  const float *w = wgt + (size_t)b * 3 * N;
  const float w1 = w[j], w2 = w[j + N], w3 = w[j + 2 * N];
  const int a1 = min(max(id[j], 0), M - 1), a2 = min(max(id[j + N], 0), M - 1),
            a3 = min(max(id[j + 2 * N], 0), M - 1);
  const int c0 = blockIdx.y * CT, c1 = min(C, c0 + CT);
  const float *f = cfeat + ((size_t)b * C + c0) * M;
  float *o = out + ((size_t)b * C + c0) * N + j;
#pragma unroll 4
  for (int c = c0; c < c1; ++c, f += M, o += N) // :110-112
    *o = add_rn(add_rn(mul_rn(f[a1], w1), mul_rn(f[a2], w2)), mul_rn(f[a3], w3));
}

// The FP module's [3-NN interpolation of cat(centre features, time embedding) ; skip features] in ONE pass
// (models/pvcnn2_ada.py PointNetFPModule.forward :403-411 fed by models/latent_points_ada.py's torch.cat([features, temb])):
// rows [0, C1) interpolate cfeat, rows [C1, C1 + C2) interpolate the per-sample constants temb[b * ld_t + c] -- the same three
// products and two sums as on a materialised [B, C2, M] broadcast -- rows [C1 + C2, C1 + C2 + C3) copy skip[b][c][j].
template <int CT>
__global__ __launch_bounds__(256) void three_nn_interp_cat_kernel(const float *__restrict__ cfeat,
                                                                  const float *__restrict__ temb, int ld_t,
                                                                  const float *__restrict__ skip,
                                                                  const int32_t *__restrict__ idx,
                                                                  const float *__restrict__ wgt, int C1, int C2, int C3,
                                                                  int N, int M, float *__restrict__ out) {
  const int b = blockIdx.z, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  const int32_t *id = idx + (size_t)b * 3 * N;
  const float *w = wgt + (size_t)b * 3 * N;
  const float w1 = w[j], w2 = w[j + N], w3 = w[j + 2 * N];
  const int a1 = min(max(id[j], 0), M - 1), a2 = min(max(id[j + N], 0), M - 1),
            a3 = min(max(id[j + 2 * N], 0), M - 1);
  const int C = C1 + C2 + C3;
  const int c0 = blockIdx.y * CT, c1 = min(C, c0 + CT);
  float *o = out + ((size_t)b * C + c0) * N + j;
#pragma unroll 4
  for (int c = c0; c < c1; ++c, o += N) {
    if (c < C1) {
      const float *f = cfeat + ((size_t)b * C1 + c) * M;
      *o = add_rn(add_rn(mul_rn(f[a1], w1), mul_rn(f[a2], w2)), mul_rn(f[a3], w3));
    } else if (c < C1 + C2) {
      const float t = temb[(size_t)b * ld_t + c - C1];
      *o = add_rn(add_rn(mul_rn(t, w1), mul_rn(t, w2)), mul_rn(t, w3));
    } else {
      *o = skip[((size_t)b * C3 + c - C1 - C2) * N + j];
    }
  }
}

// gradient: rows[CT][M] accumulated in LDS, written once
__global__ __launch_bounds__(512) void three_nn_interp_grad_kernel(const float *__restrict__ gy,
                                                                   const int32_t *__restrict__ idx,
                                                                   const float *__restrict__ wgt,
                                                                   int C, int N, int M, int CT,
                                                                   float *__restrict__ gx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *rows = reinterpret_cast<float *>(smem);
  const int tid = threadIdx.x, nt = blockDim.x, b = blockIdx.y;
  const int c0 = blockIdx.x * CT, nc = min(C, c0 + CT) - c0;
  for (int v = tid; v < nc * M; v += nt) rows[v] = 0.f;
  __syncthreads();
  const int32_t *id = idx + (size_t)b * 3 * N;
  const float *w = wgt + (size_t)b * 3 * N;
  for (int j = tid; j < N; j += nt) {
    const float w1 = w[j], w2 = w[j + N], w3 = w[j + 2 * N];
    const int a1 = min(max(id[j], 0), M - 1), a2 = min(max(id[j + N], 0), M - 1),
              a3 = min(max(id[j + 2 * N], 0), M - 1);
    for (int c = 0; c < nc; ++c) {
      const float g = gy[((size_t)b * C + c0 + c) * N + j];
      atomicAdd(rows + c * M + a1, mul_rn(g, w1)); // :166-168
      atomicAdd(rows + c * M + a2, mul_rn(g, w2));
      atomicAdd(rows + c * M + a3, mul_rn(g, w3));
    }
  }
  __syncthreads();
  for (int v = tid; v < nc * M; v += nt) gx[((size_t)b * C + c0) * M + v] = rows[v];
}

__global__ void three_nn_interp_grad_atomic_kernel(const float *__restrict__ gy,
                                                   const int32_t *__restrict__ idx,
                                                   const float *__restrict__ wgt, int C, int N,
                                                   int M, float *__restrict__ gx) {
  const int b = blockIdx.z, c = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const float g = gy[((size_t)b * C + c) * N + j];
  for (int q = 0; q < 3; ++q) {
    const int a = min(max(idx[((size_t)b * 3 + q) * N + j], 0), M - 1);
    atomicAdd(gx + ((size_t)b * C + c) * M + a, mul_rn(g, wgt[((size_t)b * 3 + q) * N + j]));
  }
}

} // namespace

extern "C" {

int lion_three_nn_interpolate_forward(const float *points, const float *centers, const float *cfeat,
                                      int B, int C, int N, int M, float *out, int32_t *idx,
                                      float *wgt, lionStream_t stream) {
  if (!points || !centers || !idx || !wgt || B <= 0 || N <= 0 || M <= 0) return LION_EINVAL;
  if (cfeat && (!out || C <= 0)) return LION_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  three_nn_kernel<<<dim3(lion_cdiv(N, 256 / NN_SPLIT), B), 256, 0, st>>>(points, centers, N, M, idx, wgt);
  LION_LAUNCH_CHECK();
  if (!cfeat) return 0;
  const int pt = lion_cdiv(N, 256);
  int ct = 16;
  while (ct > 2 && (long)B * pt * lion_cdiv(C, ct) < 2048) ct >>= 1;
  dim3 grid(pt, lion_cdiv(C, ct), B);
  switch (ct) {
  case 16: three_nn_interp_kernel<16><<<grid, 256, 0, st>>>(cfeat, idx, wgt, C, N, M, out); break;
  case 8:  three_nn_interp_kernel<8><<<grid, 256, 0, st>>>(cfeat, idx, wgt, C, N, M, out); break;
  case 4:  three_nn_interp_kernel<4><<<grid, 256, 0, st>>>(cfeat, idx, wgt, C, N, M, out); break;
  default: three_nn_interp_kernel<2><<<grid, 256, 0, st>>>(cfeat, idx, wgt, C, N, M, out); break;
  }
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_three_nn_interpolate_cat_forward(const float *points, const float *centers, const float *cfeat, const float *temb,
                                          int ld_t, const float *skip, int B, int C1, int C2, int C3, int N, int M,
                                          float *out, int32_t *idx, float *wgt, lionStream_t stream) {
  if (!points || !centers || !cfeat || !out || !idx || !wgt || B <= 0 || N <= 0 || M <= 0 || C1 <= 0) return LION_EINVAL;
  if (C2 < 0 || C3 < 0 || (C2 > 0 && !temb) || (C3 > 0 && !skip) || ld_t < 0) return LION_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  three_nn_kernel<<<dim3(lion_cdiv(N, 256 / NN_SPLIT), B), 256, 0, st>>>(points, centers, N, M, idx, wgt);
  LION_LAUNCH_CHECK();
  const int C = C1 + C2 + C3, pt = lion_cdiv(N, 256);
  int ct = 16;
  while (ct > 2 && (long)B * pt * lion_cdiv(C, ct) < 2048) ct >>= 1;
  dim3 grid(pt, lion_cdiv(C, ct), B);
#define LION_NN_CAT(CT_) three_nn_interp_cat_kernel<CT_><<<grid, 256, 0, st>>>(cfeat, temb, ld_t, skip, idx, wgt, C1, C2, C3, N, M, out)
  switch (ct) {
  case 16: LION_NN_CAT(16); break;
  case 8:  LION_NN_CAT(8); break;
  case 4:  LION_NN_CAT(4); break;
  default: LION_NN_CAT(2); break;
  }
#undef LION_NN_CAT
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_three_nn_interpolate_backward(const float *gy, const int32_t *idx, const float *wgt, int B,
                                       int C, int N, int M, float *gx, lionStream_t stream) {
  if (!gy || !idx || !wgt || !gx || B <= 0 || C <= 0 || N <= 0 || M <= 0) return LION_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if ((size_t)M * 4 <= 64 * 1024) {
    int CT = (int)((64 * 1024) / ((size_t)M * 4));
    if (CT > 8) CT = 8;
    if (CT > C) CT = C;
    while (CT > 1 && (long)B * lion_cdiv(C, CT) < 1024) CT >>= 1;
    three_nn_interp_grad_kernel<<<dim3(lion_cdiv(C, CT), B), 512, (size_t)CT * M * 4, st>>>(
        gy, idx, wgt, C, N, M, CT, gx);
    LION_LAUNCH_CHECK();
    return 0;
  }
  hipError_t e = hipMemsetAsync(gx, 0, (size_t)B * C * M * 4, st);
  if (e != hipSuccess) return (int)e;
  three_nn_interp_grad_atomic_kernel<<<dim3(lion_cdiv(N, 256), C, B), 256, 0, st>>>(gy, idx, wgt, C,
                                                                                  N, M, gx);
  LION_LAUNCH_CHECK();
  return 0;
}

} // extern "C"
