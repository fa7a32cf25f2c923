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

__global__ __launch_bounds__(256) void three_nn_kernel(const float *__restrict__ points,
                                                       const float *__restrict__ centers, int N,
                                                       int M, int32_t *__restrict__ idx,
                                                       float *__restrict__ wgt) {
  __shared__ float sx[NN_TILE], sy[NN_TILE], sz[NN_TILE];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int j = blockIdx.x * 256 + tid;
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
    for (int k = 0; k < tn; ++k) {
      const float d = sqdist3(ux, uy, uz, sx[k], sy[k], sz[k]); // :44
      if (d < best2) {                                           // :45-59
        best2 = d; i2 = t0 + k;
        if (d < best1) {
          best2 = best1; i2 = i1; best1 = d; i1 = t0 + k;
          if (d < best0) { best1 = best0; i1 = i0; best0 = d; i0 = t0 + k; }
        }
      }
    }
  }
  if (j >= N) return;
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
  three_nn_kernel<<<dim3(lion_cdiv(N, 256), B), 256, 0, st>>>(points, centers, N, M, idx, wgt);
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
