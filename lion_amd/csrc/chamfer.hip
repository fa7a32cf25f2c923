// chamfer.hip -- E1 Chamfer nearest-neighbour distances (+ arg-min) and gradient.
//
// Reference: third_party/ChamferDistancePytorch/chamfer3D/chamfer3D.cu:12-134 (forward, a fixed
// 32x16 grid of 512-thread blocks whatever the batch size), :155-185 (gradient, 6 float atomics
// per point into a caller-zeroed buffer), chamfer_cuda.cpp:17-33.
//
// MI355X design: grid = (query tiles, batch, 2 directions) so one launch covers both directions
// and small batches still spread over the chip; a lane owns QPL query points (register tiling: one
// LDS broadcast read of a target feeds QPL distance evaluations), targets are staged in LDS as
// float4 tiles.  Scan order is ascending target index with strict '<', i.e. the lowest index wins
// ties exactly as in the reference (strict '<' inside a tile, strict '>' across tiles).  Distances
// use the reference's expression without FMA contraction -> dist and idx are bit-exact.
// The gradient is the reference's arithmetic in two launches per direction: the own term is written
// once per point (plain stores), the term handed to the matched point of the other cloud goes through
// global float atomics (chamfer3D.cu:169-171), so the gradient buffers are written by the own-term
// pass first and need no pre-zeroing; the atomic order is free, as in the reference.
#include "common.h"
#include <math.h>

namespace {

constexpr int CH_TILE = 1024; // targets per LDS tile (16 KiB as float4)
constexpr int CH_QPL = 4;     // queries per lane
constexpr int CH_QPB = 64 * CH_QPL; // queries per workgroup

// Round 4.  B = 32 pairs of 2048-point clouds are only 131 k query points: with one query pair per lane (round 1-3:
// 256 workgroups x 4 waves) every SIMD held ONE wave and waited out its own dependent-issue and LDS latencies (17.6 TF
// of distance arithmetic, 11 % of the vector peak).  Now a workgroup's four waves share the SAME 256 queries -- four per
// lane: four independent distance / compare chains per LDS broadcast -- and each wave scans its own quarter of every
// target tile; the four partial (distance, index) results of a query are merged through LDS with the scan's own rule
// (smaller distance wins, equal distances keep the lower index -- what strict '<' over ascending indices yields), so
// dist and idx stay bit-exact.  512 workgroups at B = 32: two waves per SIMD.
__global__ __launch_bounds__(256) void chamfer_fwd_kernel(const float *__restrict__ xyz1,
                                                          const float *__restrict__ xyz2, int N,
                                                          int M, float *__restrict__ dist1,
                                                          float *__restrict__ dist2,
                                                          int32_t *__restrict__ idx1,
                                                          int32_t *__restrict__ idx2) {
  __shared__ float4 tile[CH_TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y, dir = blockIdx.z;
  const int nq = dir == 0 ? N : M, nt = dir == 0 ? M : N;
  if (blockIdx.x * CH_QPB >= nq) return; // uniform per block
  const float *q = (dir == 0 ? xyz1 : xyz2) + (size_t)b * nq * 3;
  const float *t = (dir == 0 ? xyz2 : xyz1) + (size_t)b * nt * 3;
  float *dout = (dir == 0 ? dist1 : dist2) + (size_t)b * nq;
  int32_t *iout = (dir == 0 ? idx1 : idx2) + (size_t)b * nq;

  float qx[CH_QPL], qy[CH_QPL], qz[CH_QPL], best[CH_QPL];
  int bi[CH_QPL];
#pragma unroll
  for (int p = 0; p < CH_QPL; ++p) {
    const int j = blockIdx.x * CH_QPB + p * 64 + lane;
    qx[p] = qy[p] = qz[p] = 0.f;
    if (j < nq) { qx[p] = q[j * 3]; qy[p] = q[j * 3 + 1]; qz[p] = q[j * 3 + 2]; }
    best[p] = INFINITY; // the reference accepts element 0 unconditionally ("k==0 ||")
    bi[p] = 0;
  }
  constexpr int SUB = CH_TILE / 4; // targets of a tile per wave
  for (int t0 = 0; t0 < nt; t0 += CH_TILE) {
    const int tn = min(CH_TILE, nt - t0);
    __syncthreads();
    for (int k = tid; k < tn; k += 256)
      tile[k] = make_float4(t[(size_t)(t0 + k) * 3], t[(size_t)(t0 + k) * 3 + 1],
                            t[(size_t)(t0 + k) * 3 + 2], 0.f);
    __syncthreads();
    const int k1 = min(tn, (wave + 1) * SUB);
#pragma unroll 4
    for (int k = wave * SUB; k < k1; ++k) {
      const float4 v = tile[k];
#pragma unroll
      for (int p = 0; p < CH_QPL; ++p) {
        // chamfer3D.cu:31-34: x2 = buf - x1; d = x2*x2 + y2*y2 + z2*z2
        const float d = sqdist3(v.x, v.y, v.z, qx[p], qy[p], qz[p]);
        if (d < best[p]) { best[p] = d; bi[p] = t0 + k; }
      }
    }
  }
  // merge the four waves' candidates of each query
  __syncthreads(); // the last tile is no longer read: its memory carries the candidates
  float *sd = reinterpret_cast<float *>(tile);                 // [4][CH_QPB]
  int *si = reinterpret_cast<int *>(tile) + 4 * CH_QPB;        // [4][CH_QPB]
#pragma unroll
  for (int p = 0; p < CH_QPL; ++p) {
    sd[wave * CH_QPB + p * 64 + lane] = best[p];
    si[wave * CH_QPB + p * 64 + lane] = bi[p];
  }
  __syncthreads();
  for (int qq = tid; qq < CH_QPB; qq += 256) {
    float d = sd[qq];
    int i = si[qq];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float dw = sd[w * CH_QPB + qq];
      const int iw = si[w * CH_QPB + qq];
      if (dw < d || (dw == d && iw < i)) { d = dw; i = iw; }
    }
    const int j = blockIdx.x * CH_QPB + qq;
    if (j < nq) { dout[j] = d; iout[j] = i; }
  }
}

// Gradient, step 1: own-direction term, written (not accumulated):
//   g1[j] = 2*gd1[j]*(x1[j]-x2[idx1[j]]);   g2[j] = 2*gd2[j]*(x2[j]-x1[idx2[j]])
// step 2 adds the scattered terms with float atomics (few collisions: <= N adds per cloud).
__global__ void chamfer_grad_own_kernel(const float *__restrict__ xa, const float *__restrict__ xb,
                                        const float *__restrict__ gda,
                                        const int32_t *__restrict__ idxa, int na, int nb,
                                        float *__restrict__ ga) {
  const int b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= na) return;
  const float *pa = xa + ((size_t)b * na + j) * 3;
  const int j2 = min(max(idxa[(size_t)b * na + j], 0), nb - 1);
  const float *pb = xb + ((size_t)b * nb + j2) * 3;
  const float g = mul_rn(gda[(size_t)b * na + j], 2.0f);
  float *o = ga + ((size_t)b * na + j) * 3;
  o[0] = mul_rn(g, sub_rn(pa[0], pb[0]));
  o[1] = mul_rn(g, sub_rn(pa[1], pb[1]));
  o[2] = mul_rn(g, sub_rn(pa[2], pb[2]));
}

__global__ void chamfer_grad_scatter_kernel(const float *__restrict__ xa,
                                            const float *__restrict__ xb,
                                            const float *__restrict__ gda,
                                            const int32_t *__restrict__ idxa, int na, int nb,
                                            float *__restrict__ gb) {
  const int b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= na) return;
  const float *pa = xa + ((size_t)b * na + j) * 3;
  const int j2 = min(max(idxa[(size_t)b * na + j], 0), nb - 1);
  const float *pb = xb + ((size_t)b * nb + j2) * 3;
  const float g = mul_rn(gda[(size_t)b * na + j], 2.0f);
  float *o = gb + ((size_t)b * nb + j2) * 3;
  atomicAdd(o + 0, -mul_rn(g, sub_rn(pa[0], pb[0]))); // chamfer3D.cu:169-171
  atomicAdd(o + 1, -mul_rn(g, sub_rn(pa[1], pb[1])));
  atomicAdd(o + 2, -mul_rn(g, sub_rn(pa[2], pb[2])));
}

} // namespace

extern "C" {

int lion_chamfer_forward(const float *xyz1, const float *xyz2, int B, int N, int M, float *dist1,
                         float *dist2, int32_t *idx1, int32_t *idx2, lionStream_t stream) {
  if (!xyz1 || !xyz2 || !dist1 || !dist2 || !idx1 || !idx2 || B <= 0 || N <= 0 || M <= 0)
    return LION_EINVAL;
  const int nmax = N > M ? N : M;
  chamfer_fwd_kernel<<<dim3(lion_cdiv(nmax, CH_QPB), B, 2), 256, 0,
                       static_cast<hipStream_t>(stream)>>>(xyz1, xyz2, N, M, dist1, dist2, idx1,
                                                           idx2);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_chamfer_backward(const float *xyz1, const float *xyz2, const float *gdist1,
                          const float *gdist2, const int32_t *idx1, const int32_t *idx2, int B,
                          int N, int M, float *gxyz1, float *gxyz2, lionStream_t stream) {
  if (!xyz1 || !xyz2 || !gdist1 || !gdist2 || !idx1 || !idx2 || !gxyz1 || !gxyz2 || B <= 0 ||
      N <= 0 || M <= 0)
    return LION_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  chamfer_grad_own_kernel<<<dim3(lion_cdiv(N, 256), B), 256, 0, st>>>(xyz1, xyz2, gdist1, idx1, N, M, gxyz1);
  chamfer_grad_own_kernel<<<dim3(lion_cdiv(M, 256), B), 256, 0, st>>>(xyz2, xyz1, gdist2, idx2, M, N, gxyz2);
  chamfer_grad_scatter_kernel<<<dim3(lion_cdiv(N, 256), B), 256, 0, st>>>(xyz1, xyz2, gdist1, idx1, N, M, gxyz2);
  chamfer_grad_scatter_kernel<<<dim3(lion_cdiv(M, 256), B), 256, 0, st>>>(xyz2, xyz1, gdist2, idx2, M, N, gxyz1);
  LION_LAUNCH_CHECK();
  return 0;
}

} // extern "C"
