// ref_driver.cpp -- C entry points that run the REFERENCE'S OWN kernel bodies on the CPU
// (TEST INFRASTRUCTURE; built only where /root/reference exists, output oracle/_ref/libref.so).
// Host-side allocation / zero-initialisation and launch shapes restate the reference's launchers:
//   vox.cu:112-125 + vox.cpp:33-38, trilinear_devox.cu:164-178 + .cpp:33-54,85, ball_query.cu:52-58 +
//   .cpp:20-22, grouping.cu:38-43,79-84, sampling.cu:33-38,68-73,169-173 + .cpp:21,38,51-54,
//   neighbor_interpolate.cu:118-130,172-180, chamfer3D.cu:142-143,184-185, emd_kernel.cu:186,272,390-391,
//   cuda_utils.cuh:13-26 (optimal_num_threads / optimal_block_config).
#include "ref_shim.h"
#include "_ref/gen/kernels.inc"
#include <cstdint>

static int optimal_num_threads(int work_size) {            // cuda_utils.cuh:13-18
  const int pow_2 = std::log2(static_cast<double>(work_size));
  return max(min(1 << pow_2, 512), 1);
}
static dim3 optimal_block_config(int x, int y) {           // cuda_utils.cuh:20-26
  const int xt = optimal_num_threads(x);
  const int yt = max(min(optimal_num_threads(y), 512 / xt), 1);
  return dim3(xt, yt, 1);
}
template <typename T> static void zero(T *p, size_t n) { memset(p, 0, n * sizeof(T)); }
#define API extern "C" __attribute__((visibility("default")))

API void ref_avg_voxelize_forward(int b, int c, int n, int r, const int *coords, const float *feat, int *ind, int *cnt, float *out) {
  const int r2 = r * r, r3 = r2 * r;
  zero(ind, (size_t)b * n); zero(cnt, (size_t)b * r3); zero(out, (size_t)b * c * r3);
  REF_LAUNCH(dim3(b), dim3(optimal_num_threads(n)), grid_stats_kernel(b, n, r, r2, r3, coords, ind, cnt));
  REF_LAUNCH(dim3(b), dim3(optimal_num_threads(n)), avg_voxelize_kernel(b, c, n, r3, ind, cnt, feat, out));
}
API void ref_avg_voxelize_backward(int b, int c, int n, int r3, const float *gy, const int *ind, const int *cnt, float *gx) {
  zero(gx, (size_t)b * c * n);
  REF_LAUNCH(dim3(b), dim3(optimal_num_threads(n)), avg_voxelize_grad_kernel(b, c, n, r3, ind, cnt, gy, gx));
}
API void ref_trilinear_devoxelize_forward(int b, int c, int n, int r, int training, const float *coords, const float *feat, int *inds, float *wgts, float *outs) {
  const int r2 = r * r, r3 = r2 * r;
  zero(outs, (size_t)b * c * n);
  if (training) { zero(inds, (size_t)b * 8 * n); zero(wgts, (size_t)b * 8 * n); }
  REF_LAUNCH(dim3(b), dim3(optimal_num_threads(n)), trilinear_devoxelize_kernel(b, c, n, r, r2, r3, training != 0, coords, feat, inds, wgts, outs));
}
API void ref_trilinear_devoxelize_backward(int b, int c, int n, int r3, const float *gy, const int *inds, const float *wgts, float *gx) {
  zero(gx, (size_t)b * c * r3);
  REF_LAUNCH(dim3(b), dim3(optimal_num_threads(n)), trilinear_devoxelize_grad_kernel(b, c, n, r3, inds, wgts, gy, gx));
}
API void ref_ball_query(int b, int n, int m, float radius, int u, const float *centers, const float *points, int *idx) {
  zero(idx, (size_t)b * m * u);
  REF_LAUNCH(dim3(b), dim3(optimal_num_threads(m)), ball_query_kernel(b, n, m, radius * radius, u, centers, points, idx));
}
API void ref_grouping_forward(int b, int c, int n, int m, int u, const float *feat, const int *idx, float *out) {
  zero(out, (size_t)b * c * m * u);
  REF_LAUNCH(dim3(b), optimal_block_config(m, c), grouping_kernel(b, c, n, m, u, feat, idx, out));
}
API void ref_grouping_backward(int b, int c, int n, int m, int u, const float *gy, const int *idx, float *gx) {
  zero(gx, (size_t)b * c * n);
  REF_LAUNCH(dim3(b), optimal_block_config(m, c), grouping_grad_kernel(b, c, n, m, u, gy, idx, gx));
}
API void ref_furthest_point_sampling(int b, int n, int m, const float *coords, int *idx) {
  zero(idx, (size_t)b * m);
  std::vector<float> dist((size_t)b * n, 1e38f);
  REF_LAUNCH(dim3(b), dim3(512), furthest_point_sampling_kernel(b, n, m, coords, dist.data(), idx));
}
API void ref_gather_features_forward(int b, int c, int n, int m, const float *feat, const int *idx, float *out) {
  zero(out, (size_t)b * c * m);
  REF_LAUNCH(dim3(b, c, 1), dim3(optimal_num_threads(m)), gather_features_kernel(b, c, n, m, feat, idx, out));
}
API void ref_gather_features_backward(int b, int c, int n, int m, const float *gy, const int *idx, float *gx) {
  zero(gx, (size_t)b * c * n);
  REF_LAUNCH(dim3(b, c, 1), dim3(optimal_num_threads(m)), gather_features_grad_kernel(b, c, n, m, gy, idx, gx));
}
API void ref_three_nn_interpolate_forward(int b, int c, int m, int n, const float *points, const float *centers, const float *cfeat, int *idx, float *wgt, float *out) {
  zero(idx, (size_t)b * 3 * n); zero(wgt, (size_t)b * 3 * n); zero(out, (size_t)b * c * n);
  REF_LAUNCH(dim3(b), dim3(optimal_num_threads(n)), three_nearest_neighbors_kernel(b, n, m, points, centers, wgt, idx));
  REF_LAUNCH(dim3(b), optimal_block_config(n, c), three_nearest_neighbors_interpolate_kernel(b, c, m, n, cfeat, idx, wgt, out));
}
API void ref_three_nn_interpolate_backward(int b, int c, int n, int m, const float *gy, const int *idx, const float *wgt, float *gx) {
  zero(gx, (size_t)b * c * m);
  REF_LAUNCH(dim3(b), optimal_block_config(n, c), three_nearest_neighbors_interpolate_grad_kernel(b, c, n, m, gy, idx, wgt, gx));
}
API void ref_chamfer_forward(int b, int n, int m, const float *xyz1, const float *xyz2, float *d1, float *d2, int *i1, int *i2) {
  zero(d1, (size_t)b * n); zero(d2, (size_t)b * m); zero(i1, (size_t)b * n); zero(i2, (size_t)b * m);
  REF_LAUNCH(dim3(32, 16, 1), dim3(512), NmDistanceKernel(b, n, xyz1, m, xyz2, d1, i1));
  REF_LAUNCH(dim3(32, 16, 1), dim3(512), NmDistanceKernel(b, m, xyz2, n, xyz1, d2, i2));
}
API void ref_chamfer_backward(int b, int n, int m, const float *xyz1, const float *xyz2, const float *gd1, const float *gd2, const int *i1, const int *i2, float *g1, float *g2) {
  zero(g1, (size_t)b * n * 3); zero(g2, (size_t)b * m * 3);   // dist_chamfer_3D.py:82-83
  REF_LAUNCH(dim3(1, 16, 1), dim3(256), NmDistanceGradKernel(b, n, xyz1, m, xyz2, gd1, i1, g1, g2));
  REF_LAUNCH(dim3(1, 16, 1), dim3(256), NmDistanceGradKernel(b, m, xyz2, n, xyz1, gd2, i2, g2, g1));
}
API void ref_approxmatch(int b, int n, int m, const float *xyz1, const float *xyz2, float *match) {
  zero(match, (size_t)b * n * m);
  std::vector<float> temp((size_t)b * (n + m) * 2, 0.f);
  REF_LAUNCH(dim3(32), dim3(512), approxmatch<float>(b, n, m, xyz1, xyz2, match, temp.data()));
}
API void ref_matchcost(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *cost) {
  zero(cost, (size_t)b);
  REF_LAUNCH(dim3(32), dim3(512), matchcost<float>(b, n, m, xyz1, xyz2, match, cost));
}
API void ref_matchcost_backward(int b, int n, int m, const float *gc, const float *xyz1, const float *xyz2, const float *match, float *g1, float *g2) {
  zero(g1, (size_t)b * n * 3); zero(g2, (size_t)b * m * 3);
  REF_LAUNCH(dim3(32), dim3(512), matchcostgrad1<float>(b, n, m, gc, xyz1, xyz2, match, g1));
  REF_LAUNCH(dim3(32, 32), dim3(256), matchcostgrad2<float>(b, n, m, gc, xyz1, xyz2, match, g2));
}
