// ball_query.hip -- K6: first-U-in-radius neighbour lists.
//
// Reference: third_party/pvcnn/functional/src/ball_query/ball_query.cu:19-50 (one *thread* per
// centre scanning all N points from global memory, 32 blocks at B=32) and ball_query.cpp:7-33.
//
// MI355X design: one wave64 per centre, 4 waves per workgroup, CPW centres per wave, grid =
// (centre tiles, batch).  The point cloud is staged once per workgroup in LDS (SoA); a wave tests
// 64 points per step, __ballot + prefix popcount keeps "first U hits in ascending point index"
// exactly, and a centre stops scanning as soon as it has U hits.  The distance expression is
// evaluated exactly like the reference's (no FMA contraction), so the integer output is bit-exact.
#include "common.h"

namespace {

constexpr int BQ_TILE = 2048; // points per LDS tile (24 KiB)
constexpr int BQ_CPW = 4;     // centres per wave

__global__ __launch_bounds__(256) void ball_query_kernel(const float *__restrict__ centers,
                                                         const float *__restrict__ points, int M,
                                                         int N, float r2, int U,
                                                         int32_t *__restrict__ idx) {
  __builtin_amdgcn_s_setprio(2); // runs on the geometry side stream next to the convolutions: see sampling.hip
  __shared__ float px[BQ_TILE], py[BQ_TILE], pz[BQ_TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const float *pc = points + (size_t)b * 3 * N;
  const float *cc = centers + (size_t)b * 3 * M;
  const int j0 = (blockIdx.x * 4 + wave) * BQ_CPW;

  float cx[BQ_CPW], cy[BQ_CPW], cz[BQ_CPW];
  int cnt[BQ_CPW], first[BQ_CPW];
#pragma unroll
  for (int q = 0; q < BQ_CPW; ++q) {
    const int j = j0 + q;
    cnt[q] = (j < M) ? 0 : U; // out-of-range centres are "done"
    first[q] = 0;
    cx[q] = cy[q] = cz[q] = 0.f;
    if (j < M) { cx[q] = cc[j]; cy[q] = cc[j + M]; cz[q] = cc[j + 2 * M]; }
  }

  for (int t0 = 0; t0 < N; t0 += BQ_TILE) {
    const int tn = min(BQ_TILE, N - t0);
    __syncthreads();
    for (int k = tid; k < tn; k += 256) {
      px[k] = pc[t0 + k]; py[k] = pc[t0 + k + N]; pz[k] = pc[t0 + k + 2 * N];
    }
    __syncthreads();
    bool all_done = true;
#pragma unroll
    for (int q = 0; q < BQ_CPW; ++q) all_done = all_done && (cnt[q] >= U);
    if (all_done) continue; // wave-uniform; still takes part in the barriers above
    for (int s = 0; s < tn; s += 64) {
      const int k = s + lane;
      const bool valid = k < tn;
      const float x = valid ? px[k] : 0.f, y = valid ? py[k] : 0.f, z = valid ? pz[k] : 0.f;
      bool done = true;
#pragma unroll
      for (int q = 0; q < BQ_CPW; ++q) {
        if (cnt[q] < U) { // wave-uniform
          const float d2 = sqdist3(cx[q], cy[q], cz[q], x, y, z); // ball_query.cu:35-38
          const bool hit = valid && (d2 < r2);
          const unsigned long long mask = __ballot(hit);
          if (mask) {
            const int pre = __popcll(mask & ((1ull << lane) - 1ull));
            if (cnt[q] == 0) first[q] = t0 + s + (__ffsll((long long)mask) - 1);
            const int slot = cnt[q] + pre;
            if (hit && slot < U) idx[((size_t)b * M + j0 + q) * U + slot] = t0 + k;
            cnt[q] += __popcll(mask);
          }
          done = done && (cnt[q] >= U);
        }
      }
      if (done) break;
    }
  }
  // padding: slots beyond the hit count repeat the first hit; no hit at all -> zeros
#pragma unroll
  for (int q = 0; q < BQ_CPW; ++q) {
    const int j = j0 + q;
    if (j < M) {
      const int have = min(cnt[q], U);
      for (int s = have + lane; s < U; s += 64) idx[((size_t)b * M + j) * U + s] = first[q];
    }
  }
}

} // namespace

extern "C" int lion_ball_query(const float *centers, const float *points, int B, int M, int N,
                               float radius, int U, int32_t *idx, lionStream_t stream) {
  if (!centers || !points || !idx || B <= 0 || M <= 0 || N <= 0 || U <= 0) return LION_EINVAL;
  const float r2 = radius * radius; // ball_query.cpp:24
  ball_query_kernel<<<dim3(lion_cdiv(M, 4 * BQ_CPW), B), 256, 0, static_cast<hipStream_t>(stream)>>>(
      centers, points, M, N, r2, U, idx);
  LION_LAUNCH_CHECK();
  return 0;
}
