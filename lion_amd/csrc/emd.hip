// emd.hip -- E2 approximate earth mover's distance (auction-style soft matching).
//
// Reference: third_party/PyTorchEMD/cuda/emd_kernel.cu:24-156 (approxmatch: ONE 512-thread block
// per cloud pair walks 10 temperature levels x 3 O(N*M) passes), :199-241 (matchcost),
// :285-353 (gradients), cuda/emd.cpp:7-27.
//
// MI355X design: the three passes of a level are separate launches with grid = (row tiles, batch)
// so a batch of pairs fills the chip instead of 32 blocks; per row the loop over the other cloud
// runs in ascending index order inside each of four fixed quarters of a tile (one per wave), the
// four partial sums added in a fixed order (round 4; a single sequential chain before).  The other cloud is staged in LDS as float4 {x,y,z,w}
// tiles (w = remainR / ratioL / ratioR) and read as broadcasts.  exp() is the hardware v_exp_f32
// path (__expf), the same class of approximation as the reference's CUDA __expf.
#include "common.h"

namespace {

constexpr int EMD_TILE = 1024;

// level * ((x2-x1)^2 + (y2-y1)^2 + (z2-z1)^2), reference operand order
__device__ __forceinline__ float lvl_d(float level, float x2, float y2, float z2, float x1,
                                       float y1, float z1) {
  return mul_rn(level, sqdist3(x2, y2, z2, x1, y1, z1));
}

__global__ void emd_init_kernel(int n, int m, float multiL, float multiR, float *__restrict__ remainL,
                                float *__restrict__ remainR) {
  const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) remainL[(size_t)b * n + i] = multiL;
  if (i < m) remainR[(size_t)b * m + i] = multiR;
}

// Round 4: the four waves of a workgroup share 64 rows (one per lane) and each scans its own quarter of every tile of the
// other cloud; the four partial sums of a row are added in the fixed order ((w0 + w1) + w2) + w3.  With one row per thread
// and whole tiles per wave (rounds 1-3) B = 32 pairs of 2048 points were 256 workgroups = ONE wave per SIMD, which waited
// out its own LDS round trips and exp latencies (97 cycles per evaluation against ~34 of issue).  The sums are no longer
// the reference's single sequential chain: they differ from it by fp32 rounding (1e-7 relative), far inside the tolerance
// v_exp_f32 already imposes on this operator (tests: cost 1e-4, match 2e-3), and they do not depend on the batch size.
constexpr int EMD_ROWS = 64;           // rows per workgroup
constexpr int EMD_SUB = EMD_TILE / 4;  // elements of a tile per wave

#define EMD_MERGE(red_, wave_, lane_, v_)        \
  __syncthreads();                               \
  red_[wave_ * EMD_ROWS + lane_] = v_;           \
  __syncthreads();                               \
  v_ = add_rn(add_rn(add_rn(red_[lane_], red_[EMD_ROWS + lane_]), red_[2 * EMD_ROWS + lane_]), red_[3 * EMD_ROWS + lane_]);

// pass 1 (emd_kernel.cu:50-81): ratioL[k] = remainL[k] / (1e-9 + sum_l exp(level*d) * remainR[l])
__global__ __launch_bounds__(256) void emd_pass1_kernel(const float *__restrict__ xyz1,
                                                        const float *__restrict__ xyz2, int n, int m,
                                                        float level,
                                                        const float *__restrict__ remainL,
                                                        const float *__restrict__ remainR,
                                                        float *__restrict__ ratioL) {
  __shared__ float4 tile[EMD_TILE];
  __shared__ float red[4 * EMD_ROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y, k = blockIdx.x * EMD_ROWS + lane;
  const float *p1 = xyz1 + (size_t)b * n * 3, *p2 = xyz2 + (size_t)b * m * 3;
  float x1 = 0, y1 = 0, z1 = 0;
  if (k < n) { x1 = p1[k * 3]; y1 = p1[k * 3 + 1]; z1 = p1[k * 3 + 2]; }
  float suml = wave == 0 ? 1e-9f : 0.f;
  for (int l0 = 0; l0 < m; l0 += EMD_TILE) {
    const int ln = min(EMD_TILE, m - l0);
    __syncthreads();
    for (int l = tid; l < ln; l += 256)
      tile[l] = make_float4(p2[(size_t)(l0 + l) * 3], p2[(size_t)(l0 + l) * 3 + 1],
                            p2[(size_t)(l0 + l) * 3 + 2], remainR[(size_t)b * m + l0 + l]);
    __syncthreads();
    const int l1 = min(ln, (wave + 1) * EMD_SUB);
#pragma unroll 8
    for (int l = wave * EMD_SUB; l < l1; ++l) {
      const float4 v = tile[l];
      const float w = mul_rn(__expf(lvl_d(level, v.x, v.y, v.z, x1, y1, z1)), v.w);
      suml = add_rn(suml, w);
    }
  }
  EMD_MERGE(red, wave, lane, suml)
  if (wave == 0 && k < n) ratioL[(size_t)b * n + k] = div_rn(remainL[(size_t)b * n + k], suml);
}

// pass 2 (emd_kernel.cu:83-117)
__global__ __launch_bounds__(256) void emd_pass2_kernel(const float *__restrict__ xyz1,
                                                        const float *__restrict__ xyz2, int n, int m,
                                                        float level,
                                                        const float *__restrict__ ratioL,
                                                        float *__restrict__ remainR,
                                                        float *__restrict__ ratioR) {
  __shared__ float4 tile[EMD_TILE];
  __shared__ float red[4 * EMD_ROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y, l = blockIdx.x * EMD_ROWS + lane;
  const float *p1 = xyz1 + (size_t)b * n * 3, *p2 = xyz2 + (size_t)b * m * 3;
  float x2 = 0, y2 = 0, z2 = 0;
  if (l < m) { x2 = p2[l * 3]; y2 = p2[l * 3 + 1]; z2 = p2[l * 3 + 2]; }
  float sumr = 0;
  for (int k0 = 0; k0 < n; k0 += EMD_TILE) {
    const int kn = min(EMD_TILE, n - k0);
    __syncthreads();
    for (int k = tid; k < kn; k += 256)
      tile[k] = make_float4(p1[(size_t)(k0 + k) * 3], p1[(size_t)(k0 + k) * 3 + 1],
                            p1[(size_t)(k0 + k) * 3 + 2], ratioL[(size_t)b * n + k0 + k]);
    __syncthreads();
    const int k1 = min(kn, (wave + 1) * EMD_SUB);
#pragma unroll 8
    for (int k = wave * EMD_SUB; k < k1; ++k) {
      const float4 v = tile[k];
      const float w = mul_rn(__expf(lvl_d(level, x2, y2, z2, v.x, v.y, v.z)), v.w);
      sumr = add_rn(sumr, w);
    }
  }
  EMD_MERGE(red, wave, lane, sumr)
  if (wave == 0 && l < m) {
    const float rr = remainR[(size_t)b * m + l];
    sumr = mul_rn(sumr, rr);
    const float consumption = fminf(div_rn(rr, add_rn(sumr, 1e-9f)), 1.0f);
    ratioR[(size_t)b * m + l] = mul_rn(consumption, rr);
    remainR[(size_t)b * m + l] = fmaxf(0.0f, sub_rn(rr, sumr));
  }
}

// pass 3 (emd_kernel.cu:119-153): match[l][k] += w ; remainL[k] = max(0, remainL[k] - sum_l w)
template <bool FIRST>
__global__ __launch_bounds__(256) void emd_pass3_kernel(const float *__restrict__ xyz1,
                                                        const float *__restrict__ xyz2, int n, int m,
                                                        float level,
                                                        const float *__restrict__ ratioL,
                                                        const float *__restrict__ ratioR,
                                                        float *__restrict__ remainL,
                                                        float *__restrict__ match) {
  __shared__ float4 tile[EMD_TILE];
  __shared__ float red[4 * EMD_ROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y, k = blockIdx.x * EMD_ROWS + lane;
  const float *p1 = xyz1 + (size_t)b * n * 3, *p2 = xyz2 + (size_t)b * m * 3;
  float x1 = 0, y1 = 0, z1 = 0, rl = 0;
  if (k < n) { x1 = p1[k * 3]; y1 = p1[k * 3 + 1]; z1 = p1[k * 3 + 2]; rl = ratioL[(size_t)b * n + k]; }
  float suml = 0;
  float *mt = match + (size_t)b * n * m;
  for (int l0 = 0; l0 < m; l0 += EMD_TILE) {
    const int ln = min(EMD_TILE, m - l0);
    __syncthreads();
    for (int l = tid; l < ln; l += 256)
      tile[l] = make_float4(p2[(size_t)(l0 + l) * 3], p2[(size_t)(l0 + l) * 3 + 1],
                            p2[(size_t)(l0 + l) * 3 + 2], ratioR[(size_t)b * m + l0 + l]);
    __syncthreads();
    if (k < n) {
      const int l1 = min(ln, (wave + 1) * EMD_SUB);
#pragma unroll 4
      for (int l = wave * EMD_SUB; l < l1; ++l) {
        const float4 v = tile[l];
        const float w = mul_rn(mul_rn(__expf(lvl_d(level, v.x, v.y, v.z, x1, y1, z1)), rl), v.w);
        float *dst = mt + (size_t)(l0 + l) * n + k;
        if (FIRST) *dst = w;               // first level: match starts at zero (0 + w == w)
        else *dst = add_rn(*dst, w);
        suml = add_rn(suml, w);
      }
    }
  }
  EMD_MERGE(red, wave, lane, suml)
  if (wave == 0 && k < n) remainL[(size_t)b * n + k] = fmaxf(0.0f, sub_rn(remainL[(size_t)b * n + k], suml));
}

// pass 3 without the match matrix (evaluation path, emd_nograd.py:19-44 only needs the cost): the reference builds
// match[b][l][k] (16.8 MB per pair at 2048^2, read and rewritten at each of the 10 levels) and then sums
// match * d^2; the sum over levels commutes with that product, so row k accumulates sum_l w * d^2 on the fly
// (levels in order) and nothing of size N*M ever reaches memory.
template <bool FIRST>
__global__ __launch_bounds__(256) void emd_pass3_cost_kernel(const float *__restrict__ xyz1,
                                                             const float *__restrict__ xyz2, int n, int m,
                                                             float level, const float *__restrict__ ratioL,
                                                             const float *__restrict__ ratioR,
                                                             float *__restrict__ remainL,
                                                             float *__restrict__ costrow) {
  __shared__ float4 tile[EMD_TILE];
  __shared__ float red[4 * EMD_ROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y, k = blockIdx.x * EMD_ROWS + lane;
  const float *p1 = xyz1 + (size_t)b * n * 3, *p2 = xyz2 + (size_t)b * m * 3;
  float x1 = 0, y1 = 0, z1 = 0, rl = 0;
  if (k < n) { x1 = p1[k * 3]; y1 = p1[k * 3 + 1]; z1 = p1[k * 3 + 2]; rl = ratioL[(size_t)b * n + k]; }
  float suml = 0, crow = 0;
  for (int l0 = 0; l0 < m; l0 += EMD_TILE) {
    const int ln = min(EMD_TILE, m - l0);
    __syncthreads();
    for (int l = tid; l < ln; l += 256)
      tile[l] = make_float4(p2[(size_t)(l0 + l) * 3], p2[(size_t)(l0 + l) * 3 + 1],
                            p2[(size_t)(l0 + l) * 3 + 2], ratioR[(size_t)b * m + l0 + l]);
    __syncthreads();
    const int l1 = min(ln, (wave + 1) * EMD_SUB);
#pragma unroll 8
    for (int l = wave * EMD_SUB; l < l1; ++l) {
      const float4 v = tile[l];
      const float d2 = sqdist3(v.x, v.y, v.z, x1, y1, z1);
      const float w = mul_rn(mul_rn(__expf(mul_rn(level, d2)), rl), v.w);
      suml = add_rn(suml, w);
      crow = add_rn(crow, mul_rn(d2, w));
    }
  }
  EMD_MERGE(red, wave, lane, suml)
  EMD_MERGE(red, wave, lane, crow)
  if (wave == 0 && k < n) {
    remainL[(size_t)b * n + k] = fmaxf(0.0f, sub_rn(remainL[(size_t)b * n + k], suml));
    costrow[(size_t)b * n + k] = FIRST ? crow : add_rn(costrow[(size_t)b * n + k], crow);
  }
}

// ---- round 6: the evaluation path (lion_emd_cost) on a shorter instruction stream ---------------------------------------
// The three passes of a level evaluate exp(level * d(k, l)) for all N x M pairs each: 30 N M evaluations per pair of clouds,
// every one 8 VALU operations for the distance + 2 multiplies + v_exp_f32 (quarter rate) + accumulate = 16-19 issue slots of
// the non-packed fp32 pipe -- the launches sit AT that issue rate (1.45 ms at 32 x 2048^2 = 64 G slots / 39 T slots per second),
// not at the transcendental rate the roofline is priced against.  Fewer slots per evaluation:
//   * coordinates are pre-scaled by s = sqrt(-level * log2(e)) when a tile is staged (and the row's own point once), so that
//     exp(level * d) = exp2(-d') with d' the squared distance of the scaled points: 3 subtractions, 1 multiply, 2 fused
//     multiply-adds, and v_exp_f32 on the negated operand -- no level multiply, no log2(e) multiply;
//   * pass 3 of a level and pass 1 of the NEXT level walk the same row over the same tiles: one kernel, one distance per
//     pair, the next level's exponent is d' * (level' / level) (0.25, or 0 before the last level);
//   * true squared distances for the cost come back as d' / s^2.
// Sums keep their order (ascending l inside each wave's quarter of a tile, the four quarters merged in a fixed order).  The
// weights differ from the literal expression by fp32 rounding of the exponent (<= 1e-5 relative), inside what v_exp_f32
// itself imposes (tests: cost 1e-4, the materialised match 2e-3); lion_emd_approxmatch keeps the literal kernels.
__device__ __forceinline__ float sqd_fma(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx));
}
__device__ __forceinline__ float exp2_neg(float a) { return __builtin_amdgcn_exp2f(-a); }

// Rows per lane: two (128 rows per workgroup, 512 workgroups at B = 32: two waves per SIMD) -- one 16-byte LDS broadcast read
// feeds two evaluations.  Measured (32 x 2048^2, rocprofv3): pass 2 37.3 -> 39.7 us, pass 3 + 1 76.7 -> 70.0 us, whole call
// 1169 -> 1132 us: the passes are NOT bound by LDS return bandwidth; what is left is the issue stream itself -- per pair
// 7.4 full-rate VALU instructions + one v_exp_f32 (two in the fused pass) -- at the clock the board sustains under it.
constexpr int EMD_RPL = 2;
constexpr int EMD_FROWS = EMD_ROWS * EMD_RPL;

#define EMD_MERGE2(red_, wave_, lane_, v_)                                                        \
  __syncthreads();                                                                                 \
  _Pragma("unroll") for (int q_ = 0; q_ < EMD_RPL; ++q_) red_[(wave_ * EMD_RPL + q_) * EMD_ROWS + lane_] = v_[q_]; \
  __syncthreads();                                                                                 \
  _Pragma("unroll") for (int q_ = 0; q_ < EMD_RPL; ++q_)                                           \
    v_[q_] = add_rn(add_rn(add_rn(red_[q_ * EMD_ROWS + lane_], red_[(EMD_RPL + q_) * EMD_ROWS + lane_]), \
                           red_[(2 * EMD_RPL + q_) * EMD_ROWS + lane_]), red_[(3 * EMD_RPL + q_) * EMD_ROWS + lane_]);

// scale / inv_s2: s and 1 / s^2 of THIS level (s = 0 at level 0: every weight is exp(0) = 1, distances unscaled)
__global__ __launch_bounds__(256) void emd_fast_pass1_kernel(const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                             int n, int m, float scale, const float *__restrict__ remainL,
                                                             const float *__restrict__ remainR, float *__restrict__ ratioL) {
  __shared__ float4 tile[EMD_TILE];
  __shared__ float red[4 * EMD_FROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y, k0r = blockIdx.x * EMD_FROWS + lane;
  const float *p1 = xyz1 + (size_t)b * n * 3, *p2 = xyz2 + (size_t)b * m * 3;
  float x1[EMD_RPL], y1[EMD_RPL], z1[EMD_RPL], suml[EMD_RPL];
#pragma unroll
  for (int q = 0; q < EMD_RPL; ++q) {
    const int k = k0r + q * EMD_ROWS;
    x1[q] = y1[q] = z1[q] = 0.f;
    if (k < n) { x1[q] = p1[k * 3] * scale; y1[q] = p1[k * 3 + 1] * scale; z1[q] = p1[k * 3 + 2] * scale; }
    suml[q] = wave == 0 ? 1e-9f : 0.f;
  }
  for (int l0 = 0; l0 < m; l0 += EMD_TILE) {
    const int ln = min(EMD_TILE, m - l0);
    __syncthreads();
    for (int l = tid; l < ln; l += 256)
      tile[l] = make_float4(p2[(size_t)(l0 + l) * 3] * scale, p2[(size_t)(l0 + l) * 3 + 1] * scale,
                            p2[(size_t)(l0 + l) * 3 + 2] * scale, remainR[(size_t)b * m + l0 + l]);
    __syncthreads();
    const int l1 = min(ln, (wave + 1) * EMD_SUB);
#pragma unroll 4
    for (int l = wave * EMD_SUB; l < l1; ++l) {
      const float4 v = tile[l];
#pragma unroll
      for (int q = 0; q < EMD_RPL; ++q)
        suml[q] = __fmaf_rn(exp2_neg(sqd_fma(v.x, v.y, v.z, x1[q], y1[q], z1[q])), v.w, suml[q]);
    }
  }
  EMD_MERGE2(red, wave, lane, suml)
  if (wave == 0)
#pragma unroll
    for (int q = 0; q < EMD_RPL; ++q) {
      const int k = k0r + q * EMD_ROWS;
      if (k < n) ratioL[(size_t)b * n + k] = div_rn(remainL[(size_t)b * n + k], suml[q]);
    }
}

__global__ __launch_bounds__(256) void emd_fast_pass2_kernel(const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                             int n, int m, float scale, const float *__restrict__ ratioL,
                                                             float *__restrict__ remainR, float *__restrict__ ratioR) {
  __shared__ float4 tile[EMD_TILE];
  __shared__ float red[4 * EMD_FROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y, l0r = blockIdx.x * EMD_FROWS + lane;
  const float *p1 = xyz1 + (size_t)b * n * 3, *p2 = xyz2 + (size_t)b * m * 3;
  float x2[EMD_RPL], y2[EMD_RPL], z2[EMD_RPL], sumr[EMD_RPL];
#pragma unroll
  for (int q = 0; q < EMD_RPL; ++q) {
    const int l = l0r + q * EMD_ROWS;
    x2[q] = y2[q] = z2[q] = 0.f;
    if (l < m) { x2[q] = p2[l * 3] * scale; y2[q] = p2[l * 3 + 1] * scale; z2[q] = p2[l * 3 + 2] * scale; }
    sumr[q] = 0.f;
  }
  for (int k0 = 0; k0 < n; k0 += EMD_TILE) {
    const int kn = min(EMD_TILE, n - k0);
    __syncthreads();
    for (int k = tid; k < kn; k += 256)
      tile[k] = make_float4(p1[(size_t)(k0 + k) * 3] * scale, p1[(size_t)(k0 + k) * 3 + 1] * scale,
                            p1[(size_t)(k0 + k) * 3 + 2] * scale, ratioL[(size_t)b * n + k0 + k]);
    __syncthreads();
    const int k1 = min(kn, (wave + 1) * EMD_SUB);
#pragma unroll 4
    for (int k = wave * EMD_SUB; k < k1; ++k) {
      const float4 v = tile[k];
#pragma unroll
      for (int q = 0; q < EMD_RPL; ++q)
        sumr[q] = __fmaf_rn(exp2_neg(sqd_fma(x2[q], y2[q], z2[q], v.x, v.y, v.z)), v.w, sumr[q]);
    }
  }
  EMD_MERGE2(red, wave, lane, sumr)
  if (wave == 0)
#pragma unroll
    for (int q = 0; q < EMD_RPL; ++q) {
      const int l = l0r + q * EMD_ROWS;
      if (l < m) {
        const float rr = remainR[(size_t)b * m + l];
        const float sr = mul_rn(sumr[q], rr);
        const float consumption = fminf(div_rn(rr, add_rn(sr, 1e-9f)), 1.0f);
        ratioR[(size_t)b * m + l] = mul_rn(consumption, rr);
        remainR[(size_t)b * m + l] = fmaxf(0.0f, sub_rn(rr, sr));
      }
    }
}

// pass 3 of this level (cost form) + pass 1 of the next level (NEXT: next_ratio = level' / level, 0 for the step onto level 0)
template <bool FIRST, bool NEXT>
__global__ __launch_bounds__(256) void emd_fast_pass31_kernel(const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                              int n, int m, float scale, float inv_s2, float next_ratio,
                                                              float *__restrict__ ratioL, const float *__restrict__ ratioR,
                                                              const float *__restrict__ remainR, float *__restrict__ remainL,
                                                              float *__restrict__ costrow) {
  __shared__ float4 tile[EMD_TILE];
  __shared__ float trem[EMD_TILE];
  __shared__ float red[4 * EMD_FROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y, k0r = blockIdx.x * EMD_FROWS + lane;
  const float *p1 = xyz1 + (size_t)b * n * 3, *p2 = xyz2 + (size_t)b * m * 3;
  const float cs = scale > 0.f ? scale : 1.f;          // level 0: plain coordinates, weights exp(0) = 1
  const float ws = scale > 0.f ? 1.f : 0.f, c2 = scale > 0.f ? inv_s2 : 1.f;
  float x1[EMD_RPL], y1[EMD_RPL], z1[EMD_RPL], rl[EMD_RPL], suml[EMD_RPL], crow[EMD_RPL], sum1[EMD_RPL];
#pragma unroll
  for (int q = 0; q < EMD_RPL; ++q) {
    const int k = k0r + q * EMD_ROWS;
    x1[q] = y1[q] = z1[q] = rl[q] = 0.f;
    if (k < n) { x1[q] = p1[k * 3] * cs; y1[q] = p1[k * 3 + 1] * cs; z1[q] = p1[k * 3 + 2] * cs; rl[q] = ratioL[(size_t)b * n + k]; }
    suml[q] = crow[q] = 0.f;
    sum1[q] = wave == 0 ? 1e-9f : 0.f;
  }
  for (int l0 = 0; l0 < m; l0 += EMD_TILE) {
    const int ln = min(EMD_TILE, m - l0);
    __syncthreads();
    for (int l = tid; l < ln; l += 256) {
      tile[l] = make_float4(p2[(size_t)(l0 + l) * 3] * cs, p2[(size_t)(l0 + l) * 3 + 1] * cs,
                            p2[(size_t)(l0 + l) * 3 + 2] * cs, ratioR[(size_t)b * m + l0 + l]);
      if (NEXT) trem[l] = remainR[(size_t)b * m + l0 + l];
    }
    __syncthreads();
    const int l1 = min(ln, (wave + 1) * EMD_SUB);
#pragma unroll 4
    for (int l = wave * EMD_SUB; l < l1; ++l) {
      const float4 v = tile[l];
      const float tr = NEXT ? trem[l] : 0.f;
#pragma unroll
      for (int q = 0; q < EMD_RPL; ++q) {
        const float dp = sqd_fma(v.x, v.y, v.z, x1[q], y1[q], z1[q]);
        const float w = (exp2_neg(dp * ws) * rl[q]) * v.w;
        suml[q] += w;
        crow[q] = __fmaf_rn(dp * c2, w, crow[q]);
        if (NEXT) sum1[q] = __fmaf_rn(exp2_neg(dp * next_ratio), tr, sum1[q]);
      }
    }
  }
  EMD_MERGE2(red, wave, lane, suml)
  EMD_MERGE2(red, wave, lane, crow)
  if (NEXT) { EMD_MERGE2(red, wave, lane, sum1) }
  if (wave == 0)
#pragma unroll
    for (int q = 0; q < EMD_RPL; ++q) {
      const int k = k0r + q * EMD_ROWS;
      if (k < n) {
        const float rem = fmaxf(0.0f, sub_rn(remainL[(size_t)b * n + k], suml[q]));
        remainL[(size_t)b * n + k] = rem;
        costrow[(size_t)b * n + k] = FIRST ? crow[q] : add_rn(costrow[(size_t)b * n + k], crow[q]);
        if (NEXT) ratioL[(size_t)b * n + k] = div_rn(rem, sum1[q]);   // pass 1 of the next level (emd_kernel.cu:50-81)
      }
    }
}

// cost[b] = sum_k costrow[b][k]: 256 strided partials, then a fixed tree (deterministic)
__global__ __launch_bounds__(256) void emd_costrow_sum_kernel(const float *__restrict__ costrow, int n,
                                                              float *__restrict__ cost) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float s = 0;
  for (int k = tid; k < n; k += 256) s = add_rn(s, costrow[(size_t)b * n + k]);
  for (int q = 32; q >= 1; q >>= 1) s = add_rn(s, __shfl_xor(s, q, 64));
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) cost[b] = add_rn(add_rn(red[0], red[1]), add_rn(red[2], red[3]));
}

// matchcost (emd_kernel.cu:199-241): per-k partial sums over l, reduced per block into
// partial[b][block]; a second tiny kernel sums the partials in a fixed order (deterministic).
__global__ __launch_bounds__(256) void emd_cost_partial_kernel(const float *__restrict__ xyz1,
                                                               const float *__restrict__ xyz2,
                                                               const float *__restrict__ match, int n,
                                                               int m, float *__restrict__ partial) {
  __shared__ float4 tile[EMD_TILE];
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y;
  const int k = blockIdx.x * 256 + tid;
  const float *p1 = xyz1 + (size_t)b * n * 3, *p2 = xyz2 + (size_t)b * m * 3;
  const float *mt = match + (size_t)b * n * m;
  float x1 = 0, y1 = 0, z1 = 0;
  if (k < n) { x1 = p1[k * 3]; y1 = p1[k * 3 + 1]; z1 = p1[k * 3 + 2]; }
  float sub = 0;
  for (int l0 = 0; l0 < m; l0 += EMD_TILE) {
    const int ln = min(EMD_TILE, m - l0);
    __syncthreads();
    for (int l = tid; l < ln; l += 256)
      tile[l] = make_float4(p2[(size_t)(l0 + l) * 3], p2[(size_t)(l0 + l) * 3 + 1],
                            p2[(size_t)(l0 + l) * 3 + 2], 0.f);
    __syncthreads();
    if (k < n)
#pragma unroll 4
      for (int l = 0; l < ln; ++l) {
        const float4 v = tile[l];
        sub = add_rn(sub, mul_rn(sqdist3(v.x, v.y, v.z, x1, y1, z1), mt[(size_t)(l0 + l) * n + k]));
      }
  }
  for (int s = 32; s >= 1; s >>= 1) sub = add_rn(sub, __shfl_xor(sub, s, 64));
  if (lane == 0) red[wave] = sub;
  __syncthreads();
  if (tid == 0)
    partial[(size_t)b * gridDim.x + blockIdx.x] = add_rn(add_rn(red[0], red[1]), add_rn(red[2], red[3]));
}

__global__ void emd_cost_final_kernel(const float *__restrict__ partial, int nblk, int B,
                                      float *__restrict__ cost) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s = 0;
  for (int i = 0; i < nblk; ++i) s = add_rn(s, partial[(size_t)b * nblk + i]);
  cost[b] = s;
}

// grad wrt xyz1 (emd_kernel.cu:332-353): thread per point l of cloud 1, k ascending
__global__ __launch_bounds__(256) void emd_grad1_kernel(const float *__restrict__ grad_cost,
                                                        const float *__restrict__ xyz1,
                                                        const float *__restrict__ xyz2,
                                                        const float *__restrict__ match, int n, int m,
                                                        float *__restrict__ grad1) {
  __shared__ float4 tile[EMD_TILE];
  const int tid = threadIdx.x, b = blockIdx.y, l = blockIdx.x * 256 + tid;
  const float *p1 = xyz1 + (size_t)b * n * 3, *p2 = xyz2 + (size_t)b * m * 3;
  const float *mt = match + (size_t)b * n * m;
  float x1 = 0, y1 = 0, z1 = 0;
  if (l < n) { x1 = p1[l * 3]; y1 = p1[l * 3 + 1]; z1 = p1[l * 3 + 2]; }
  float dx = 0, dy = 0, dz = 0;
  for (int k0 = 0; k0 < m; k0 += EMD_TILE) {
    const int kn = min(EMD_TILE, m - k0);
    __syncthreads();
    for (int k = tid; k < kn; k += 256)
      tile[k] = make_float4(p2[(size_t)(k0 + k) * 3], p2[(size_t)(k0 + k) * 3 + 1],
                            p2[(size_t)(k0 + k) * 3 + 2], 0.f);
    __syncthreads();
    if (l < n)
#pragma unroll 4
      for (int k = 0; k < kn; ++k) {
        const float4 v = tile[k];
        const float d = mul_rn(mt[(size_t)(k0 + k) * n + l], 2.0f);
        dx = add_rn(dx, mul_rn(sub_rn(x1, v.x), d));
        dy = add_rn(dy, mul_rn(sub_rn(y1, v.y), d));
        dz = add_rn(dz, mul_rn(sub_rn(z1, v.z), d));
      }
  }
  if (l < n) {
    const float g = grad_cost[b];
    grad1[((size_t)b * n + l) * 3 + 0] = mul_rn(dx, g);
    grad1[((size_t)b * n + l) * 3 + 1] = mul_rn(dy, g);
    grad1[((size_t)b * n + l) * 3 + 2] = mul_rn(dz, g);
  }
}

// grad wrt xyz2 (emd_kernel.cu:285-325): one wave per point k of cloud 2, lanes stride over j
__global__ __launch_bounds__(256) void emd_grad2_kernel(const float *__restrict__ grad_cost,
                                                        const float *__restrict__ xyz1,
                                                        const float *__restrict__ xyz2,
                                                        const float *__restrict__ match, int n, int m,
                                                        float *__restrict__ grad2) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y;
  const int k = blockIdx.x * 4 + wave;
  if (k >= m) return;
  const float *p1 = xyz1 + (size_t)b * n * 3, *p2 = xyz2 + (size_t)b * m * 3;
  const float *row = match + (size_t)b * n * m + (size_t)k * n;
  const float x2 = p2[k * 3], y2 = p2[k * 3 + 1], z2 = p2[k * 3 + 2];
  float sx = 0, sy = 0, sz = 0;
  for (int j = lane; j < n; j += 64) {
    const float d = mul_rn(row[j], 2.0f);
    sx = add_rn(sx, mul_rn(sub_rn(x2, p1[j * 3]), d));
    sy = add_rn(sy, mul_rn(sub_rn(y2, p1[j * 3 + 1]), d));
    sz = add_rn(sz, mul_rn(sub_rn(z2, p1[j * 3 + 2]), d));
  }
  for (int s = 32; s >= 1; s >>= 1) {
    sx = add_rn(sx, __shfl_xor(sx, s, 64));
    sy = add_rn(sy, __shfl_xor(sy, s, 64));
    sz = add_rn(sz, __shfl_xor(sz, s, 64));
  }
  if (lane == 0) {
    const float g = grad_cost[b];
    grad2[((size_t)b * m + k) * 3 + 0] = mul_rn(sx, g);
    grad2[((size_t)b * m + k) * 3 + 1] = mul_rn(sy, g);
    grad2[((size_t)b * m + k) * 3 + 2] = mul_rn(sz, g);
  }
}

static size_t emd_ws_floats(int B, int N, int M) {
  const size_t nb = (size_t)lion_cdiv(N, 256);
  // remainL[B,N] remainR[B,M] ratioL[B,N] ratioR[B,M] partial[B,nb] costrow[B,N]
  return (size_t)B * (3 * (size_t)N + 2 * (size_t)M + nb) + 64;
}

} // namespace

extern "C" {

size_t lion_emd_workspace_bytes(int B, int N, int M) {
  if (B <= 0 || N <= 0 || M <= 0) return 0;
  return emd_ws_floats(B, N, M) * 4;
}

int lion_emd_approxmatch(const float *xyz1, const float *xyz2, int B, int N, int M, float *match,
                         void *ws, size_t ws_bytes, lionStream_t stream) {
  if (!xyz1 || !xyz2 || !match || B <= 0 || N <= 0 || M <= 0) return LION_EINVAL;
  if (!ws || ws_bytes < lion_emd_workspace_bytes(B, N, M)) return LION_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float *remainL = static_cast<float *>(ws);
  float *remainR = remainL + (size_t)B * N;
  float *ratioL = remainR + (size_t)B * M;
  float *ratioR = ratioL + (size_t)B * N;
  float multiL, multiR; // emd_kernel.cu:27-33 (integer division)
  if (N >= M) { multiL = 1.f; multiR = (float)(N / M); }
  else { multiL = (float)(M / N); multiR = 1.f; }
  const int nmax = N > M ? N : M;
  emd_init_kernel<<<dim3(lion_cdiv(nmax, 256), B), 256, 0, st>>>(N, M, multiL, multiR, remainL, remainR);
  const dim3 gn(lion_cdiv(N, EMD_ROWS), B), gm(lion_cdiv(M, EMD_ROWS), B);
  for (int j = 7; j >= -2; --j) {
    float level = -powf(4.0f, (float)j); // :45-48
    if (j == -2) level = 0.f;
    emd_pass1_kernel<<<gn, 256, 0, st>>>(xyz1, xyz2, N, M, level, remainL, remainR, ratioL);
    emd_pass2_kernel<<<gm, 256, 0, st>>>(xyz1, xyz2, N, M, level, ratioL, remainR, ratioR);
    if (j == 7)
      emd_pass3_kernel<true><<<gn, 256, 0, st>>>(xyz1, xyz2, N, M, level, ratioL, ratioR, remainL, match);
    else
      emd_pass3_kernel<false><<<gn, 256, 0, st>>>(xyz1, xyz2, N, M, level, ratioL, ratioR, remainL, match);
  }
  LION_LAUNCH_CHECK();
  return 0;
}

// approxmatch + matchcost in one call without materialising match (no-grad evaluation path): cost f32[B].
int lion_emd_cost(const float *xyz1, const float *xyz2, int B, int N, int M, float *cost, void *ws,
                  size_t ws_bytes, lionStream_t stream) {
  if (!xyz1 || !xyz2 || !cost || B <= 0 || N <= 0 || M <= 0) return LION_EINVAL;
  if (!ws || ws_bytes < lion_emd_workspace_bytes(B, N, M)) return LION_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float *remainL = static_cast<float *>(ws);
  float *remainR = remainL + (size_t)B * N;
  float *ratioL = remainR + (size_t)B * M;
  float *ratioR = ratioL + (size_t)B * N;
  float *costrow = ratioR + (size_t)B * M + (size_t)B * lion_cdiv(N, 256);
  float multiL, multiR; // emd_kernel.cu:27-33 (integer division)
  if (N >= M) { multiL = 1.f; multiR = (float)(N / M); }
  else { multiL = (float)(M / N); multiR = 1.f; }
  const int nmax = N > M ? N : M;
  emd_init_kernel<<<dim3(lion_cdiv(nmax, 256), B), 256, 0, st>>>(N, M, multiL, multiR, remainL, remainR);
  const dim3 gn(lion_cdiv(N, EMD_FROWS), B), gm(lion_cdiv(M, EMD_FROWS), B);
  // round 6 (see emd_fast_*): pass 1 of the first level, then per level [pass 2, pass 3 + pass 1 of the next level]
  const float LOG2E = 1.4426950408889634f;
  auto level_of = [](int j) { return j == -2 ? 0.f : -powf(4.0f, (float)j); };   // :45-48
  auto scale_of = [&](float level) { return level < 0.f ? sqrtf(-level * LOG2E) : 0.f; };
  emd_fast_pass1_kernel<<<gn, 256, 0, st>>>(xyz1, xyz2, N, M, scale_of(level_of(7)), remainL, remainR, ratioL);
  for (int j = 7; j >= -2; --j) {
    const float level = level_of(j), sc = scale_of(level);
    const float inv_s2 = sc > 0.f ? 1.0f / (sc * sc) : 1.f;
    emd_fast_pass2_kernel<<<gm, 256, 0, st>>>(xyz1, xyz2, N, M, sc, ratioL, remainR, ratioR);
    if (j == -2) {
      emd_fast_pass31_kernel<false, false><<<gn, 256, 0, st>>>(xyz1, xyz2, N, M, sc, inv_s2, 0.f, ratioL, ratioR, remainR,
                                                                remainL, costrow);
    } else {
      const float next_ratio = level_of(j - 1) / level;     // 0.25; 0 for the step onto level 0
      if (j == 7)
        emd_fast_pass31_kernel<true, true><<<gn, 256, 0, st>>>(xyz1, xyz2, N, M, sc, inv_s2, next_ratio, ratioL, ratioR,
                                                               remainR, remainL, costrow);
      else
        emd_fast_pass31_kernel<false, true><<<gn, 256, 0, st>>>(xyz1, xyz2, N, M, sc, inv_s2, next_ratio, ratioL, ratioR,
                                                                remainR, remainL, costrow);
    }
  }
  emd_costrow_sum_kernel<<<B, 256, 0, st>>>(costrow, N, cost);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_emd_matchcost(const float *xyz1, const float *xyz2, const float *match, int B, int N, int M,
                       float *cost, void *ws, size_t ws_bytes, lionStream_t stream) {
  if (!xyz1 || !xyz2 || !match || !cost || B <= 0 || N <= 0 || M <= 0) return LION_EINVAL;
  if (!ws || ws_bytes < lion_emd_workspace_bytes(B, N, M)) return LION_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float *partial = static_cast<float *>(ws) + (size_t)B * (2 * (size_t)N + 2 * (size_t)M);
  const int nb = lion_cdiv(N, 256);
  emd_cost_partial_kernel<<<dim3(nb, B), 256, 0, st>>>(xyz1, xyz2, match, N, M, partial);
  emd_cost_final_kernel<<<lion_cdiv(B, 64), 64, 0, st>>>(partial, nb, B, cost);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_emd_matchcost_backward(const float *grad_cost, const float *xyz1, const float *xyz2,
                                const float *match, int B, int N, int M, float *grad1, float *grad2,
                                lionStream_t stream) {
  if (!grad_cost || !xyz1 || !xyz2 || !match || !grad1 || !grad2 || B <= 0 || N <= 0 || M <= 0)
    return LION_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  emd_grad1_kernel<<<dim3(lion_cdiv(N, 256), B), 256, 0, st>>>(grad_cost, xyz1, xyz2, match, N, M, grad1);
  emd_grad2_kernel<<<dim3(lion_cdiv(M, 4), B), 256, 0, st>>>(grad_cost, xyz1, xyz2, match, N, M, grad2);
  LION_LAUNCH_CHECK();
  return 0;
}

} // extern "C"
