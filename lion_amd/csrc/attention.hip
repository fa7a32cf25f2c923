// attention.hip -- P5: the core of LinearAttention (models/pvcnn2_ada.py:43-71) between its two 1x1 convolutions:
//   q, k, v = split(to_qkv(x))                    [B, heads, 32, N] each  (channel order (qkv, head, d))
//   p = softmax over the N points of k, per row d
//   ctx[d][e] = sum_n p[d][n] v[e][n]             [32, 32] per (batch, head)
//   out[e][n] = sum_d ctx[d][e] q[d][n]           -> [B, heads * 32, N]
// The reference issues a rearrange copy, a softmax and two batched einsums (bmm) per call.  Here one workgroup per
// (batch, head) does all of it on the fp32 matrix cores: the row maxima / sums in two sweeps over the L2-resident k
// rows, ctx as a 32 x 32 x N GEMM whose operands go through LDS transposed (n-major, padded stride 33: conflict-free
// MFMA operand reads), out as a 32 x N x 32 GEMM whose B operand (q) is read straight from memory in 128-byte rows.
// Exact fp32 (fmaf chains), accurate expf: compared with a float64 evaluation at 1e-5 (tests).
#include "common.h"
#include <math.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int AD = 32;   // dim_head of every LinearAttention in the models
constexpr int AT = 64;   // points per staged tile

__global__ __launch_bounds__(256) void linattn_core_kernel(const float *__restrict__ qkv, int H, int N,
                                                           float *__restrict__ out) {
  __shared__ float pT[AT * 33], vT[AT * 33]; // [n][d] / [n][e], stride 33
  __shared__ float part[4][1024];
  __shared__ float rmax[AD], rinv[AD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cl = lane & 31, kh = lane >> 5;
  const int h = blockIdx.x, b = blockIdx.y;
  const size_t bs = (size_t)b * 3 * H * AD * N;
  const float *q = qkv + bs + (size_t)(0 * H + h) * AD * N;
  const float *k = qkv + bs + (size_t)(1 * H + h) * AD * N;
  const float *v = qkv + bs + (size_t)(2 * H + h) * AD * N;

  // ---- softmax statistics of the 32 rows of k: 8 lanes per row ----
  {
    const int d = tid >> 3, sub = tid & 7;
    const float *kr = k + (size_t)d * N;
    float m = -INFINITY;
    for (int n = sub; n < N; n += 8) { const float t = kr[n]; m = t > m ? t : m; }
    for (int s = 1; s < 8; s <<= 1) { const float o = __shfl_xor(m, s, 64); m = o > m ? o : m; }
    float sum = 0.f;
    for (int n = sub; n < N; n += 8) sum += expf(kr[n] - m);
    for (int s = 1; s < 8; s <<= 1) sum += __shfl_xor(sum, s, 64);
    if (sub == 0) { rmax[d] = m; rinv[d] = 1.0f / sum; }
  }
  __syncthreads();

  // ---- ctx[d][e] = sum_n p[d][n] v[e][n]: tiles of AT points through LDS, each wave 8 of the tile's 32 k-steps ----
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int tn = tid & 63, d0 = (tid >> 6) * 8; // staging: thread = (point of the tile, 8 rows)
  for (int n0 = 0; n0 < N; n0 += AT) {
    const int n = n0 + tn;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = d0 + j;
      float pv = 0.f, vv = 0.f;
      if (n < N) {
        pv = expf(k[(size_t)d * N + n] - rmax[d]) * rinv[d];
        vv = v[(size_t)d * N + n];
      }
      pT[tn * 33 + d] = pv;
      vT[tn * 33 + d] = vv;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int nn = 2 * (wave * 8 + u) + kh;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pT[nn * 33 + cl], vT[nn * 33 + cl], acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // acc register i of lane l: row d = (i&3) + 8*(i>>2) + 4*kh, column e = cl
#pragma unroll
  for (int i = 0; i < 16; ++i) part[wave][((i & 3) + 8 * (i >> 2) + 4 * kh) * 32 + cl] = acc[i];
  __syncthreads();
  float *ctx = pT; // [d][e], stride 33 (the tiles are done)
#pragma unroll
  for (int e = tid; e < 1024; e += 256)
    ctx[(e >> 5) * 33 + (e & 31)] = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
  __syncthreads();

  // ---- out[e][n] = sum_d ctx[d][e] q[d][n]: rows e, 32-point column blocks round-robin over the waves ----
  float av[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) av[s] = ctx[(2 * s + kh) * 33 + cl]; // A[e = cl][d = 2s + kh]
  float *ob = out + ((size_t)b * H + h) * AD * N;
  for (int c0 = wave * 32; c0 < N; c0 += 128) {
    const int n = c0 + cl;
    const bool ok = n < N;
    const int nc = ok ? n : N - 1;
    float bv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) bv[s] = q[(size_t)(2 * s + kh) * N + nc];
    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) o = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], ok ? bv[s] : 0.f, o, 0, 0, 0);
    if (ok) {
#pragma unroll
      for (int i = 0; i < 16; ++i) ob[(size_t)((i & 3) + 8 * (i >> 2) + 4 * kh) * N + n] = o[i];
    }
  }
}

// Backward of the core (training; reference models/pvcnn2_ada.py:62-68 differentiated).  With p = softmax_n(k), g = the
// gradient of out:
//   gctx[d][e] = sum_n q[d][n] g[e][n]          gq[d][n] = sum_e ctx[d][e] g[e][n]        gv[e][n] = sum_d p[d][n] gctx[d][e]
//   gp[d][n]   = sum_e gctx[d][e] v[e][n]       gk[d][n] = p[d][n] (gp[d][n] - dot[d]),   dot[d] = sum_n p gp = sum_e gctx[d][e] ctx[d][e]
// (the softmax backward's row dot product needs no second sweep over the points: it is a 32 x 32 contraction).  One
// workgroup per (batch, head), as the forward: the row statistics of k, ctx and gctx as two 32 x 32 x N fp32-MFMA GEMMs
// over tiles staged through LDS, then per 32-point column block three 32 x 32 x 32 GEMMs (gq, gp, gv) whose B operands
// are read straight from memory in 128-byte rows.  Exact fp32 products (fmaf chains); compared with float64 autograd.
__global__ __launch_bounds__(256) void linattn_core_bwd_kernel(const float *__restrict__ qkv, const float *__restrict__ gout,
                                                               int H, int N, float *__restrict__ gqkv) {
  __shared__ float pT[AT * 33], vT[AT * 33], qT[AT * 33], gT[AT * 33]; // [n][row], stride 33
  __shared__ float part[4][1024];
  __shared__ float ctxs[AD * 33], gctxs[AD * 33];
  __shared__ float rmax[AD], rinv[AD], dots[AD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cl = lane & 31, kh = lane >> 5;
  const int h = blockIdx.x, b = blockIdx.y;
  const size_t bs = (size_t)b * 3 * H * AD * N;
  const float *q = qkv + bs + (size_t)(0 * H + h) * AD * N;
  const float *k = qkv + bs + (size_t)(1 * H + h) * AD * N;
  const float *v = qkv + bs + (size_t)(2 * H + h) * AD * N;
  const float *g = gout + ((size_t)b * H + h) * AD * N;
  float *gq = gqkv + bs + (size_t)(0 * H + h) * AD * N;
  float *gk = gqkv + bs + (size_t)(1 * H + h) * AD * N;
  float *gv = gqkv + bs + (size_t)(2 * H + h) * AD * N;

  { // softmax statistics of the 32 rows of k (as the forward)
    const int d = tid >> 3, sub = tid & 7;
    const float *kr = k + (size_t)d * N;
    float m = -INFINITY;
    for (int n = sub; n < N; n += 8) { const float t = kr[n]; m = t > m ? t : m; }
    for (int s = 1; s < 8; s <<= 1) { const float o = __shfl_xor(m, s, 64); m = o > m ? o : m; }
    float sum = 0.f;
    for (int n = sub; n < N; n += 8) sum += expf(kr[n] - m);
    for (int s = 1; s < 8; s <<= 1) sum += __shfl_xor(sum, s, 64);
    if (sub == 0) { rmax[d] = m; rinv[d] = 1.0f / sum; }
  }
  __syncthreads();

  // ctx[d][e] = sum_n p[d][n] v[e][n] and gctx[d][e] = sum_n q[d][n] g[e][n]
  f32x16 acc, gacc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = gacc[i] = 0.f;
  const int tn = tid & 63, d0 = (tid >> 6) * 8;
  for (int n0 = 0; n0 < N; n0 += AT) {
    const int n = n0 + tn;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = d0 + j;
      float pv = 0.f, vv = 0.f, qv = 0.f, gg = 0.f;
      if (n < N) {
        pv = expf(k[(size_t)d * N + n] - rmax[d]) * rinv[d];
        vv = v[(size_t)d * N + n];
        qv = q[(size_t)d * N + n];
        gg = g[(size_t)d * N + n];
      }
      pT[tn * 33 + d] = pv;
      vT[tn * 33 + d] = vv;
      qT[tn * 33 + d] = qv;
      gT[tn * 33 + d] = gg;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int nn = 2 * (wave * 8 + u) + kh;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pT[nn * 33 + cl], vT[nn * 33 + cl], acc, 0, 0, 0);
      gacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qT[nn * 33 + cl], gT[nn * 33 + cl], gacc, 0, 0, 0);
    }
    __syncthreads();
  }
  // the four waves' partial sums in a fixed order, one matrix after the other through the same buffer
#pragma unroll
  for (int i = 0; i < 16; ++i) part[wave][((i & 3) + 8 * (i >> 2) + 4 * kh) * 32 + cl] = acc[i];
  __syncthreads();
  for (int e = tid; e < 1024; e += 256)
    ctxs[(e >> 5) * 33 + (e & 31)] = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) part[wave][((i & 3) + 8 * (i >> 2) + 4 * kh) * 32 + cl] = gacc[i];
  __syncthreads();
  for (int e = tid; e < 1024; e += 256)
    gctxs[(e >> 5) * 33 + (e & 31)] = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
  __syncthreads();
  if (tid < AD) { // dot[d] = sum_e gctx[d][e] ctx[d][e], ascending e
    float s = 0.f;
    for (int e = 0; e < AD; ++e) s += gctxs[tid * 33 + e] * ctxs[tid * 33 + e];
    dots[tid] = s;
  }
  __syncthreads();

  // per 32-point column block: gq = ctx g, gp = gctx v (rows d, k = e); gv = gctx^T p (rows e, k = d)
  float aq[16], ap[16], av[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    aq[s] = ctxs[cl * 33 + 2 * s + kh];   // A[d = cl][e = 2s + kh]
    ap[s] = gctxs[cl * 33 + 2 * s + kh];
    av[s] = gctxs[(2 * s + kh) * 33 + cl]; // A[e = cl][d = 2s + kh]
  }
  for (int c0 = wave * 32; c0 < N; c0 += 128) {
    const int n = c0 + cl;
    const bool ok = n < N;
    const int nc = ok ? n : N - 1;
    float bg[16], bv[16], bp[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int r_ = 2 * s + kh;
      bg[s] = ok ? g[(size_t)r_ * N + nc] : 0.f;
      bv[s] = ok ? v[(size_t)r_ * N + nc] : 0.f;
      bp[s] = ok ? expf(k[(size_t)r_ * N + nc] - rmax[r_]) * rinv[r_] : 0.f;
    }
    f32x16 oq, op, ov;
#pragma unroll
    for (int i = 0; i < 16; ++i) oq[i] = op[i] = ov[i] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      oq = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[s], bg[s], oq, 0, 0, 0);
      op = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[s], bv[s], op, 0, 0, 0);
      ov = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bp[s], ov, 0, 0, 0);
    }
    if (ok) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * kh;
        gq[(size_t)row * N + n] = oq[i];
        gv[(size_t)row * N + n] = ov[i];
        const float pr = expf(k[(size_t)row * N + n] - rmax[row]) * rinv[row];
        gk[(size_t)row * N + n] = pr * (op[i] - dots[row]);
      }
    }
  }
}

} // namespace

extern "C" {

// qkv f32[B, 3*H*32, N] (output of to_qkv, channel order (qkv, head, d)) -> out f32[B, H*32, N] (input of to_out).
int lion_linear_attention_core(const float *qkv, int B, int H, int D, int N, float *out, lionStream_t stream) {
  if (!qkv || !out || B <= 0 || H <= 0 || N <= 0) return LION_EINVAL;
  if (D != AD) return LION_EUNSUPPORTED;
  linattn_core_kernel<<<dim3(H, B), 256, 0, static_cast<hipStream_t>(stream)>>>(qkv, H, N, out);
  LION_LAUNCH_CHECK();
  return 0;
}

// gradient of lion_linear_attention_core: qkv as the forward's, gout f32[B, H*32, N] -> gqkv f32[B, 3*H*32, N]
int lion_linear_attention_core_backward(const float *qkv, const float *gout, int B, int H, int D, int N, float *gqkv,
                                        lionStream_t stream) {
  if (!qkv || !gout || !gqkv || B <= 0 || H <= 0 || N <= 0) return LION_EINVAL;
  if (D != AD) return LION_EUNSUPPORTED;
  linattn_core_bwd_kernel<<<dim3(H, B), 256, 0, static_cast<hipStream_t>(stream)>>>(qkv, gout, H, N, gqkv);
  LION_LAUNCH_CHECK();
  return 0;
}

} // extern "C"
