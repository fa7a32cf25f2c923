// norm_train.hip -- the TRAINING forms of P2/P3 (GroupNorm / AdaGN + Swish around the voxel and point convolutions;
// reference models/adagn.py:45-65, models/pvcnn2_ada.py:78-84, :120-164, :211-222).
//
// ATen runs y = swish(GroupNorm8(x) * factor + bias) as group_norm, mul, add, sigmoid, mul -- five passes over the
// activation forward (268 MB at [32, 64, 32^3]) and about ten backward.  A GroupNorm followed by per-(batch, channel)
// scalars and an activation is, per row (b, c) of length L,
//     y = act(A x + Bs),   A = rstd_g gw_c f_bc,   Bs = (gb_c - mean_g rstd_g gw_c) f_bc + b_bc
// so the forward is lion_row_stats (one read) + lion_affine_act (one read, one write), and the backward -- with
// da = gy act'(A x + Bs), n = (x - mean) rstd, dn = gw f da --
//     dx = rstd (dn - mean_g(dn) - n mean_g(dn n)) = A da + Q_bg + R_bg x
// needs the per-row sums S1 = sum da, S2 = sum da x (lion_affine_act_bwd_stats: reads x, gy) and one elementwise pass
// (lion_affine_act_bwd_apply: reads x, gy, writes dx); every parameter gradient is a combination of S1, S2 on [B, C]
// scalars (lion_amd/train_ops.py).  act: 0 identity, 1 swish.
#include "common.h"

namespace {

__device__ __forceinline__ float act_f(float a, int act) { return act ? swish_fast(a) : a; }
// d act(a) / da;  swish'(a) = s (1 + a (1 - s)),  s = sigmoid(a)
__device__ __forceinline__ float act_d(float a, int act) {
  if (!act) return 1.0f;
  const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-a));
  return s * (1.0f + a * (1.0f - s));
}

// Inverted dropout behind the activation (nn.Dropout after Swish in PVConv's voxel branch, reference pvcnn2_ada.py:211-222), DROP
// variants: element e of the tensor is kept when word (e & 3) of Philox4x32-10(counter = e >> 2, key = the 64-bit seed read from
// device memory) is below thr = keep * 2^32, and scaled by 1 / keep.  The seed is drawn per call by torch's generator (graph-safe:
// a replayed step draws a new one), forward and both backward kernels regenerate the same mask from it -- no mask tensor, no
// extra pass (ATen: fused_dropout reads x, writes y + mask; masked_scale reads gy + mask, writes).
__device__ __forceinline__ void drop4(const unsigned long long *__restrict__ seed, size_t e, unsigned thr, float scale, float m[4]) {
  uint32_t r[4];
  const unsigned long long s = seed[0], quad = (unsigned long long)e >> 2;
  philox4x32_10((uint32_t)quad, (uint32_t)(quad >> 32), 0x44524f50u, 0u, (uint32_t)s, (uint32_t)(s >> 32), r);
#pragma unroll
  for (int j = 0; j < 4; ++j) m[j] = r[j] < thr ? scale : 0.f;
}
__device__ __forceinline__ float drop1(const unsigned long long *__restrict__ seed, size_t e, unsigned thr, float scale) {
  float m[4];
  drop4(seed, e, thr, scale, m);
  const int k = (int)(e & 3);
  return k == 0 ? m[0] : k == 1 ? m[1] : k == 2 ? m[2] : m[3];
}

template <bool DROP>
__global__ __launch_bounds__(256) void affine_act_kernel(const float *__restrict__ x, const float *__restrict__ A,
                                                         const float *__restrict__ Bs, int L, int act,
                                                         float *__restrict__ y,
                                                         const unsigned long long *__restrict__ seed, unsigned thr,
                                                         float scale) {
  const int bpr = (((L + 3) >> 2) + 255) >> 8; // workgroups per row: the row index rides on blockIdx.x (B C > 65535 rows exist)
  const int row = blockIdx.x / bpr;
  const float a = A[row], b = Bs[row];
  const float *p = x + (size_t)row * L;
  float *q = y + (size_t)row * L;
  const int i = ((blockIdx.x - row * bpr) * 256 + threadIdx.x) * 4;
  if (i + 3 < L && (L & 3) == 0) {
    const float4 v = *reinterpret_cast<const float4 *>(p + i);
    float4 o = make_float4(act_f(v.x * a + b, act), act_f(v.y * a + b, act), act_f(v.z * a + b, act), act_f(v.w * a + b, act));
    if (DROP) {
      float m[4];
      drop4(seed, (size_t)row * L + i, thr, scale, m);
      o = make_float4(o.x * m[0], o.y * m[1], o.z * m[2], o.w * m[3]);
    }
    *reinterpret_cast<float4 *>(q + i) = o;
  } else {
    for (int j = i; j < L && j < i + 4; ++j)
      q[j] = act_f(p[j] * a + b, act) * (DROP ? drop1(seed, (size_t)row * L + j, thr, scale) : 1.0f);
  }
}

// one workgroup per row: stats[row] = {sum x, sum x^2} as DOUBLES.  The sums run over x - x0 (x0 = the row's first
// element) in fp32 and are moved back in double: E[x^2] - E[x]^2 from plain fp32 sums loses (mean / std)^2 x 1e-7 of the
// variance, i.e. everything once |mean| >> std (round-3 advisor finding; ATen's group_norm does not).
__global__ __launch_bounds__(256) void row_stats64_kernel(const float *__restrict__ x, int L, double *__restrict__ stats) {
  __shared__ float r1[4], r2[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *p = x + (size_t)row * L;
  const float x0 = p[0];
  float s1 = 0.f, s2 = 0.f;
  if ((L & 3) == 0) {
    for (int i = tid * 4; i < L; i += 1024) {
      const float4 v = *reinterpret_cast<const float4 *>(p + i);
      const float a = v.x - x0, b = v.y - x0, c = v.z - x0, d = v.w - x0;
      s1 += (a + b) + (c + d);
      s2 += (a * a + b * b) + (c * c + d * d);
    }
  } else {
    for (int i = tid; i < L; i += 256) { const float v = p[i] - x0; s1 += v; s2 += v * v; }
  }
  s1 = row16_sum_rn(s1); s2 = row16_sum_rn(s2);
#pragma unroll
  for (int m = 16; m < 64; m <<= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
  if (lane == 0) { r1[wave] = s1; r2[wave] = s2; }
  __syncthreads();
  if (tid == 0) {
    const double t1 = ((double)r1[0] + (double)r1[1]) + ((double)r1[2] + (double)r1[3]);
    const double t2 = ((double)r2[0] + (double)r2[1]) + ((double)r2[2] + (double)r2[3]);
    const double s = (double)x0, n = (double)L;
    stats[(size_t)row * 2] = t1 + n * s;                       // sum x
    stats[(size_t)row * 2 + 1] = t2 + 2.0 * s * t1 + n * s * s; // sum x^2
  }
}

// one workgroup per row: S[row] = {sum da, sum da x}
template <bool DROP>
__global__ __launch_bounds__(256) void affine_act_bwd_stats_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                   const float *__restrict__ A,
                                                                   const float *__restrict__ Bs, int L, int act,
                                                                   float *__restrict__ S,
                                                                   const unsigned long long *__restrict__ seed, unsigned thr,
                                                                   float scale) {
  __shared__ float r1[4], r2[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float a = A[row], b = Bs[row];
  const float *p = x + (size_t)row * L, *g = gy + (size_t)row * L;
  float s1 = 0.f, s2 = 0.f;
  if ((L & 3) == 0) {
    for (int i = tid * 4; i < L; i += 1024) {
      const float4 v = *reinterpret_cast<const float4 *>(p + i);
      float4 w = *reinterpret_cast<const float4 *>(g + i);
      if (DROP) {
        float m[4];
        drop4(seed, (size_t)row * L + i, thr, scale, m);
        w = make_float4(w.x * m[0], w.y * m[1], w.z * m[2], w.w * m[3]);
      }
      const float d0 = w.x * act_d(v.x * a + b, act), d1 = w.y * act_d(v.y * a + b, act);
      const float d2 = w.z * act_d(v.z * a + b, act), d3 = w.w * act_d(v.w * a + b, act);
      s1 += (d0 + d1) + (d2 + d3);
      s2 += (d0 * v.x + d1 * v.y) + (d2 * v.z + d3 * v.w);
    }
  } else {
    for (int i = tid; i < L; i += 256) {
      const float d = g[i] * (DROP ? drop1(seed, (size_t)row * L + i, thr, scale) : 1.0f) * act_d(p[i] * a + b, act);
      s1 += d;
      s2 += d * p[i];
    }
  }
  s1 = row16_sum_rn(s1); s2 = row16_sum_rn(s2);
#pragma unroll
  for (int m = 16; m < 64; m <<= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
  if (lane == 0) { r1[wave] = s1; r2[wave] = s2; }
  __syncthreads();
  if (tid == 0) {
    S[(size_t)row * 2] = (r1[0] + r1[1]) + (r1[2] + r1[3]);
    S[(size_t)row * 2 + 1] = (r2[0] + r2[1]) + (r2[2] + r2[3]);
  }
}

template <bool DROP>
__global__ __launch_bounds__(256) void affine_act_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                   const float *__restrict__ A,
                                                                   const float *__restrict__ Bs,
                                                                   const float *__restrict__ Q,
                                                                   const float *__restrict__ R, int L, int act,
                                                                   float *__restrict__ dx,
                                                                   const unsigned long long *__restrict__ seed, unsigned thr,
                                                                   float scale) {
  const int bpr = (((L + 3) >> 2) + 255) >> 8;
  const int row = blockIdx.x / bpr;
  const float a = A[row], b = Bs[row], qq = Q[row], rr = R[row];
  const float *p = x + (size_t)row * L, *g = gy + (size_t)row * L;
  float *o = dx + (size_t)row * L;
  const int i = ((blockIdx.x - row * bpr) * 256 + threadIdx.x) * 4;
  if (i + 3 < L && (L & 3) == 0) {
    const float4 v = *reinterpret_cast<const float4 *>(p + i);
    float4 w = *reinterpret_cast<const float4 *>(g + i);
    if (DROP) {
      float m[4];
      drop4(seed, (size_t)row * L + i, thr, scale, m);
      w = make_float4(w.x * m[0], w.y * m[1], w.z * m[2], w.w * m[3]);
    }
    *reinterpret_cast<float4 *>(o + i) = make_float4(a * (w.x * act_d(v.x * a + b, act)) + qq + rr * v.x,
                                                     a * (w.y * act_d(v.y * a + b, act)) + qq + rr * v.y,
                                                     a * (w.z * act_d(v.z * a + b, act)) + qq + rr * v.z,
                                                     a * (w.w * act_d(v.w * a + b, act)) + qq + rr * v.w);
  } else {
    for (int j = i; j < L && j < i + 4; ++j)
      o[j] = a * (g[j] * (DROP ? drop1(seed, (size_t)row * L + j, thr, scale) : 1.0f) * act_d(p[j] * a + b, act)) + qq + rr * p[j];
  }
}

// dgw[c] = sum_b pw[b][c][0], dgb[c] = sum_b pw[b][c][1], dxs[c] = sum_b pw[b][c][2] in ascending b (what pw.sum(0) + strided copies
// did in three ATen launches per AdaGN layer and backward)
__global__ __launch_bounds__(256) void pw_batch_sum_kernel(const float *__restrict__ pw, int B, int C, float *__restrict__ dgw,
                                                           float *__restrict__ dgb, float *__restrict__ dxs) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int b0 = 0; b0 < B; b0 += 16) {   // 48 loads in flight (one dependent round trip per sample made this a 10-us kernel), summed in order
    float v[16][3];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float *p = pw + ((size_t)min(b0 + k, B - 1) * C + c) * 3;
      v[k][0] = p[0]; v[k][1] = p[1]; v[k][2] = p[2];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (b0 + k < B) { s0 += v[k][0]; s1 += v[k][1]; s2 += v[k][2]; }
  }
  dgw[c] = s0;
  dgb[c] = s1;
  if (dxs) dxs[c] = s2;
}

// ---- round 6: y[row][m] = max_u act(A x[row][m][u] + Bs) -- the set-abstraction pooling behind the last AdaGN + Swish of a
// grouped SharedMLP (reference models/pvcnn2_ada.py:375-377) as ONE differentiable op.  The activated [B,C,M,U] tensor is never
// written (forward) and its gradient -- zero except at each group's arg-max -- never materialised (backward): every pass
// recomputes act(A x + Bs) over the U neighbours of a group from x.  Arg-max = the FIRST maximal neighbour (ball query pads a
// group with copies of its first hit: which copy receives the gradient does not change the sum grouping's backward forms).
template <int U4>   // U = 4 * U4 neighbours per group
__device__ __forceinline__ int group_argmax(const float4 *__restrict__ p, float a, float b, int act, float &vmax, float &xmax) {
  int best = 0;
  vmax = -INFINITY; xmax = 0.f;
#pragma unroll
  for (int q = 0; q < U4; ++q) {
    const float4 v = p[q];
    const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float y = act_f(xs[j] * a + b, act);
      if (y > vmax) { vmax = y; xmax = xs[j]; best = 4 * q + j; }   // strict: the first maximum wins
    }
  }
  return best;
}
template <int U4>
__global__ __launch_bounds__(256) void affine_act_max_kernel(const float *__restrict__ x, const float *__restrict__ A,
                                                             const float *__restrict__ Bs, int M, int act, float *__restrict__ y) {
  const int row = blockIdx.y, m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float vmax, xmax;
  group_argmax<U4>(reinterpret_cast<const float4 *>(x + ((size_t)row * M + m) * (4 * U4)), A[row], Bs[row], act, vmax, xmax);
  y[(size_t)row * M + m] = vmax;
}
// S[row] = {sum da, sum da x} with da = gy[row][m] act'(A x* + Bs) at the group's arg-max x*, 0 elsewhere
template <int U4>
__global__ __launch_bounds__(256) void affine_act_max_bwd_stats_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                       const float *__restrict__ A, const float *__restrict__ Bs,
                                                                       int M, int act, float *__restrict__ S) {
  __shared__ float r1[4], r2[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float a = A[row], b = Bs[row];
  float s1 = 0.f, s2 = 0.f;
  for (int m = tid; m < M; m += 256) {
    float vmax, xmax;
    group_argmax<U4>(reinterpret_cast<const float4 *>(x + ((size_t)row * M + m) * (4 * U4)), a, b, act, vmax, xmax);
    const float da = gy[(size_t)row * M + m] * act_d(xmax * a + b, act);
    s1 += da;
    s2 += da * xmax;
  }
  s1 = row16_sum_rn(s1); s2 = row16_sum_rn(s2);
#pragma unroll
  for (int q = 16; q < 64; q <<= 1) { s1 += __shfl_xor(s1, q, 64); s2 += __shfl_xor(s2, q, 64); }
  if (lane == 0) { r1[wave] = s1; r2[wave] = s2; }
  __syncthreads();
  if (tid == 0) {
    S[(size_t)row * 2] = (r1[0] + r1[1]) + (r1[2] + r1[3]);
    S[(size_t)row * 2 + 1] = (r2[0] + r2[1]) + (r2[2] + r2[3]);
  }
}
// dx[row][m][u] = (u == arg-max ? A da : 0) + Q + R x
template <int U4>
__global__ __launch_bounds__(256) void affine_act_max_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                       const float *__restrict__ A, const float *__restrict__ Bs,
                                                                       const float *__restrict__ Q, const float *__restrict__ R,
                                                                       int M, int act, float *__restrict__ dx) {
  const int row = blockIdx.y, m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const float a = A[row], b = Bs[row], qq = Q[row], rr = R[row];
  const float4 *p = reinterpret_cast<const float4 *>(x + ((size_t)row * M + m) * (4 * U4));
  float vmax, xmax;
  const int best = group_argmax<U4>(p, a, b, act, vmax, xmax);
  const float g = a * (gy[(size_t)row * M + m] * act_d(xmax * a + b, act));
  float4 *o = reinterpret_cast<float4 *>(dx + ((size_t)row * M + m) * (4 * U4));
#pragma unroll
  for (int q = 0; q < U4; ++q) {
    const float4 v = p[q];
    float4 d = make_float4(qq + rr * v.x, qq + rr * v.y, qq + rr * v.z, qq + rr * v.w);
    if (best == 4 * q) d.x += g;
    if (best == 4 * q + 1) d.y += g;
    if (best == 4 * q + 2) d.z += g;
    if (best == 4 * q + 3) d.w += g;
    o[q] = d;
  }
}

// the [B, C] scalar algebra of both directions, one workgroup per sample (C <= 1024 threads), in double
// forward: stats [B,C,2] (row sums) -> A, Bs, mean, rstd [B,C]
template <typename ST>
__global__ void gn_train_fold_kernel(const ST *__restrict__ stats, const float *__restrict__ gw,
                                     const float *__restrict__ gb, const float *__restrict__ fac, int fs,
                                     const float *__restrict__ bia, int bs, int C, int G, int L, float eps,
                                     float *__restrict__ A, float *__restrict__ Bs, float *__restrict__ mean,
                                     float *__restrict__ rstd) {
  extern __shared__ double sh[]; // [2 C]
  const int b = blockIdx.x, c = threadIdx.x, cpg = C / G;
  if (c < C) { sh[c] = stats[((size_t)b * C + c) * 2]; sh[C + c] = stats[((size_t)b * C + c) * 2 + 1]; }
  __syncthreads();
  if (c >= C) return;
  const int g0 = (c / cpg) * cpg;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < cpg; ++k) { s1 += sh[g0 + k]; s2 += sh[C + g0 + k]; }
  const double cnt = (double)cpg * (double)L, m = s1 / cnt;
  double var = s2 / cnt - m * m;
  var = var > 0.0 ? var : 0.0;
  const double rs = 1.0 / sqrt(var + (double)eps);
  const double f = fac ? (double)fac[(size_t)b * fs + c] : 1.0, bb = bia ? (double)bia[(size_t)b * bs + c] : 0.0;
  const double w = gw[c], o = gb[c];
  A[(size_t)b * C + c] = (float)(rs * w * f);
  Bs[(size_t)b * C + c] = (float)((o - m * rs * w) * f + bb);
  mean[(size_t)b * C + c] = (float)m;
  rstd[(size_t)b * C + c] = (float)rs;
}

// backward: S [B,C,2] = {sum da, sum da x} -> Q, R (dx = A da + Q + R x), dfac, dbias [B,C], and the per-sample
// terms of the GroupNorm parameter gradients pw [B,C,2] = {f T, f S1} (summed over the batch by the caller)
__global__ void gn_train_bwd_fold_kernel(const float *__restrict__ S, const float *__restrict__ mean,
                                         const float *__restrict__ rstd, const float *__restrict__ gw,
                                         const float *__restrict__ gb, const float *__restrict__ fac, int fs, int C,
                                         int G, int L, float *__restrict__ Q, float *__restrict__ R,
                                         float *__restrict__ dfac, float *__restrict__ dbias, int ds, float *__restrict__ pw,
                                         const float *__restrict__ A, const double *__restrict__ xstats,
                                         const float *__restrict__ gate, const float *__restrict__ qse,
                                         float *__restrict__ Aout) {
  extern __shared__ double sh[]; // [2 C]: coef S1, coef T
  const int b = blockIdx.x, c = threadIdx.x, cpg = C / G;
  double s1 = 0.0, T = 0.0, m = 0.0, rs = 0.0, f = 1.0, w = 0.0;
  if (c < C) {
    s1 = S[((size_t)b * C + c) * 2];
    const double s2 = S[((size_t)b * C + c) * 2 + 1];
    m = mean[(size_t)b * C + c]; rs = rstd[(size_t)b * C + c];
    f = fac ? (double)fac[(size_t)b * fs + c] : 1.0;
    w = gw[c];
    T = rs * (s2 - m * s1); // sum da n
    sh[c] = w * f * s1;
    sh[C + c] = w * f * T;
  }
  __syncthreads();
  if (c >= C) return;
  const int g0 = (c / cpg) * cpg;
  double a1 = 0.0, a2 = 0.0;
  for (int k = 0; k < cpg; ++k) { a1 += sh[g0 + k]; a2 += sh[C + g0 + k]; }
  const double cnt = (double)cpg * (double)L, m1 = a1 / cnt, m2 = a2 / cnt;
  // gate / qse (an SE3d gate behind this AdaGN, folded into one op): upstream was du = g gy + qse, so dx = (A g) gy + (A qse + Q) + R x
  const double a_row = A ? (double)A[(size_t)b * C + c] : 0.0;
  Q[(size_t)b * C + c] = (float)(-rs * m1 + rs * rs * m * m2 + (gate ? a_row * (double)qse[(size_t)b * C + c] : 0.0));
  R[(size_t)b * C + c] = (float)(-rs * rs * m2);
  if (gate) Aout[(size_t)b * C + c] = (float)(a_row * (double)gate[(size_t)b * C + c]);
  if (dfac) dfac[(size_t)b * ds + c] = (float)(w * T + (double)gb[c] * s1);   // ds: row stride (the two may be the halves of one [B, 2C] buffer)
  if (dbias) dbias[(size_t)b * ds + c] = (float)s1;
  pw[((size_t)b * C + c) * 3] = (float)(f * T);
  pw[((size_t)b * C + c) * 3 + 1] = (float)(f * s1);
  // sum over the row of dx = A da + Q + R x (what lion_affine_act_bwd_apply writes): the bias gradient of the convolution that
  // produced x, per sample -- a [B, C] by-product here, a pass over the 268-MB gradient when taken from dx itself
  const double q = -rs * m1 + rs * rs * m * m2, r = -rs * rs * m2;
  pw[((size_t)b * C + c) * 3 + 2] =
      (A && xstats) ? (float)((double)A[(size_t)b * C + c] * s1 + (double)L * q + r * xstats[((size_t)b * C + c) * 2]) : 0.f;
}

} // namespace

// S[row] = {sum_p g[row][p] ws[b][p], sum_p g[row][p] v[row][p]}: the two row sums of the AdaGN(+SE) backward when the upstream
// gradient is a devoxelisation's scatter of g (ws = the 8 corner weights of point p summed, v = the devoxelised input): one wave
// per row of N points instead of a pass over the r^3 grid (lion_amd/train_ops.py::adagn_se_devox)
__global__ __launch_bounds__(256) void rows_dot2_kernel(const float *__restrict__ g, const float *__restrict__ v,
                                                        const float *__restrict__ ws, int rows, int C, int N,
                                                        float *__restrict__ S) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float *gr = g + (size_t)row * N, *vr = v + (size_t)row * N, *wr = ws + (size_t)(row / C) * N;
  float s1 = 0.f, s2 = 0.f;
  for (int p = lane; p < N; p += 64) {
    const float t = gr[p];
    s1 = fmaf(t, wr[p], s1);
    s2 = fmaf(t, vr[p], s2);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  if (lane == 0) { S[(size_t)row * 2] = s1; S[(size_t)row * 2 + 1] = s2; }
}

// ---- SE3d gate (reference models/pvcnn2_ada.py:27-41: x * sigmoid(W2 relu(W1 mean_voxels(x)))) -- the [B, C] algebra between the
// row-sum pass and the scaling pass, training form (round 6).  ATen ran it as ~7 tiny launches forward (div, mm, relu, mm,
// sigmoid, two fills) and ~13 backward per SE layer, 28 layers per VAE step.  One block per sample; C <= 1024, Cr <= 128.
// GN: the gate sits directly behind an AdaGN WITHOUT activation (the tail of every PVConv's voxel branch: Conv3d -> AdaGN -> SE3d,
// reference pvcnn2_ada.py:211-226) -- both are affine per (sample, channel), so u = A x + Bs never has to exist: its channel
// mean is A mean(x) + Bs (xstats = the row sums of x in double), and the gated result is (g A) x + (g Bs): A2 / B2 out.
template <bool GN>
__global__ __launch_bounds__(256) void se_gate_fwd_kernel(const float *__restrict__ stats, const float *__restrict__ w1,
                                                          const float *__restrict__ w2, int C, int Cr, float L,
                                                          float *__restrict__ mean, float *__restrict__ h,
                                                          float *__restrict__ g, float *__restrict__ zero,
                                                          const double *__restrict__ xstats, const float *__restrict__ A,
                                                          const float *__restrict__ Bs, float *__restrict__ A2,
                                                          float *__restrict__ B2) {
  __shared__ float m[1024], hh[128];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < C; c += 256) {
    const size_t row = (size_t)b * C + c;
    const float v = GN ? (float)((double)A[row] * (xstats[row * 2] / (double)L) + (double)Bs[row]) : stats[row * 2] / L;
    m[c] = v;
    mean[row] = v;
    if (!GN) zero[row] = 0.f;
  }
  __syncthreads();
  for (int j = wave; j < Cr; j += 4) {
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc = fmaf(m[c], w1[(size_t)j * C + c], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
      const float r = fmaxf(acc, 0.f);
      hh[j] = r;
      h[(size_t)b * Cr + j] = r;
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float acc = 0.f;
    for (int j = 0; j < Cr; ++j) acc = fmaf(hh[j], w2[(size_t)c * Cr + j], acc);
    const float gg = 1.f / (1.f + expf(-acc));
    g[(size_t)b * C + c] = gg;
    if (GN) {
      A2[(size_t)b * C + c] = gg * A[(size_t)b * C + c];
      B2[(size_t)b * C + c] = gg * Bs[(size_t)b * C + c];
    }
  }
}

// backward, per sample: dpre2 = dg g (1 - g) with dg = sum_voxels gy x (S[:, 1] of the reduction pass);
// dpre1 = (dpre2 W2) [h > 0];  Q = (dpre1 W1) / L (the gradient every voxel of a channel receives through the mean)
// GN (see se_gate_fwd_kernel): S = {sum gy, sum gy x} over the row; the gate's gradient is sum gy u = A S2 + Bs S1, and the row sums
// the AdaGN backward needs of du = g gy + Qse follow without another pass: S'1 = g S1 + L Qse, S'2 = g S2 + Qse sum x  (Sp out).
template <bool GN>
__global__ __launch_bounds__(256) void se_gate_bwd_kernel(const float *__restrict__ S, const float *__restrict__ g,
                                                          const float *__restrict__ h, const float *__restrict__ w1,
                                                          const float *__restrict__ w2, int C, int Cr, float L,
                                                          float *__restrict__ dpre2, float *__restrict__ dpre1,
                                                          float *__restrict__ Q, const double *__restrict__ xstats,
                                                          const float *__restrict__ A, const float *__restrict__ Bs,
                                                          float *__restrict__ Sp) {
  __shared__ float s2[1024], s1[128];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < C; c += 256) {
    const size_t row = (size_t)b * C + c;
    const float gg = g[row];
    const float dg = GN ? (float)((double)A[row] * (double)S[row * 2 + 1] + (double)Bs[row] * (double)S[row * 2]) : S[row * 2 + 1];
    const float d = dg * gg * (1.f - gg);
    s2[c] = d;
    dpre2[(size_t)b * C + c] = d;
  }
  __syncthreads();
  for (int j = wave; j < Cr; j += 4) {
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc = fmaf(s2[c], w2[(size_t)c * Cr + j], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
      const float r = h[(size_t)b * Cr + j] > 0.f ? acc : 0.f;
      s1[j] = r;
      dpre1[(size_t)b * Cr + j] = r;
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float acc = 0.f;
    for (int j = 0; j < Cr; ++j) acc = fmaf(s1[j], w1[(size_t)j * C + c], acc);
    const float q = acc / L;
    Q[(size_t)b * C + c] = q;
    if (GN) {
      const size_t row = (size_t)b * C + c;
      const double gg = (double)g[row];
      Sp[row * 2] = (float)(gg * (double)S[row * 2] + (double)L * (double)q);
      Sp[row * 2 + 1] = (float)(gg * (double)S[row * 2 + 1] + (double)q * xstats[row * 2]);
    }
  }
}

// weight gradients, summed over the batch in ascending order: dW2[c, j] = sum_b dpre2[b, c] h[b, j];  dW1[j, c] = sum_b dpre1[b, j] mean[b, c]
__global__ __launch_bounds__(256) void se_gate_wgrad_kernel(const float *__restrict__ dpre2, const float *__restrict__ dpre1,
                                                            const float *__restrict__ h, const float *__restrict__ mean, int B,
                                                            int C, int Cr, float *__restrict__ dw1, float *__restrict__ dw2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= C * Cr) return;
  const float *p, *q;   // out[i] = sum_b p[b * ps] * q[b * qs]
  int ps, qs;
  if (blockIdx.y == 0) {   // dW2 [C, Cr]
    const int c = i / Cr, j = i - c * Cr;
    p = dpre2 + c, ps = C, q = h + j, qs = Cr;
  } else {                 // dW1 [Cr, C]
    const int j = i / C, c = i - j * C;
    p = dpre1 + j, ps = Cr, q = mean + c, qs = C;
  }
  float acc = 0.f;
  for (int b0 = 0; b0 < B; b0 += 8) {   // 16 loads in flight, summed in ascending b
    float u[8], v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool in = b0 + k < B;
      u[k] = in ? p[(size_t)(b0 + k) * ps] : 0.f;
      v[k] = in ? q[(size_t)(b0 + k) * qs] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (b0 + k < B) acc = fmaf(u[k], v[k], acc);
  }
  (blockIdx.y == 0 ? dw2 : dw1)[i] = acc;
}

extern "C" {

int lion_gn_train_fold(const float *stats, const float *gw, const float *gb, const float *fac, int fac_stride,
                       const float *bias, int bias_stride, int B, int C, int G, int L, float eps, float *A, float *Bs,
                       float *mean, float *rstd, lionStream_t stream) {
  if (!stats || !gw || !gb || !A || !Bs || !mean || !rstd || B <= 0 || C <= 0 || G <= 0 || C % G || L <= 0) return LION_EINVAL;
  if (C > 1024) return LION_EUNSUPPORTED;
  const int T = (C + 63) / 64 * 64;
  gn_train_fold_kernel<float><<<B, T, (size_t)2 * C * sizeof(double), static_cast<hipStream_t>(stream)>>>(
      stats, gw, gb, fac, fac_stride, bias, bias_stride, C, G, L, eps, A, Bs, mean, rstd);
  LION_LAUNCH_CHECK();
  return 0;
}

// the training path's statistics: row sums carried out on shifted values and handed over in double (see
// row_stats64_kernel); lion_gn_train_fold64 = lion_gn_train_fold on those
int lion_row_stats64(const float *x, int rows, int L, double *stats, lionStream_t stream) {
  if (!x || !stats || rows <= 0 || L <= 0) return LION_EINVAL;
  row_stats64_kernel<<<rows, 256, 0, static_cast<hipStream_t>(stream)>>>(x, L, stats);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_gn_train_fold64(const double *stats, const float *gw, const float *gb, const float *fac, int fac_stride,
                         const float *bias, int bias_stride, int B, int C, int G, int L, float eps, float *A, float *Bs,
                         float *mean, float *rstd, lionStream_t stream) {
  if (!stats || !gw || !gb || !A || !Bs || !mean || !rstd || B <= 0 || C <= 0 || G <= 0 || C % G || L <= 0) return LION_EINVAL;
  if (C > 1024) return LION_EUNSUPPORTED;
  const int T = (C + 63) / 64 * 64;
  gn_train_fold_kernel<double><<<B, T, (size_t)2 * C * sizeof(double), static_cast<hipStream_t>(stream)>>>(
      stats, gw, gb, fac, fac_stride, bias, bias_stride, C, G, L, eps, A, Bs, mean, rstd);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_gn_train_bwd_fold(const float *S, const float *mean, const float *rstd, const float *gw, const float *gb,
                           const float *fac, int fac_stride, int B, int C, int G, int L, float *Q, float *R, float *dfac,
                           float *dbias, int d_stride, float *pw, const float *A, const double *xstats, const float *gate,
                           const float *qse, float *Aout, lionStream_t stream) {
  if (!S || !mean || !rstd || !gw || !gb || !Q || !R || !pw || B <= 0 || C <= 0 || G <= 0 || C % G || L <= 0) return LION_EINVAL;
  if ((dfac || dbias) && d_stride < C) return LION_EINVAL;
  if ((gate != nullptr) != (qse != nullptr) || (gate && (!A || !Aout))) return LION_EINVAL;
  if (C > 1024) return LION_EUNSUPPORTED;
  const int T = (C + 63) / 64 * 64;
  gn_train_bwd_fold_kernel<<<B, T, (size_t)2 * C * sizeof(double), static_cast<hipStream_t>(stream)>>>(
      S, mean, rstd, gw, gb, fac, fac_stride, C, G, L, Q, R, dfac, dbias, d_stride, pw, A, xstats, gate, qse, Aout);
  LION_LAUNCH_CHECK();
  return 0;
}


// keep in (0, 1]: thr = keep * 2^32 (saturated), scale = 1 / keep
static bool drop_args(const void *seed, float keep, unsigned *thr, float *scale) {
  if (!seed || !(keep > 0.f) || keep > 1.f) return false;
  const double t = (double)keep * 4294967296.0;
  *thr = t >= 4294967295.0 ? 0xffffffffu : (unsigned)t;
  *scale = 1.0f / keep;
  return true;
}

static int affine_act_launch(const float *x, const float *A, const float *Bs, int rows, int L, int act, float *y,
                             const unsigned long long *seed, unsigned thr, float scale, lionStream_t stream) {
  if (!x || !A || !Bs || !y || rows <= 0 || L <= 0 || (act != 0 && act != 1)) return LION_EINVAL;
  if ((long)rows * lion_cdiv(lion_cdiv(L, 4), 256) > 0x7fffffffL) return LION_EUNSUPPORTED;
  const unsigned grid = (unsigned)(rows * lion_cdiv(lion_cdiv(L, 4), 256));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (seed) affine_act_kernel<true><<<grid, 256, 0, st>>>(x, A, Bs, L, act, y, seed, thr, scale);
  else affine_act_kernel<false><<<grid, 256, 0, st>>>(x, A, Bs, L, act, y, nullptr, 0u, 1.f);
  LION_LAUNCH_CHECK();
  return 0;
}
static int affine_act_bwd_stats_launch(const float *x, const float *gy, const float *A, const float *Bs, int rows, int L, int act,
                                       float *S, const unsigned long long *seed, unsigned thr, float scale, lionStream_t stream) {
  if (!x || !gy || !A || !Bs || !S || rows <= 0 || L <= 0 || (act != 0 && act != 1)) return LION_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (seed) affine_act_bwd_stats_kernel<true><<<rows, 256, 0, st>>>(x, gy, A, Bs, L, act, S, seed, thr, scale);
  else affine_act_bwd_stats_kernel<false><<<rows, 256, 0, st>>>(x, gy, A, Bs, L, act, S, nullptr, 0u, 1.f);
  LION_LAUNCH_CHECK();
  return 0;
}
static int affine_act_bwd_apply_launch(const float *x, const float *gy, const float *A, const float *Bs, const float *Q,
                                       const float *R, int rows, int L, int act, float *dx, const unsigned long long *seed,
                                       unsigned thr, float scale, lionStream_t stream) {
  if (!x || !gy || !A || !Bs || !Q || !R || !dx || rows <= 0 || L <= 0 || (act != 0 && act != 1)) return LION_EINVAL;
  if ((long)rows * lion_cdiv(lion_cdiv(L, 4), 256) > 0x7fffffffL) return LION_EUNSUPPORTED;
  const unsigned grid = (unsigned)(rows * lion_cdiv(lion_cdiv(L, 4), 256));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (seed) affine_act_bwd_apply_kernel<true><<<grid, 256, 0, st>>>(x, gy, A, Bs, Q, R, L, act, dx, seed, thr, scale);
  else affine_act_bwd_apply_kernel<false><<<grid, 256, 0, st>>>(x, gy, A, Bs, Q, R, L, act, dx, nullptr, 0u, 1.f);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_affine_act(const float *x, const float *A, const float *Bs, int rows, int L, int act, float *y,
                    lionStream_t stream) {
  return affine_act_launch(x, A, Bs, rows, L, act, y, nullptr, 0u, 1.f, stream);
}
int lion_affine_act_bwd_stats(const float *x, const float *gy, const float *A, const float *Bs, int rows, int L, int act,
                              float *S, lionStream_t stream) {
  return affine_act_bwd_stats_launch(x, gy, A, Bs, rows, L, act, S, nullptr, 0u, 1.f, stream);
}
int lion_affine_act_bwd_apply(const float *x, const float *gy, const float *A, const float *Bs, const float *Q,
                              const float *R, int rows, int L, int act, float *dx, lionStream_t stream) {
  return affine_act_bwd_apply_launch(x, gy, A, Bs, Q, R, rows, L, act, dx, nullptr, 0u, 1.f, stream);
}
// the same three passes with inverted dropout (keep probability `keep`, mask from the device-resident 64-bit `seed`) behind the activation
int lion_affine_act_dropout(const float *x, const float *A, const float *Bs, int rows, int L, int act, const uint64_t *seed,
                            float keep, float *y, lionStream_t stream) {
  unsigned thr; float scale;
  if (!drop_args(seed, keep, &thr, &scale)) return LION_EINVAL;
  return affine_act_launch(x, A, Bs, rows, L, act, y, reinterpret_cast<const unsigned long long *>(seed), thr, scale, stream);
}
int lion_affine_act_dropout_bwd_stats(const float *x, const float *gy, const float *A, const float *Bs, int rows, int L, int act,
                                      const uint64_t *seed, float keep, float *S, lionStream_t stream) {
  unsigned thr; float scale;
  if (!drop_args(seed, keep, &thr, &scale)) return LION_EINVAL;
  return affine_act_bwd_stats_launch(x, gy, A, Bs, rows, L, act, S, reinterpret_cast<const unsigned long long *>(seed), thr, scale, stream);
}
int lion_affine_act_dropout_bwd_apply(const float *x, const float *gy, const float *A, const float *Bs, const float *Q,
                                      const float *R, int rows, int L, int act, const uint64_t *seed, float keep, float *dx,
                                      lionStream_t stream) {
  unsigned thr; float scale;
  if (!drop_args(seed, keep, &thr, &scale)) return LION_EINVAL;
  return affine_act_bwd_apply_launch(x, gy, A, Bs, Q, R, rows, L, act, dx, reinterpret_cast<const unsigned long long *>(seed), thr, scale, stream);
}

// U in {8, 16, 32, 64}; x f32[rows, M, U] 16-byte aligned
#define LION_AAM_DISPATCH(KERNEL, GRID, ...)                                                       \
  switch (U) {                                                                                     \
  case 8:  KERNEL<2><<<GRID, 256, 0, st>>>(__VA_ARGS__); break;                                    \
  case 16: KERNEL<4><<<GRID, 256, 0, st>>>(__VA_ARGS__); break;                                    \
  case 32: KERNEL<8><<<GRID, 256, 0, st>>>(__VA_ARGS__); break;                                    \
  case 64: KERNEL<16><<<GRID, 256, 0, st>>>(__VA_ARGS__); break;                                   \
  default: return LION_EUNSUPPORTED;                                                               \
  }
int lion_affine_act_max(const float *x, const float *A, const float *Bs, int rows, int M, int U, int act, float *y,
                        lionStream_t stream) {
  if (!x || !A || !Bs || !y || rows <= 0 || M <= 0 || (act != 0 && act != 1)) return LION_EINVAL;
  if ((((uintptr_t)x) & 15) != 0 || rows > 65535) return LION_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  LION_AAM_DISPATCH(affine_act_max_kernel, dim3(lion_cdiv(M, 256), rows), x, A, Bs, M, act, y)
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_affine_act_max_bwd_stats(const float *x, const float *gy, const float *A, const float *Bs, int rows, int M, int U,
                                  int act, float *S, lionStream_t stream) {
  if (!x || !gy || !A || !Bs || !S || rows <= 0 || M <= 0 || (act != 0 && act != 1)) return LION_EINVAL;
  if ((((uintptr_t)x) & 15) != 0) return LION_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  LION_AAM_DISPATCH(affine_act_max_bwd_stats_kernel, rows, x, gy, A, Bs, M, act, S)
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_affine_act_max_bwd_apply(const float *x, const float *gy, const float *A, const float *Bs, const float *Q,
                                  const float *R, int rows, int M, int U, int act, float *dx, lionStream_t stream) {
  if (!x || !gy || !A || !Bs || !Q || !R || !dx || rows <= 0 || M <= 0 || (act != 0 && act != 1)) return LION_EINVAL;
  if (((((uintptr_t)x) | ((uintptr_t)dx)) & 15) != 0 || rows > 65535) return LION_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  LION_AAM_DISPATCH(affine_act_max_bwd_apply_kernel, dim3(lion_cdiv(M, 256), rows), x, gy, A, Bs, Q, R, M, act, dx)
  LION_LAUNCH_CHECK();
  return 0;
}
#undef LION_AAM_DISPATCH

int lion_gn_train_param_grads(const float *pw, int B, int C, float *dgw, float *dgb, float *dxs, lionStream_t stream) {
  if (!pw || !dgw || !dgb || B <= 0 || C <= 0) return LION_EINVAL;
  pw_batch_sum_kernel<<<lion_cdiv(C, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(pw, B, C, dgw, dgb, dxs);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_rows_dot2(const float *g, const float *v, const float *ws, int B, int C, int N, float *S, lionStream_t stream) {
  if (!g || !v || !ws || !S || B <= 0 || C <= 0 || N <= 0) return LION_EINVAL;
  const int rows = B * C;
  rows_dot2_kernel<<<lion_cdiv(rows, 4), 256, 0, static_cast<hipStream_t>(stream)>>>(g, v, ws, rows, C, N, S);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_se_gate_fwd(const float *stats, const float *w1, const float *w2, int B, int C, int Cr, int L, float *mean, float *h,
                     float *g, float *zero, lionStream_t stream) {
  if (!stats || !w1 || !w2 || !mean || !h || !g || !zero || B <= 0 || C <= 0 || Cr <= 0 || L <= 0) return LION_EINVAL;
  if (C > 1024 || Cr > 128) return LION_EUNSUPPORTED;
  se_gate_fwd_kernel<false><<<B, 256, 0, static_cast<hipStream_t>(stream)>>>(stats, w1, w2, C, Cr, (float)L, mean, h, g, zero,
                                                                             nullptr, nullptr, nullptr, nullptr, nullptr);
  LION_LAUNCH_CHECK();
  return 0;
}
// the gate directly behind an AdaGN without activation (see se_gate_fwd_kernel<true>): xstats f64[B*C,2] (lion_row_stats64 of x),
// A / Bs f32[B,C] of the AdaGN fold -> mean (of u = A x + Bs), h, g, and A2 = g A, B2 = g Bs for ONE lion_affine_act pass over x
int lion_gn_se_gate_fwd(const double *xstats, const float *A, const float *Bs, const float *w1, const float *w2, int B, int C, int Cr,
                        int L, float *mean, float *h, float *g, float *A2, float *B2, lionStream_t stream) {
  if (!xstats || !A || !Bs || !w1 || !w2 || !mean || !h || !g || !A2 || !B2 || B <= 0 || C <= 0 || Cr <= 0 || L <= 0) return LION_EINVAL;
  if (C > 1024 || Cr > 128) return LION_EUNSUPPORTED;
  se_gate_fwd_kernel<true><<<B, 256, 0, static_cast<hipStream_t>(stream)>>>(nullptr, w1, w2, C, Cr, (float)L, mean, h, g, nullptr,
                                                                            xstats, A, Bs, A2, B2);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_se_gate_bwd(const float *S, const float *g, const float *h, const float *mean, const float *w1, const float *w2, int B,
                     int C, int Cr, int L, float *dpre2, float *dpre1, float *Q, float *dw1, float *dw2, lionStream_t stream) {
  if (!S || !g || !h || !mean || !w1 || !w2 || !dpre2 || !dpre1 || !Q || !dw1 || !dw2 || B <= 0 || C <= 0 || Cr <= 0 || L <= 0)
    return LION_EINVAL;
  if (C > 1024 || Cr > 128) return LION_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  se_gate_bwd_kernel<false><<<B, 256, 0, st>>>(S, g, h, w1, w2, C, Cr, (float)L, dpre2, dpre1, Q, nullptr, nullptr, nullptr, nullptr);
  LION_LAUNCH_CHECK();
  se_gate_wgrad_kernel<<<dim3(lion_cdiv(C * Cr, 256), 2), 256, 0, st>>>(dpre2, dpre1, h, mean, B, C, Cr, dw1, dw2);
  LION_LAUNCH_CHECK();
  return 0;
}
// backward of the same pair: S f32[B*C,2] = {sum gy, sum gy x} (lion_affine_act_bwd_stats with act = 0) -> Qse (the gate's
// contribution to du, per row), Sp f32[B*C,2] = the row sums {sum du, sum du x} lion_gn_train_bwd_fold needs, dw1 / dw2
int lion_gn_se_gate_bwd(const float *S, const double *xstats, const float *A, const float *Bs, const float *g, const float *h,
                        const float *mean, const float *w1, const float *w2, int B, int C, int Cr, int L, float *Sp, float *dpre2,
                        float *dpre1, float *Qse, float *dw1, float *dw2, lionStream_t stream) {
  if (!S || !xstats || !A || !Bs || !g || !h || !mean || !w1 || !w2 || !Sp || !dpre2 || !dpre1 || !Qse || !dw1 || !dw2 || B <= 0 ||
      C <= 0 || Cr <= 0 || L <= 0)
    return LION_EINVAL;
  if (C > 1024 || Cr > 128) return LION_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  se_gate_bwd_kernel<true><<<B, 256, 0, st>>>(S, g, h, w1, w2, C, Cr, (float)L, dpre2, dpre1, Qse, xstats, A, Bs, Sp);
  LION_LAUNCH_CHECK();
  se_gate_wgrad_kernel<<<dim3(lion_cdiv(C * Cr, 256), 2), 256, 0, st>>>(dpre2, dpre1, h, mean, B, C, Cr, dw1, dw2);
  LION_LAUNCH_CHECK();
  return 0;
}

} // extern "C"
