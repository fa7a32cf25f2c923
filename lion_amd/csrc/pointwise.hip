// pointwise.hip -- P2/P3/P6 for the point branch and the SA/FP SharedMLPs (inference):
// AdaGN + Swish (+ the max over the neighbourhood) after a 1x1 convolution in two passes over the
// activations instead of the ~8 ATen launches of GroupNorm, *factor, +bias, sigmoid, mul, max
// (models/pvcnn2_ada.py:120-164, :375-377; models/adagn.py:45-65).
//   row_stats_kernel        [B,C,L] -> per (b,c) sum and sum of squares (then lion_groupnorm_fold)
//   affine_swish_kernel     y = swish(x * A[b,c] + Bs[b,c])
//   affine_swish_max_kernel [B,C,M,U] -> [B,C,M]: max_u swish(x * A + Bs)   (P6 fused)
// All HBM bound: one read (+ one write) of the tensor per kernel, 16-byte lanes.
#include "common.h"

namespace {

__device__ __forceinline__ float swishf(float t) { return swish_fast(t); } // as in the conv prologue

__global__ __launch_bounds__(256) void row_stats_kernel(const float *__restrict__ x, int L,
                                                        float *__restrict__ stats) {
  __shared__ float r1[4], r2[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *p = x + (size_t)row * L;
  float s1 = 0.f, s2 = 0.f;
  if ((L & 3) == 0) {
    for (int i = tid * 4; i < L; i += 1024) {
      const float4 v = *reinterpret_cast<const float4 *>(p + i);
      s1 += (v.x + v.y) + (v.z + v.w);
      s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
  } else {
    for (int i = tid; i < L; i += 256) { const float v = p[i]; s1 += v; s2 += v * v; }
  }
  s1 = row16_sum_rn(s1); s2 = row16_sum_rn(s2);
#pragma unroll
  for (int m = 16; m < 64; m <<= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
  if (lane == 0) { r1[wave] = s1; r2[wave] = s2; }
  __syncthreads();
  if (tid == 0) {
    stats[(size_t)row * 2] = (r1[0] + r1[1]) + (r1[2] + r1[3]);
    stats[(size_t)row * 2 + 1] = (r2[0] + r2[1]) + (r2[2] + r2[3]);
  }
}

__global__ __launch_bounds__(256) void affine_swish_kernel(const float *__restrict__ x,
                                                           const float *__restrict__ A,
                                                           const float *__restrict__ Bs, int L,
                                                           float *__restrict__ y) {
  const int row = blockIdx.y;
  const float a = A[row], b = Bs[row];
  const float *p = x + (size_t)row * L;
  float *q = y + (size_t)row * L;
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < L && (L & 3) == 0) {
    const float4 v = *reinterpret_cast<const float4 *>(p + i);
    *reinterpret_cast<float4 *>(q + i) =
        make_float4(swishf(v.x * a + b), swishf(v.y * a + b), swishf(v.z * a + b), swishf(v.w * a + b));
  } else {
    for (int j = i; j < L && j < i + 4; ++j) q[j] = swishf(p[j] * a + b);
  }
}

// y = swish(x * A + Bs) + addend: the last activation pass of a PVConv's point branch with the residual sum behind it
// (fused_features = voxel_features + point_features, models/pvcnn2_ada.py:276-278) -- one launch instead of two
__global__ __launch_bounds__(256) void affine_swish_add_kernel(const float *__restrict__ x, const float *__restrict__ A,
                                                               const float *__restrict__ Bs,
                                                               const float *__restrict__ addend, int L,
                                                               float *__restrict__ y) {
  const int row = blockIdx.y;
  const float a = A[row], b = Bs[row];
  const float *p = x + (size_t)row * L, *ad = addend + (size_t)row * L;
  float *q = y + (size_t)row * L;
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < L && (L & 3) == 0) {
    const float4 v = *reinterpret_cast<const float4 *>(p + i), w = *reinterpret_cast<const float4 *>(ad + i);
    *reinterpret_cast<float4 *>(q + i) = make_float4(swishf(v.x * a + b) + w.x, swishf(v.y * a + b) + w.y,
                                                     swishf(v.z * a + b) + w.z, swishf(v.w * a + b) + w.w);
  } else {
    for (int j = i; j < L && j < i + 4; ++j) q[j] = swishf(p[j] * a + b) + ad[j];
  }
}

// sinusoidal timestep embedding (models/latent_points_ada.py get_timestep_embedding, reference :101-115):
// emb[b][i] = sin((t_b * scale) * row_i), emb[b][half + i] = cos(same), one launch instead of mul, mul, sin, cos, cat
__global__ void timestep_embedding_kernel(const float *__restrict__ t, const float *__restrict__ row, float scale, int B,
                                          int half, int D, float *__restrict__ emb) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * D) return;
  const int b = e / D, i = e - b * D;
  float v = 0.f; // odd D: the last column is zero padding
  if (i < 2 * half) {
    const float ang = mul_rn(mul_rn(t[b], scale), row[i < half ? i : i - half]);
    v = i < half ? sinf(ang) : cosf(ang);
  }
  emb[e] = v;
}

// one lane per centre m: U consecutive floats (U % 4 == 0), max of the activated values
__global__ __launch_bounds__(256) void affine_swish_max_kernel(const float *__restrict__ x,
                                                               const float *__restrict__ A,
                                                               const float *__restrict__ Bs, int M,
                                                               int U, float *__restrict__ y) {
  const int row = blockIdx.y, m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const float a = A[row], b = Bs[row];
  const float *p = x + ((size_t)row * M + m) * U;
  float best = -INFINITY;
  if ((U & 3) == 0) {
    for (int u = 0; u < U; u += 4) {
      const float4 v = *reinterpret_cast<const float4 *>(p + u);
      best = fmaxf(best, fmaxf(fmaxf(swishf(v.x * a + b), swishf(v.y * a + b)),
                               fmaxf(swishf(v.z * a + b), swishf(v.w * a + b))));
    }
  } else {
    for (int u = 0; u < U; ++u) best = fmaxf(best, swishf(p[u] * a + b));
  }
  y[(size_t)row * M + m] = best;
}

// U/4 lanes per centre (U in {8,16,32,64,...,256} -> 2..64 lanes): a wave reads 1 KiB of CONTIGUOUS
// floats per instruction (the one-lane-per-centre layout above makes every lane walk its own 128-byte
// line), the max is finished with xor-shuffles inside the lane group.
template <int LPM>
__global__ __launch_bounds__(256) void affine_swish_max_coop_kernel(const float *__restrict__ x,
                                                                    const float *__restrict__ A,
                                                                    const float *__restrict__ Bs, int M,
                                                                    float *__restrict__ y) {
  constexpr int MPW = 64 / LPM; // centres per wave instruction
  constexpr int IT = 8;         // wave instructions per thread -> 8 KiB in flight per wave
  const int row = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float a = A[row], b = Bs[row];
  const int m0 = (blockIdx.x * 4 + wave) * MPW * IT;
  const float4 *p = reinterpret_cast<const float4 *>(x + (size_t)row * M * (LPM * 4));
  float4 v[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int m = min(m0 + i * MPW + lane / LPM, M - 1);
    v[i] = p[(size_t)m * LPM + (lane % LPM)];
  }
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    float best = fmaxf(fmaxf(swishf(v[i].x * a + b), swishf(v[i].y * a + b)),
                       fmaxf(swishf(v[i].z * a + b), swishf(v[i].w * a + b)));
    // max over the LPM lanes of a centre: the same butterfly on DPP row operations (see common.h)
    if (LPM >= 2) best = fmaxf(best, LION_DPP_F32(best, 0xB1));
    if (LPM >= 4) best = fmaxf(best, LION_DPP_F32(best, 0x4E));
    if (LPM >= 8) best = fmaxf(best, LION_DPP_F32(best, 0x141));
    if (LPM >= 16) best = fmaxf(best, LION_DPP_F32(best, 0x140));
    const int m = m0 + i * MPW + lane / LPM;
    if ((lane % LPM) == 0 && m < M) y[(size_t)row * M + m] = best;
  }
}

// SE3d gate folded into the AdaGN scalars (pvcnn2_ada.py:27-41 + :219-226): the mean over the grid of
// AdaGN(y) is A*mean(y) + Bs per (batch, channel), so the gate needs no pass over the grid:
//   h = relu(W1 (A*m + Bs)),  g = sigmoid(W2 h),  A <- A*g,  Bs <- Bs*g.
// One workgroup per batch element (C <= 1024, hidden <= 128); 8 tiny launches (2 GEMMs, 6 elementwise)
// become one.
__global__ __launch_bounds__(256) void se_gate_kernel(const float *__restrict__ chmean,
                                                      const float *__restrict__ w1,
                                                      const float *__restrict__ w2, int C, int H,
                                                      float *__restrict__ A, float *__restrict__ Bs) {
  __shared__ float sm[1024], sh[128];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < C; c += 256) sm[c] = A[(size_t)b * C + c] * chmean[(size_t)b * C + c] + Bs[(size_t)b * C + c];
  __syncthreads();
  for (int j = wave; j < H; j += 4) { // one wave per hidden unit: coalesced row of W1
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc += w1[(size_t)j * C + c] * sm[c];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) sh[j] = acc > 0.f ? acc : 0.f;
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float acc = 0.f;
    for (int j = 0; j < H; ++j) acc += w2[(size_t)c * H + j] * sh[j];
    const float g = 1.0f / (1.0f + expf(-acc));
    A[(size_t)b * C + c] *= g;
    Bs[(size_t)b * C + c] *= g;
  }
}


// x f32[B][N][D] (point-major latent, models/latent_points_ada_localprior.py:72-84) -> the channel-major tensors the
// denoiser's forward takes apart with permute / slice / contiguous (three ATen copies per step): all [B][D][N], coords
// [B][3][N] = rows 0-2, rest [B][D-3][N]; any of them may be NULL.  One workgroup stages 256 points x D through registers:
// reads coalesced over (n, d), writes coalesced over n.
__global__ __launch_bounds__(256) void latent_unpack_kernel(const float *__restrict__ x, int N, int D,
                                                            float *__restrict__ all, float *__restrict__ coords,
                                                            float *__restrict__ rest) {
  __shared__ float tile[256 * 9];
  const int b = blockIdx.y, n0 = blockIdx.x * 256, tid = threadIdx.x;
  const int np = min(256, N - n0);
  const float *src = x + ((size_t)b * N + n0) * D;
  for (int i = tid; i < np * D; i += 256) tile[(i / D) * (D + 1) + i % D] = src[i];
  __syncthreads();
  if (tid >= np) return;
  for (int d = 0; d < D; ++d) {
    const float v = tile[tid * (D + 1) + d];
    if (all) all[((size_t)b * D + d) * N + n0 + tid] = v;
    if (coords && d < 3) coords[((size_t)b * 3 + d) * N + n0 + tid] = v;
    if (rest && d >= 3) rest[((size_t)b * (D - 3) + d - 3) * N + n0 + tid] = v;
  }
}

// out f32[B][Ca + Ct][N]: rows [0, Ca) = a f32[B][Ca][N], rows [Ca, Ca + Ct) = t[b * ld_t + c] broadcast along N
// (torch.cat([features, temb.expand(..., N)], dim=1), models/latent_points_ada.py forward)
__global__ __launch_bounds__(256) void concat_broadcast_kernel(const float *__restrict__ a, const float *__restrict__ t,
                                                               int Ca, int Ct, int N4, int ld_t, float4 *__restrict__ out) {
  const int row = blockIdx.y, C = Ca + Ct, b = row / C, c = row % C;
  float4 *o = out + (size_t)row * N4;
  if (c < Ca) {
    const float4 *s = reinterpret_cast<const float4 *>(a) + ((size_t)b * Ca + c) * N4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N4; i += gridDim.x * 256) o[i] = s[i];
  } else {
    const float v = t[(size_t)b * ld_t + c - Ca];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N4; i += gridDim.x * 256) o[i] = make_float4(v, v, v, v);
  }
}

} // namespace

extern "C" {

// x f32[rows, L] (rows = B*C) -> stats f32[rows, 2] (== the [B,C,T=1,2] layout of lion_groupnorm_fold)
int lion_row_stats(const float *x, int rows, int L, float *stats, lionStream_t stream) {
  if (!x || !stats || rows <= 0 || L <= 0) return LION_EINVAL;
  row_stats_kernel<<<rows, 256, 0, static_cast<hipStream_t>(stream)>>>(x, L, stats);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_affine_swish(const float *x, const float *A, const float *Bs, int rows, int L, float *y,
                      lionStream_t stream) {
  if (!x || !A || !Bs || !y || rows <= 0 || L <= 0) return LION_EINVAL;
  affine_swish_kernel<<<dim3(lion_cdiv(lion_cdiv(L, 4), 256), rows), 256, 0, static_cast<hipStream_t>(stream)>>>(
      x, A, Bs, L, y);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_timestep_embedding(const float *t, const float *row, float scale, int B, int half, int D, float *emb,
                            lionStream_t stream) {
  if (!t || !row || !emb || B <= 0 || half <= 0 || D < 2 * half) return LION_EINVAL;
  timestep_embedding_kernel<<<lion_cdiv(B * D, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(t, row, scale, B, half, D, emb);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_affine_swish_add(const float *x, const float *A, const float *Bs, const float *addend, int rows, int L, float *y,
                          lionStream_t stream) {
  if (!x || !A || !Bs || !addend || !y || rows <= 0 || L <= 0) return LION_EINVAL;
  affine_swish_add_kernel<<<dim3(lion_cdiv(lion_cdiv(L, 4), 256), rows), 256, 0, static_cast<hipStream_t>(stream)>>>(
      x, A, Bs, addend, L, y);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_affine_swish_max(const float *x, const float *A, const float *Bs, int rows, int M, int U,
                          float *y, lionStream_t stream) {
  if (!x || !A || !Bs || !y || rows <= 0 || M <= 0 || U <= 0) return LION_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool al = (((uintptr_t)x) & 15) == 0;
#define LION_ASM_COOP(LPM_)                                                                                  \
  affine_swish_max_coop_kernel<LPM_><<<dim3(lion_cdiv(M, 4 * (64 / LPM_) * 8), rows), 256, 0, st>>>(x, A, Bs, M, y)
  if (al && U == 32) LION_ASM_COOP(8);
  else if (al && U == 16) LION_ASM_COOP(4);
  else if (al && U == 64) LION_ASM_COOP(16);
  else affine_swish_max_kernel<<<dim3(lion_cdiv(M, 256), rows), 256, 0, st>>>(x, A, Bs, M, U, y);
#undef LION_ASM_COOP
  LION_LAUNCH_CHECK();
  return 0;
}

// A, Bs f32[B,C] (in place) *= sigmoid(W2 relu(W1 (A*chmean + Bs))); w1 f32[H,C], w2 f32[C,H] (nn.Linear layouts)
int lion_se_gate(const float *chmean, const float *w1, const float *w2, int B, int C, int H, float *A,
                 float *Bs, lionStream_t stream) {
  if (!chmean || !w1 || !w2 || !A || !Bs || B <= 0 || C <= 0 || H <= 0) return LION_EINVAL;
  if (C > 1024 || H > 128) return LION_EUNSUPPORTED;
  se_gate_kernel<<<B, 256, 0, static_cast<hipStream_t>(stream)>>>(chmean, w1, w2, C, H, A, Bs);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_latent_unpack(const float *x, int B, int N, int D, float *all, float *coords, float *rest, lionStream_t stream) {
  if (!x || B <= 0 || N <= 0 || D < 3 || D > 8 || (!all && !coords && !rest)) return LION_EINVAL;
  if (rest && D == 3) return LION_EINVAL;
  latent_unpack_kernel<<<dim3(lion_cdiv(N, 256), B), 256, 0, static_cast<hipStream_t>(stream)>>>(x, N, D, all, coords, rest);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_concat_broadcast(const float *a, const float *t, int B, int Ca, int Ct, int N, int ld_t, float *out,
                          lionStream_t stream) {
  if (!a || !t || !out || B <= 0 || Ca <= 0 || Ct <= 0 || N <= 0 || ld_t < 0) return LION_EINVAL;
  if (N % 4 != 0 || ((((uintptr_t)a) | ((uintptr_t)out)) & 15) != 0) return LION_EUNSUPPORTED;
  if ((long)B * (Ca + Ct) > 65535) return LION_EUNSUPPORTED;   // one grid row per (sample, channel)
  const int N4 = N / 4;
  concat_broadcast_kernel<<<dim3(lion_cdiv(N4, 256) > 4 ? 4 : lion_cdiv(N4, 256), B * (Ca + Ct)), 256, 0,
                            static_cast<hipStream_t>(stream)>>>(a, t, Ca, Ct, N4, ld_t, reinterpret_cast<float4 *>(out));
  LION_LAUNCH_CHECK();
  return 0;
}

} // extern "C"
