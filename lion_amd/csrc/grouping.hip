// grouping.hip -- K7 grouping (row gather) and K8 its gradient; K10 gather_features.
//
// Reference: third_party/pvcnn/functional/src/grouping/grouping.cu:18-36 / :58-77,
//            sampling/sampling.cu:17-31 / :52-66.
//
// Forward is a pure copy bounded by the [B,C,M,U] write (147 MB at SA-0): a lane owns 4 consecutive
// (m,u) slots, reads their indices once (one int4), then walks a channel tile issuing one 16-byte
// coalesced store per channel; the gathered feature rows (N floats) stay in L2.
// Backward: one workgroup owns a tile of (b, c) rows, accumulates them in LDS with ds_add_f32 and
// writes each row once -- no memset, no global atomics (the reference uses both).
#include "common.h"

namespace {

// out is a channel slice of a [B, Ctot, MU] tensor (Ctot = C for the plain operator); centers (optional, [B, C, M]):
// out = feat[idx] - centers[m] -- the neighbour coordinates relative to their centre (pvcnn2_ada.py:104-105), so that
// BallQuery.forward writes coordinates and features straight into ONE [B, 3 + C, M, U] tensor: no subtraction pass,
// no torch.cat of the 147 MB activation.  The subtraction is a single IEEE operation, as in the reference.
template <int CT>
__global__ __launch_bounds__(256) void grouping_fwd_kernel(const float *__restrict__ feat,
                                                           const int32_t *__restrict__ idx, int C,
                                                           int N, int MU, int Ctot, int U,
                                                           const float *__restrict__ centers,
                                                           float *__restrict__ out) {
  const int b = blockIdx.z;
  const int e = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (e >= MU) return;
  const int c0 = blockIdx.y * CT, c1 = min(C, c0 + CT);
  const int32_t *id = idx + (size_t)b * MU + e;
  const int M = MU / U;
  if (e + 3 < MU && (MU & 3) == 0) {
    const int4 t = *reinterpret_cast<const int4 *>(id);
    const int i0 = min(max(t.x, 0), N - 1), i1 = min(max(t.y, 0), N - 1),
              i2 = min(max(t.z, 0), N - 1), i3 = min(max(t.w, 0), N - 1);
    const float *f = feat + ((size_t)b * C + c0) * N;
    float *o = out + ((size_t)b * Ctot + c0) * MU + e;
    if (centers) {
      const int m0 = e / U, m1 = (e + 1) / U, m2 = (e + 2) / U, m3 = (e + 3) / U;
      const float *ct = centers + ((size_t)b * C + c0) * M;
      for (int c = c0; c < c1; ++c, f += N, o += MU, ct += M)
        *reinterpret_cast<float4 *>(o) = make_float4(sub_rn(f[i0], ct[m0]), sub_rn(f[i1], ct[m1]),
                                                     sub_rn(f[i2], ct[m2]), sub_rn(f[i3], ct[m3]));
    } else {
#pragma unroll 4
      for (int c = c0; c < c1; ++c, f += N, o += MU)
        *reinterpret_cast<float4 *>(o) = make_float4(f[i0], f[i1], f[i2], f[i3]);
    }
  } else {
    for (int q = 0; q < 4 && e + q < MU; ++q) {
      const int i = min(max(id[q], 0), N - 1);
      for (int c = c0; c < c1; ++c) {
        float v = feat[((size_t)b * C + c) * N + i];
        if (centers) v = sub_rn(v, centers[((size_t)b * C + c) * M + (e + q) / U]);
        out[((size_t)b * Ctot + c) * MU + e + q] = v;
      }
    }
  }
}

// rows[CT][N] in LDS; gy [B,C,MU] -> gx [B,C,N]
__global__ __launch_bounds__(512) void scatter_rows_lds_kernel(const float *__restrict__ gy,
                                                               const int32_t *__restrict__ idx,
                                                               int C, int N, int MU, int CT,
                                                               float *__restrict__ gx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *rows = reinterpret_cast<float *>(smem);
  const int tid = threadIdx.x, nt = blockDim.x, b = blockIdx.y;
  const int c0 = blockIdx.x * CT, nc = min(C, c0 + CT) - c0;
  for (int v = tid; v < nc * N; v += nt) rows[v] = 0.f;
  __syncthreads();
  const int32_t *id = idx + (size_t)b * MU;
  for (int e = tid; e < MU; e += nt) {
    const int i = min(max(id[e], 0), N - 1);
    for (int c = 0; c < nc; ++c) atomicAdd(rows + c * N + i, gy[((size_t)b * C + c0 + c) * MU + e]);
  }
  __syncthreads();
  for (int v = tid; v < nc * N; v += nt) gx[((size_t)b * C + c0) * N + v] = rows[v];
}

__global__ void scatter_rows_atomic_kernel(const float *__restrict__ gy,
                                           const int32_t *__restrict__ idx, int C, int N, int MU,
                                           float *__restrict__ gx) {
  const int b = blockIdx.z, c = blockIdx.y, e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= MU) return;
  const int i = min(max(idx[(size_t)b * MU + e], 0), N - 1);
  atomicAdd(gx + ((size_t)b * C + c) * N + i, gy[((size_t)b * C + c) * MU + e]);
}

static int scatter_rows(const float *gy, const int32_t *idx, int B, int C, int N, int MU, float *gx,
                        hipStream_t st) {
  // LDS rows if a row tile fits comfortably; otherwise memset + global atomics.
  if ((size_t)N * 4 <= 64 * 1024) {
    int CT = (int)((64 * 1024) / ((size_t)N * 4));
    if (CT > 8) CT = 8;
    if (CT > C) CT = C;
    while (CT > 1 && (long)B * lion_cdiv(C, CT) < 1024) CT >>= 1;
    const size_t lds = (size_t)CT * N * 4;
    scatter_rows_lds_kernel<<<dim3(lion_cdiv(C, CT), B), 512, lds, st>>>(gy, idx, C, N, MU, CT, gx);
    LION_LAUNCH_CHECK();
    return 0;
  }
  hipError_t e = hipMemsetAsync(gx, 0, (size_t)B * C * N * 4, st);
  if (e != hipSuccess) return (int)e;
  scatter_rows_atomic_kernel<<<dim3(lion_cdiv(MU, 256), C, B), 256, 0, st>>>(gy, idx, C, N, MU, gx);
  LION_LAUNCH_CHECK();
  return 0;
}

} // namespace

extern "C" {

static int grouping_launch(const float *feat, const int32_t *idx, int B, int C, int N, int M, int U, int Ctot,
                           const float *centers, float *out, lionStream_t stream);

int lion_grouping_forward(const float *feat, const int32_t *idx, int B, int C, int N, int M, int U,
                          float *out, lionStream_t stream) {
  if (!feat || !idx || !out || B <= 0 || C <= 0 || N <= 0 || M <= 0 || U <= 0) return LION_EINVAL;
  return grouping_launch(feat, idx, B, C, N, M, U, C, nullptr, out, stream);
}

// BallQuery.forward (pvcnn2_ada.py:98-114) in two launches and no concatenation: out f32[B, 3 + C, M, U] with
// channels 0..2 = coords[b, :, idx] - centers[b, :, m] and 3.. = feat[b, :, idx] (feat may be NULL with C = 0).
int lion_group_points_forward(const float *coords, const float *centers, const float *feat, const int32_t *idx,
                              int B, int C, int N, int M, int U, float *out, lionStream_t stream) {
  if (!coords || !centers || !idx || !out || B <= 0 || C < 0 || N <= 0 || M <= 0 || U <= 0) return LION_EINVAL;
  if (C > 0 && !feat) return LION_EINVAL;
  if (int e = grouping_launch(coords, idx, B, 3, N, M, U, 3 + C, centers, out, stream)) return e;
  if (C > 0) return grouping_launch(feat, idx, B, C, N, M, U, 3 + C, nullptr, out + (size_t)3 * M * U, stream);
  return 0;
}

static int grouping_launch(const float *feat, const int32_t *idx, int B, int C, int N, int M, int U, int Ctot,
                           const float *centers, float *out, lionStream_t stream) {
  const int MU = M * U;
  const int et = lion_cdiv(lion_cdiv(MU, 4), 256);
  int ct = 16;
  while (ct > 1 && (long)B * et * lion_cdiv(C, ct) < 2048) ct >>= 1;
  dim3 grid(et, lion_cdiv(C, ct), B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (ct) {
  case 16: grouping_fwd_kernel<16><<<grid, 256, 0, st>>>(feat, idx, C, N, MU, Ctot, U, centers, out); break;
  case 8:  grouping_fwd_kernel<8><<<grid, 256, 0, st>>>(feat, idx, C, N, MU, Ctot, U, centers, out); break;
  case 4:  grouping_fwd_kernel<4><<<grid, 256, 0, st>>>(feat, idx, C, N, MU, Ctot, U, centers, out); break;
  case 2:  grouping_fwd_kernel<2><<<grid, 256, 0, st>>>(feat, idx, C, N, MU, Ctot, U, centers, out); break;
  default: grouping_fwd_kernel<1><<<grid, 256, 0, st>>>(feat, idx, C, N, MU, Ctot, U, centers, out); break;
  }
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_grouping_backward(const float *gy, const int32_t *idx, int B, int C, int N, int M, int U,
                           float *gx, lionStream_t stream) {
  if (!gy || !idx || !gx || B <= 0 || C <= 0 || N <= 0 || M <= 0 || U <= 0) return LION_EINVAL;
  return scatter_rows(gy, idx, B, C, N, M * U, gx, static_cast<hipStream_t>(stream));
}

// K10 is grouping with U == 1: out[b,c,j] = feat[b,c,idx[b,j]].
int lion_gather_features_forward(const float *feat, const int32_t *idx, int B, int C, int N, int M,
                                 float *out, lionStream_t stream) {
  return lion_grouping_forward(feat, idx, B, C, N, M, 1, out, stream);
}

int lion_gather_features_backward(const float *gy, const int32_t *idx, int B, int C, int N, int M,
                                  float *gx, lionStream_t stream) {
  if (!gy || !idx || !gx || B <= 0 || C <= 0 || N <= 0 || M <= 0) return LION_EINVAL;
  return scatter_rows(gy, idx, B, C, N, M, gx, static_cast<hipStream_t>(stream));
}

} // extern "C"
