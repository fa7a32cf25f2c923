// devoxelize.hip -- trilinear devoxelization (K4 forward, K5 backward).
//
// Reference: third_party/pvcnn/functional/src/interpolate/trilinear_devox.cu:21-162,
//            trilinear_devox.cpp:18-95.
//
// Forward: grid = (point tiles, channel tiles, batch) instead of the reference's one block per
// cloud.  A lane owns one point: the 8 corner indices / weights are computed once in registers
// (same expressions, same left-to-right evaluation as the reference: bit-exact vs the oracle), then
// the lane walks a tile of channels; the 8 gathers of a channel hit an L2-resident [r^3] slab and
// the [C,N] output row is written coalesced.
//
// Backward: the reference issues 8*C global float atomics per point into a memset grid.  Here one
// workgroup owns one (batch, channel) slab, accumulates it in LDS with ds_add_f32 (16 KiB at r=16,
// 128 KiB at r=32) and writes the dense slab exactly once with coalesced 16-byte stores: no memset,
// no global atomics.  Summation order inside a voxel is not fixed (LDS atomics), so backward is
// compared with tolerance, like the reference itself (atomicAdd order is undefined there too).
#include "common.h"

namespace {

struct Corners {
  int ix[8];
  float w[8];
};

__device__ __forceinline__ Corners corners_of(float x, float y, float z, int r, int r2) {
  Corners k;
  const float xl = floorf(x), yl = floorf(y), zl = floorf(z);
  const float xd1 = sub_rn(x, xl), yd1 = sub_rn(y, yl), zd1 = sub_rn(z, zl);
  const float xd0 = sub_rn(1.0f, xd1), yd0 = sub_rn(1.0f, yd1), zd0 = sub_rn(1.0f, zd1);
  // trilinear_devox.cu:52-59 -- (a*b)*c, left to right
  k.w[0] = mul_rn(mul_rn(xd0, yd0), zd0);
  k.w[1] = mul_rn(mul_rn(xd0, yd0), zd1);
  k.w[2] = mul_rn(mul_rn(xd0, yd1), zd0);
  k.w[3] = mul_rn(mul_rn(xd0, yd1), zd1);
  k.w[4] = mul_rn(mul_rn(xd1, yd0), zd0);
  k.w[5] = mul_rn(mul_rn(xd1, yd0), zd1);
  k.w[6] = mul_rn(mul_rn(xd1, yd1), zd0);
  k.w[7] = mul_rn(mul_rn(xd1, yd1), zd1);
  const int xlo = (int)xl, ylo = (int)yl, zlo = (int)zl;
  const int xs = (xd1 > 0.0f) ? r2 : 0; // :64-66, (x_hi & r2)
  const int ys = (yd1 > 0.0f) ? r : 0;
  const int zs = (zd1 > 0.0f) ? 1 : 0;
  k.ix[0] = xlo * r2 + ylo * r + zlo;
  k.ix[1] = k.ix[0] + zs;
  k.ix[2] = k.ix[0] + ys;
  k.ix[3] = k.ix[2] + zs;
  k.ix[4] = k.ix[0] + xs;
  k.ix[5] = k.ix[4] + zs;
  k.ix[6] = k.ix[4] + ys;
  k.ix[7] = k.ix[6] + zs;
  return k;
}

// AFF: out = scale[b,c] * interp(feat) + shift[b,c] * sum(w)  ==  interp(scale*feat + shift): the
// per-(batch, channel) affine that AdaGN + the SE gate apply to the whole grid (pvcnn2_ada.py:219-226)
// commutes with the interpolation, so the two full-grid passes are replaced by two scalars here.
template <int CT, bool AFF>
__global__ __launch_bounds__(256) void devox_fwd_kernel(const float *__restrict__ coords,
                                                        const float *__restrict__ feat, int C,
                                                        int N, int r, int training,
                                                        float *__restrict__ out,
                                                        int32_t *__restrict__ inds,
                                                        float *__restrict__ wgts,
                                                        const float *__restrict__ scale,
                                                        const float *__restrict__ shift) {
  const int b = blockIdx.z, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int r2 = r * r, r3 = r2 * r;
  const float *co = coords + (size_t)b * 3 * N;
  Corners k = corners_of(co[i], co[i + N], co[i + 2 * N], r, r2);
  if (training && blockIdx.y == 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      wgts[((size_t)b * 8 + q) * N + i] = k.w[q];
      inds[((size_t)b * 8 + q) * N + i] = k.ix[q];
    }
  }
  // memory safety for out-of-contract coordinates (the reference would read out of bounds)
#pragma unroll
  for (int q = 0; q < 8; ++q) k.ix[q] = min(max(k.ix[q], 0), r3 - 1);
  const int c0 = blockIdx.y * CT, c1 = min(C, c0 + CT);
  const float *f = feat + ((size_t)b * C + c0) * r3;
  float *o = out + ((size_t)b * C + c0) * N + i;
  float wsum = 0.f;
  if (AFF) {
    wsum = k.w[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) wsum += k.w[q];
  }
#pragma unroll 4
  for (int c = c0; c < c1; ++c, f += r3, o += N) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = f[k.ix[q]];
    float acc = mul_rn(k.w[0], v[0]); // :96-103, left to right
#pragma unroll
    for (int q = 1; q < 8; ++q) acc = add_rn(acc, mul_rn(k.w[q], v[q]));
    if (AFF) acc = acc * scale[(size_t)b * C + c] + shift[(size_t)b * C + c] * wsum;
    *o = acc;
  }
}

// One workgroup per (b, c) slab; slab accumulated in LDS.
__global__ __launch_bounds__(1024) void devox_bwd_lds_kernel(const float *__restrict__ gy,
                                                             const int32_t *__restrict__ inds,
                                                             const float *__restrict__ wgts, int C,
                                                             int N, int r3,
                                                             float *__restrict__ gx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *slab = reinterpret_cast<float *>(smem);
  const int tid = threadIdx.x, nt = blockDim.x;
  const int c = blockIdx.x, b = blockIdx.y;
  for (int v = tid * 4; v < r3; v += nt * 4)
    *reinterpret_cast<float4 *>(slab + v) = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const float *g = gy + ((size_t)b * C + c) * N;
  const int32_t *id = inds + (size_t)b * 8 * N;
  const float *wg = wgts + (size_t)b * 8 * N;
  for (int i = tid; i < N; i += nt) {
    const float gv = g[i];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int ix = min(max(id[(size_t)q * N + i], 0), r3 - 1);
      atomicAdd(slab + ix, mul_rn(wg[(size_t)q * N + i], gv)); // ds_add_f32
    }
  }
  __syncthreads();
  float *o = gx + ((size_t)b * C + c) * r3;
  for (int v = tid * 4; v < r3; v += nt * 4)
    *reinterpret_cast<float4 *>(o + v) = *reinterpret_cast<const float4 *>(slab + v);
}

// Fallback for slabs that do not fit LDS (r^3 * 4 > 128 KiB or r^3 % 4 != 0).
__global__ void devox_bwd_atomic_kernel(const float *__restrict__ gy,
                                        const int32_t *__restrict__ inds,
                                        const float *__restrict__ wgts, int C, int N, int r3,
                                        int CT, float *__restrict__ gx) {
  const int b = blockIdx.z, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int ix[8];
  float w[8];
  for (int q = 0; q < 8; ++q) {
    ix[q] = min(max(inds[((size_t)b * 8 + q) * N + i], 0), r3 - 1);
    w[q] = wgts[((size_t)b * 8 + q) * N + i];
  }
  const int c0 = blockIdx.y * CT, c1 = min(C, c0 + CT);
  for (int c = c0; c < c1; ++c) {
    const float gv = gy[((size_t)b * C + c) * N + i];
    for (int q = 0; q < 8; ++q) atomicAdd(gx + ((size_t)b * C + c) * r3 + ix[q], mul_rn(w[q], gv));
  }
}

} // namespace

extern "C" {

static int devox_launch(const float *coords, const float *feat, int B, int C, int N, int r,
                        int training, float *out, int32_t *inds, float *wgts, const float *scale,
                        const float *shift, hipStream_t st) {
  const int pt = lion_cdiv(N, 256);
  // channel tile: keep >= ~2048 workgroups in flight, amortise the corner computation
  int ct = 16;
  while (ct > 2 && (long)B * pt * lion_cdiv(C, ct) < 2048) ct >>= 1;
  dim3 grid(pt, lion_cdiv(C, ct), B);
#define DEVOX_CASE(CT_)                                                                                   \
  if (scale) devox_fwd_kernel<CT_, true><<<grid, 256, 0, st>>>(coords, feat, C, N, r, training, out, inds, wgts, scale, shift); \
  else devox_fwd_kernel<CT_, false><<<grid, 256, 0, st>>>(coords, feat, C, N, r, training, out, inds, wgts, scale, shift)
  switch (ct) {
  case 16: DEVOX_CASE(16); break;
  case 8:  DEVOX_CASE(8); break;
  case 4:  DEVOX_CASE(4); break;
  default: DEVOX_CASE(2); break;
  }
#undef DEVOX_CASE
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_trilinear_devoxelize_forward(const float *coords, const float *feat, int B, int C, int N,
                                      int r, int training, float *out, int32_t *inds, float *wgts,
                                      lionStream_t stream) {
  if (!coords || !feat || !out || B <= 0 || C <= 0 || N <= 0 || r <= 0) return LION_EINVAL;
  if (training && (!inds || !wgts)) return LION_EINVAL;
  return devox_launch(coords, feat, B, C, N, r, training, out, inds, wgts, nullptr, nullptr,
                      static_cast<hipStream_t>(stream));
}

// devoxelize(scale[b,c] * feat + shift[b,c]) without materialising the scaled grid (inference path).
int lion_trilinear_devoxelize_affine_forward(const float *coords, const float *feat, const float *scale,
                                             const float *shift, int B, int C, int N, int r, float *out,
                                             lionStream_t stream) {
  if (!coords || !feat || !out || !scale || !shift || B <= 0 || C <= 0 || N <= 0 || r <= 0) return LION_EINVAL;
  return devox_launch(coords, feat, B, C, N, r, 0, out, nullptr, nullptr, scale, shift,
                      static_cast<hipStream_t>(stream));
}

int lion_trilinear_devoxelize_backward(const float *gy, const int32_t *inds, const float *wgts,
                                       int B, int C, int N, int r3, float *gx,
                                       lionStream_t stream) {
  if (!gy || !inds || !wgts || !gx || B <= 0 || C <= 0 || N <= 0 || r3 <= 0) return LION_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t lds = (size_t)r3 * 4;
  if (lds <= 128 * 1024 && (r3 % 4) == 0) {
    static size_t configured = 0;
    if (lds > configured) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&devox_bwd_lds_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      configured = lds;
    }
    const int nt = r3 >= 16384 ? 1024 : 256;
    devox_bwd_lds_kernel<<<dim3(C, B), nt, lds, st>>>(gy, inds, wgts, C, N, r3, gx);
    LION_LAUNCH_CHECK();
    return 0;
  }
  hipError_t e = hipMemsetAsync(gx, 0, (size_t)B * C * r3 * 4, st);
  if (e != hipSuccess) return (int)e;
  const int CT = 8;
  devox_bwd_atomic_kernel<<<dim3(lion_cdiv(N, 256), lion_cdiv(C, CT), B), 256, 0, st>>>(
      gy, inds, wgts, C, N, r3, CT, gx);
  LION_LAUNCH_CHECK();
  return 0;
}

} // extern "C"
