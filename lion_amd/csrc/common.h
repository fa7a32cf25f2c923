// common.h -- shared device helpers for the gfx950 kernels (wave64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "lion_hip.h"

#define LION_WAVE 64

#define LION_LAUNCH_CHECK()                         \
  do {                                              \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return (int)e__;         \
  } while (0)

static inline int lion_cdiv(int a, int b) { return (a + b - 1) / b; }

// (d, h) extent of the spatial tiles of a SPARSE (work-queue) voxel convolution at resolution r -- the geometry the per-tile
// occupancy flags (lion_conv3d_tile_occupancy*) are laid out in.  Defined next to the convolution's own tile choice
// (csrc/conv3d.hip: conv_plan / conv_tile_dims) so that every other reader of the flags (lion_voxel_scatter_read) follows
// a change of that choice instead of repeating it; library-internal (hidden), not part of the C ABI.
__attribute__((visibility("hidden"))) int lion_internal_sparse_tile_dims(int r, int *td, int *th);

// Dynamic-LDS limit of a kernel.  HIP keeps the attribute per device, so what has already been configured is
// remembered per device (one slot per kernel instantiation at its launch site); the attribute call itself is
// idempotent and the slot only grows, which makes concurrent first calls from several host threads harmless.
#define LION_MAX_DEVICES 32
struct LionLdsLimit {
  size_t bytes[LION_MAX_DEVICES];
};
static inline int lion_current_device(int *dev) {
  if (hipGetDevice(dev) != hipSuccess || *dev < 0 || *dev >= LION_MAX_DEVICES) return LION_EINVAL;
  return 0;
}
template <typename K>
static inline int lion_dynamic_lds(K kernel, size_t bytes, LionLdsLimit &limit) {
  int dev = 0;
  if (int e = lion_current_device(&dev)) return e;
  if (bytes > limit.bytes[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    limit.bytes[dev] = bytes;
  }
  return 0;
}

// Parity-critical float arithmetic: one IEEE rounding per operation, never contracted into
// an FMA, exactly like the -ffp-contract=off oracle.  (The whole library is also compiled with
// -ffp-contract=off; the intrinsics make the intent explicit at the call sites that matter.)
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float div_rn(float a, float b) { return __fdiv_rn(a, b); }
// NOTE: __fsqrt_rn() lowers to a bare v_sqrt_f32 (1 ulp) on gfx950; sqrtf() gets the correctly
// rounded refinement sequence under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt.
__device__ __forceinline__ float sqrt_rn(float a) { return sqrtf(a); }
// swish(t) = t * sigmoid(t) on the hardware's v_exp_f32 + v_rcp_f32 (1 ulp each).  ONE definition for every kernel that
// applies it (conv / 1x1 prologues, affine_swish passes, the constant-response table): the delta decomposition of
// csrc/conv3d*.hip needs the constant and the staged activation to come from the same arithmetic.  (Until round 3 this
// was written __frcp_rn(1 + __expf(-t)), which compiles to the 10-instruction IEEE division sequence: the activation cost
// 22 VALU issues per value, half the matrix time of a prologue convolution.)
// Philox4x32-10 (Salmon et al. 2011): counter (c0..c3), key (k0, k1) -> four 32-bit words
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float swish_fast(float t) { return t * __builtin_amdgcn_rcpf(1.0f + __expf(-t)); }

// (a-b)^2 + (c-d)^2 + (e-f)^2 evaluated left to right without contraction.
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by,
                                         float bz) {
  const float dx = sub_rn(ax, bx), dy = sub_rn(ay, by), dz = sub_rn(az, bz);
  return add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
}

// Inclusive wave prefix sum (wave64) with shuffles.
// ---- butterfly steps on DPP row operations --------------------------------------------------------------------
// A commutative xor-butterfly reduction over 16 consecutive lanes (x ^= 1, 2, 4, 8) without the LDS crossbar that
// __shfl_xor / ds_bpermute go through: quad_perm does the true xor-1 / xor-2 exchanges; after them the 4 lanes of a
// quad hold the same value, so pairing lane i with lane 7-i (row_half_mirror) combines the same two quad results as
// xor-4 would -- and likewise row_mirror for xor-8.  Bit-identical to the butterfly for any commutative op.
#define LION_DPP_F32(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true))
__device__ __forceinline__ float row16_sum_rn(float v) { // exact same tree as v += shfl_xor(v, 1|2|4|8) with add_rn
  v = __fadd_rn(v, LION_DPP_F32(v, 0xB1));  // quad_perm [1,0,3,2]
  v = __fadd_rn(v, LION_DPP_F32(v, 0x4E));  // quad_perm [2,3,0,1]
  v = __fadd_rn(v, LION_DPP_F32(v, 0x141)); // row_half_mirror
  v = __fadd_rn(v, LION_DPP_F32(v, 0x140)); // row_mirror
  return v;
}
// max over the 64 lanes of a wave, valid in LANE 63 only: DPP butterflies inside the 16-lane rows, then the classic
// row_bcast:15 (lane 15 of a row -> the next row, rows 1 and 3) and row_bcast:31 (lane 31 -> rows 2 and 3) steps -- no
// LDS crossbar (__shfl_xor = ds_bpermute: 6 round trips per reduction).
__device__ __forceinline__ unsigned wave_max_u32_lane63(unsigned v) {
#define LION_DPP_U32(x, ctrl, rmask) ((unsigned)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), (rmask), 0xf, true))
  unsigned o;
  o = LION_DPP_U32(v, 0xB1, 0xf); v = o > v ? o : v;
  o = LION_DPP_U32(v, 0x4E, 0xf); v = o > v ? o : v;
  o = LION_DPP_U32(v, 0x141, 0xf); v = o > v ? o : v;
  o = LION_DPP_U32(v, 0x140, 0xf); v = o > v ? o : v;
  o = LION_DPP_U32(v, 0x142, 0xa); v = o > v ? o : v; // row_bcast:15 into rows 1, 3 (disabled rows read 0: a no-op for max)
  o = LION_DPP_U32(v, 0x143, 0xc); v = o > v ? o : v; // row_bcast:31 into rows 2, 3
#undef LION_DPP_U32
  return v;
}
// v + (the value of the other 16-lane row of the same 32-lane half), valid in the ODD rows only (lanes 16-31, 48-63):
// row_bcast:15 hands lane 15 of rows 0 / 2 -- after row16_sum_rn every lane of a row holds the row's sum -- to rows
// 1 / 3.  Same two addends as v += __shfl_xor(v, 16) (addition commutes), without the LDS round trip.
__device__ __forceinline__ float row_pair_sum_odd_rows(float v) {
  return __fadd_rn(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, true)));
}
#define LION_DPP_F32_ROWS(x, ctrl, rmask) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), (rmask), 0xf, true))
__device__ __forceinline__ float row16_max(float v) {
  float o;
  o = LION_DPP_F32(v, 0xB1); v = o > v ? o : v;
  o = LION_DPP_F32(v, 0x4E); v = o > v ? o : v;
  o = LION_DPP_F32(v, 0x141); v = o > v ? o : v;
  o = LION_DPP_F32(v, 0x140); v = o > v ? o : v;
  return v;
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int d = 1; d < LION_WAVE; d <<= 1) {
    const int t = __shfl_up(v, d, LION_WAVE);
    if (lane >= d) v += t;
  }
  return v;
}
