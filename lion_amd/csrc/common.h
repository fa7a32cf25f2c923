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

// (a-b)^2 + (c-d)^2 + (e-f)^2 evaluated left to right without contraction.
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by,
                                         float bz) {
  const float dx = sub_rn(ax, bx), dy = sub_rn(ay, by), dz = sub_rn(az, bz);
  return add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
}

// Inclusive wave prefix sum (wave64) with shuffles.
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int d = 1; d < LION_WAVE; d <<= 1) {
    const int t = __shfl_up(v, d, LION_WAVE);
    if (lane >= d) v += t;
  }
  return v;
}
