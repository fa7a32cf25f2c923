// split_ops.h -- what the split-operand kernels share (conv3d_split.hip, pwconv_split.hip): an fp32 operand is cut into
// two fp16 pieces a = a_h + a_l / 2048 after an exact power-of-two block scaling, products are accumulated in fp32 on
// v_mfma_f32_32x32x16_f16:  main += A_h B_h,  corr += A_h B_l + A_l B_h,  D = main + corr / 2048.
#pragma once
#include "common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
constexpr int KS = 16; // input channels per chunk = K of one MFMA
template <bool V> struct BoolC { static constexpr bool value = V; }; // compile-time flag for generic lambdas

__device__ __forceinline__ float pro_act(float v, float pa, float pb) { // == csrc/conv3d.hip
  const float t = v * pa + pb;
  return swish_fast(t);
}
// exponent e with 2^13 <= m * 2^e < 2^14 for a finite m > 0 (from the float's exponent field; subnormal m -> +100)
__device__ __forceinline__ int scale_exp(float m) {
  const int ex = (int)((__float_as_uint(m) >> 23) & 0xff) - 127; // floor(log2 m) for normal m; 128 for inf / nan
  const int e = 13 - ex;
  return e > 100 ? 100 : e;
}
// A running block scale 2^E may be set H binades below the cut's ceiling (max * 2^E in [2^(13-H), 2^(14-H)) when it
// is chosen) and is only replaced when a later maximum reaches 2^14: growth by less than 2^H costs no accumulator
// rescale.  Nothing is lost: fp16 is normal down to 2^-14, i.e. 2^-23 of the block maximum at H = 4, and subnormal
// steps below that are 2^-33 of it.  The 1x1 kernel (one scale per COLUMN: some column of a wave grows in most chunks)
// uses H = 4; the 3x3x3 kernels (one scale per workgroup tile: growth is rare) keep H = 0 -- with H = 4 the register
// allocation of the 32-channel tile variant went from 156 to 380 bytes of scratch and the layer from 296 to 400 us.
constexpr int PW_SPLIT_HEADROOM = 4;
constexpr int CONV_SPLIT_HEADROOM = 0;
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); } // -126 <= e <= 127
__device__ __forceinline__ void cut(float v, unsigned short &hi, unsigned short &lo) {
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)((v - (float)h) * 2048.f);
  hi = __builtin_bit_cast(unsigned short, h);
  lo = __builtin_bit_cast(unsigned short, l);
}
// two values at once: v_cvt_pk_f16_f32 (gfx950) converts and packs a pair in one instruction -- the same round-to-nearest
// as the scalar conversion, so (hi2, lo2) = the scalar cuts of a (low half) and b (high half), bit for bit; 8 VALU per pair
// instead of 12.  The subtraction and the scaling stay scalar: the library holds no packed fp32 arithmetic.
__device__ __forceinline__ void cut2(float a, float b, unsigned &hi2, unsigned &lo2) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  typedef float f2_t __attribute__((ext_vector_type(2)));
  const h2_t h = __builtin_convertvector(f2_t{a, b}, h2_t);
  const float la = (a - (float)h[0]) * 2048.f, lb = (b - (float)h[1]) * 2048.f;
  const h2_t l = __builtin_convertvector(f2_t{la, lb}, h2_t);
  hi2 = __builtin_bit_cast(unsigned, h);
  lo2 = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ f32x16 mma(u4 a, u4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

// ---- the per-tensor power-of-two scale of a packed weight: tail = 4 words behind the pieces {max |w| bits, ew, 2^-ew, 0}
// with max |w| * 2^ew in [2^13, 2^14).  pass 1: max |w| (as bits: non-negative floats order like unsigned integers).
static __global__ void split_wmax_kernel(const float *__restrict__ w, int n, unsigned *__restrict__ tail) {
  // grid-stride, ONE atomic per workgroup: with one element per thread and one atomicMax per wave a 442 k-element tensor
  // queued 6.9 k atomics on one word (54 us -- the training step repacks ~120 weights per step)
  __shared__ unsigned sm[4];
  unsigned m = 0u;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const unsigned a = __float_as_uint(w[i]) & 0x7fffffffu;
    m = a > m ? a : m;
  }
  for (int s = 32; s > 0; s >>= 1) { const unsigned o = __shfl_xor(m, s, 64); m = o > m ? o : m; }
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < (int)(blockDim.x >> 6); ++k) m = sm[k] > m ? sm[k] : m;
    if (m) atomicMax(tail, m);
  }
}
// pass 2 rides on the pack kernels: every thread derives ew from the maximum, the first one records {ew, 2^-ew} for the
// consumers (was a one-thread launch of its own between the two passes)
__device__ __forceinline__ int split_tail_scale(unsigned *__restrict__ tail, bool writer) {
  const float m = __uint_as_float(tail[0]);
  const int ew = m > 0.f ? scale_exp(m) : 0;
  if (writer) {
    tail[1] = (unsigned)ew;
    tail[2] = __float_as_uint(pow2f(-ew));
    tail[3] = 0u;
  }
  return ew;
}

// the former stand-alone pass 2 (tools/exp variants of the kernels still launch it)
static __global__ void split_wscale_kernel(unsigned *__restrict__ tail) { split_tail_scale(tail, true); }

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

} // namespace
