// sampling.hip -- K9 furthest point sampling.
//
// Reference: third_party/pvcnn/functional/src/sampling/sampling.cu:86-167, sampling.cpp:43-58
// (512 threads per cloud, two __syncthreads tree reductions per round, coordinates in 36 KiB of
// shared memory, running distances in global memory).
//
// M-1 dependent rounds per cloud make this latency bound, so the design minimises the length of
// one round: one 256-thread workgroup per cloud, the cloud's coordinates AND running min-distances
// live in registers (PPT points per lane), the arg-max is a single 64-bit key max
//   key = (float bits of d2) << 32 | ~((k & 511) << 20 | k)
// (d2 >= 0 so its bit pattern is monotone) reduced with 6 xor-shuffles per wave and ONE barrier per
// round (double-buffered 4-entry LDS exchange).  The low word reproduces the reference's tie-break
// exactly: its strided per-thread scan + left-biased tree picks the lowest (k mod 512, k) among
// equal distances (sampling.cu:141-159), so the indices are bit-exact, ties included.
// Clouds with N > 256*8 take the generic strided variant below (distances in LDS, up to 32 K points).
// The per-wave arg-max runs on DPP row operations + v_readlane (two 32-bit reductions: the distance bits, then the
// tie-break word among the lanes that hold the maximum) instead of 64-bit xor-shuffles, which go through the LDS
// crossbar (ds_bpermute, ~12 dependent LDS round trips per round).
#include "common.h"

namespace {

__device__ __forceinline__ unsigned long long fps_key(float d2, int k) {
  const unsigned int tb = 0xffffffffu - (((unsigned)(k & 511) << 20) | (unsigned)k);
  return ((unsigned long long)__float_as_uint(d2) << 32) | tb;
}
__device__ __forceinline__ int fps_key_index(unsigned long long key) {
  return (int)((0xffffffffu - (unsigned)(key & 0xffffffffull)) & 0xfffffu);
}
// max over the 64 lanes, result uniform: xor-1 / xor-2 inside quads, half-mirror (8), mirror (16), then the 4 rows
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  unsigned o;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
  v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
  v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true); // row_half_mirror
  v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true); // row_mirror
  v = o > v ? o : v;
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const unsigned long long o = __shfl_xor(v, s, 64);
    v = o > v ? o : v;
  }
  return v;
}

// NW waves per cloud, PPT points per lane.  Four waves: eight (two per SIMD, half the arithmetic per wave, eight exchange
// words) were measured in round 3 at 549 us against 530 us for 2048 -> 1024, B = 32 -- a round is bound by its chain of
// dependent latencies (LDS lookup, two DPP reductions, exchange + barrier + LDS read: ~800 of 1250 cycles), not by
// VALU issue.  With the sampling step on one stream (lion_amd/geometry.py) the chain is on the critical path of every
// denoiser step (0.68 ms of 8.7).
template <int PPT, int NW>
__global__ __launch_bounds__(64 * NW) void fps_reg_kernel(const float *__restrict__ coords, int N, int M,
                                                           int32_t *__restrict__ idx) {
  constexpr int TPB = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4 *sx = reinterpret_cast<float4 *>(smem); // [N] {x, y, z, -} copy for the "coords[old]" lookup (one 16-byte read)
  __shared__ unsigned long long wkey[2][NW];
  // 1023 dependent rounds on one workgroup per cloud: pure latency.  When the sampler runs this chain on a side stream
  // under the MFMA convolutions (lion_amd/geometry.py, optional) the raised wave priority lets its few instructions per
  // round issue ahead of the convolution waves sharing the SIMD instead of queueing behind them.
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const float *co = coords + (size_t)b * 3 * N;
  float x[PPT], y[PPT], z[PPT], td[PPT];
  unsigned tb[PPT]; // tie-break word of point k: larger wins, 0 for lanes past N (see fps_key)
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int k = tid + p * TPB;
    x[p] = y[p] = z[p] = 0.f;
    td[p] = 1e38f; // sampling.cpp:53-54
    tb[p] = 0u;
    if (k < N) {
      x[p] = co[k]; y[p] = co[k + N]; z[p] = co[k + 2 * N];
      sx[k] = make_float4(x[p], y[p], z[p], 0.f);
      tb[p] = 0xffffffffu - (((unsigned)(k & 511) << 20) | (unsigned)k);
    }
  }
  int32_t *out = idx + (size_t)b * M;
  if (tid == 0) out[0] = 0;
  __syncthreads();
  int old = 0;
  for (int j = 1; j < M; ++j) {
    const float4 c1 = sx[old];
    unsigned md = 0u;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const float d = sqdist3(x[p], y[p], z[p], c1.x, c1.y, c1.z); // sampling.cu:133-134
      const float d2 = d < td[p] ? d : td[p];
      td[p] = tb[p] ? d2 : 0.f; // lanes past N never compete (distance bits 0, tie-break word 0)
      const unsigned db = __float_as_uint(td[p]); // d2 >= 0: the bit pattern is monotone
      md = db > md ? db : md;
    }
    const unsigned wmax = wave_max_u32(md);
    unsigned mt = 0u;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const unsigned cand = __float_as_uint(td[p]) == wmax ? tb[p] : 0u;
      mt = cand > mt ? cand : mt;
    }
    const unsigned wtb = wave_max_u32(mt);
    if (lane == 0) wkey[j & 1][wave] = ((unsigned long long)wmax << 32) | wtb;
    __syncthreads();
    unsigned long long k0 = wkey[j & 1][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) { const unsigned long long kw = wkey[j & 1][w]; k0 = kw > k0 ? kw : k0; }
    old = fps_key_index(k0);
    if (tid == 0) out[j] = old;
  }
}

// Generic variant: running distances in LDS (N <= 32768), coordinates re-read from global / L2.
__global__ __launch_bounds__(1024) void fps_lds_kernel(const float *__restrict__ coords, int N,
                                                       int M, int32_t *__restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *td = reinterpret_cast<float *>(smem);
  __shared__ unsigned long long wkey[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const float *co = coords + (size_t)b * 3 * N;
  for (int k = tid; k < N; k += 1024) td[k] = 1e38f;
  int32_t *out = idx + (size_t)b * M;
  if (tid == 0) out[0] = 0;
  __syncthreads();
  int old = 0;
  for (int j = 1; j < M; ++j) {
    const float x1 = co[old], y1 = co[old + N], z1 = co[old + 2 * (size_t)N];
    unsigned long long best = 0ull;
    for (int k = tid; k < N; k += 1024) {
      const float d = sqdist3(co[k], co[k + N], co[k + 2 * (size_t)N], x1, y1, z1);
      const float t = td[k];
      const float d2 = d < t ? d : t;
      td[k] = d2;
      const unsigned long long key = fps_key(d2, k);
      best = key > best ? key : best;
    }
    best = wave_max_u64(best);
    if (lane == 0) wkey[j & 1][wave] = best;
    __syncthreads();
    unsigned long long m = wkey[j & 1][lane & 15];
#pragma unroll
    for (int s = 1; s < 16; s <<= 1) {
      const unsigned long long o = __shfl_xor(m, s, 64);
      m = o > m ? o : m;
    }
    old = fps_key_index(m);
    if (tid == 0) out[j] = old;
  }
}

template <int PPT, int NW>
static int launch_fps_reg(const float *coords, int B, int N, int M, int32_t *idx, hipStream_t st) {
  // History.  Round 2: inside a captured sampling step this kernel ran on a side stream beside the
  // convolutions and returned 300-1800 wrong indices of 2048 per graph replay whenever conv3d_split_kernel workgroups
  // shared its CUs; it was then given a whole CU's LDS so that nothing could share.  Round 3 named the mechanism with
  // builds that differ in one property each (tools/victims_beside_conv.py, profiles/archive/r03_fps_variants.txt): the SLP
  // vectoriser had turned the distance arithmetic into PACKED fp32 VALU (v_pk_add_f32 / v_pk_mul_f32); any build that
  // contains them -- even a single float2 expression in an otherwise scalar kernel -- fails in 37-40 of 40 replays beside
  // the fp16-MFMA stream, every build without them (52, 200 or 256 registers, with or without s_setprio) is exact in 100+.
  // The library is compiled with -fno-slp-vectorize (csrc/build.sh, tests/test_isa_cpu.py), the kernel asks for the
  // N * 16 bytes it uses, and it shares its CUs again (tests/test_concurrency_gpu.py replays it beside the convolution).
  const size_t lds = (size_t)N * 16;
  static LionLdsLimit configured = {};
  if (int e = lion_dynamic_lds(&fps_reg_kernel<PPT, NW>, lds, configured)) return e;
  fps_reg_kernel<PPT, NW><<<B, 64 * NW, lds, st>>>(coords, N, M, idx);
  LION_LAUNCH_CHECK();
  return 0;
}

} // namespace

extern "C" int lion_furthest_point_sampling(const float *coords, int B, int N, int M, int32_t *idx,
                                            lionStream_t stream) {
  if (!coords || !idx || B <= 0 || N <= 0 || M < 0) return LION_EINVAL;
  if (M == 0) return 0;
  if (N > (1 << 20)) return LION_EUNSUPPORTED; // tie-break key holds 20 index bits
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (N <= 256) return launch_fps_reg<1, 4>(coords, B, N, M, idx, st);
  if (N <= 512) return launch_fps_reg<2, 4>(coords, B, N, M, idx, st);
  if (N <= 1024) return launch_fps_reg<4, 4>(coords, B, N, M, idx, st);
  if (N <= 2048) return launch_fps_reg<8, 4>(coords, B, N, M, idx, st);
  if (N <= 4096) return launch_fps_reg<16, 4>(coords, B, N, M, idx, st);
  if (N <= 32768) {
    const size_t lds = (size_t)N * 4;
    static LionLdsLimit configured = {};
    if (int e = lion_dynamic_lds(&fps_lds_kernel, lds, configured)) return e;
    fps_lds_kernel<<<B, 1024, lds, st>>>(coords, N, M, idx);
    LION_LAUNCH_CHECK();
    return 0;
  }
  return LION_EUNSUPPORTED;
}
