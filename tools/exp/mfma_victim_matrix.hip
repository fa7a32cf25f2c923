// mfma_victim_matrix.hip -- round 3: which instruction CLASS of a co-resident wave is disturbed by a neighbour's dense
// matrix-pipe stream?  (fps_reg_kernel returned wrong samples beside conv3d_split_kernel.)
// The compiled FPS round holds, besides plain VALU: SLP-packed fp32 arithmetic (v_pk_add_f32 / v_pk_mul_f32 with op_sel /
// neg modifiers -- 116 of them in sampling.hip at -O3), v_cmp_*_e64 -> SGPR-pair masks -> v_cndmask, DPP row operations,
// v_readlane, s_setprio 3, one barrier per round.  Victims isolate these; aggressors vary the matrix instruction and the
// register footprint.  Every victim runs alone first (reference hash per workgroup), then `reps` times beside each
// aggressor; printed: workgroup runs whose hash differs.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/exp/mfma_victim_matrix.hip -o tools/exp/mfma_matrix
// (-fno-slp-vectorize: the only packed fp32 operations are the float2 expressions written below)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

// ---- aggressors: MODE 0 fp16 32x32x16 (4 accumulators), 1 fp16 with 8 accumulators (128 registers, 2 waves / SIMD like
// the convolution), 2 fp32 32x32x2, 3 bf16 32x32x16, 4 fp16 16x16x32
template <int MODE>
__global__ __launch_bounds__(256, 2) void aggressor(float *sink, int iters) {
  h8 a, b;
  b8 ab, bb;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (i + 1));
    ab[i] = (__bf16)(0.001f * (threadIdx.x + i)); bb[i] = (__bf16)(0.002f * (i + 1));
  }
  constexpr int NA = MODE == 1 ? 8 : 4;
  f32x16 c[NA];
  for (int k = 0; k < NA; ++k) c[k] = f32x16{};
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 d[4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      if (MODE <= 1) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[k], 0, 0, 0);
      else if (MODE == 2) c[k] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[0], (float)b[0], c[k], 0, 0, 0);
      else if (MODE == 3) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c[k], 0, 0, 0);
      else d[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d[k & 3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int k = 0; k < NA; ++k) for (int i = 0; i < 16; ++i) s += c[k][i];
  for (int k = 0; k < 4; ++k) for (int i = 0; i < 4; ++i) s += d[k][i];
  if (s == 1.2345e30f) sink[blockIdx.x * 256 + threadIdx.x] = s;
}

__device__ __forceinline__ unsigned dpp_row_max(unsigned v) {
  unsigned o;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true); v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true); v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true); v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true); v = o > v ? o : v;
  return v;
}

// ---- victims (256 threads, `rounds` rounds with one barrier each, 8 running minima per lane kept in registers)
//  0 scalar fp32 VALU (v_sub / v_mul / v_add / v_min)           1 the same arithmetic written on float2: v_pk_*_f32
//  2 scalar + v_cmp_lt_f32_e64 -> SGPR pair -> v_cndmask (the select form of the FPS round)
//  3 scalar + DPP row maxima                                      4 scalar + DPP + v_readlane
//  5 = 1 + 2 + 4 (everything)                                     6 = 5 with s_setprio 3
template <int MODE>
__global__ __launch_bounds__(256) void victim(const float *coords, int N, int rounds, unsigned *out) {
  constexpr int PPT = 8;
  const bool PK = MODE == 1 || MODE >= 5, SEL = MODE == 2 || MODE >= 5, DPP = MODE >= 3, RL = MODE >= 4;
  if (MODE == 6) __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, b = blockIdx.x;
  const float *co = coords + (size_t)b * 3 * N;
  float x[PPT], y[PPT], z[PPT], td[PPT];
  for (int p = 0; p < PPT; ++p) {
    const int k = tid + p * 256;
    x[p] = co[k]; y[p] = co[k + N]; z[p] = co[k + 2 * N];
    td[p] = 1e38f;
  }
  unsigned trace = 0u;
  for (int j = 1; j < rounds; ++j) {
    const float cx = 0.001f * (float)((j * 37) % 997) - 0.5f, cy = 0.5f - 0.002f * (float)((j * 11) % 499),
                cz = 0.0005f * (float)(j % 613);
    unsigned md = 0u;
    if (PK) {
#pragma unroll
      for (int p = 0; p < PPT; p += 2) {
        const f2 dx = f2{x[p], x[p + 1]} - f2{cx, cx}, dy = f2{y[p], y[p + 1]} - f2{cy, cy}, dz = f2{z[p], z[p + 1]} - f2{cz, cz};
        const f2 d = dx * dx + dy * dy + dz * dz;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float dd = h ? d.y : d.x;
          if (SEL) { const bool lt = dd < td[p + h]; td[p + h] = lt ? dd : td[p + h]; }
          else td[p + h] = fminf(dd, td[p + h]);
        }
      }
    } else {
#pragma unroll
      for (int p = 0; p < PPT; ++p) {
        const float dx = x[p] - cx, dy = y[p] - cy, dz = z[p] - cz;
        const float dd = dx * dx + dy * dy + dz * dz;
        if (SEL) { const bool lt = dd < td[p]; td[p] = lt ? dd : td[p]; }
        else td[p] = fminf(dd, td[p]);
      }
    }
#pragma unroll
    for (int p = 0; p < PPT; ++p) { const unsigned db = __float_as_uint(td[p]); md = db > md ? db : md; }
    unsigned pick = md;
    if (DPP) {
      pick = dpp_row_max(md);
      if (RL) {
        const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)pick, 0), b2 = (unsigned)__builtin_amdgcn_readlane((int)pick, 16);
        const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)pick, 32), d2 = (unsigned)__builtin_amdgcn_readlane((int)pick, 48);
        const unsigned ab = a > b2 ? a : b2, cd = c > d2 ? c : d2;
        pick = ab > cd ? ab : cd;
      }
    }
    __syncthreads();
    trace = trace * 1664525u + 1013904223u + pick;
  }
  unsigned h = trace;
  for (int p = 0; p < PPT; ++p) h = h * 31u + __float_as_uint(td[p]);
  out[(size_t)b * 256 + tid] = h;
}

template <int VM>
static void launch_victim(const float *c, int B, int N, int rounds, unsigned *out, hipStream_t s) { victim<VM><<<B, 256, 0, s>>>(c, N, rounds, out); }
static void launch_aggr(int am, float *sink, int iters, hipStream_t s) {
  switch (am) {
    case 0: aggressor<0><<<512, 256, 0, s>>>(sink, iters); break;
    case 1: aggressor<1><<<512, 256, 0, s>>>(sink, iters / 2); break;
    case 2: aggressor<2><<<512, 256, 0, s>>>(sink, iters / 2); break;
    case 3: aggressor<3><<<512, 256, 0, s>>>(sink, iters); break;
    default: aggressor<4><<<512, 256, 0, s>>>(sink, iters * 2); break;
  }
}

int main() {
  const int B = 128, N = 2048, rounds = 1024, reps = 8, NV = 7, NA = 5;
  std::vector<float> h((size_t)B * 3 * N);
  unsigned s = 12345u;
  for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; }
  float *coords, *sink; unsigned *out;
  (void)hipMalloc(&coords, h.size() * 4); (void)hipMemcpy(coords, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMalloc(&sink, 512 * 256 * 4); (void)hipMalloc(&out, (size_t)B * 256 * 4);
  hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
  void (*lv[NV])(const float *, int, int, int, unsigned *, hipStream_t) = {launch_victim<0>, launch_victim<1>, launch_victim<2>, launch_victim<3>,
                                                                        launch_victim<4>, launch_victim<5>, launch_victim<6>};
  const char *vn[NV] = {"scalar f32 VALU", "packed f32 (v_pk_*_f32)", "scalar + v_cmp_e64/v_cndmask", "scalar + DPP", "scalar + DPP + v_readlane",
                        "packed + select + DPP + readlane", "the same + s_setprio 3"};
  const char *an[NA] = {"fp16 32x32x16 x4 acc", "fp16 32x32x16 x8 acc", "fp32 32x32x2", "bf16 32x32x16", "fp16 16x16x32"};
  std::vector<unsigned> ref((size_t)B * 256), got((size_t)B * 256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int v = 0; v < NV; ++v) {
    lv[v](coords, B, N, rounds, out, s1);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(ref.data(), out, ref.size() * 4, hipMemcpyDeviceToHost);
    for (int a = 0; a < NA; ++a) {
      int bad = 0;
      float ms_a = 0.f;
      for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(e0, s2);
        launch_aggr(a, sink, 12000, s2);
        (void)hipEventRecord(e1, s2);
        lv[v](coords, B, N, rounds, out, s1);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms_a, e0, e1);
        (void)hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost);
        for (int b = 0; b < B; ++b) {
          bool w = false;
          for (int t = 0; t < 256; ++t) w |= got[(size_t)b * 256 + t] != ref[(size_t)b * 256 + t];
          bad += w;
        }
      }
      printf("victim %d (%-34s) beside %-22s (%.2f ms): %4d of %d workgroup runs differ\n", v, vn[v], an[a], ms_a, bad, B * reps);
    }
  }
  return 0;
}
