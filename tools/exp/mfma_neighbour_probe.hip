// mfma_neighbour_probe.hip -- which instruction mix is disturbed by a neighbour's dense v_mfma_f32_32x32x16_f16 stream?
// (the register FPS kernel was wrong beside conv3d_split_kernel inside graph replays, and of that
// kernel only the MFMA stream mattered.)  NOT YET RUN -- written at the end of round 2 after the GPU budget was spent.
//
// Aggressor: every wave issues independent 32x32x16 fp16 MFMAs back to back for `iters` rounds (no memory traffic).
// Victims (one 256-thread workgroup per "cloud", a few hundred rounds with one barrier each, like fps_reg_kernel), each
// built from one ingredient of the FPS round so that a failing variant names the vulnerable instruction:
//   0  VALU only: 8 running minima per lane updated from a per-round centre derived from the round number (registers live
//      for the whole kernel), no cross-lane traffic, no LDS
//   1  + the centre comes from a 16-byte broadcast LDS read of a data-dependent slot
//   2  + DPP butterfly (quad_perm / row_half_mirror / row_mirror) + v_readlane wave maximum
//   3  + the LDS key exchange between the four waves with its barrier (= the whole FPS round)
// Every victim runs once ALONE (reference) and then `reps` times beside the aggressor on a second stream; the number of
// workgroups whose result differs from the reference is printed per variant.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_neighbour_probe.hip -o tools/exp/mfma_probe && ./tools/exp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 2) void mfma_aggressor(float *sink, int iters) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (i + 1)); }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 1.2345e30f) sink[blockIdx.x * 256 + threadIdx.x] = s; // keep the MFMAs alive
}

__device__ __forceinline__ unsigned dpp_wave_max(unsigned v) {
  unsigned o;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true); v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true); v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true); v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true); v = o > v ? o : v;
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}

template <int MODE>
__global__ __launch_bounds__(256) void victim(const float *coords, int N, int rounds, unsigned *out) {
  extern __shared__ __attribute__((aligned(16))) float4 sx[];
  __shared__ unsigned long long wkey[2][4];
  constexpr int PPT = 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const float *co = coords + (size_t)b * 3 * N;
  float x[PPT], y[PPT], z[PPT], td[PPT];
  for (int p = 0; p < PPT; ++p) {
    const int k = tid + p * 256;
    x[p] = co[k]; y[p] = co[k + N]; z[p] = co[k + 2 * N];
    td[p] = 1e38f;
    sx[k] = make_float4(x[p], y[p], z[p], 0.f);
  }
  __syncthreads();
  int old = 0;
  unsigned trace = 0u;
  for (int j = 1; j < rounds; ++j) {
    float4 c1;
    if (MODE >= 1) c1 = sx[old];
    else c1 = make_float4(0.001f * (float)((j * 37) % 997), -0.002f * (float)((j * 11) % 499), 0.0005f * (float)(j % 613), 0.f);
    unsigned md = 0u;
    for (int p = 0; p < PPT; ++p) {
      const float dx = x[p] - c1.x, dy = y[p] - c1.y, dz = z[p] - c1.z;
      const float d = dx * dx + dy * dy + dz * dz;
      td[p] = d < td[p] ? d : td[p];
      const unsigned db = __float_as_uint(td[p]);
      md = db > md ? db : md;
    }
    unsigned pick;
    if (MODE >= 2) {
      const unsigned wmax = dpp_wave_max(md);
      unsigned mt = 0u;
      for (int p = 0; p < PPT; ++p) {
        const unsigned cand = __float_as_uint(td[p]) == wmax ? (unsigned)(tid + p * 256) + 1u : 0u;
        mt = cand > mt ? cand : mt;
      }
      const unsigned widx = dpp_wave_max(mt);
      if (MODE >= 3) {
        if (lane == 0) wkey[j & 1][wave] = ((unsigned long long)wmax << 32) | widx;
        __syncthreads();
        unsigned long long k0 = wkey[j & 1][0], k1 = wkey[j & 1][1], k2 = wkey[j & 1][2], k3 = wkey[j & 1][3];
        k0 = k1 > k0 ? k1 : k0; k2 = k3 > k2 ? k3 : k2; k0 = k2 > k0 ? k2 : k0;
        pick = (unsigned)(k0 & 0xffffffffull) - 1u;
      } else {
        pick = widx - 1u;
        __syncthreads();
      }
    } else {
      pick = (unsigned)((j * 131) % N);
      __syncthreads();
    }
    old = (int)(pick % (unsigned)N);
    trace = trace * 1664525u + 1013904223u + md + pick;
  }
  unsigned h = trace;
  for (int p = 0; p < PPT; ++p) h = h * 31u + __float_as_uint(td[p]);
  out[(size_t)b * 256 + tid] = h;
}

template <int MODE>
static int run_mode(const float *coords, int B, int N, int rounds, int reps, unsigned *out_d, float *sink, hipStream_t s1,
                    hipStream_t s2) {
  const size_t lds = (size_t)N * 16;
  std::vector<unsigned> ref((size_t)B * 256), got((size_t)B * 256);
  victim<MODE><<<B, 256, lds, s1>>>(coords, N, rounds, out_d);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(ref.data(), out_d, ref.size() * 4, hipMemcpyDeviceToHost);
  int bad_wgs = 0;
  for (int r = 0; r < reps; ++r) {
    mfma_aggressor<<<512, 256, 0, s2>>>(sink, 20000);
    victim<MODE><<<B, 256, lds, s1>>>(coords, N, rounds, out_d);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(got.data(), out_d, got.size() * 4, hipMemcpyDeviceToHost);
    for (int b = 0; b < B; ++b) {
      bool bad = false;
      for (int t = 0; t < 256; ++t) bad |= got[(size_t)b * 256 + t] != ref[(size_t)b * 256 + t];
      bad_wgs += bad;
    }
  }
  return bad_wgs;
}

int main() {
  const int B = 64, N = 2048, rounds = 1024, reps = 10;
  std::vector<float> h((size_t)B * 3 * N);
  unsigned s = 12345u;
  for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; }
  float *coords, *sink; unsigned *out;
  (void)hipMalloc(&coords, h.size() * 4); (void)hipMemcpy(coords, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMalloc(&sink, 512 * 256 * 4); (void)hipMalloc(&out, (size_t)B * 256 * 4);
  hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
  const char *names[4] = {"VALU running minima only", "+ broadcast LDS centre read", "+ DPP / readlane wave maximum",
                          "+ LDS key exchange and barrier (the whole FPS round)"};
  const int bad[4] = {run_mode<0>(coords, B, N, rounds, reps, out, sink, s1, s2), run_mode<1>(coords, B, N, rounds, reps, out, sink, s1, s2),
                      run_mode<2>(coords, B, N, rounds, reps, out, sink, s1, s2), run_mode<3>(coords, B, N, rounds, reps, out, sink, s1, s2)};
  for (int m = 0; m < 4; ++m)
    printf("victim %d (%-52s): %d of %d workgroup runs differ from the run without the MFMA neighbour\n", m, names[m], bad[m], B * reps);
  return 0;
}
