// skinny_probe.hip -- where does the time of the global denoiser's 2048x2048 layers go?  (0.47 TB/s)
// Standalone: hipcc --offload-arch=gfx950 -O3 -I include -I lion_amd/csrc tools/exp/skinny_probe.hip -o tools/exp/skinny_probe
// Cycles through NBUF weight buffers (> 256 MB Infinity Cache) so that every launch streams HBM-cold weights, like the
// chain does (309 MB per forward), and times: the product kernel (ks_in = 4 and 1), pure streaming reads with the
// same geometry, and the batch-major variant (float4 operand loads, all loads up front).
#include "../../lion_amd/csrc/skinny.hip"
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_nt(const float4 *p) {
  const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

// pure stream: grid (tiles, KS), 1024 threads, every lane reads its `nload` float4 (all in flight), one float out
template <int NLOAD, bool NT>
__global__ __launch_bounds__(1024) void stream_kernel(const float4 *__restrict__ w, float *__restrict__ out) {
  const size_t base = ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 1024 + threadIdx.x);
  float4 v[NLOAD];
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) {
    const float4 *p = w + base + (size_t)i * gridDim.x * gridDim.y * 1024;
    v[i] = NT ? ld_nt(p) : *p;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
  if (s == 12345.678f) out[base] = s;
}

// batch-major variant.  act / partials: [q][32 b][C]; weights packed [tile][g][kh][32][4]: k = 8g + 4kh + e.
template <int KSIN>
__global__ __launch_bounds__(1024) void gemm_bm_kernel(const float *__restrict__ pin, const float *__restrict__ bias_in,
                                                       int act_in, const float *__restrict__ addT,
                                                       const float *__restrict__ wp, int Cin, int Cout,
                                                       float *__restrict__ pout) {
  extern __shared__ __attribute__((aligned(16))) float smem[]; // [16][32*36]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int o0 = blockIdx.x * 32, ks = blockIdx.y, KS = gridDim.y;
  const int cl = lane & 31, kh = lane >> 5;
  const int groups = Cin >> 3;                       // 8 k per group
  const int per = (groups + KS * 16 - 1) / (KS * 16); // groups per wave
  const int g_lo = min(groups, (ks * 16 + wave) * per), g_hi = min(groups, g_lo + per);
  const float4 *wt4 = reinterpret_cast<const float4 *>(wp + (size_t)blockIdx.x * groups * 256);
  const size_t in_stride = (size_t)32 * Cin;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  constexpr int GP = 4; // groups per round: 4 weight + 4*KSIN operand float4 loads in flight per lane
  for (int g0 = g_lo; g0 < g_hi; g0 += GP) {
    float4 a4[GP], x4[GP][KSIN], b4[GP], t4[GP];
#pragma unroll
    for (int g = 0; g < GP; ++g) {
      const int gg = min(g0 + g, groups - 1);
      a4[g] = ld_nt(&wt4[((size_t)gg * 2 + kh) * 32 + cl]);
      const size_t off = (size_t)cl * Cin + gg * 8 + kh * 4;
#pragma unroll
      for (int q = 0; q < KSIN; ++q) x4[g][q] = *reinterpret_cast<const float4 *>(pin + q * in_stride + off);
      b4[g] = bias_in ? *reinterpret_cast<const float4 *>(bias_in + gg * 8 + kh * 4) : make_float4(0, 0, 0, 0);
      t4[g] = addT ? *reinterpret_cast<const float4 *>(addT + off) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < GP; ++g) {
      float v[4] = {b4[g].x, b4[g].y, b4[g].z, b4[g].w};
#pragma unroll
      for (int q = 0; q < KSIN; ++q) { v[0] += x4[g][q].x; v[1] += x4[g][q].y; v[2] += x4[g][q].z; v[3] += x4[g][q].w; }
      const float tt[4] = {t4[g].x, t4[g].y, t4[g].z, t4[g].w};
      const float aa[4] = {a4[g].x, a4[g].y, a4[g].z, a4[g].w};
      const bool live = g0 + g < g_hi;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = v[e];
        if (act_in == 1) x = x > 0.f ? x : 0.f;
        x += tt[e];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[e], live ? x : 0.f, acc, 0, 0, 0);
      }
    }
  }
  float *part = smem + wave * (32 * 36);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    *reinterpret_cast<float4 *>(part + cl * 36 + 8 * j + 4 * kh) = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
  __syncthreads();
  const int b = tid >> 5, o = tid & 31;
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) s += smem[w * (32 * 36) + b * 36 + o];
  pout[(size_t)ks * 32 * Cout + (size_t)b * Cout + o0 + o] = s;
}

__global__ void pack_bm_kernel(const float *__restrict__ w, int Cout, int Cin, float *__restrict__ wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, groups = Cin >> 3;
  if (i >= Cout * Cin) return;
  const int e = i & 3, j = (i >> 2) & 31, kh = (i >> 7) & 1, g = (i >> 8) % groups, t = i / (256 * groups);
  wp[i] = w[(size_t)(t * 32 + j) * Cin + 8 * g + 4 * kh + e];
}

} // namespace

int main() {
  const int C = 2048, NBUF = 20, REPS = 5;
  const size_t wfl = (size_t)C * C;
  std::vector<float *> wp(NBUF), wb(NBUF);
  float *w, *pin, *pout, *bias, *pinb, *poutb;
  CK(hipMalloc(&w, wfl * 4));
  std::vector<float> hw(wfl);
  for (size_t i = 0; i < wfl; ++i) hw[i] = (float)((int)((i * 2654435761u) >> 20 & 1023) - 512) / 8192.f;
  CK(hipMemcpy(w, hw.data(), wfl * 4, hipMemcpyHostToDevice));
  for (int i = 0; i < NBUF; ++i) {
    CK(hipMalloc(&wp[i], wfl * 4));
    CK(hipMalloc(&wb[i], wfl * 4));
    if (lion_skinny_pack_weights(w, C, C, wp[i], nullptr)) return 2;
    pack_bm_kernel<<<(C * C + 255) / 256, 256>>>(w, C, C, wb[i]);
  }
  CK(hipMalloc(&pin, (size_t)4 * C * 32 * 4)); CK(hipMalloc(&pout, (size_t)8 * C * 32 * 4));
  CK(hipMalloc(&pinb, (size_t)4 * C * 32 * 4)); CK(hipMalloc(&poutb, (size_t)8 * C * 32 * 4));
  CK(hipMalloc(&bias, C * 4));
  std::vector<float> hx((size_t)4 * C * 32), hxb((size_t)4 * C * 32), hb(C);
  for (int q = 0; q < 4; ++q)
    for (int k = 0; k < C; ++k)
      for (int b = 0; b < 32; ++b) {
        const float v = (float)(((q * 131 + k * 31 + b * 7) % 257) - 128) / 256.f;
        hx[((size_t)q * C + k) * 32 + b] = v;          // channel-major [q][C][32]
        hxb[((size_t)q * 32 + b) * C + k] = v;         // batch-major   [q][32][C]
      }
  for (int k = 0; k < C; ++k) hb[k] = (float)(k % 13) / 13.f - 0.5f;
  CK(hipMemcpy(pin, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(pinb, hxb.data(), hxb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hb.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bm_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 32 * 36 * 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bm_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 32 * 36 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char *name, auto fn, bool cold) -> int {
    for (int i = 0; i < NBUF; ++i) fn(cold ? i : 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < REPS; ++r)
      for (int i = 0; i < NBUF; ++i) fn(cold ? i : 0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (REPS * NBUF);
    printf("%-58s %s  %7.2f us/launch  %6.2f TB/s\n", name, cold ? "cold" : "warm", us, wfl * 4 / us / 1e6);
    return 0;
  };
  const size_t lds_bm = 16 * 32 * 36 * 4;
  for (int cold = 1; cold >= 0; --cold) {
    timeit("product skinny_gemm ks_in=4 (+bias, relu)", [&](int i) { lion_skinny_gemm(pin, 4, bias, 1, nullptr, wp[i], 1, C, C, pout, nullptr); }, cold);
    timeit("product skinny_gemm ks_in=1", [&](int i) { lion_skinny_gemm(pin, 1, nullptr, 0, nullptr, wp[i], 1, C, C, pout, nullptr); }, cold);
    timeit("stream (64,4)x1024, 4 float4/lane", [&](int i) { stream_kernel<4, false><<<dim3(64, 4), 1024>>>((const float4 *)wp[i], pout); }, cold);
    timeit("stream (64,4)x1024, 4 float4/lane, nt", [&](int i) { stream_kernel<4, true><<<dim3(64, 4), 1024>>>((const float4 *)wp[i], pout); }, cold);
    timeit("stream (64,8)x1024, 2 float4/lane", [&](int i) { stream_kernel<2, false><<<dim3(64, 8), 1024>>>((const float4 *)wp[i], pout); }, cold);
    timeit("stream (64,2)x1024, 8 float4/lane, nt", [&](int i) { stream_kernel<8, true><<<dim3(64, 2), 1024>>>((const float4 *)wp[i], pout); }, cold);
    timeit("batch-major gemm ks_in=4 (+bias, relu), KS=4", [&](int i) { gemm_bm_kernel<4><<<dim3(64, 4), 1024, lds_bm>>>(pinb, bias, 1, nullptr, wb[i], C, C, poutb); }, cold);
    timeit("batch-major gemm ks_in=1, KS=4", [&](int i) { gemm_bm_kernel<1><<<dim3(64, 4), 1024, lds_bm>>>(pinb, nullptr, 0, nullptr, wb[i], C, C, poutb); }, cold);
    timeit("batch-major gemm ks_in=4, KS=8", [&](int i) { gemm_bm_kernel<4><<<dim3(64, 8), 1024, lds_bm>>>(pinb, bias, 1, nullptr, wb[i], C, C, poutb); }, cold);
  }
  // correctness of the batch-major variant vs the product kernel (same operand, same weights)
  lion_skinny_gemm(pin, 4, bias, 1, nullptr, wp[0], 1, C, C, pout, nullptr);
  gemm_bm_kernel<4><<<dim3(64, 4), 1024, lds_bm>>>(pinb, bias, 1, nullptr, wb[0], C, C, poutb);
  CK(hipDeviceSynchronize());
  std::vector<float> ho((size_t)4 * C * 32), hob((size_t)4 * C * 32);
  CK(hipMemcpy(ho.data(), pout, ho.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hob.data(), poutb, hob.size() * 4, hipMemcpyDeviceToHost));
  double maxd = 0, maxv = 0;
  for (int o = 0; o < C; ++o)
    for (int b = 0; b < 32; ++b) {
      double a = 0, c = 0;
      for (int q = 0; q < 4; ++q) { a += ho[((size_t)q * C + o) * 32 + b]; c += hob[((size_t)q * 32 + b) * C + o]; }
      maxd = fmax(maxd, fabs(a - c)); maxv = fmax(maxv, fabs(a));
    }
  printf("batch-major vs product: max |diff| %.3g (max |value| %.3g)\n", maxd, maxv);
  return 0;
}
