// mfma_issue_probe.hip -- how fast does ONE wave issue v_mfma_f32_32x32x16_f16, alone on its SIMD and beside a partner,
// with and without the fragment reads of the split convolution's tap loop?  (profiles/HISTORY.md 4c claims 64-65 cycles per MFMA
// and wave "with or without a second wave"; the microarchitecture guide says 32 for a lone wave.  The answer decides
// whether a one-wave-per-SIMD tap loop can feed the matrix pipe.)
//
// Every configuration runs the tap pattern of conv3d_split_kernel<.., CB, VB, ..>: per "tap" CB*VB main products, CB*VB
// W_h X_l products, CB*VB W_l X_h products into 2*CB*VB accumulators; READS = the 2*(CB+VB) ds_read_b128 of the next tap
// spread one per MFMA slot (double buffered in registers), one lgkmcnt(0) per tap.
//   role 0: MFMA wave (the measured one); role 1: partner = VALU loop (v_exp / fma / cvt: a staging wave); role 2:
//   partner = idle (ends at once); role 3: partner runs the same MFMA pattern.
// One workgroup per CU (LDS request), 256 workgroups x rounds; clock64 (s_memtime) per wave, wall_clock64 for the clock.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_issue_probe.hip -o tools/exp/mfma_issue_probe && ./tools/exp/mfma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mma(u4 a, u4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

template <int CB, int VB, bool READS, int PRIO>
__device__ __forceinline__ void tap_stream(const u4 *lds, int lane, int taps, float *sink, unsigned long long *cyc, int rnd) {
  f32x16 acc[CB][VB], cor[CB][VB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int vb = 0; vb < VB; ++vb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][vb][i] = cor[cb][vb][i] = 0.f;
  u4 wf[2][CB][2], xf[2][VB][2];
  constexpr int NR = 2 * VB + 2 * CB, NM = 3 * CB * VB;
  typedef __attribute__((address_space(3))) const u4 lds_u4;
  uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)reinterpret_cast<const unsigned char *>(lds) + (uint32_t)lane * 16u;
  asm volatile("" : "+v"(base));
  auto frag = [&](int s_, int r_, int tsel) {
    const int pc = r_ >= VB + CB, rr = pc ? r_ - (VB + CB) : r_;
    if (rr < VB) xf[s_][rr][pc] = *(lds_u4 *)(uintptr_t)(base + (uint32_t)(((tsel * 8 + rr * 2 + pc) * 64) * 16));
    else wf[s_][rr - VB][pc] = *(lds_u4 *)(uintptr_t)(base + (uint32_t)(((32 + tsel * 8 + (rr - VB) * 2 + pc) * 64) * 16));
  };
#pragma unroll
  for (int r_ = 0; r_ < NR; ++r_) frag(0, r_, 0);
  if (PRIO) __builtin_amdgcn_s_setprio(2);
  const unsigned long long t0 = clock64();
  const uint32_t base0 = base;
  for (int t2 = 0; t2 < taps; t2 += 2) {
    if (rnd) base = base0 + (uint32_t)((t2 * 1040) & 0x7ff0); // random-data runs: the fragments change from tap to tap
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int cur = tt, nxt = tt ^ 1;
      asm volatile("" : "+v"(base)); // the reads are not loop invariant for the compiler
      if (READS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        constexpr int PER = (NR + NM - 2) / (NM - 1);
        if (READS && m >= 1) {
#pragma unroll
          for (int r_ = (m - 1) * PER; r_ < m * PER && r_ < NR; ++r_) frag(nxt, r_, nxt);
        }
        const int kind = m / (CB * VB), cb = (m / VB) % CB, vb = m % VB;
        if (kind == 0) acc[cb][vb] = mma(wf[cur][cb][0], xf[cur][vb][0], acc[cb][vb]);
        else if (kind == 1) cor[cb][vb] = mma(wf[cur][cb][0], xf[cur][vb][1], cor[cb][vb]);
        else cor[cb][vb] = mma(wf[cur][cb][1], xf[cur][vb][0], cor[cb][vb]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = clock64();
  if (PRIO) __builtin_amdgcn_s_setprio(0);
  *cyc = t1 - t0;
  float s = 0.f;
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int vb = 0; vb < VB; ++vb)
#pragma unroll
      for (int i = 0; i < 16; ++i) s += acc[cb][vb][i] + cor[cb][vb][i];
  if (s == 1.2345e30f) *sink = s;
}

__device__ __forceinline__ void valu_stream(int iters, float *sink, int lane) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * (lane + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { // ~ the staging arithmetic: fma, exp, rcp, mul, cvt pair
      const float t = v[i] * 1.0001f + 0.5f;
      const float e = __builtin_amdgcn_exp2f(-t);
      const float r = __builtin_amdgcn_rcpf(1.f + e);
      const _Float16 h = (_Float16)(t * r);
      v[i] = (t * r - (float)h) * 2048.f + 0.25f;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 1.2345e30f) *sink = s;
}

// NW waves per workgroup (4 = one per SIMD, 8 = two per SIMD); waves >= 4 play PARTNER
template <int NW, int CB, int VB, bool READS, int PARTNER, int PRIO, int REGS_OCC>
__global__ __launch_bounds__(64 * NW, REGS_OCC) void probe(unsigned long long *out, float *sink, int taps, int valu_iters,
                                                          unsigned long long *wall, int rnd) {
  extern __shared__ __attribute__((aligned(16))) u4 lds[];
  for (int i = threadIdx.x; i < 96 * 64; i += blockDim.x) {
    const unsigned short h = 0x2c00 + (i & 7); // small fp16 numbers
    u4 v = u4{(unsigned)h | ((unsigned)h << 16), (unsigned)h | ((unsigned)h << 16), (unsigned)h | ((unsigned)h << 16), (unsigned)h | ((unsigned)h << 16)};
    if (rnd) { // pseudo-random fp16 of magnitude 2^-4 .. 2^0, both signs, full mantissas: the operand statistics of real data
      unsigned x = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
      for (int k = 0; k < 4; ++k) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const unsigned lo = (x & 0x83ffu) | ((11u + ((x >> 10) & 3u)) << 10);
        const unsigned hi = ((x >> 16) & 0x83ffu) | ((11u + ((x >> 26) & 3u)) << 10);
        v[k] = lo | (hi << 16);
      }
    }
    lds[i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long w0 = wall_clock64();
  unsigned long long cyc = 0;
  if (wave < 4 || PARTNER == 3) tap_stream<CB, VB, READS, PRIO>(lds, lane, taps, sink, &cyc, rnd);
  else if (PARTNER == 1) valu_stream(valu_iters, sink, lane);
  if (lane == 0) out[blockIdx.x * NW + wave] = cyc;
  if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) wall[0] = wall_clock64() - w0; // a workgroup from the middle of the launch
}

struct Res { double cyc_per_mfma_med, cyc_per_mfma_max, ms, tf; };

template <int NW, int CB, int VB, bool READS, int PARTNER, int PRIO, int REGS_OCC>
static void run(const char *name, int taps, int valu_iters, int rnd = 0) {
  const int blocks = 256 * 4;
  unsigned long long *out, *wall;
  float *sink;
  hipMalloc(&out, blocks * NW * 8);
  hipMalloc(&wall, 8);
  hipMalloc(&sink, 4);
  auto k = probe<NW, CB, VB, READS, PARTNER, PRIO, REGS_OCC>;
  const size_t LDS = 100 * 1024; // one workgroup per CU
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<<<blocks, 64 * NW, LDS>>>(out, sink, taps, valu_iters, wall, rnd); // warm
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<<<blocks, 64 * NW, LDS>>>(out, sink, taps, valu_iters, wall, rnd);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * NW);
  hipMemcpy(h.data(), out, blocks * NW * 8, hipMemcpyDeviceToHost);
  unsigned long long hw = 0;
  hipMemcpy(&hw, wall, 8, hipMemcpyDeviceToHost);
  std::vector<double> c;
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < NW; ++w)
      if (w < 4 || PARTNER == 3) c.push_back((double)h[b * NW + w] / ((double)taps * 3 * CB * VB));
  std::sort(c.begin(), c.end());
  const int mfma_waves = PARTNER == 3 ? NW : 4;
  const double flops = (double)blocks * mfma_waves * taps * 3 * CB * VB * 32768.0;
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k));
  printf("%-58s regs %3d scratch %3zu | cyc/MFMA/wave med %6.1f p10 %6.1f p90 %6.1f | %7.3f ms  %7.1f TF fp16 | blk0 wall %.1f us\n",
         name, fa.numRegs, (size_t)fa.localSizeBytes, c[c.size() / 2], c[c.size() / 10], c[c.size() * 9 / 10], ms,
         flops / (ms * 1e-3) / 1e12, hw / 100.0);
  // clock = s_memtime ticks of the middle workgroup's MFMA wave / its wall time (the tap stream is nearly its whole life)
  printf("%58s   in-kernel clock of a mid-launch workgroup: %.0f MHz\n", "", (double)h[(blocks / 2) * NW] / (hw / 100.0));
  hipFree(out);
  hipFree(wall);
  hipFree(sink);
}

// what bench.py calls (tools/exp/libmfma_probe.so, built by __graft_entry__.build()): the rate the matrix pipe sustains on
// the bare tap stream (1 wave per SIMD, 2 x 2 pattern with its fragment reads) for taps = 432 * mult taps per wave
extern "C" int mfma_ceiling(int random_operands, int mult, double *tf, double *mhz) {
  constexpr int NW = 4, CB = 2, VB = 2;
  const int blocks = 256 * 4, taps = 27 * 16 * (mult < 1 ? 1 : mult);
  unsigned long long *out, *wall;
  float *sink;
  if (hipMalloc(&out, blocks * NW * 8) != hipSuccess || hipMalloc(&wall, 8) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return -1;
  auto k = probe<NW, CB, VB, true, 0, 0, 1>;
  const size_t LDS = 100 * 1024;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS) != hipSuccess) return -2;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<<<blocks, 64 * NW, LDS>>>(out, sink, taps, 0, wall, random_operands); // warm: the clock settles
  hipEventRecord(e0);
  k<<<blocks, 64 * NW, LDS>>>(out, sink, taps, 0, wall, random_operands);
  hipEventRecord(e1);
  if (hipDeviceSynchronize() != hipSuccess) return -3;
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long hw = 0, c0 = 0;
  hipMemcpy(&hw, wall, 8, hipMemcpyDeviceToHost);
  hipMemcpy(&c0, out + (blocks / 2) * NW, 8, hipMemcpyDeviceToHost);
  *tf = (double)blocks * NW * taps * 3 * CB * VB * 32768.0 / (ms * 1e-3) / 1e12;
  *mhz = (double)c0 / (hw / 100.0);
  hipFree(out); hipFree(wall); hipFree(sink);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return 0;
}

#ifndef MFMA_PROBE_LIB
int main(int argc, char **argv) {
  const int mult = argc > 1 ? atoi(argv[1]) : 1;
  const int taps = 27 * 16 * mult; // 4 chunks of 27 taps x 4 (x mult: sustained runs show the clock the chip settles at)
  // valu partner iterations sized to last about as long as the MFMA wave: 16 values x ~7 VALU per iteration
  const int vi = 1500;
  printf("tap pattern CB x VB: per tap 3 CB VB MFMAs, 2 (CB + VB) ds_read_b128; one workgroup per CU, 1024 workgroups\n");
  if (argc > 2) { // random-operand runs only: what the matrix pipe sustains under the board's power cap on real-looking data
    run<4, 2, 2, true, 0, 0, 1>("1 wave/SIMD  2x2 + reads, RANDOM fp16 operands", taps, vi, 1);
    run<8, 2, 2, true, 3, 0, 1>("2 waves/SIMD 2x2 + reads, both MFMA, RANDOM fp16 operands", taps, vi, 1);
    run<4, 2, 4, true, 0, 0, 1>("1 wave/SIMD  2x4 + reads (512 regs), RANDOM fp16 operands", taps, vi, 1);
    run<4, 2, 2, true, 0, 0, 1>("1 wave/SIMD  2x2 + reads, constant operands (again)", taps, vi, 0);
    return 0;
  }
  run<4, 2, 2, false, 0, 0, 1>("1 wave/SIMD  2x2 bare", taps, vi);
  run<4, 2, 2, true, 0, 0, 1>("1 wave/SIMD  2x2 + reads", taps, vi);
  run<4, 2, 2, true, 0, 2, 1>("1 wave/SIMD  2x2 + reads prio2", taps, vi);
  run<4, 2, 4, false, 0, 0, 1>("1 wave/SIMD  2x4 bare (512 regs)", taps, vi);
  run<4, 2, 4, true, 0, 0, 1>("1 wave/SIMD  2x4 + reads (512 regs)", taps, vi);
  run<4, 4, 2, true, 0, 0, 1>("1 wave/SIMD  4x2 + reads (512 regs)", taps, vi);
  run<8, 2, 2, false, 3, 0, 1>("2 waves/SIMD 2x2 bare, both MFMA", taps, vi);
  run<8, 2, 2, true, 3, 0, 1>("2 waves/SIMD 2x2 + reads, both MFMA", taps, vi);
  run<8, 2, 2, false, 2, 0, 1>("2 waves/SIMD 2x2 bare, partner idle", taps, vi);
  run<8, 2, 2, true, 2, 0, 1>("2 waves/SIMD 2x2 + reads, partner idle", taps, vi);
  run<8, 2, 2, false, 1, 0, 1>("2 waves/SIMD 2x2 bare, partner VALU", taps, vi);
  run<8, 2, 2, true, 1, 0, 1>("2 waves/SIMD 2x2 + reads, partner VALU", taps, vi);
  run<8, 2, 2, true, 1, 2, 1>("2 waves/SIMD 2x2 + reads prio2, partner VALU", taps, vi);
  run<8, 1, 2, true, 3, 0, 1>("2 waves/SIMD 1x2 + reads, both MFMA", taps, vi);
  run<8, 1, 1, true, 3, 0, 1>("2 waves/SIMD 1x1 + reads, both MFMA", taps, vi);
  return 0;
}
#endif
