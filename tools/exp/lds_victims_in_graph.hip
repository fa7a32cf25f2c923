// debug_victims.hip -- EXPERIMENT ONLY (built into tools/exp/liblion_victims.so, never into liblion_hip.so): LDS self-checking
// kernels to run beside a suspect kernel inside a captured graph (tools/_det6.py).
#include "common.h"
namespace {
__device__ unsigned g_victim_bad[8];
// mode 0: broadcast 16-byte reads of a data-dependent slot; mode 1: publish / barrier / read with round tags
__global__ __launch_bounds__(256) void victim_kernel(int slots, int rounds, int mode) {
  extern __shared__ __attribute__((aligned(16))) float4 l4[];
  __shared__ uint4 wkey[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < slots; i += 256) l4[i] = make_float4((float)i, (float)(i + 1), (float)(i + 2), 0.f);
  __syncthreads();
  unsigned nread = 0, nstale = 0, nfinal = 0;
  int slot = 0;
  for (int j = 1; j < rounds; ++j) {
    const float4 v = l4[slot];
    nread += !(v.x == (float)slot && v.y == (float)(slot + 1) && v.z == (float)(slot + 2));
    slot = (slot * 5 + 7 + j) % slots;
    if (mode == 1) {
      if (lane == 0) wkey[j & 1][wave] = make_uint4((unsigned)wave, 0x1234u, (unsigned)j, 0u);
      __syncthreads();
      const uint4 e0 = wkey[j & 1][0], e1 = wkey[j & 1][1], e2 = wkey[j & 1][2], e3 = wkey[j & 1][3];
      nstale += (e0.z != (unsigned)j) + (e1.z != (unsigned)j) + (e2.z != (unsigned)j) + (e3.z != (unsigned)j);
    }
  }
  for (int i = tid; i < slots; i += 256) { const float4 v = l4[i]; nfinal += !(v.x == (float)i && v.y == (float)(i + 1)); }
  if (nread) atomicAdd(&g_victim_bad[0], nread);
  if (nstale) atomicAdd(&g_victim_bad[1], nstale);
  if (nfinal) atomicAdd(&g_victim_bad[2], nfinal);
  if (tid == 0) atomicAdd(&g_victim_bad[3], 1u);
}
} // namespace
extern "C" int lion_debug_victim(int B, int slots, int rounds, int mode, lionStream_t stream) {
  victim_kernel<<<B, 256, (size_t)slots * 16, static_cast<hipStream_t>(stream)>>>(slots, rounds, mode);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int lion_debug_victim_read(unsigned *h8, int reset) {
  if (hipMemcpyFromSymbol(h8, HIP_SYMBOL(g_victim_bad), 32) != hipSuccess) return -1;
  if (reset) { unsigned z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_victim_bad), z, 32) != hipSuccess) return -1; }
  return 0;
}
