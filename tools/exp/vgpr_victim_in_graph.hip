// vgpr_victim_in_graph.hip -- EXPERIMENT (linked into tools/exp/liblion_victims.so only): a kernel that parks a known pattern
// in 192 VGPRs per lane, keeps them live through a long dependent VALU loop that maps the pattern onto itself, and counts
// the registers that changed.  Run beside a suspect kernel inside a captured graph (tools/_det7.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace {
__device__ unsigned g_vv_bad[256];
constexpr int NR = 176;
__global__ __launch_bounds__(256) void vgpr_victim(int rounds, unsigned seed) {
  unsigned r[NR];
  const unsigned t = threadIdx.x + blockIdx.x * 256u;
#pragma unroll
  for (int i = 0; i < NR; ++i) r[i] = (seed + t) * 2654435761u + (unsigned)i * 40503u;
  for (int j = 0; j < rounds; ++j) {
    // every register is read and rewritten with its own value through an opaque identity (x ^ k ^ k, k changes per round)
    unsigned k = (unsigned)j * 0x9e3779b9u;
    asm volatile("" : "+v"(k));
#pragma unroll
    for (int i = 0; i < NR; ++i) { r[i] ^= k; asm volatile("" : "+v"(r[i])); r[i] ^= k; }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < NR; ++i)
    if (r[i] != (seed + t) * 2654435761u + (unsigned)i * 40503u) atomicAdd(&g_vv_bad[i], 1u);
  if (threadIdx.x == 0) atomicAdd(&g_vv_bad[255], 1u);
}
} // namespace
extern "C" int lion_debug_vgpr_victim(int B, int rounds, unsigned seed, void *stream) {
  vgpr_victim<<<B, 256, 0, static_cast<hipStream_t>(stream)>>>(rounds, seed);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int lion_debug_vgpr_victim_read(unsigned *h256, int reset) {
  if (hipMemcpyFromSymbol(h256, HIP_SYMBOL(g_vv_bad), 1024) != hipSuccess) return -1;
  if (reset) { unsigned z[256] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_vv_bad), z, 1024) != hipSuccess) return -1; }
  return 0;
}
