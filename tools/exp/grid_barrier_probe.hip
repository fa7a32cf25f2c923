// grid_barrier_probe.hip -- what would a persistent PVConv pay per stage boundary?  ("persistent r <= 16
// PVConv"; round 4.)  A stage boundary inside one launch is a grid-wide barrier with device-scope release / acquire (the
// next stage reads what other CUs -- on other XCDs, behind other L2s -- wrote); between launches it is the kernel boundary
// of a hipGraph replay.  This probe measures both on the same box:
//   (a) K dependent tiny kernels replayed from one hipGraph           -> us per kernel boundary
//   (b) ONE launch of G workgroups (one per CU) doing K rounds of {touch a 1 MiB buffer, grid barrier}: a monotone
//       counter in global memory, release fence + atomic add, spin on an acquire load                 -> us per barrier
//   (c) the same with the buffer traffic removed (barrier only)
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/grid_barrier_probe.hip -o tools/exp/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void tiny_kernel(float *buf, int n, int it) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] = buf[(i + 4099) % n] * 0.5f + (float)it;
}

__device__ __forceinline__ void grid_barrier(unsigned *counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();                       // release: this workgroup's stores are visible device-wide
    atomicAdd(counter, 1u);
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

template <bool TOUCH>
__global__ __launch_bounds__(256) void persistent_kernel(float *buf, int n, int K, unsigned *counter) {
  const int G = gridDim.x;
  for (int it = 0; it < K; ++it) {
    if (TOUCH) {
      for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += G * 256) buf[i] = buf[(i + 4099) % n] * 0.5f + (float)it;
    }
    grid_barrier(counter, (unsigned)(it + 1) * G);
  }
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int G = prop.multiProcessorCount, n = 1 << 18, K = 200;
  float *buf; unsigned *counter;
  CK(hipMalloc(&buf, n * 4)); CK(hipMemset(buf, 0, n * 4));
  CK(hipMalloc(&counter, 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float ms;
  // (a) graph of K dependent tiny kernels
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int it = 0; it < K; ++it) tiny_kernel<<<n / 256, 256, 0, st>>>(buf, n, it);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
  CK(hipEventElapsedTime(&ms, a, b));
  printf("(a) hipGraph of %d dependent 1-MiB kernels: %.2f us per kernel boundary (kernel + boundary)\n", K, ms * 1e3 / K);
  // (b), (c) persistent kernel with grid barriers
  for (int touch = 1; touch >= 0; --touch) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemsetAsync(counter, 0, 4, st));
      CK(hipEventRecord(a, st));
      if (touch) persistent_kernel<true><<<G, 256, 0, st>>>(buf, n, K, counter);
      else persistent_kernel<false><<<G, 256, 0, st>>>(buf, n, K, counter);
      CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
      CK(hipEventElapsedTime(&ms, a, b));
    }
    printf("(%c) one launch, %d workgroups, %d rounds of {%s grid barrier}: %.2f us per round\n", touch ? 'b' : 'c', G, K,
           touch ? "touch 1 MiB," : "", ms * 1e3 / K);
  }
  return 0;
}
