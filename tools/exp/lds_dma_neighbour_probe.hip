// lds_dma_neighbour_probe.hip -- can global_load_lds_dwordx4 of one workgroup damage the LDS of ANOTHER workgroup resident on
// the same CU?  Victim: fills its LDS with a pattern, spins, re-checks.  Aggressor (other stream): LDS-DMA into its own
// allocation at offsets [lo, hi).  Prints corrupted words seen by the victims.
// hipcc --offload-arch=gfx950 -O3 tools/exp/lds_dma_neighbour_probe.hip -o tools/exp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ void victim(unsigned *bad, int words, long spin) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = 0xA5000000u + i;
  __syncthreads();
  unsigned n = 0;
  for (long it = 0; it < spin; ++it) {
    for (int i = threadIdx.x; i < words; i += blockDim.x) n += lds[i] != 0xA5000000u + i;
    __syncthreads();
  }
  if (n) atomicAdd(bad, n);
}
// victim 2: the access pattern of the FPS kernel -- broadcast 16-byte reads of a data-dependent slot, raised wave priority
__global__ void victim_b128(unsigned *bad, int slots, long spin) {
  extern __shared__ __attribute__((aligned(16))) float4 l4[];
  __builtin_amdgcn_s_setprio(3);
  for (int i = threadIdx.x; i < slots; i += blockDim.x) l4[i] = make_float4((float)i, (float)(i + 1), (float)(i + 2), 0.f);
  __syncthreads();
  unsigned n = 0;
  int slot = 0;
  for (long it = 0; it < spin; ++it) {
    const float4 v = l4[slot];                       // every lane reads the same slot
    n += !(v.x == (float)slot && v.y == (float)(slot + 1) && v.z == (float)(slot + 2));
    slot = (slot * 5 + 7 + (int)v.w) % slots;
    __syncthreads();
  }
  if (n) atomicAdd(bad, n);
}
// victim 3: FPS's round structure -- lane 0 of every wave publishes a 64-bit key, ONE barrier, every thread reads the 4 keys
// (double buffered by round parity); a stale key of round j - 2 is a failure
__global__ void victim_publish(unsigned *bad, int unused, long spin) {
  __shared__ unsigned long long wkey[2][4];
  extern __shared__ __attribute__((aligned(16))) float4 l4[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  l4[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  unsigned n = 0;
  for (long j = 1; j < spin; ++j) {
    if (lane == 0) wkey[j & 1][wave] = ((unsigned long long)j << 8) | (unsigned)wave;
    __syncthreads();
    const unsigned long long k0 = wkey[j & 1][0], k1 = wkey[j & 1][1], k2 = wkey[j & 1][2], k3 = wkey[j & 1][3];
    n += (k0 != (((unsigned long long)j << 8) | 0)) + (k1 != (((unsigned long long)j << 8) | 1)) +
         (k2 != (((unsigned long long)j << 8) | 2)) + (k3 != (((unsigned long long)j << 8) | 3));
  }
  if (n) atomicAdd(bad, n);
}
__global__ void aggressor(const u4 *src, unsigned *sink, int lo, int hi, int iters, int dynamic_bytes) {
  extern __shared__ unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const uint32_t base = (uint32_t)(uintptr_t)(lds_byte *)smem;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    for (int off = lo + wave * 1024; off + 1024 <= hi; off += 4096) {
      const u4 *gp = src + (size_t)((it * 64 + off / 16 + lane) & 4095);
      const uint32_t dst = __builtin_amdgcn_readfirstlane(base + (uint32_t)off);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gp), "s"(dst) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc += reinterpret_cast<unsigned *>(smem)[(lo / 4 + threadIdx.x) % (dynamic_bytes / 4)];
    __syncthreads();
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  u4 *src; unsigned *sink, *bad;
  (void)hipMalloc(&src, 4096 * 16); (void)hipMemset(src, 0x11, 4096 * 16);
  (void)hipMalloc(&sink, 1 << 22); (void)hipMalloc(&bad, 4);
  hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
  (void)hipFuncSetAttribute((const void *)victim, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void *)victim_b128, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void *)victim_publish, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void *)aggressor, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  struct C { const char *name; int vbytes, abytes, lo, hi; };
  const C cases[] = {
    {"aggressor 66 KiB alloc, DMA into [52K, 64K)", 32 * 1024, 65 * 1024 + 1152, 53248, 65536},
    {"aggressor 66 KiB alloc, DMA into [0, 16K)", 32 * 1024, 65 * 1024 + 1152, 0, 16384},
    {"aggressor 32 KiB alloc, DMA into [0, 32K)", 32 * 1024, 32 * 1024, 0, 32768},
    {"aggressor 140 KiB alloc, DMA into [80K, 136K)", 16 * 1024, 140 * 1024, 81920, 139264},
    {"no DMA at all (hi = lo)", 32 * 1024, 65 * 1024 + 1152, 0, 0},
  };
  for (int mode = 0; mode < 3; ++mode)
  for (const C &c : cases) {
    (void)hipMemset(bad, 0, 4);
    (void)hipDeviceSynchronize();
    if (mode == 0) victim<<<256, 256, c.vbytes, s1>>>(bad, c.vbytes / 4, 3000);
    else if (mode == 1) victim_b128<<<256, 256, c.vbytes, s1>>>(bad, c.vbytes / 16, 300000);
    else victim_publish<<<256, 256, c.vbytes, s1>>>(bad, 0, 300000);
    for (int r = 0; r < 40; ++r) aggressor<<<512, 256, c.abytes, s2>>>(src, sink, c.lo, c.hi, 40, c.abytes);
    (void)hipDeviceSynchronize();
    unsigned h; (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("%s %-52s victim LDS %3d KiB: corrupted word reads = %u\n", mode == 2 ? "[publish/barrier/read]" : mode ? "[b128 bcast, prio 3]" : "[b32 sweep]", c.name, c.vbytes / 1024, h);
  }
  return 0;
}
