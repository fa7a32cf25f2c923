// residency_probe.hip -- how many 256-thread workgroups with a given LDS request does a CU of gfx950 hold at once?
// (conv3d_split_kernel asks for 80,908 bytes and is built for two workgroups per CU: 2 x 80 KiB = the whole 160 KiB.)
// Every workgroup notes its (XCC, SE, CU) from the hardware id registers and its start / end time; the host counts the
// maximum number of workgroups whose lifetimes overlap on one CU.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/residency_probe.hip -o tools/exp/residency_probe && ./tools/exp/residency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>
struct Rec { unsigned long long t0, t1; unsigned hwid, xcc; };
__global__ __launch_bounds__(256, 2) void spin(Rec *out, int spin_us) {
  extern __shared__ unsigned char lds[];
  __shared__ int s_pad[3]; // the product kernel's 12 static bytes
  const unsigned long long t0 = wall_clock64();
  lds[threadIdx.x] = (unsigned char)threadIdx.x;
  s_pad[threadIdx.x % 3] = 1;
  while (wall_clock64() - t0 < (unsigned long long)spin_us * 100ull) __builtin_amdgcn_s_sleep(32);
  __syncthreads();
  if (threadIdx.x == 0) {
    Rec r;
    r.t0 = t0; r.t1 = wall_clock64();
    r.hwid = __builtin_amdgcn_s_getreg(4 | (31 << 11));  // HW_REG_HW_ID
    r.xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));  // HW_REG_XCC_ID
    out[blockIdx.x] = r;
  }
  if (lds[5] == 77 && s_pad[0] == 5) out[0].t0 = 0;
}
int main() {
  const int blocks = 2048;
  Rec *d; hipMalloc(&d, blocks * sizeof(Rec));
  for (int lds : {16384, 65536, 78000, 80896, 81408, 81908, 81920, 83000, 100000}) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    int occ = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin, 256, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    spin<<<blocks, 256, lds>>>(d, 50);
    hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("lds %d: launch failed\n", lds); continue; }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Rec> h(blocks); hipMemcpy(h.data(), d, blocks * sizeof(Rec), hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
    for (auto &r : h) {
      const unsigned cu = (r.hwid >> 8) & 0xf, sh = (r.hwid >> 12) & 1, se = (r.hwid >> 13) & 7;
      const unsigned long long key = ((unsigned long long)(r.xcc & 0xf) << 16) | (se << 8) | (sh << 4) | cu;
      ev[key].push_back({r.t0, +1}); ev[key].push_back({r.t1, -1});
    }
    int mx = 0; 
    for (auto &kv : ev) { auto &v = kv.second; std::sort(v.begin(), v.end()); int c = 0; for (auto &e : v) { c += e.second; mx = std::max(mx, c); } }
    printf("dynamic LDS %6d (+12 static): occupancy API %d workgroups/CU | distinct CUs seen %zu | max concurrent on one CU %d | %d x 50 us workgroups took %.3f ms (=> %.1f resident per CU on 256 CUs)\n",
           lds, occ, ev.size(), mx, blocks, ms, blocks * 0.05 / ms / 256.0);
  }
  return 0;
}
