// lds_b128_probe.hip -- which lane -> address patterns of ds_read_b128 are bank-conflict free on gfx950?
// One wave reads 16 bytes per lane from LDS slot pattern[lane] (in 16-byte slots) in a dependent-free loop; prints
// shader clocks per read.  hipcc --offload-arch=gfx950 -O3 tools/exp/lds_b128_probe.hip -o tools/exp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ void probe(const int *pat, unsigned long long *out, u4 *sink, int iters) {
  extern __shared__ u4 lds[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = u4{(unsigned)i, 0u, 0u, 0u};
  __syncthreads();
  const unsigned addr = (unsigned)pat[threadIdx.x & 63] * 16u; // byte address; the 8 reads of a round are 1 KiB * 8 apart
  u4 a = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    u4 v0, v1, v2, v3, v4, v5, v6, v7;
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:8192\n\tds_read_b128 %2, %8 offset:16384\n\t"
                 "ds_read_b128 %3, %8 offset:24576\n\tds_read_b128 %4, %8 offset:32768\n\tds_read_b128 %5, %8 offset:40960\n\t"
                 "ds_read_b128 %6, %8 offset:49152\n\tds_read_b128 %7, %8 offset:57344\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(addr) : "memory");
    a += v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  sink[threadIdx.x] = a;
  if (threadIdx.x == 0) out[0] = t1 - t0;
}
int main() {
  struct P { const char *name; std::vector<int> v; };
  std::vector<P> ps;
  auto mk = [&](const char *n, auto f) { P p{n, std::vector<int>(64)}; for (int l = 0; l < 64; ++l) p.v[l] = f(l); ps.push_back(p); };
  mk("consecutive slots", [](int l) { return l; });
  mk("all lanes same slot (broadcast)", [](int) { return 5; });
  mk("stride 16 slots (256 B): worst case", [](int l) { return l * 16; });
  mk("lanes 32+ at +640 slots (plane offset = 0 mod 256 B)", [](int l) { return (l & 31) + (l >> 5) * 640; });
  mk("lanes 32+ at +648 slots (plane offset = 128 mod 256 B)", [](int l) { return (l & 31) + (l >> 5) * 648; });
  mk("8-wide rows, stride 10 (h, h+1, h+2, h+3), planes +640", [](int l) { return (l & 7) + ((l >> 3) & 3) * 10 + (l >> 5) * 640; });
  mk("8-wide rows h, h+4, h+1, h+5 (stride 10), planes +640", [](int l) { return (l & 7) + (((l >> 4) & 1) + 4 * ((l >> 3) & 1)) * 10 + (l >> 5) * 640; });
  mk("8-wide rows h, h+4, h+1, h+5 (stride 10), planes +648", [](int l) { return (l & 7) + (((l >> 4) & 1) + 4 * ((l >> 3) & 1)) * 10 + (l >> 5) * 648; });
  mk("8-wide rows h, h+2, h+4, h+6 (stride 12), planes +768", [](int l) { return (l & 7) + ((l >> 3) & 3) * 24 + (l >> 5) * 768; });
  mk("8-wide rows stride 10, planes +644", [](int l) { return (l & 7) + ((l >> 3) & 3) * 10 + (l >> 5) * 644; });
  mk("8-wide rows stride 12, planes +640", [](int l) { return (l & 7) + ((l >> 3) & 3) * 12 + (l >> 5) * 640; });
  mk("8-wide rows stride 12, planes +648", [](int l) { return (l & 7) + ((l >> 3) & 3) * 12 + (l >> 5) * 648; });
  mk("8-wide rows stride 8 (dense), planes +640", [](int l) { return (l & 7) + ((l >> 3) & 3) * 8 + (l >> 5) * 640; });
  mk("16-wide rows stride 18, planes +704", [](int l) { return (l & 15) + ((l >> 4) & 1) * 18 + (l >> 5) * 704; });
  mk("16-wide rows stride 18, planes +712", [](int l) { return (l & 15) + ((l >> 4) & 1) * 18 + (l >> 5) * 712; });
  mk("32-wide row, planes +832", [](int l) { return (l & 31) + (l >> 5) * 832; });
  mk("32-wide row, planes +840", [](int l) { return (l & 31) + (l >> 5) * 840; });
  int *dp; unsigned long long *dout; u4 *sink;
  (void)hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  (void)hipMalloc(&dp, 256); (void)hipMalloc(&dout, 8); (void)hipMalloc(&sink, 16384);
  const int iters = 2000;
  for (int nthreads = 64; nthreads <= 512; nthreads *= 2)
  for (auto &p : ps) {
    if (nthreads > 64 && &p != &ps[0] && &p != &ps[6] && &p != &ps[5] && &p != &ps[8]) continue;
    (void)hipMemcpy(dp, p.v.data(), 256, hipMemcpyHostToDevice);
    unsigned long long best = ~0ull;
    for (int rep = 0; rep < 3; ++rep) {
      probe<<<1, nthreads, 131072>>>(dp, dout, sink, iters);
      unsigned long long c; (void)hipMemcpy(&c, dout, 8, hipMemcpyDeviceToHost);
      if (c < best) best = c;
    }
    printf("%d waves: %-62s %6.2f clocks / ds_read_b128 per wave -> %.1f B/clk\n", nthreads / 64, p.name, (double)best / (iters * 8.0), nthreads * 16.0 * iters * 8.0 / (double)best);
  }
  return 0;
}
