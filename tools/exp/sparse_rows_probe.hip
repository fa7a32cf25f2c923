// sparse_rows_probe.hip -- how fast can the chip read a sparse set of 128-byte rows?  (devoxelize at r = 32 needs
// ~43 % of the 1024 z-rows of each [32^3] channel grid; both the predicated slab kernel and the compacted-rows
// kernel move ~143 MB in ~50 us = 2.9 TB/s, no faster than streaming all 285 MB at 5.5 TB/s.)
// Each workgroup reads the needed rows of one (cloud, channel) grid with plain 16-byte loads, all of a lane's loads
// in flight (registers as landing zone, several workgroups per CU): the memory system's rate for this access pattern,
// free of LDS-DMA / barrier effects.  Variants: row fraction, run length (rows needed in runs of L consecutive rows).
// hipcc --offload-arch=gfx950 -O3 tools/exp/sparse_rows_probe.hip -o tools/exp/sparse_rows_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NL>
__global__ __launch_bounds__(256) void read_rows(const float4 *__restrict__ grid, const unsigned short *__restrict__ rows,
                                                 const int *__restrict__ nrows, int C, float *__restrict__ out) {
  const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
  const float4 *g = grid + ((size_t)b * C + c) * 8192; // 32^3 floats = 8192 float4
  const unsigned short *rl = rows + b * 1024;
  const int n4 = nrows[b] * 8;
  float4 v[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int f = tid + j * 256;
    v[j] = make_float4(0, 0, 0, 0);
    if (f < n4) v[j] = g[(int)rl[f >> 3] * 8 + (f & 7)];
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NL; ++j) s += v[j].x + v[j].y + v[j].z + v[j].w;
  if (s == 123.456f) out[blockIdx.x] = s;
}

int main() {
  const int B = 32, C = 64;
  float4 *grid; float *out; unsigned short *rows; int *nrows;
  CK(hipMalloc(&grid, (size_t)B * C * 8192 * 16)); CK(hipMemset(grid, 0, (size_t)B * C * 8192 * 16));
  CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&rows, B * 1024 * 2)); CK(hipMalloc(&nrows, B * 4));
  // second, equally large buffer read in between to flush the 256 MB Infinity Cache
  float4 *flush; CK(hipMalloc(&flush, (size_t)B * C * 8192 * 16)); CK(hipMemset(flush, 0, (size_t)B * C * 8192 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int run : {1, 2, 4, 8, 32}) {
    for (double frac : {0.25, 0.43, 0.60, 1.0}) {
      std::vector<unsigned short> hr(B * 1024); std::vector<int> hn(B);
      srand(7);
      for (int b = 0; b < B; ++b) {
        int n = 0;
        for (int r0 = 0; r0 < 1024; r0 += run)
          if (frac >= 1.0 || (rand() / (double)RAND_MAX) < frac)
            for (int k = 0; k < run; ++k) hr[b * 1024 + n++] = (unsigned short)(r0 + k);
        hn[b] = n;
      }
      CK(hipMemcpy(rows, hr.data(), hr.size() * 2, hipMemcpyHostToDevice));
      CK(hipMemcpy(nrows, hn.data(), B * 4, hipMemcpyHostToDevice));
      double bytes = 0; for (int b = 0; b < B; ++b) bytes += (double)hn[b] * 128 * C;
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        std::vector<int> full(B, 1024);
        // flush: stream the other buffer
        int *nfull; CK(hipMalloc(&nfull, B * 4)); CK(hipMemcpy(nfull, full.data(), B * 4, hipMemcpyHostToDevice));
        unsigned short *rfull; CK(hipMalloc(&rfull, B * 1024 * 2));
        std::vector<unsigned short> id(B * 1024); for (int i = 0; i < B * 1024; ++i) id[i] = i & 1023;
        CK(hipMemcpy(rfull, id.data(), id.size() * 2, hipMemcpyHostToDevice));
        read_rows<32><<<dim3(C, B), 256>>>(flush, rfull, nfull, C, out);
        CK(hipEventRecord(e0));
        read_rows<32><<<dim3(C, B), 256>>>(grid, rows, nrows, C, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        CK(hipFree(nfull)); CK(hipFree(rfull));
      }
      printf("run %2d rows  frac %.2f  %7.1f MB  %6.1f us  %5.2f TB/s\n", run, frac, bytes / 1e6, best * 1e3, bytes / best / 1e9);
    }
  }
  return 0;
}
