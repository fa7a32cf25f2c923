// ref_shim.h -- a tiny CPU execution model for CUDA __global__ functions (TEST INFRASTRUCTURE).
//
// oracle/ref_build.py extracts the *unmodified text* of the reference's __global__ kernels from
// /root/reference at build time into oracle/_ref/gen/kernels.inc (git-ignored, never committed) and
// compiles them with g++ against this header: blocks run one after another, the threads of a block
// are cooperative fibers (ucontext) that yield at __syncthreads(), __shared__ variables are statics,
// atomics are plain read-modify-writes (one thread runs at a time).  That executes the reference's own
// kernel bodies -- index arithmetic, tie-breaks, tiling, reduction trees -- on the CPU, which is what
// pins oracle/lion_oracle.c ("outputs of the reference itself run here").  Float expressions are
// evaluated without FMA contraction (g++ -ffp-contract=off), i.e. the C++ semantics of the source.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <ucontext.h>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint3_ { unsigned x, y, z; };
static uint3_ threadIdx, blockIdx;
static dim3 blockDim, gridDim;

#define __global__
#define __device__
#define __restrict__
#define __shared__ static

inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline float min(float a, float b) { return fminf(a, b); }
inline float max(float a, float b) { return fmaxf(a, b); }
inline double min(double a, double b) { return fmin(a, b); }
inline double max(double a, double b) { return fmax(a, b); }
inline double min(float a, double b) { return fmin((double)a, b); }
inline double max(double a, float b) { return fmax(a, (double)b); }
inline double min(double a, float b) { return fmin(a, (double)b); }
inline double max(float a, double b) { return fmax((double)a, b); }
inline float __expf(float x) { return expf(x); }
template <typename T> inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }

namespace refshim {
struct Fiber {
  ucontext_t ctx;
  std::vector<char> stack;
  bool done = false;
};
static ucontext_t g_main;
static Fiber *g_cur = nullptr;
static std::function<void()> g_body;
static void trampoline() {
  g_body();
  g_cur->done = true;
  swapcontext(&g_cur->ctx, &g_main);
}
inline void yield_barrier() { swapcontext(&g_cur->ctx, &g_main); }

// run `body` for every thread of every block of the grid
inline void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
  gridDim = grid;
  blockDim = block;
  const unsigned nt = block.x * block.y * block.z;
  std::vector<Fiber> fibers(nt);
  for (auto &f : fibers) f.stack.resize(96 * 1024);
  g_body = body;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        for (unsigned t = 0; t < nt; ++t) {
          Fiber &f = fibers[t];
          f.done = false;
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = f.stack.data();
          f.ctx.uc_stack.ss_size = f.stack.size();
          f.ctx.uc_link = &g_main;
          makecontext(&f.ctx, trampoline, 0);
        }
        bool any = true;
        while (any) { // one pass = every live thread runs up to its next __syncthreads()
          any = false;
          for (unsigned t = 0; t < nt; ++t) {
            Fiber &f = fibers[t];
            if (f.done) continue;
            blockIdx = {bx, by, bz};
            threadIdx = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
            g_cur = &f;
            swapcontext(&g_main, &f.ctx);
            any = any || !f.done;
          }
        }
      }
}
} // namespace refshim

inline void __syncthreads() { refshim::yield_barrier(); }
#define REF_LAUNCH(grid, block, call) refshim::launch((grid), (block), [&]() { call; })
