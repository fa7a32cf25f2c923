// fold.h -- GroupNorm fold (+ SE gate) in the TAIL of the kernel that produced the tile sums (round 6).
//
// A layer with GroupNorm statistics used to be two launches: the convolution writes per-tile channel sums, then
// gn_fold_kernel / gn_fold_se_kernel (one small workgroup per sample, ~5-8 us of launch latency for a few kflop) turns
// them into the per-(sample, channel) affine the next kernel applies in its prologue: 61 of the ~230 launches of a
// denoiser step.  Here the workgroup that finishes the LAST work item of a sample does that sample's fold itself:
//   * every workgroup writes its tile sums with agent-scope (sc1, write-through) stores, waits for them to complete
//     (s_waitcnt vmcnt(0): memory operations of a wave retire in order) and then bumps the sample's arrival counter with a
//     relaxed agent-scope atomic -- no release fence: a release at agent scope would write back every dirty line of this
//     XCD's L2 (the convolution's own output), which is what made last-arriver schemes 2-3 x slower when tried with
//     __threadfence() in round 4;
//   * the workgroup that reads items - 1 from the counter owns the sample: it reads all tile sums with agent-scope loads
//     (they bypass the non-coherent lines of its own L2), folds them with EXACTLY the arithmetic and summation order of
//     gn_fold_se_kernel (csrc/conv3d.hip; double accumulation, fixed order -- deterministic whichever workgroup arrives
//     last), writes A / Bs with plain stores (visible to the next kernel at the kernel boundary) and re-arms the counter.
// Counters: int32[B], zero before the first launch, zero again after every launch (the caller keeps one buffer per layer).
#pragma once
#include "common.h"

struct LionFold {            // by-value kernel argument built from lion_fold_t (include/lion_hip.h); counters == nullptr: off
  int32_t *counters;
  float *A, *Bs;             // [B, C] outputs
  const float *gamma, *beta; // GroupNorm affine [C]
  const float *fac, *gbias;  // AdaGN style factor / bias [B, ld_fg] views
  const float *w1, *w2;      // SE3d gate (w1 [H, C], w2 [C, H]) or nullptr
  int C, T, G, ld_fg, H, items;
  float count, eps;
};
constexpr int LION_FOLD_SCRATCH = 16 + 256 * 2 * 8 + 3 * 256 * 4 + 128 * 4;   // bytes of LDS lion_fold_sample needs

static inline LionFold lion_make_fold(const lion_fold_t *f, int B, int C, int T, int items) {
  LionFold k = {};
  if (!f || !f->counters) return k;
  k.counters = f->counters; k.A = f->A; k.Bs = f->Bs; k.gamma = f->gamma; k.beta = f->beta; k.fac = f->fac;
  k.gbias = f->gbias; k.w1 = f->w1; k.w2 = f->w2; k.C = C; k.T = T; k.G = f->G; k.ld_fg = f->ld_fg; k.H = f->H;
  k.items = items; k.count = (float)f->count; k.eps = f->eps;
  return k;
}
static inline int lion_check_fold(const lion_fold_t *f, int C) {
  if (!f || !f->counters) return 0;
  if (!f->A || !f->Bs || !f->gamma || !f->beta || !f->fac || !f->gbias || f->G <= 0 || C % f->G != 0 || f->ld_fg < C ||
      f->count <= 0)
    return LION_EINVAL;
  if ((f->w1 == nullptr) != (f->w2 == nullptr) || (f->w1 && (f->H <= 0 || f->H > 128))) return LION_EINVAL;
  if (C > 256 || C < 4) return LION_EUNSUPPORTED;
  return 0;
}

__device__ __forceinline__ void lion_fold_store2(float *o, float s1, float s2, bool coherent) {
  if (coherent) {
    __hip_atomic_store(o, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(o + 1, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    o[0] = s1;
    o[1] = s2;
  }
}

// the fold of sample b by one workgroup (all its threads must call; the first 256 work).  stats f32[B][C][T][2].
__device__ __forceinline__ void lion_fold_sample(const LionFold &f, int b, const float *stats, unsigned char *scratch) {
  double(*cs)[2] = reinterpret_cast<double(*)[2]>(scratch + 16);
  float *sA = reinterpret_cast<float *>(scratch + 16 + 256 * 2 * 8), *sB = sA + 256, *sm = sB + 256, *sh = sm + 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, C = f.C, T = f.T, cpg = C / f.G;
  int cp2 = 1;
  while (cp2 < C) cp2 <<= 1;
  const int LPC = 256 / cp2, ch = tid / LPC, sub = tid % LPC; // LPC in {1, ..., 64}
  if (tid < 256) {
    double s1 = 0.0, s2 = 0.0;
    if (ch < C) {
      const unsigned long long *p = reinterpret_cast<const unsigned long long *>(stats + (((size_t)b * C + ch) * T) * 2);
      // the loads bypass this XCD's L2 (~1-2 us each): eight in flight, then summed in the fixed order t = sub, sub + LPC, ...
      for (int t0 = sub; t0 < T; t0 += 8 * LPC) {
        unsigned long long v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          v[k] = t0 + k * LPC < T ? __hip_atomic_load(p + t0 + k * LPC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (t0 + k * LPC < T) {
            s1 += (double)__uint_as_float((unsigned)v[k]);
            s2 += (double)__uint_as_float((unsigned)(v[k] >> 32));
          }
      }
    }
    for (int m = 1; m < LPC; m <<= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
    if (ch < C && sub == 0) { cs[ch][0] = s1; cs[ch][1] = s2; }
  }
  __syncthreads();
  if (tid < C) {
    const int g0 = (tid / cpg) * cpg;
    double g1 = 0.0, g2 = 0.0;
    for (int k = 0; k < cpg; ++k) { g1 += cs[g0 + k][0]; g2 += cs[g0 + k][1]; }
    const double n = (double)f.count * cpg, mean = g1 / n;
    double var = g2 / n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)f.eps));
    const float fa = f.fac[(size_t)b * f.ld_fg + tid], gb = f.gbias[(size_t)b * f.ld_fg + tid];
    const float a0 = rstd * f.gamma[tid];
    const float a = a0 * fa, bb = (f.beta[tid] - (float)mean * a0) * fa + gb;
    sA[tid] = a;
    sB[tid] = bb;
    sm[tid] = a * (float)(cs[tid][0] / f.count) + bb; // mean over the grid of AdaGN(y)
    if (!f.w1) {
      f.A[(size_t)b * C + tid] = a;
      f.Bs[(size_t)b * C + tid] = bb;
    }
  }
  if (!f.w1) return;    // (block-uniform)
  __syncthreads();
  if (tid < 256)
    for (int j = wave; j < f.H; j += 4) { // one wave per hidden unit: coalesced row of W1
      float acc = 0.f;
      for (int c = lane; c < C; c += 64) acc += f.w1[(size_t)j * C + c] * sm[c];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
      if (lane == 0) sh[j] = acc > 0.f ? acc : 0.f;
    }
  __syncthreads();
  if (tid < C) {
    float acc = 0.f;
    for (int j = 0; j < f.H; ++j) acc += f.w2[(size_t)tid * f.H + j] * sh[j];
    const float g = 1.0f / (1.0f + expf(-acc));
    f.A[(size_t)b * C + tid] = sA[tid] * g;
    f.Bs[(size_t)b * C + tid] = sB[tid] * g;
  }
}

// End of a work item of sample b.  ALL threads of the workgroup call it after the item's tile sums were stored (coherently)
// by threads of `writer_waves` leading waves; `flag` is one int of LDS, `scratch` LION_FOLD_SCRATCH bytes of LDS that
// nothing else uses until the call returns.  Returns after a barrier.
__device__ __forceinline__ void lion_fold_arrive(const LionFold &f, int b, const float *stats, int *flag,
                                                 unsigned char *scratch, bool single_writer_wave) {
  if (!f.counters) return;
  const int tid = threadIdx.x;
  if (single_writer_wave) {          // every tile sum was stored by wave 0: only it has to wait for its stores
    if (tid < 64) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (tid == 0) *flag = __hip_atomic_fetch_add(f.counters + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == f.items - 1;
    }
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) *flag = __hip_atomic_fetch_add(f.counters + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == f.items - 1;
  }
  __syncthreads();
  if (*flag) {                       // (block-uniform)
    lion_fold_sample(f, b, stats, scratch);
    if (tid == 0) __hip_atomic_store(f.counters + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
}
