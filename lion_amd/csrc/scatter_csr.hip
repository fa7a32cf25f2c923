// scatter_csr.hip -- the backward scatters of the training path without float atomics (round 3):
//   K8   grouping backward        gx[b,c,n] = sum_{(m,u): idx[b,m,u] = n} gy[b,c,m,u]              (grouping.cu:58-80)
//   K12g 3-NN interpolation bwd   gx[b,c,m] = sum_{(q,j): idx[b,q,j] = m} w[b,q,j] gy[b,c,j]       (neighbor_interpolate.cu:145-170)
//   K5   devoxelize backward      gx[b,c,v] = sum_{(q,i): ind[b,q,i] = v} w[b,q,i] gy[b,c,i]       (trilinear_devox.cu:119-162)
// The reference issues one global float atomicAdd per (entry, channel); the round-1/2 kernels here accumulated rows in LDS
// with ds_add_f32 -- ~35 M of them per call at the largest shapes, and the LDS float-atomic rate (~0.3 per cycle and CU)
// made all three take ~200 us (0.05-0.18 of the HBM rate).  All three are  gx[b,c,bin] = sum over the entries e of the
// bin of w[b,e] * gy[b,c, e mod S]  with an index that does NOT depend on the channel, so the inverse index is built once
// per sample (a counting sort of the entries by bin: integer LDS atomics on E entries, not on E x C values) and every
// channel is a gather-sum over it -- in ascending entry order (the lists are sorted), i.e. deterministic, which the
// atomics of the reference are not.
//   build: workgroup (sample, 1 / P of the bins): histogram of the entries that fall into its bin range, scan, fill,
//          every entry ranks itself inside its bin (ascending entry order);  start / count per bin, entry lists per (sample, range).
//   apply: workgroup = (sample, CT channels): the gy rows streamed into LDS once, thread = bin gathers its entries from
//          LDS (the entry list is read once per CT channels), gx written coalesced.
#include "common.h"

namespace {

constexpr int CSR_P = 8;     // bin ranges (workgroups) per sample in the build
constexpr int CSR_NT = 1024;

__global__ __launch_bounds__(CSR_NT) void csr_build_kernel(const int32_t *__restrict__ idx, int E, int bins, int span,
                                                           int32_t *__restrict__ start, int32_t *__restrict__ count,
                                                           int32_t *__restrict__ perm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int *hist = reinterpret_cast<int *>(smem); // [span] counts, then cursors
  __shared__ int part[CSR_NT];
  const int tid = threadIdx.x, b = blockIdx.y, range = blockIdx.x, lo = range * span;
  const int nb = min(span, bins - lo); // bins of this range
  const int32_t *id = idx + (size_t)b * E;
  int32_t *pm = perm + ((size_t)b * CSR_P + range) * E;
  for (int v = tid; v < span; v += CSR_NT) hist[v] = 0;
  __syncthreads();
  for (int e = tid; e < E; e += CSR_NT) {
    const int t = min(max(id[e], 0), bins - 1) - lo;
    if (t >= 0 && t < nb) atomicAdd(&hist[t], 1);
  }
  __syncthreads();
  // exclusive scan: thread t owns bins [t*per, (t+1)*per)
  const int per = (span + CSR_NT - 1) / CSR_NT;
  int s = 0;
  for (int k = 0; k < per; ++k) { const int v = tid * per + k; if (v < nb) s += hist[v]; }
  part[tid] = s;
  __syncthreads();
  for (int d = 1; d < CSR_NT; d <<= 1) { // Hillis-Steele inclusive scan of the partials
    const int add = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += add;
    __syncthreads();
  }
  int run = part[tid] - s;
  for (int k = 0; k < per; ++k) {
    const int v = tid * per + k;
    if (v < nb) {
      const int c = hist[v];
      start[(size_t)b * bins + lo + v] = run;
      count[(size_t)b * bins + lo + v] = c;
      hist[v] = run; // cursor
      run += c;
    }
  }
  __syncthreads();
  for (int e = tid; e < E; e += CSR_NT) {
    const int t = min(max(id[e], 0), bins - 1) - lo;
    if (t >= 0 && t < nb) pm[atomicAdd(&hist[t], 1)] = e;
  }
  __threadfence_block();
  __syncthreads();
  // Ascending entry order inside every bin.  Short lists (the usual case: a handful of entries) by a thread-per-bin
  // insertion sort.  Round 4 (advisor finding): that sort is O(c^2) DEPENDENT global steps for a bin of c entries, and ball
  // query pads empty slots with the first neighbour, so a clustered cloud puts thousands of entries into a few bins --
  // bins longer than CSR_SHORT are ordered by their ENTRIES instead: each finds its own rank (the number of smaller entry
  // ids in its bin, 8 list reads in flight), all ranks first, then -- behind a barrier; the list region belongs to this
  // workgroup -- all writes: O(c) independent reads per entry, entries in parallel.  (Ranking every bin this way costs the
  // common case: K8 at SA-0 166 -> 218 us.)  Up to CSR_EPT entries per thread stay in registers; larger entry sets keep the
  // insertion sort for their long bins too.
  constexpr int CSR_EPT = 32, CSR_SHORT = 32;
  const bool rank_long = E <= CSR_EPT * CSR_NT;
  for (int v = tid; v < nb; v += CSR_NT) {
    const int c = count[(size_t)b * bins + lo + v], st = start[(size_t)b * bins + lo + v];
    if (c > CSR_SHORT && rank_long) continue;
    for (int i = 1; i < c; ++i) {
      const int key = pm[st + i];
      int j = i - 1;
      while (j >= 0 && pm[st + j] > key) { pm[st + j + 1] = pm[st + j]; --j; }
      pm[st + j + 1] = key;
    }
  }
  if (!rank_long) return;
  int slot[CSR_EPT];
#pragma unroll
  for (int k = 0; k < CSR_EPT; ++k) {
    const int e = tid + k * CSR_NT;
    slot[k] = -1;
    if (e < E) {
      const int t = min(max(id[e], 0), bins - 1) - lo;
      if (t >= 0 && t < nb) {
        const int c = count[(size_t)b * bins + lo + t], st = start[(size_t)b * bins + lo + t];
        if (c > CSR_SHORT) {
          int rank = 0, q = 0;
          for (; q + 8 <= c; q += 8) {
            int v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = pm[st + q + j];
#pragma unroll
            for (int j = 0; j < 8; ++j) rank += (v[j] < e) ? 1 : 0;
          }
          for (; q < c; ++q) rank += (pm[st + q] < e) ? 1 : 0;
          slot[k] = st + rank;
        }
      }
    }
  }
  __syncthreads(); // every rank is known: the long lists may be rewritten
#pragma unroll
  for (int k = 0; k < CSR_EPT; ++k)
    if (slot[k] >= 0) pm[slot[k]] = tid + k * CSR_NT;
}

// apply: a workgroup owns CT channels of a sample; their gy rows are streamed into LDS once (coalesced -- 4-byte gathers from
// global would pull a 64-byte sector per value), then thread = bin gathers its entries from LDS and writes gx coalesced.
constexpr int CSR_MAXC = 16;
template <bool W>
__global__ __launch_bounds__(1024) void csr_apply_kernel(const float *__restrict__ gy, const float *__restrict__ w,
                                                         const int32_t *__restrict__ start,
                                                         const int32_t *__restrict__ count,
                                                         const int32_t *__restrict__ perm, int C, int S, int E, int bins,
                                                         int span, int CT, float *__restrict__ gx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *rows = reinterpret_cast<float *>(smem); // [nc][S]
  const int tid = threadIdx.x, c0 = blockIdx.x * CT, b = blockIdx.y;
  const int nc = min(CT, C - c0);
  const float *g = gy + ((size_t)b * C + c0) * S;
  const int total = nc * S;
  if ((S & 3) == 0) {
    // 8 x 16 bytes per thread in flight (one workgroup per CU at 128-KiB rows: nothing else hides the latency)
    for (int v0 = tid * 4; v0 < total; v0 += 1024 * 4 * 8) {
      float4 t[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int v = v0 + q * 1024 * 4;
        t[q] = v < total ? *reinterpret_cast<const float4 *>(g + v) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int v = v0 + q * 1024 * 4;
        if (v < total) *reinterpret_cast<float4 *>(rows + v) = t[q];
      }
    }
  } else {
    for (int v = tid; v < total; v += 1024) rows[v] = g[v];
  }
  __syncthreads();
  const float *wb = W ? w + (size_t)b * E : nullptr;
  const bool ident = S == E; // grouping: entry e reads element e
  // four bins per thread and pass with their (start, count) loads in flight together, and the entry list of a bin read four
  // entries at a time: both were chains of dependent global round trips (one per bin, one per entry)
  constexpr int NBB = 4;
  for (int bin0 = tid; bin0 < bins; bin0 += 1024 * NBB) {
    int stv[NBB], nv[NBB];
#pragma unroll
    for (int q = 0; q < NBB; ++q) {
      const int bin = min(bin0 + q * 1024, bins - 1);
      stv[q] = start[(size_t)b * bins + bin];
      nv[q] = bin0 + q * 1024 < bins ? count[(size_t)b * bins + bin] : -1;
    }
#pragma unroll
    for (int q = 0; q < NBB; ++q) {
      const int bin = bin0 + q * 1024, n = nv[q];
      if (n < 0) continue;
      const int32_t *pm = perm + ((size_t)b * CSR_P + bin / span) * E + stv[q];
      float acc[CSR_MAXC];
#pragma unroll
      for (int c = 0; c < CSR_MAXC; ++c) acc[c] = 0.f;
      auto one = [&](int e) {
        const int src = ident ? e : e % S;
        const float wv = W ? wb[e] : 1.0f;
#pragma unroll
        for (int c = 0; c < CSR_MAXC; ++c)
          if (c < nc) acc[c] = add_rn(acc[c], W ? mul_rn(wv, rows[c * S + src]) : rows[c * S + src]);
      };
      int k = 0;
      for (; k + 4 <= n; k += 4) {
        const int e0 = pm[k], e1 = pm[k + 1], e2 = pm[k + 2], e3 = pm[k + 3];
        one(e0); one(e1); one(e2); one(e3); // ascending entry order, as the lists are sorted
      }
      for (; k < n; ++k) one(pm[k]);
#pragma unroll
      for (int c = 0; c < CSR_MAXC; ++c)
        if (c < nc) gx[((size_t)b * C + c0 + c) * bins + bin] = acc[c];
    }
  }
}

static int csr_span(int bins) { return (bins + CSR_P - 1) / CSR_P; }

} // namespace

extern "C" {

// start + count (bins ints each) + CSR_P entry lists of capacity E, per sample
size_t lion_scatter_csr_workspace_bytes(int B, int E, int bins) {
  if (B <= 0 || E <= 0 || bins <= 0) return 0;
  return ((size_t)B * bins * 2 + (size_t)B * CSR_P * E) * sizeof(int32_t);
}

// gx f32[B,C,bins] = sum over entries e with idx[b,e] = bin of (w ? w[b,e] : 1) * gy[b,c, e mod S];  idx int32[B,E]
// (clamped to [0, bins)), w f32[B,E] or NULL, gy f32[B,C,S].  ws from lion_scatter_csr_workspace_bytes.
int lion_scatter_csr(const float *gy, const int32_t *idx, const float *w, int B, int C, int S, int E, int bins, void *ws,
                     size_t ws_bytes, float *gx, lionStream_t stream) {
  if (!gy || !idx || !gx || B <= 0 || C <= 0 || S <= 0 || E <= 0 || bins <= 0) return LION_EINVAL;
  if (!ws || ws_bytes < lion_scatter_csr_workspace_bytes(B, E, bins)) return LION_EWORKSPACE;
  const int span = csr_span(bins);
  if ((size_t)span * 4 > 60 * 1024) return LION_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int32_t *start = static_cast<int32_t *>(ws), *count = start + (size_t)B * bins, *perm = count + (size_t)B * bins;
  csr_build_kernel<<<dim3(CSR_P, B), CSR_NT, (size_t)span * 4, st>>>(idx, E, bins, span, start, count, perm);
  // channels per workgroup: as many gy rows as fit 128 KiB of LDS, fewer while that leaves CUs without a workgroup
  int CT = (int)((128 * 1024) / ((size_t)S * 4));
  if (CT < 1) return LION_EUNSUPPORTED;
  if (CT > CSR_MAXC) CT = CSR_MAXC;
  if (CT > C) CT = C;
  while (CT > 1 && (long)B * lion_cdiv(C, CT) < 256) CT >>= 1;
  const size_t lds = (size_t)CT * S * 4;
  const dim3 grid(lion_cdiv(C, CT), B);
  if (w) {
    static LionLdsLimit cfg = {};
    if (int e = lion_dynamic_lds(&csr_apply_kernel<true>, lds, cfg)) return e;
    csr_apply_kernel<true><<<grid, 1024, lds, st>>>(gy, w, start, count, perm, C, S, E, bins, span, CT, gx);
  } else {
    static LionLdsLimit cfg = {};
    if (int e = lion_dynamic_lds(&csr_apply_kernel<false>, lds, cfg)) return e;
    csr_apply_kernel<false><<<grid, 1024, lds, st>>>(gy, w, start, count, perm, C, S, E, bins, span, CT, gx);
  }
  LION_LAUNCH_CHECK();
  return 0;
}

} // extern "C"
