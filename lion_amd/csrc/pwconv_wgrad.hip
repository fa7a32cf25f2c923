// pwconv_wgrad.hip -- weight gradient of the kernel-size-1 convolutions (training path of G1: SharedMLP / point branch
// / classifier layers, models/pvcnn2_ada.py:120-164):   gw[o][i] = sum_b sum_l gy[b][o][l] x[b][i][l].
//
// autograd differentiates the matrix-product form of these layers into `mm([O, B L] x [B L, I])`: two transposing copies
// of the activations and a GEMM whose K is 1-2 million -- the library runs [32, 1048576] x [1048576, 35] in 1.45 ms,
// where streaming both operands once is 45 us at HBM rate.  Both operands are read here as what they are, rows along l:
// a workgroup owns a 64 x 64 tile of gw and a slice of (b, l), stages [64 rows x 64 l] of gy and of x in LDS (coalesced
// 256-byte row segments, row stride 68 floats: 16 consecutive rows of a ds_read_b128 cover the 256-byte LDS window
// once), and its four waves (one 32 x 32 quadrant each) run v_mfma_f32_32x32x2_f32 with l on the k axis -- exact fp32
// products, fp32 accumulation.  Partial tiles go to a workspace [slices][Cout][Cin]; a second kernel sums the slices in
// fixed order (deterministic, no atomics).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int PWG_T = 64;       // tile edge (channels) and l per chunk
constexpr int PWG_S = 68;       // LDS row stride in floats

__global__ __launch_bounds__(256, 2) void pwconv_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                              int Cin, int Cout, int L, int LK, int spb,
                                                              float *__restrict__ part, float *__restrict__ bpart) {
  __shared__ __attribute__((aligned(16))) float sg[2][PWG_T * PWG_S], sx[2][PWG_T * PWG_S];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slice = blockIdx.x, b = slice / spb, l_begin = (slice % spb) * LK;
  const int l_end = min(L, l_begin + LK);
  const int o0 = blockIdx.y * PWG_T, i0 = blockIdx.z * PWG_T;
  const float *gyb = gy + ((size_t)b * Cout + o0) * L, *xb = x + ((size_t)b * Cin + i0) * L;
  const int orows = min(PWG_T, Cout - o0), irows = min(PWG_T, Cin - i0);
  const bool vec = (L & 3) == 0;
  // staging: thread -> (row r0 + 16 j, 4 consecutive l at c4): 16 lanes cover 256 contiguous bytes of a row
  const int r0 = tid >> 4, c4 = (tid & 15) * 4;
  float4 rg[4], rx[4];
  // bias gradient (row sums of gy) from the staging registers of the workgroups of the first input-channel tile: the rows
  // are in flight anyway -- was a separate streaming pass over gy + a batch sum per layer
  const bool want_b = bpart != nullptr && blockIdx.z == 0;
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  auto load = [&](int l0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = r0 + 16 * j, l = l0 + c4;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
      if (vec && l + 3 < l_end) {
        if (r < orows) a = *reinterpret_cast<const float4 *>(gyb + (size_t)r * L + l);
        if (r < irows) c = *reinterpret_cast<const float4 *>(xb + (size_t)r * L + l);
      } else {
        float av[4] = {0.f, 0.f, 0.f, 0.f}, cv[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < 4; ++k)
          if (l + k < l_end) {
            if (r < orows) av[k] = gyb[(size_t)r * L + l + k];
            if (r < irows) cv[k] = xb[(size_t)r * L + l + k];
          }
        a = make_float4(av[0], av[1], av[2], av[3]);
        c = make_float4(cv[0], cv[1], cv[2], cv[3]);
      }
      rg[j] = a;
      rx[j] = c;
      if (want_b) bs[j] += (a.x + a.y) + (a.z + a.w);
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *reinterpret_cast<float4 *>(&sg[buf][(r0 + 16 * j) * PWG_S + c4]) = rg[j];
      *reinterpret_cast<float4 *>(&sx[buf][(r0 + 16 * j) * PWG_S + c4]) = rx[j];
    }
  };
  f32x16 acc;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  const int wo = (wave & 1) * 32, wi = (wave >> 1) * 32, l32 = lane & 31, half = lane >> 5;
  const int nchunks = (l_end - l_begin + PWG_T - 1) / PWG_T;
  if (nchunks > 0) {
    load(l_begin);
    store(0);
  }
  __syncthreads();
  for (int q = 0; q < nchunks; ++q) {
    const int buf = q & 1;
    if (q + 1 < nchunks) load(l_begin + (q + 1) * PWG_T); // in flight under the MFMAs
    const float *ag = &sg[buf][(wo + l32) * PWG_S + 4 * half], *bx = &sx[buf][(wi + l32) * PWG_S + 4 * half];
#pragma unroll
    for (int kk = 0; kk < PWG_T / 8; ++kk) {
      // lanes 0-31 hold l = 8 kk .. + 3 of their row, lanes 32-63 l = 8 kk + 4 .. + 7: MFMA j pairs (8 kk + j, 8 kk + 4 + j)
      const float4 a = *reinterpret_cast<const float4 *>(ag + 8 * kk), c = *reinterpret_cast<const float4 *>(bx + 8 * kk);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, c.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, c.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, c.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, c.w, acc, 0, 0, 0);
    }
    if (q + 1 < nchunks) store(buf ^ 1); // last read two barriers ago
    __syncthreads();
  }
  // acc register k of lane l: row (k & 3) + 8 (k >> 2) + 4 (l >> 5), column l & 31
  float *pt = part + (size_t)slice * Cout * Cin;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int o = o0 + wo + (k & 3) + 8 * (k >> 2) + 4 * half, i = i0 + wi + l32;
    if (o < Cout && i < Cin) pt[(size_t)o * Cin + i] = acc[k];
  }
  if (want_b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = bs[j];
      v += __shfl_xor(v, 8, 64);   // the 16 lanes that staged one row, fixed order
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 1, 64);
      const int r = r0 + 16 * j;
      if ((tid & 15) == 0 && r < orows) bpart[(size_t)slice * Cout + o0 + r] = v;
    }
  }
}

// 16 elements x 16 slice groups per workgroup: the slices of an element are summed by 16 threads (stride 16) and combined
// through LDS in fixed order -- with one thread per element a 32 x 35 gradient spread over 2048 slices was a 90-us serial loop
// elements n .. n + nb - 1 are the bias gradient (partials bpart [slices][nb])
__global__ __launch_bounds__(256) void pwconv_wgrad_reduce_kernel(const float *__restrict__ part, int n, int slices,
                                                                  float *__restrict__ gw, const float *__restrict__ bpart,
                                                                  int nb, float *__restrict__ gb) {
  __shared__ double sh[16][17];
  const int el = threadIdx.x & 15, grp = threadIdx.x >> 4, e = blockIdx.x * 16 + el;
  const bool is_w = e < n, live = e < n + nb;
  const float *src = is_w ? part + e : bpart + (e - n);
  const size_t stride = is_w ? (size_t)n : (size_t)nb;
  double s = 0.0;
  if (live)
    for (int k = grp; k < slices; k += 16) s += (double)src[(size_t)k * stride];
  sh[grp][el] = s;
  __syncthreads();
  if (grp == 0 && live) {
    double t = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += sh[g][el];
    (is_w ? gw + e : gb + (e - n))[0] = (float)t;
  }
}

struct PwgPlan { int LK, spb, slices; };
static PwgPlan pwg_plan(int B, int Cin, int Cout, int L) {
  const long tiles = (long)lion_cdiv(Cout, PWG_T) * lion_cdiv(Cin, PWG_T);
  // ~2048 workgroups in flight; slices of whole 64-l chunks inside one sample
  long want = 2048 / tiles;
  if (want < 1) want = 1;
  long per_b = (want + B - 1) / B;
  if (per_b < 1) per_b = 1;
  int LK = (int)(((long)L + per_b - 1) / per_b);
  LK = (LK + PWG_T - 1) / PWG_T * PWG_T;
  if (LK < 4 * PWG_T && L > 4 * PWG_T) LK = 4 * PWG_T;
  PwgPlan p;
  p.LK = LK;
  p.spb = lion_cdiv(L, LK);
  p.slices = B * p.spb;
  return p;
}

} // namespace

extern "C" {

size_t lion_pwconv_wgrad_workspace_bytes(int B, int Cin, int Cout, int L) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || L <= 0) return 0;
  return (size_t)pwg_plan(B, Cin, Cout, L).slices * Cout * (Cin + 1) * sizeof(float);   // + 1: the bias gradient's partials
}

int lion_pwconv_wgrad(const float *x, const float *gy, int B, int Cin, int Cout, int L, void *ws, size_t ws_bytes,
                      float *gw, float *gb, lionStream_t stream) {
  if (!x || !gy || !gw || B <= 0 || Cin <= 0 || Cout <= 0 || L <= 0) return LION_EINVAL;
  const PwgPlan p = pwg_plan(B, Cin, Cout, L);
  const size_t need = (size_t)p.slices * Cout * (Cin + 1) * sizeof(float);
  if (!ws || ws_bytes < need) return LION_EWORKSPACE;
  if ((((uintptr_t)x | (uintptr_t)gy) & 15) != 0) return LION_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float *part = static_cast<float *>(ws);
  float *bpart = gb ? part + (size_t)p.slices * Cout * Cin : nullptr;
  pwconv_wgrad_kernel<<<dim3(p.slices, lion_cdiv(Cout, PWG_T), lion_cdiv(Cin, PWG_T)), 256, 0, st>>>(
      x, gy, Cin, Cout, L, p.LK, p.spb, part, bpart);
  const int n = Cout * Cin, nb = gb ? Cout : 0;
  pwconv_wgrad_reduce_kernel<<<lion_cdiv(n + nb, 16), 256, 0, st>>>(part, n, p.slices, gw, bpart, nb, gb);
  LION_LAUNCH_CHECK();
  return 0;
}

} // extern "C"
