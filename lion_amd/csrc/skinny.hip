// skinny.hip -- D2: the 1x1 convolutions of the global (style-latent) denoiser, models/score_sde/resnet.py:60-90,
// 124-218.  The activation is [B, C, 1, 1] with B = 32 and C = 2048: every layer is a GEMM with 32 rows
// whose cost is streaming its weights once (16.8 MB for 2048x2048).  The library path spends ~15
// launches per residual block (GEMM, bias, relu, SE GEMMs, sigmoid, mul, add ...), 154 per forward,
// each a few microseconds of mostly launch latency.  Here a block is 4 launches of one kernel:
//
//   yT[o][b] = epi( bias[o] + sum_k Wt[k][o] * (xT[k][b] (+ addT[k][b])) )
//
// * activations are kept channel-major, [C][32] with the batch on the fast axis, for the whole
//   network: both MFMA operands of v_mfma_f32_32x32x2_f32 are then plain 128-byte row reads (A = 32
//   output channels of the k-major packed weights, B = the 32 batch columns of xT) and the
//   accumulator tile [32 outputs x 32 batch] is written back with coalesced rows;
// * one workgroup of 16 waves owns a tile of 32 output channels and splits K over its waves (the
//   partial tiles are reduced through LDS in a fixed order), so the weight stream of a layer is
//   spread over Cout/32 CUs with >= 256 KB in flight each.  (Splitting K across workgroups with a
//   last-arriver reduction was measured 2-3x slower: a device-scope fence writes back / invalidates
//   the XCD's L2 on this multi-die part.);
// * prologue: + time embedding (ResBlockSEDrop: conv1(x + t)); epilogue: bias, ReLU, or the whole
//   squeeze-excite tail x + h * sigmoid(acc).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// act: 0 none, 1 relu.  gate != NULL: y = resid + gate * sigmoid(acc + bias).
__global__ __launch_bounds__(1024) void skinny_gemm_kernel(const float *__restrict__ xT, const float *__restrict__ wp,
                                                           const float *__restrict__ bias, int Cin, int Cout,
                                                           const float *__restrict__ addT, int act,
                                                           const float *__restrict__ gate,
                                                           const float *__restrict__ resid, float *__restrict__ yT) {
  __shared__ float part[16][1024]; // per-wave partial tiles (64 KiB); row writes / column sums are conflict-free
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int o0 = blockIdx.x * 32, nb = blockIdx.y;
  const int cl = lane & 31, kh = lane >> 5;
  const float *xb = xT + (size_t)nb * Cin * 32;
  const float *ab = addT ? addT + (size_t)nb * Cin * 32 : nullptr;
  const int ksteps = (Cin + 1) >> 1;
  const int per = (ksteps + 15) >> 4, s_lo = wave * per, s_hi = min(ksteps, s_lo + per);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  constexpr int UN = 16; // k-steps whose loads are all issued before the first MFMA (4 / 8 / 16 measure the same)
  for (int s0 = s_lo; s0 < s_hi; s0 += UN) {
    float av[UN], bv[UN], cv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = 2 * min(s0 + u, ksteps - 1) + kh; // packed weights have 2*ksteps rows (zero padded)
      const int kc = min(k, Cin - 1);
      av[u] = wp[(size_t)k * Cout + o0 + cl];
      bv[u] = xb[(size_t)kc * 32 + cl];
      cv[u] = ab ? ab[(size_t)kc * 32 + cl] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = 2 * (s0 + u) + kh;
      const float v = (s0 + u < s_hi && k < Cin) ? bv[u] + cv[u] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], v, acc, 0, 0, 0);
    }
  }
  // acc register i of lane l: output row (i&3) + 8*(i>>2) + 4*(l>>5), batch column l&31
#pragma unroll
  for (int i = 0; i < 16; ++i) part[wave][((i & 3) + 8 * (i >> 2) + 4 * kh) * 32 + cl] = acc[i];
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) s += part[w][tid]; // element (o = tid / 32, b = tid % 32), fixed order
  const int o = o0 + (tid >> 5);
  s += bias ? bias[o] : 0.f;
  const size_t at = ((size_t)nb * Cout + o) * 32 + (tid & 31);
  if (gate) s = resid[at] + gate[at] * (1.0f / (1.0f + expf(-s)));
  else if (act == 1) s = s > 0.f ? s : 0.f;
  yT[at] = s;
}

} // namespace

extern "C" {

// xT f32[nb][Cin][32] (channel-major activations, batch padded to 32 per slab), wp = lion_pwconv_pack_weights
// of w f32[Cout,Cin] (k-major [ceil2(Cin)][Cout]), bias f32[Cout] or NULL, addT f32[nb][Cin][32] or NULL
// (added to the input), act 0 none / 1 relu; gate/resid f32[nb][Cout][32] (both or neither):
// y = resid + gate * sigmoid(acc + bias).  Cout % 32 == 0.  -> yT f32[nb][Cout][32].
int lion_skinny_gemm(const float *xT, const float *wp, const float *bias, int nb, int Cin, int Cout,
                     const float *addT, int act, const float *gate, const float *resid, float *yT,
                     lionStream_t stream) {
  if (!xT || !wp || !yT || nb <= 0 || Cin <= 0 || Cout <= 0) return LION_EINVAL;
  if ((gate == nullptr) != (resid == nullptr) || act < 0 || act > 1) return LION_EINVAL;
  if (Cout % 32 != 0) return LION_EUNSUPPORTED;
  skinny_gemm_kernel<<<dim3(Cout / 32, nb), 1024, 0, static_cast<hipStream_t>(stream)>>>(
      xT, wp, bias, Cin, Cout, addT, act, gate, resid, yT);
  LION_LAUNCH_CHECK();
  return 0;
}

} // extern "C"
