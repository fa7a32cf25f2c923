// skinny.hip -- D2: the 1x1 convolutions of the global (style-latent) denoiser, models/score_sde/resnet.py:60-90,
// 124-218.  The activation is [B, C, 1, 1] with B = 32 and C = 2048: every layer is a GEMM with 32 rows
// whose cost is streaming its weights once (16.8 MB for 2048x2048).  The library path spends ~15
// launches per residual block (GEMM, bias, relu, SE GEMMs, sigmoid, mul, add ...), 154 per forward.
//
// * activations are kept channel-major, [C][32] with the batch on the fast axis, for the whole
//   network: both MFMA operands of v_mfma_f32_32x32x2_f32 are then plain 128-byte row reads (A = 32
//   output channels of the tile-major packed weights, B = the 32 batch columns) and the accumulator
//   tile [32 outputs x 32 batch] is written back with coalesced rows;
// * split-K WITHOUT in-kernel synchronisation: a layer is cut into (output tile, k-split) workgroups
//   of 16 waves (one round of loads per wave -- 4 x 16 B of weights + one 16 B operand load per input partial and
//   8-k chunk per lane --, all in flight at once, then 16 MFMAs fed from the wave's LDS slice) that write RAW
//   partial tiles P[ks][Cout][32]; the consumer sums the partials -- with the producer's bias and
//   ReLU -- while it loads its operand, so the hand-over is the kernel boundary.  (A last-arriver
//   reduction inside one launch was measured 2-3x slower: a device-scope fence writes back /
//   invalidates the XCD's L2 on this multi-die part; 64 workgroups without split-K are MFMA bound
//   at 17 us per layer.)
// * the squeeze-excite tail x + h * sigmoid(.) and the last layer's bias go through one small
//   element-wise kernel on the partials.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4 *p) { // read-once weight stream: non-temporal
  const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

// operand element k of batch column b: act_in( sum_q Pin[q][k][b] + bias_in[k] ) + addT[k][b]
//
// What bounds a layer (tools/exp/skinny_probe.hip, 2048 x 2048, HBM-cold weights): streaming the 16.8 MB alone takes
// 3.9 us (4.3 TB/s, all loads of a lane in flight, non-temporal); the first version of this kernel took 16.5 us
// because every lane fetched its operand elements with 4-byte loads -- 64 wave-level load instructions per wave
// for 4 input partials against 4 for the weights, and the CU's one texture-address unit serialises them (11.1 us
// with a single partial).  Now a wave brings its operand slice (8 k x 32 batch per chunk = 1 KiB per partial) in
// with ONE coalesced 16-byte load per lane, partial and chunk, reduces / biases / activates it in registers, parks
// it in its private 4 KiB of LDS and feeds the MFMAs from there; all weight loads of the wave (<= 4 x 16 B per
// lane and round) are issued before anything waits.
// FIN (round 6): the layer is the second squeeze-excite GEMM of a block (one k-split: the tile's sum is complete in this
// workgroup) and its epilogue IS the block's tail -- y = resid + relu(sum_q A[q] + bias_a) * sigmoid(tile sum), the arithmetic of
// skinny_finish_kernel mode 1 in the same order -- instead of raw partials followed by that launch (8 launches per step).
template <bool FIN>
__global__ __launch_bounds__(1024) void skinny_gemm_kernel(const float *__restrict__ pin, int ks_in,
                                                           const float *__restrict__ bias_in, int act_in,
                                                           const float *__restrict__ addT,
                                                           const float *__restrict__ wp, int Cin, int Cout,
                                                           float *__restrict__ pout, const float *__restrict__ fa, int ks_a,
                                                           const float *__restrict__ bias_a,
                                                           const float *__restrict__ resid) {
  extern __shared__ __attribute__((aligned(16))) float smem[]; // [16][1024]: operand slices, then partial tiles
  float(*part)[1024] = reinterpret_cast<float(*)[1024]>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int o0 = blockIdx.x * 32, ks = blockIdx.y, KS = gridDim.y, nb = blockIdx.z, NB = gridDim.z;
  const int cl = lane & 31, kh = lane >> 5;
  const int ksteps = (Cin + 1) >> 1;
  const int ksteps4 = (ksteps + 3) >> 2; // k-steps padded to a multiple of 4
  const float4 *wt4 = reinterpret_cast<const float4 *>(wp + (size_t)blockIdx.x * (ksteps4 * 8) * 32);
  const size_t in_stride = (size_t)NB * Cin * 32; // one k-split slab of the input partials
  const float *xb = pin + (size_t)nb * Cin * 32;
  const float *ab = addT ? addT + (size_t)nb * Cin * 32 : nullptr;
  const int per = ((ksteps + KS * 16 - 1) / (KS * 16) + 3) & ~3; // multiple of 4 k-steps per wave
  const int s_lo = min(ksteps, (ks * 16 + wave) * per), s_hi = min(ksteps, s_lo + per);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // weights: packed [tile][k-step / 4][k-half][32 channels][4 k-steps] -> one 16-byte load per lane covers 4 k-steps
  // (2 KiB per wave instruction); per-wave slices start on multiples of 4 k-steps.  A round = 16 k-steps = 32 k.
  for (int r0 = 0; r0 < per; r0 += 16) {
    const int s0 = s_lo + r0;
    float4 a4[4], x4[2][4], t4[2];
    float bq[2];
    // the weights of the whole round (HBM, the long latency) are requested first; the operand partials (L2 / MALL
    // resident, written by the previous layer) follow two chunks ahead of their use, so that chunk g's MFMAs run
    // while later chunks are still in flight and the kernel stays under the 128 registers of a 1024-thread workgroup
#pragma unroll
    for (int g = 0; g < 4; ++g)
      a4[g] = ld_stream(&wt4[((size_t)(min(s0 + 4 * g, ksteps4 * 4 - 4) >> 2) * 2 + kh) * 32 + cl]);
    // operand: chunk g = k in [2*(s0 + 4g), +8) x 32 batch columns = 256 consecutive floats: lane -> (k, 4 columns)
    auto fetch = [&](int g, int slot) {
      const int kc = min(2 * (s0 + 4 * g) + (lane >> 3), Cin - 1);
      const size_t off = (size_t)kc * 32 + (lane & 7) * 4;
      bq[slot] = bias_in ? bias_in[kc] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        x4[slot][q] = q < ks_in ? *reinterpret_cast<const float4 *>(xb + q * in_stride + off) : make_float4(0.f, 0.f, 0.f, 0.f);
      t4[slot] = ab ? *reinterpret_cast<const float4 *>(ab + off) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    fetch(0, 0);
    fetch(1, 1);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int sl = g & 1;
      const int k = 2 * (s0 + 4 * g) + (lane >> 3);
      float4 v = make_float4(bq[sl], bq[sl], bq[sl], bq[sl]);
#pragma unroll
      for (int q = 0; q < 4; ++q) { v.x += x4[sl][q].x; v.y += x4[sl][q].y; v.z += x4[sl][q].z; v.w += x4[sl][q].w; } // fixed order
      if (ks_in > 4) { // more input partials than the unrolled four (not produced by lion_skinny_splits today)
        const size_t off = (size_t)min(k, Cin - 1) * 32 + (lane & 7) * 4;
        for (int q = 4; q < ks_in; ++q) {
          const float4 t = *reinterpret_cast<const float4 *>(xb + q * in_stride + off);
          v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
      }
      if (act_in == 1) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
      v.x += t4[sl].x; v.y += t4[sl].y; v.z += t4[sl].z; v.w += t4[sl].w;
      const bool live = k < Cin && (k >> 1) < s_hi;
      if (!live) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g + 2 < 4) fetch(g + 2, sl);
      // the slice is private to the wave and a wave's LDS operations execute in order: no workgroup barrier, only
      // a compiler fence between the lanes' writes and the (other lanes') reads
      *reinterpret_cast<float4 *>(&part[wave][g * 256 + lane * 4]) = v;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int u = 4 * g + e;
        const float a = e == 0 ? a4[g].x : e == 1 ? a4[g].y : e == 2 ? a4[g].z : a4[g].w;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, part[wave][(2 * u + kh) * 32 + cl], acc, 0, 0, 0);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // next round rewrites the slice after these reads
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads(); // every wave is done reading its operand slice before the slices become partial tiles
  // acc register i of lane l: output row (i&3) + 8*(i>>2) + 4*(l>>5), batch column l&31
#pragma unroll
  for (int i = 0; i < 16; ++i) part[wave][((i & 3) + 8 * (i >> 2) + 4 * kh) * 32 + cl] = acc[i];
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) s += part[w][tid]; // element (o = tid / 32, b = tid % 32), fixed order
  if (FIN) { // (KS == 1: checked by the launcher)
    const size_t n = (size_t)NB * Cout * 32, i = ((size_t)nb * Cout + o0 + (tid >> 5)) * 32 + (tid & 31);
    float a = bias_a ? bias_a[o0 + (tid >> 5)] : 0.f;
    for (int q = 0; q < ks_a; ++q) a += fa[(size_t)q * n + i];
    float g = 0.f;
    g += s;
    pout[i] = resid[i] + (a > 0.f ? a : 0.f) * (1.0f / (1.0f + expf(-g)));
    return;
  }
  pout[((size_t)(ks * NB + nb) * Cout + o0 + (tid >> 5)) * 32 + (tid & 31)] = s;
}

// mode 0: y = sum_q A[q] + bias_a                               (last layer)
// mode 1: y = resid + relu(sum_q A[q] + bias_a) * sigmoid(sum_q Bp[q])   (ResBlockSEDrop tail, resnet.py:77-86)
__global__ __launch_bounds__(256) void skinny_finish_kernel(const float *__restrict__ A, int ks_a,
                                                            const float *__restrict__ bias_a,
                                                            const float *__restrict__ Bp, int ks_b,
                                                            const float *__restrict__ resid, int n, int C,
                                                            int mode, float *__restrict__ y) {
  const int i = blockIdx.x * 256 + threadIdx.x; // element of [nb][C][32]
  if (i >= n) return;
  const int c = (i >> 5) % C;
  float a = bias_a ? bias_a[c] : 0.f;
  for (int q = 0; q < ks_a; ++q) a += A[(size_t)q * n + i];
  if (mode == 1) {
    float g = 0.f;
    for (int q = 0; q < ks_b; ++q) g += Bp[(size_t)q * n + i];
    a = resid[i] + (a > 0.f ? a : 0.f) * (1.0f / (1.0f + expf(-g)));
  }
  y[i] = a;
}

// [B][C] (row stride ld floats; ld = 0: one row for the whole batch) -> [nb][C][32] with the batch on the fast axis, zero beyond
// B; two tensors per launch (the block's input x and the time embedding t), either may be absent.  And back.  (Round 6: these
// were three ATen transposing copies per step of the global prior's chain.)
__global__ __launch_bounds__(256) void to_channel_major_kernel(const float *__restrict__ a, int lda, int Ca, float *__restrict__ oa,
                                                               const float *__restrict__ b, int ldb, int Cb, float *__restrict__ ob,
                                                               int B, int nb) {
  const int na = a ? nb * Ca * 32 : 0, nbb = b ? nb * Cb * 32 : 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < na + nbb; i += gridDim.x * 256) {
    const bool second = i >= na;
    const int j = second ? i - na : i, C = second ? Cb : Ca, ld = second ? ldb : lda;
    const float *src = second ? b : a;
    const int col = j & 31, c = (j >> 5) % C, blk = j / (32 * C), row = blk * 32 + col;
    (second ? ob : oa)[j] = row < B ? src[(size_t)row * ld + c] : 0.f;
  }
}
__global__ __launch_bounds__(256) void from_channel_major_kernel(const float *__restrict__ x, int B, int C, float *__restrict__ y) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * C) return;
  const int row = i / C, c = i - row * C;
  y[i] = x[((size_t)(row >> 5) * C + c) * 32 + (row & 31)];
}

// element i of wp = [tile][s4][kh][j][e]: channel tile*32 + j, input k = 2*(4*s4 + e) + kh
__global__ void skinny_pack_kernel(const float *__restrict__ w, int Cout, int Cin, int ksteps4,
                                   float *__restrict__ wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cout * ksteps4 * 8) return;
  const int e = i & 3, j = (i >> 2) & 31, kh = (i >> 7) & 1, s4 = (i >> 8) % ksteps4, t = i / (256 * ksteps4);
  const int k = 2 * (4 * s4 + e) + kh;
  wp[i] = k < Cin ? w[(size_t)(t * 32 + j) * Cin + k] : 0.f;
}

} // namespace

extern "C" {

// floats of the packed copy: Cout * 8 * ceil(ceil(Cin / 2) / 4)
size_t lion_skinny_packed_floats(int Cout, int Cin) { return (size_t)Cout * 8 * (((Cin + 1) / 2 + 3) / 4); }

// w f32[Cout,Cin] -> wp f32[Cout/32][ceil(ksteps/4)][2][32][4] (zero padded k), Cout % 32 == 0
int lion_skinny_pack_weights(const float *w, int Cout, int Cin, float *wp, lionStream_t stream) {
  if (!w || !wp || Cout <= 0 || Cin <= 0) return LION_EINVAL;
  if (Cout % 32 != 0) return LION_EUNSUPPORTED;
  const int ksteps4 = ((Cin + 1) / 2 + 3) / 4, total = Cout * ksteps4 * 8;
  skinny_pack_kernel<<<lion_cdiv(total, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(w, Cout, Cin, ksteps4, wp);
  LION_LAUNCH_CHECK();
  return 0;
}

// k-splits of a layer: ~256 workgroups, at least 8 k-steps per wave
int lion_skinny_splits(int Cin, int Cout) {
  if (Cout <= 0 || Cout % 32 != 0 || Cin <= 0) return 0;
  const int tiles = Cout / 32, ksteps = (Cin + 1) / 2;
  int ks = 1;
  while (ks < 8 && tiles * ks * 2 <= 256 && ksteps / (ks * 2 * 16) >= 8) ks *= 2; // >= 8 k-steps per wave
  return ks;
}

// One layer: pout f32[lion_skinny_splits(Cin,Cout)][nb][Cout][32] (raw partial sums, no bias) from the operand
// act_in(sum_q pin[q] + bias_in) + addT, pin f32[ks_in][nb][Cin][32] (ks_in = 1 for a plain activation),
// bias_in f32[Cin] or NULL, act_in 0 none / 1 relu, addT f32[nb][Cin][32] or NULL; wp = lion_skinny_pack_weights.
int lion_skinny_gemm(const float *pin, int ks_in, const float *bias_in, int act_in, const float *addT,
                     const float *wp, int nb, int Cin, int Cout, float *pout, lionStream_t stream) {
  if (!pin || !wp || !pout || nb <= 0 || Cin <= 0 || Cout <= 0 || ks_in < 1 || act_in < 0 || act_in > 1)
    return LION_EINVAL;
  if (Cout % 32 != 0) return LION_EUNSUPPORTED;
  const size_t lds = (size_t)16 * 1024 * 4;
  static LionLdsLimit cfg = {};
  if (int e = lion_dynamic_lds(&skinny_gemm_kernel<false>, lds, cfg)) return e;
  skinny_gemm_kernel<false><<<dim3(Cout / 32, lion_skinny_splits(Cin, Cout), nb), 1024, lds, static_cast<hipStream_t>(stream)>>>(
      pin, ks_in, bias_in, act_in, addT, wp, Cin, Cout, pout, nullptr, 0, nullptr, nullptr);
  LION_LAUNCH_CHECK();
  return 0;
}

// The block's last GEMM with its tail: y f32[nb][Cout][32] = resid + relu(sum_q A[q] + bias_a) * sigmoid(W act_in(sum pin + bias_in)),
// A f32[ks_a][nb][Cout][32] (the block's conv2 partials), resid f32[nb][Cout][32].  == lion_skinny_gemm followed by
// lion_skinny_finish(mode 1), bit for bit, in one launch; needs lion_skinny_splits(Cin, Cout) == 1 (LION_EUNSUPPORTED otherwise).
int lion_skinny_gemm_se_finish(const float *pin, int ks_in, const float *bias_in, int act_in, const float *wp, int nb, int Cin,
                               int Cout, const float *A, int ks_a, const float *bias_a, const float *resid, float *y,
                               lionStream_t stream) {
  if (!pin || !wp || !A || !resid || !y || nb <= 0 || Cin <= 0 || Cout <= 0 || ks_in < 1 || ks_a < 1 || act_in < 0 || act_in > 1)
    return LION_EINVAL;
  if (Cout % 32 != 0 || lion_skinny_splits(Cin, Cout) != 1) return LION_EUNSUPPORTED;
  const size_t lds = (size_t)16 * 1024 * 4;
  static LionLdsLimit cfg = {};
  if (int e = lion_dynamic_lds(&skinny_gemm_kernel<true>, lds, cfg)) return e;
  skinny_gemm_kernel<true><<<dim3(Cout / 32, 1, nb), 1024, lds, static_cast<hipStream_t>(stream)>>>(
      pin, ks_in, bias_in, act_in, nullptr, wp, Cin, Cout, y, A, ks_a, bias_a, resid);
  LION_LAUNCH_CHECK();
  return 0;
}

// y f32[nb][C][32] from partials: mode 0: sum_q A[q] + bias_a; mode 1: resid + relu(sum_q A[q] + bias_a) *
// sigmoid(sum_q Bp[q]).  A f32[ks_a][nb][C][32], Bp f32[ks_b][nb][C][32].
int lion_skinny_finish(const float *A, int ks_a, const float *bias_a, const float *Bp, int ks_b, const float *resid,
                       int nb, int C, int mode, float *y, lionStream_t stream) {
  if (!A || !y || nb <= 0 || C <= 0 || ks_a < 1 || (mode != 0 && mode != 1)) return LION_EINVAL;
  if (mode == 1 && (!Bp || !resid || ks_b < 1)) return LION_EINVAL;
  const int n = nb * C * 32;
  skinny_finish_kernel<<<lion_cdiv(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(A, ks_a, bias_a, Bp, ks_b,
                                                                                        resid, n, C, mode, y);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_to_channel_major(const float *a, int lda, int Ca, float *oa, const float *b, int ldb, int Cb, float *ob, int B,
                          lionStream_t stream) {
  if (B <= 0 || (!a && !b) || (a && (!oa || Ca <= 0 || lda < 0)) || (b && (!ob || Cb <= 0 || ldb < 0))) return LION_EINVAL;
  const int nb = (B + 31) / 32, total = nb * 32 * ((a ? Ca : 0) + (b ? Cb : 0));
  to_channel_major_kernel<<<lion_cdiv(total, 256) > 64 ? 64 : lion_cdiv(total, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      a, lda, Ca, oa, b, ldb, Cb, ob, B, nb);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_from_channel_major(const float *x, int B, int C, float *y, lionStream_t stream) {
  if (!x || !y || B <= 0 || C <= 0) return LION_EINVAL;
  from_channel_major_kernel<<<lion_cdiv(B * C, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(x, B, C, y);
  LION_LAUNCH_CHECK();
  return 0;
}

} // extern "C"
