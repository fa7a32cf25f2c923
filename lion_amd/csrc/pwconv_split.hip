// pwconv_split.hip -- G1 on the 16-bit matrix pipe at fp32 accuracy: the 1x1 convolutions of SharedMLP
// (models/pvcnn2_ada.py:120-164) with both fp32 operands cut into two fp16 pieces (split_ops.h), same contract as
// pwconv.hip::pwconv_kernel:
//
//   y[b, co, l] = bias[co] + sum_ci W[co, ci] * act(x[b, ci, l]),   act(v) = swish(v * A[b,ci] + Bs[b,ci])  (PRO)
//
// Why: at B x L = 65536 columns these layers are GEMMs of 1-7 GFLOP; on v_mfma_f32_32x32x2_f32 (157 TF) that is 10-45 us
// of matrix time per layer -- more than the 8-17 us their activations need at HBM rate -- and the fp32 kernels reach a
// third of it (pwconv_small_kernel: 74 us at 192->128, L = 2048; kernel trace of a serialized step, round 2).  Three
// fp16 MFMAs per K = 16 cost 96 cycles where eight fp32 MFMAs cost 512: the layers become what they should be, one read
// and one write of the activation.
//
// Layout: the columns l sit on the MFMA columns (B operand), so a lane's B fragment -- 8 consecutive k of ONE column --
// is 8 plain coalesced row loads of x (128 bytes per half-wave), cut into the hi / lo pieces in registers: no LDS for
// the activation.  Block scaling per COLUMN: the two lanes of a column (k-halves) agree on max |act(x)| of the chunk
// with one cross-lane exchange, the power-of-two scale 2^E (max * 2^E in [2^13, 2^14)) is kept monotone along K and the
// column's accumulators (all in those two lanes) are rescaled when it grows -- as conv3d_split.hip, without LDS
// atomics or barriers.  Weights: packed once as pieces [chunk][piece][k-half][Cpad][8] with one power-of-two scale per
// tensor; a workgroup's slice of a chunk (64 B x channels) travels by LDS-DMA D chunks ahead (ring of D + 1), the
// activation rows D chunks ahead in registers; one barrier per chunk.
// A wave owns VB x 32 columns x CB x 32 channels; 4 waves = 128 VB columns per workgroup.
#include "split_ops.h"

namespace {

#ifdef PWS_TIMING
__device__ unsigned long long g_pws_t[128];
#define PWS_T(i) do { if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (i) < 128) g_pws_t[i] = __builtin_readcyclecounter(); } while (0)
#else
#define PWS_T(i) do { } while (0)
#endif

template <int CB, int VB, bool PRO, bool STATS, int NR>
__global__ __launch_bounds__(256, 2) void pwconv_split_kernel(const float *__restrict__ x, const u4 *__restrict__ wp,
                                                           const float *__restrict__ wtail,
                                                           const float *__restrict__ bias, float *__restrict__ y,
                                                           int Cin, int Cpad, int CoutY, int L,
                                                           const float *__restrict__ pro_a,
                                                           const float *__restrict__ pro_b, float *__restrict__ stats) {
  constexpr int COT = 32 * CB, WPL = 4 * COT;        // u4 per chunk slice: [piece][k-half][COT]
  constexpr int NDMA = WPL / 64, DMIN = NDMA / 4;    // wave instructions per slice; at least DMIN by every wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int D = NR - 1;                          // prefetch distance in chunks (rows in registers, slices in LDS)
  u4 *sw = reinterpret_cast<u4 *>(smem);             // [NR][WPL]
  const int nchunks = (Cin + KS - 1) / KS;
  constexpr int RING_U4 = NR * WPL > 4 * 32 * 9 ? NR * WPL : 4 * 32 * 9; // the statistics reuse the ring: 4 x 32 x 36 floats
  float *spa = reinterpret_cast<float *>(sw + RING_U4);  // [nchunks * 16] prologue scale / shift (PRO)
  float *spb = spa + (PRO ? nchunks * KS : 0);
  float *sbias = spb + (PRO ? nchunks * KS : 0);     // [COT] (zero beyond Cout)
  float *sred = sbias + COT;                         // [4 waves x 2 column halves][COT][2] (STATS)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, kh = lane >> 5;
  const int b = blockIdx.z, co0 = blockIdx.y * COT;
  PWS_T(0);
  if (PRO) {
    for (int c = tid; c < nchunks * KS; c += 256) {
      spa[c] = c < Cin ? pro_a[(size_t)b * Cin + c] : 0.f;
      spb[c] = c < Cin ? pro_b[(size_t)b * Cin + c] : 0.f;
    }
  }
  // the bias goes through LDS: read from global in the epilogue, every load would sit in front of a store's vmcnt wait
  if (tid < COT) sbias[tid] = (bias && co0 + tid < CoutY) ? bias[co0 + tid] : 0.f;
  int colc[VB];
  bool cok[VB];
#pragma unroll
  for (int vb = 0; vb < VB; ++vb) {
    const int col = ((blockIdx.x * 4 + wave) * VB + vb) * 32 + l32;
    cok[vb] = col < L;
    colc[vb] = cok[vb] ? col : L - 1; // clamped load, zeroed afterwards
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(x + (size_t)b * Cin * L), 0, Cin * L * 4, 0x00020000);
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const uint32_t sw_lds = (uint32_t)(uintptr_t)(lds_byte *)reinterpret_cast<unsigned char *>(sw);
  auto weights_dma = [&](int q) { // slice of chunk q -> ring slot q % NR; lane-linear element e = (plane, channel)
    if (q >= nchunks) return;
    const uint32_t dst0 = sw_lds + (uint32_t)((q % NR) * WPL * 16);
    for (int i = wave; i < NDMA; i += 4) {
      const int e = i * 64 + lane;
      const u4 *gp = wp + ((size_t)q * 4 + e / COT) * Cpad + co0 + e % COT;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(dst0 + (uint32_t)(i * 1024));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gp), "s"(dst) : "memory");
    }
  };
  // rows k = 16 q + 8 kh + j of this lane's columns, clamped past Cin.  UNCONDITIONAL, also for the D chunks past the end:
  // the compiler's waitcnt insertion counts the loads issued behind a register's load along every path into its use, and
  // with a path that skips loads it falls back to waiting for the loads just issued (measured: 2.6 us per chunk, the
  // whole memory latency, whatever the prefetch distance)
  auto issue_x = [&](int q, float (&v)[VB][8]) {
    // byte offset of row 16 q + 8 kh + j, column colc: rows past Cin lie past num_records of this batch entry's slice and
    // read as 0 (the range check covers the VGPR offset); one add per load
#pragma unroll
    for (int vb = 0; vb < VB; ++vb) {
      const int off0 = ((q * KS + kh * 8) * L + colc[vb]) * 4;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[vb][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, off0 + j * L * 4, 0, 0));
    }
  };
  f32x16 acc[CB][VB], cor[CB][VB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int vb = 0; vb < VB; ++vb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][vb][i] = cor[cb][vb][i] = 0.f;
  int E[VB];
#pragma unroll
  for (int vb = 0; vb < VB; ++vb) E[vb] = 127; // 127 = no scale yet (everything so far was zero)

  auto chunk = [&](int q, float (&v)[VB][8], float (&vnext)[VB][8]) {
    // chunk q's slice and rows must have arrived.  Issued behind them by this wave, in order: the slices and rows of the
    // chunks q + 1 .. q + D - 1 (8 VB row loads each, and at least DMIN slice instructions where the chunk exists) -- memory operations retire in order,
    // the count can only be too strict
    static_assert((D - 1) * (DMIN + 8 * VB) <= 63, "vmcnt is a 6-bit counter");
    const int behind = max(0, min(D - 1, nchunks - 1 - q)); // chunks behind q whose slices were issued
    if (behind >= 3 && D > 3) wait_vm<(D - 1) * 8 * VB + (D > 3 ? 3 : 0) * DMIN>();
    else if (behind == 2 && D > 2) wait_vm<(D - 1) * 8 * VB + (D > 2 ? 2 : 0) * DMIN>();
    else if (behind == 1 && D > 1) wait_vm<(D - 1) * 8 * VB + DMIN>();
    else wait_vm<(D - 1) * 8 * VB>();
    PWS_T(2 + 4 * q);
    __syncthreads(); // slice q visible to all waves; ring slot (q + D) % NR (chunk q - 1) no longer read
    PWS_T(3 + 4 * q);
    weights_dma(q + D);
    issue_x(q + D, vnext);
    PWS_T(4 + 4 * q);
    if (q >= nchunks) return; // a padding chunk of the last round: barrier and prefetch only (uniform)
    const u4 *swq = sw + (q % NR) * WPL;
    u4 wh[CB], wl[CB]; // the weight fragments first: their LDS latency runs under the cut of the activation rows
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) { wh[cb] = swq[(0 + kh) * COT + cb * 32 + l32]; wl[cb] = swq[(2 + kh) * COT + cb * 32 + l32]; }
    u4 xh[VB], xl[VB];
#pragma unroll
    for (int vb = 0; vb < VB; ++vb) {
      float t[8];
      unsigned m = 0u;
      float pa[8], pb[8];
      if (PRO) {
        const float4 a0 = *reinterpret_cast<const float4 *>(spa + q * KS + kh * 8), a1 = *reinterpret_cast<const float4 *>(spa + q * KS + kh * 8 + 4);
        const float4 b0 = *reinterpret_cast<const float4 *>(spb + q * KS + kh * 8), b1 = *reinterpret_cast<const float4 *>(spb + q * KS + kh * 8 + 4);
        pa[0] = a0.x; pa[1] = a0.y; pa[2] = a0.z; pa[3] = a0.w; pa[4] = a1.x; pa[5] = a1.y; pa[6] = a1.z; pa[7] = a1.w;
        pb[0] = b0.x; pb[1] = b0.y; pb[2] = b0.z; pb[3] = b0.w; pb[4] = b1.x; pb[5] = b1.y; pb[6] = b1.z; pb[7] = b1.w;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float u = v[vb][j];
        if (PRO) u = pro_act(u, pa[j], pb[j]);
        u = (q * KS + kh * 8 + j < Cin && cok[vb]) ? u : 0.f;
        t[j] = u;
        const unsigned a = __float_as_uint(u) & 0x7fffffffu; // inf / nan do not set the scale, they pass through the cut
        m = (a > m && a <= 0x7f7fffffu) ? a : m;
      }
      const unsigned mo = __shfl_xor(m, 32, 64); // the column's other k-half
      m = mo > m ? mo : m;
      if (m) {
        const int e = scale_exp(__uint_as_float(m));
        if (e < E[vb]) { // the column's maximum grew: bring its accumulators onto the new (smaller) scale first
          if (E[vb] != 127) {
            const float f = pow2f(max(e - PW_SPLIT_HEADROOM - E[vb], -126));
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
              for (int i = 0; i < 16; ++i) { acc[cb][vb][i] *= f; cor[cb][vb][i] *= f; }
          }
          E[vb] = e - PW_SPLIT_HEADROOM;
        }
      }
      const float xs = E[vb] == 127 ? 1.0f : pow2f(E[vb]);
#pragma unroll
      for (int k = 0; k < 4; ++k) { unsigned h2, l2; cut2(t[2 * k] * xs, t[2 * k + 1] * xs, h2, l2); xh[vb][k] = h2; xl[vb][k] = l2; }
    }
    PWS_T(64 + q);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int vb = 0; vb < VB; ++vb) acc[cb][vb] = mma(wh[cb], xh[vb], acc[cb][vb]);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int vb = 0; vb < VB; ++vb) cor[cb][vb] = mma(wh[cb], xl[vb], cor[cb][vb]);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int vb = 0; vb < VB; ++vb) cor[cb][vb] = mma(wl[cb], xh[vb], cor[cb][vb]);
    PWS_T(5 + 4 * q);
  };

  static_assert(D >= 1 && D <= 4, "the wait above distinguishes up to 3 chunks in flight behind the awaited one");
  float ring[NR][VB][8];
#pragma unroll
  for (int p = 0; p < D; ++p) { weights_dma(p); issue_x(p, ring[p]); }
  PWS_T(1);
  for (int q0 = 0; q0 < nchunks; q0 += NR) {
#pragma unroll
    for (int p = 0; p < NR; ++p)
      chunk(q0 + p, ring[p], ring[(p + D) % NR]); // static ring slots: chunk q lives in slot q % NR
  }

  // epilogue: D = (main + corr / 2048) * 2^-(E + ew) + bias, [B, Cout, L] store.  acc register i of lane l: channel row
  // (i&3) + 8*(i>>2) + 4*(l>>5), column l&31 -> 32 consecutive columns per (register, half-wave).
  PWS_T(100);
  const float us_w = wtail[2];
  float *yb = y + ((size_t)b * CoutY + co0) * L;
  const bool full = co0 + COT <= CoutY; // uniform: no padded channel rows in this tile (the common case: one branch per
#pragma unroll                          // column block instead of one per store)
  for (int vb = 0; vb < VB; ++vb) {
    const float us_x = E[vb] == 127 ? 1.0f : pow2f(-E[vb]);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
        const float o = ((acc[cb][vb][i] + cor[cb][vb][i] * (1.f / 2048.f)) * us_x) * us_w + sbias[co];
        acc[cb][vb][i] = cok[vb] ? o : 0.f;
      }
    float *yc = yb + colc[vb];
    if (full) {
      if (cok[vb]) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int i = 0; i < 16; ++i) yc[(size_t)(cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh) * L] = acc[cb][vb][i];
      }
    } else {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
          if (cok[vb] && co0 + co < CoutY) yc[(size_t)co * L] = acc[cb][vb][i]; // padded rows: zero weights, never stored
        }
    }
  }
  PWS_T(102);
  if (STATS) { // per (batch, channel, column tile) sum and sum of squares.
    // 64 channel rows per lane pair x a 32-lane butterfly each was 7.7k cycles of DPP per wave (s_memtime); instead the
    // wave transposes a 32-channel x 32-column block through LDS (its own 4.5 KiB of the weight ring, rows padded to 36
    // floats) and lane (channel, column half) sums 16 columns in a fixed order; the two halves
    // are separate partials.
    __syncthreads(); // the weight ring is no longer read
    float *T = reinterpret_cast<float *>(sw) + wave * (32 * 36);
    const int hc = lane >> 5; // which 16 columns of the row this lane sums
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int vb = 0; vb < VB; ++vb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) T[((i & 3) + 8 * (i >> 2) + 4 * kh) * 36 + l32] = acc[cb][vb][i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 t = *reinterpret_cast<const float4 *>(T + l32 * 36 + hc * 16 + 4 * j);
          s1 += t.x; s2 += t.x * t.x;
          s1 += t.y; s2 += t.y * t.y;
          s1 += t.z; s2 += t.z * t.z;
          s1 += t.w; s2 += t.w * t.w;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the next block rewrites T after these reads
        __builtin_amdgcn_wave_barrier();
      }
      // {columns 0-15, columns 16-31} of channel cb * 32 + l32: two partials per wave, summed in a fixed order below
      sred[((wave * 2 + hc) * COT + cb * 32 + l32) * 2] = s1;
      sred[((wave * 2 + hc) * COT + cb * 32 + l32) * 2 + 1] = s2;
    }
    PWS_T(103);
    __syncthreads();
    for (int c = tid; c < COT && co0 + c < CoutY; c += 256) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { s1 += sred[(w * COT + c) * 2]; s2 += sred[(w * COT + c) * 2 + 1]; }
      float *o = stats + (((size_t)b * CoutY + co0 + c) * gridDim.x + blockIdx.x) * 2;
      o[0] = s1;
      o[1] = s2;
    }
  }
  PWS_T(101);
}

// W f32[Cout][Cin] -> pieces u16 [ceil16(Cin)/16][piece][k-half][Cpad][8] (zero beyond Cout / Cin), cut from W * 2^ew.
// TR: the pieces of W^T (the data gradient's matrix) straight from W as stored -- w is then f32[Cin][Cout], read along its
// rows; its scale is W's (max |w| does not care about the transposition): `tail_src` = the tail of W's own packed form.
template <bool TR>
__global__ void pw_split_pack_kernel(const float *__restrict__ w, int Cout, int Cpad, int Cin, int nchunks,
                                     unsigned short *__restrict__ wp, unsigned *__restrict__ tail,
                                     const unsigned *__restrict__ tail_src) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int ew;
  if (TR) {
    ew = (int)tail_src[1];
    if (i < 4) tail[i] = tail_src[i];
  } else {
    ew = split_tail_scale(tail, i == 0);
  }
  if (i >= nchunks * KS * Cpad) return;
  const int co = i % Cpad, k = i / Cpad, q = k / KS, g = (k % KS) / 8, j = k % 8;
  unsigned short hi = 0, lo = 0;
  if (co < Cout && k < Cin) cut((TR ? w[(size_t)k * Cout + co] : w[(size_t)co * Cin + k]) * pow2f(ew), hi, lo);
  const size_t base = (size_t)q * 4;
  wp[((base + 0 + g) * Cpad + co) * 8 + j] = hi;
  wp[((base + 2 + g) * Cpad + co) * 8 + j] = lo;
}

// channel tile: the largest of 128 / 64 / 32 rows that divides ceil32(Cout); a wave's columns so that CB x VB <= 4
// accumulator pairs (128 registers)
#ifndef PWS_NR
#define PWS_NR 5
#endif
struct PwSplitPlan { int cb, vb; };
static int pws_pad(int Cout) { return (Cout + 31) / 32 * 32; }
static PwSplitPlan pws_plan(int Cout) {
  const int blocks = pws_pad(Cout) / 32;
  if (blocks <= 0) return {0, 0};
  if (blocks % 4 == 0) return {4, 1};
  if (blocks % 2 == 0) return {2, 2};
  return {1, 2}; // 4 column blocks per wave would need 96 registers of row prefetch on top of the accumulators
}
static size_t pws_halfs(int Cout, int Cin) { return (size_t)((Cin + KS - 1) / KS) * 4 * pws_pad(Cout) * 8; }

template <int CB, int VB, int NR>
static int launch_pws(const float *x, const u4 *wp, const float *wtail, const float *bias, float *y, int B, int Cin,
                      int Cout, int L, const float *pa, const float *pb, float *stats, hipStream_t st) {
  const int Cpad = pws_pad(Cout), nchunks = (Cin + KS - 1) / KS;
  const dim3 grid(lion_cdiv(L, 4 * VB * 32), Cpad / (CB * 32), B);
  const size_t ring = (size_t)NR * 4 * CB * 32 * 16, tile = (size_t)4 * 32 * 36 * 4; // weight ring, reused by the statistics
  const size_t lds = (ring > tile ? ring : tile) + (size_t)((pa ? 2 * nchunks * KS : 0) + CB * 32 + 8 * CB * 32 * 2) * 4;
#define LION_PWS_GO(PRO_, ST_)                                                                             \
  {                                                                                                        \
    static LionLdsLimit cfg = {};                                                                          \
    if (int e = lion_dynamic_lds(&pwconv_split_kernel<CB, VB, PRO_, ST_, NR>, lds, cfg)) return e;             \
    pwconv_split_kernel<CB, VB, PRO_, ST_, NR><<<grid, 256, lds, st>>>(x, wp, wtail, bias, y, Cin, Cpad, Cout, L, pa, pb, stats); \
  }
  if (pa && stats) LION_PWS_GO(true, true)
  else if (pa) LION_PWS_GO(true, false)
  else if (stats) LION_PWS_GO(false, true)
  else LION_PWS_GO(false, false)
#undef LION_PWS_GO
  LION_LAUNCH_CHECK();
  return 0;
}

} // namespace

extern "C" {

#ifdef PWS_TIMING
int lion_debug_pws_times(unsigned long long *host128) {
  return hipMemcpyFromSymbol(host128, HIP_SYMBOL(g_pws_t), 1024) == hipSuccess ? 0 : -1;
}
#endif

// number of uint16 of the packed pieces + the 8-halfword tail {max |w| bits, ew, 2^-ew, 0}
size_t lion_pwconv_split_packed_halfs(int Cout, int Cin) { return pws_halfs(Cout, Cin) + 8; }

int lion_pwconv_split_pack_weights(const float *w, int Cout, int Cin, uint16_t *wp, lionStream_t stream) {
  if (!w || !wp || Cout <= 0 || Cin <= 0) return LION_EINVAL;
  if ((((uintptr_t)wp) & 15) != 0) return LION_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned *tail = reinterpret_cast<unsigned *>(wp + pws_halfs(Cout, Cin));
  const int n = Cout * Cin, nchunks = (Cin + KS - 1) / KS, Cpad = pws_pad(Cout);
  if (hipMemsetAsync(tail, 0, 16, st) != hipSuccess) return LION_EINVAL;
  split_wmax_kernel<<<min(lion_cdiv(n, 2048), 128), 256, 0, st>>>(w, n, tail);
  pw_split_pack_kernel<false><<<lion_cdiv(nchunks * KS * Cpad, 256), 256, 0, st>>>(w, Cout, Cpad, Cin, nchunks, wp, tail, nullptr);
  LION_LAUNCH_CHECK();
  return 0;
}

// The packed form of W^T f32[Cin][Cout] (what the data gradient multiplies by) from W f32[Cout][Cin] as stored and W's own
// packed form wp (its tail holds the scale): one launch, no transposed copy, no second maximum.  wpt: lion_pwconv_split_packed_halfs(Cin, Cout).
int lion_pwconv_split_pack_weights_t(const float *w, int Cout, int Cin, const uint16_t *wp, uint16_t *wpt, lionStream_t stream) {
  if (!w || !wp || !wpt || Cout <= 0 || Cin <= 0) return LION_EINVAL;
  if (((((uintptr_t)wp) | ((uintptr_t)wpt)) & 15) != 0) return LION_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned *tail_src = reinterpret_cast<const unsigned *>(wp + pws_halfs(Cout, Cin));
  unsigned *tail = reinterpret_cast<unsigned *>(wpt + pws_halfs(Cin, Cout));
  // the transposed matrix has Cin rows ("output channels") and Cout columns
  const int nchunks = (Cout + KS - 1) / KS, Cpad = pws_pad(Cin);
  pw_split_pack_kernel<true><<<lion_cdiv(nchunks * KS * Cpad, 256), 256, 0, st>>>(w, Cin, Cpad, Cout, nchunks, wpt, tail, tail_src);
  LION_LAUNCH_CHECK();
  return 0;
}

// column tiles per batch element = rows of the stats tensor per channel
int lion_pwconv_split_stat_tiles(int Cout, int Cin, int L) {
  const PwSplitPlan p = pws_plan(Cout);
  return (p.cb && Cin > 0 && L > 0) ? lion_cdiv(L, 4 * p.vb * 32) : 0;
}

// Arguments as lion_pwconv_forward (include/lion_hip.h) with wp from lion_pwconv_split_pack_weights and stats sized by
// lion_pwconv_split_stat_tiles.  Any Cout, Cin, L with Cin * L < 2^29 elements per batch entry (32-bit byte offsets).
int lion_pwconv_split_forward(const float *x, const uint16_t *wp, const float *bias, int B, int Cin, int Cout, int L,
                              const float *pro_a, const float *pro_b, float *y, float *stats, lionStream_t stream) {
  if (!x || !wp || !y || B <= 0 || Cin <= 0 || Cout <= 0 || L <= 0) return LION_EINVAL;
  if ((pro_a == nullptr) != (pro_b == nullptr)) return LION_EINVAL;
  if ((long)Cin * L >= (1L << 29) || Cin > 4096) return LION_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const u4 *w4 = reinterpret_cast<const u4 *>(wp);
  const float *wtail = reinterpret_cast<const float *>(wp + pws_halfs(Cout, Cin));
  const PwSplitPlan p = pws_plan(Cout);
  switch (p.cb) {
  case 4: return launch_pws<4, 1, PWS_NR>(x, w4, wtail, bias, y, B, Cin, Cout, L, pro_a, pro_b, stats, st);
  case 2: return launch_pws<2, 2, PWS_NR>(x, w4, wtail, bias, y, B, Cin, Cout, L, pro_a, pro_b, stats, st);
  case 1: return launch_pws<1, 2, PWS_NR>(x, w4, wtail, bias, y, B, Cin, Cout, L, pro_a, pro_b, stats, st);
  default: return LION_EUNSUPPORTED;
  }
}

} // extern "C"
