// split_mfma_conv_tile.hip -- EXPERIMENT (round-2 preparation, not part of liblion_hip.so).
//
// Question: how fast can the 3x3x3 convolution's inner product run if the fp32 operands are split into 16-bit
// pieces and multiplied on the 16-bit MFMA pipe (32 cycles per 32x32x16, 16x the fp32 MFMA rate), with fp32
// accumulation and fp32-level accuracy (tools/exp/split_precision_numerics.py: both schemes below are at least as
// accurate as the native v_mfma_f32_32x32x2_f32 chain)?
//   MODE 0  fp16 x2:  a = ah + al/2048 (fp16 each, residual rescaled by 2^11), main += ah*bh, corr += ah*bl + al*bh,
//                     result = main + corr/2048                      -> 3 MFMAs per K=16 (5.3x fewer MFMA cycles)
//   MODE 1  bf16 x3:  a = a1 + a2 + a3 (bf16 each), a1b1 a1b2 a2b1 a2b2 a1b3 a3b1 in one accumulator
//                                                                    -> 6 MFMAs per K=16 (2.7x fewer MFMA cycles)
// One workgroup (4 waves) computes one output tile of the production kernel's geometry: 2x4x32 voxels x 64 output
// channels, input halo 4x6x34, Cin in chunks of 16 (= one MFMA K).  M = output channel, N = voxel (so that a lane's
// accumulator column is a voxel and stores are coalesced along W); every wave owns 2 W-rows (64 voxels) x 64 channels.
// Operand planes in LDS are [piece][k-half g][position][8 x 16 bit]: a lane's MFMA fragment is ONE ds_read_b128 and
// consecutive lanes read consecutive 16 bytes (conflict free).  Weights of the next tap are fetched into registers
// while the current tap is multiplied (double buffer, one barrier per tap).
// The split itself is done by prep kernels here (in production: activations at LDS-staging time after the
// AdaGN+Swish prologue, weights once at pack time).
//
// Build + run on an MI355X:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/split_mfma_conv_tile.hip -o /tmp/split_conv && /tmp/split_conv
// prints, per variant: error against a float64 host convolution (relative to the output rms) and the effective
// fp32-equivalent TFLOP/s with every CU busy (to compare with 141 TF of conv3d_k3_kernel and the 157.3 TF fp32 peak).
// Variants: MODE (0 fp16x2, 1 bf16x3, 2 = one fp16 piece only: NOT accurate, shows the data-movement ceiling),
// OCC = workgroups per CU asked of the compiler (register budget 512/OCC per lane);
// WSRC (0 = weights through an LDS double buffer, one barrier per tap; 1 = every wave loads its weight fragments
// straight from L2 in operand order, prefetched one tap ahead, barriers only per 16-channel chunk),
// TD (2 = 2x4x32 tile, 2 W-rows per wave; 4 = 4x4x32 tile, 4 W-rows per wave: each weight fragment feeds twice the MFMAs).
// First measurement (profiles/r01_exp_split_mfma_conv_tile_v0.txt, MODE 0/1, WSRC 0, TD 2, untuned):
//   fp16x2 269.5 TF (rms err 2.7e-7), bf16x3 193.1 TF (rms err 6.3e-7).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int TH = 4, TW = 32, COUT = 64, KC = 16;
constexpr int HH = TH + 2, HW = TW + 2;
constexpr int halo_of(int td) { return (td + 2) * HH * HW; } // TD=2: 816 positions, TD=4: 1224
constexpr int nvox_of(int td) { return td * TH * TW; }       // TD=2: 256 voxels (2 W-rows per wave), TD=4: 512 (4 per wave)

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

template <int MODE> struct Split;
template <> struct Split<0> { // fp16 hi + rescaled lo
  static constexpr int S = 2;
  __device__ static void cut(float v, unsigned short *p) {
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)((v - (float)hi) * 2048.f);
    p[0] = __builtin_bit_cast(unsigned short, hi);
    p[1] = __builtin_bit_cast(unsigned short, lo);
  }
};
template <> struct Split<2> { // fp16 hi only -- NOT fp32 accurate, shows what the data movement alone allows
  static constexpr int S = 1;
  __device__ static void cut(float v, unsigned short *p) { p[0] = __builtin_bit_cast(unsigned short, (_Float16)v); }
};
template <> struct Split<1> { // bf16 x3
  static constexpr int S = 3;
  __device__ static void cut(float v, unsigned short *p) {
    float r = v;
    for (int s = 0; s < 3; ++s) {
      const __bf16 q = (__bf16)r;
      p[s] = __builtin_bit_cast(unsigned short, q);
      r -= (float)q;
    }
  }
};

// x f32[Cin][HALO] -> xp u16[chunk][piece][g][HALO][8]   (channel c = chunk*16 + g*8 + j)
template <int MODE, int HALO>
__global__ void split_activations(const float *x, int Cin, unsigned short *xp) {
  constexpr int S = Split<MODE>::S;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cin * HALO) return;
  const int c = i / HALO, p = i % HALO, chunk = c / KC, g = (c % KC) / 8, j = c % 8;
  unsigned short piece[3];
  Split<MODE>::cut(x[i], piece);
  for (int s = 0; s < S; ++s) xp[((((size_t)chunk * S + s) * 2 + g) * HALO + p) * 8 + j] = piece[s];
}
// w f32[COUT][Cin][27] -> wp u16[chunk][tap][piece][g][COUT][8]
template <int MODE>
__global__ void split_weights(const float *w, int Cin, unsigned short *wp) {
  constexpr int S = Split<MODE>::S;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= COUT * Cin * 27) return;
  const int t = i % 27, c = (i / 27) % Cin, co = i / (27 * Cin), chunk = c / KC, g = (c % KC) / 8, j = c % 8;
  unsigned short piece[3];
  Split<MODE>::cut(w[i], piece);
  for (int s = 0; s < S; ++s)
    wp[(((((size_t)chunk * 27 + t) * S + s) * 2 + g) * COUT + co) * 8 + j] = piece[s];
}

template <int MODE> __device__ __forceinline__ f16v mma(u4 a, u4 b, f16v c);
template <> __device__ __forceinline__ f16v mma<0>(u4 a, u4 b, f16v c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f16v mma<2>(u4 a, u4 b, f16v c) { return mma<0>(a, b, c); }
template <> __device__ __forceinline__ f16v mma<1>(u4 a, u4 b, f16v c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
}

// grid = G workgroups, all computing the same tile (weights and halo come from L2 like the production kernel's
// re-reads do); rep repeats the whole reduction to reach steady state (the result is then rep x the convolution).
template <int MODE, int WSRC, int TD, int OCC, int UNR>
__global__ __launch_bounds__(256, OCC) void conv_tile_kernel(const u4 *__restrict__ xp, const u4 *__restrict__ wp,
                                                        int chunks, int rep, float *__restrict__ out) {
  constexpr int S = Split<MODE>::S;
  constexpr int HALO = halo_of(TD), NVOX = nvox_of(TD);
  constexpr int NR = TD * TH / 4;    // W-rows (32-voxel MFMA column blocks) per wave
  constexpr int XPL = S * 2 * HALO;  // u4 per chunk of activation planes
  constexpr int WPL = S * 2 * COUT;  // u4 per (chunk, tap) of weight planes
  constexpr int WLD = (WPL + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u4 *sx = reinterpret_cast<u4 *>(smem); // [S][2][HALO]
  u4 *sw = sx + XPL;                     // [2][S][2][COUT]   (WSRC 0 only)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, l32 = lane & 31;

  f16v acc[2][NR], cor[2][MODE == 0 ? NR : 1];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NR; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[m][n][r] = 0.f;
        if (MODE == 0) cor[m][n][r] = 0.f;
      }

  int xbase[NR]; // the wave's W-rows: row index NR*wave + n -> (d, h)
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    const int row = NR * wave + n, d = row / TH, h = row % TH;
    xbase[n] = (d * HH + h) * HW + l32;
  }
  const int steps = chunks * 27;
  for (int it = 0; it < rep; ++it) {
    u4 wreg[WLD];     // WSRC 0: this thread's share of the next tap's weight planes
    u4 wnext[2][S];   // WSRC 1: this lane's fragments of the next tap
    if (WSRC == 0) {
#pragma unroll
      for (int i = 0; i < WLD; ++i) {
        const int e = tid + i * 256;
        if (e < WPL) wreg[i] = wp[e];
      }
    } else {
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int m = 0; m < 2; ++m) wnext[m][s] = wp[(s * 2 + g) * COUT + m * 32 + l32];
    }
    for (int chunk = 0; chunk < chunks; ++chunk) {
      // new 16-channel chunk: restage the activation planes
      __syncthreads();
      for (int e = tid; e < XPL; e += 256) sx[e] = xp[(size_t)chunk * XPL + e];
      if (WSRC == 1) __syncthreads();
      constexpr int UF = UNR ? 27 : 1; // UNR: all tap offsets become immediates and the compiler may pipeline across taps
#pragma unroll UF
      for (int tap = 0; tap < 27; ++tap) {
      const int q = chunk * 27 + tap;
      u4 wf[2][S], xf[NR][S];
      if (WSRC == 0) {
        u4 *swb = sw + (q & 1) * WPL;
#pragma unroll
        for (int i = 0; i < WLD; ++i) {
          const int e = tid + i * 256;
          if (e < WPL) swb[e] = wreg[i];
        }
        __syncthreads();
        if (q + 1 < steps) { // next tap's weights travel while this tap is multiplied
#pragma unroll
          for (int i = 0; i < WLD; ++i) {
            const int e = tid + i * 256;
            if (e < WPL) wreg[i] = wp[(size_t)(q + 1) * WPL + e];
          }
        }
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
          for (int m = 0; m < 2; ++m) wf[m][s] = swb[(s * 2 + g) * COUT + m * 32 + l32];
      } else {
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
          for (int m = 0; m < 2; ++m) wf[m][s] = wnext[m][s];
        if (q + 1 < steps) {
#pragma unroll
          for (int s = 0; s < S; ++s)
#pragma unroll
            for (int m = 0; m < 2; ++m)
              wnext[m][s] = wp[(size_t)(q + 1) * WPL + (s * 2 + g) * COUT + m * 32 + l32];
        }
      }
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
      const int toff = (dz * HH + dy) * HW + dx;
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int n = 0; n < NR; ++n) xf[n][s] = sx[(s * 2 + g) * HALO + xbase[n] + toff];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NR; ++n) {
          if (MODE == 0) {
            acc[m][n] = mma<MODE>(wf[m][0], xf[n][0], acc[m][n]);
            cor[m][n] = mma<MODE>(wf[m][0], xf[n][S - 1], cor[m][n]);
            cor[m][n] = mma<MODE>(wf[m][S - 1], xf[n][0], cor[m][n]);
          } else if (MODE == 2) {
            acc[m][n] = mma<MODE>(wf[m][0], xf[n][0], acc[m][n]);
          } else {
            // small terms first, so that the large one lands on an accumulator that already holds them
            acc[m][n] = mma<MODE>(wf[m][S - 1], xf[n][0], acc[m][n]);
            acc[m][n] = mma<MODE>(wf[m][0], xf[n][S - 1], acc[m][n]);
            acc[m][n] = mma<MODE>(wf[m][S > 1 ? 1 : 0], xf[n][S > 1 ? 1 : 0], acc[m][n]);
            acc[m][n] = mma<MODE>(wf[m][S > 1 ? 1 : 0], xf[n][0], acc[m][n]);
            acc[m][n] = mma<MODE>(wf[m][0], xf[n][S > 1 ? 1 : 0], acc[m][n]);
            acc[m][n] = mma<MODE>(wf[m][0], xf[n][0], acc[m][n]);
          }
        }
      }
    }
  }
  // D[row = channel][col = voxel]: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float *o = out + (size_t)blockIdx.x * COUT * NVOX;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NR; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        const int vox = (NR * wave + n) * TW + l32;
        const float v = MODE == 0 ? acc[m][n][r] + cor[m][n][r] * (1.f / 2048.f) : acc[m][n][r];
        o[(size_t)co * NVOX + vox] = v;
      }
}

// Hand-pipelined form of the fp16x2 / weights-from-L2 loop (the shape the compiler does NOT produce by itself: it sinks
// every load to its first use).  Weight fragments travel two taps ahead in a ring of three register sets (27 taps per
// chunk keep the ring phase), the activation fragments of tap t+1 are read from LDS into the other half of a double
// buffer before the 12 MFMAs of tap t are issued; scheduling fences keep that order.  One bubble per 16-channel chunk
// (the first tap's activation fragments cannot be read before the planes are staged).
template <int TD, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_tile_pipe_kernel(const u4 *__restrict__ xp, const u4 *__restrict__ wp,
                                                             int chunks, int rep, float *__restrict__ out) {
  constexpr int S = 2;
  constexpr int HALO = halo_of(TD), NVOX = nvox_of(TD);
  constexpr int NR = TD * TH / 4;
  constexpr int XPL = S * 2 * HALO, WPL = S * 2 * COUT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u4 *sx = reinterpret_cast<u4 *>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, l32 = lane & 31;
  f16v acc[2][NR], cor[2][NR];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NR; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = cor[m][n][r] = 0.f;
  int xbase[NR];
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    const int row = NR * wave + n, d = row / TH, h = row % TH;
    xbase[n] = (d * HH + h) * HW + l32;
  }
  const int steps = chunks * 27;
  const int wbase = g * COUT + l32;
  for (int it = 0; it < rep; ++it) {
    u4 wfr[3][2][S], xfb[2][NR][S];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int m = 0; m < 2; ++m) wfr[a][m][s] = wp[(size_t)min(a, steps - 1) * WPL + s * 2 * COUT + wbase + m * 32];
    for (int chunk = 0; chunk < chunks; ++chunk) {
      __syncthreads();
      for (int e = tid; e < XPL; e += 256) sx[e] = xp[(size_t)chunk * XPL + e];
      __syncthreads();
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int n = 0; n < NR; ++n) xfb[0][n][s] = sx[(s * 2 + g) * HALO + xbase[n]]; // tap 0: offset 0
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
        const int q = chunk * 27 + tap;
        const int qn = min(q + 2, steps - 1); // the last two requests repeat the last slice (never used)
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
          for (int m = 0; m < 2; ++m) wfr[(tap + 2) % 3][m][s] = wp[(size_t)qn * WPL + s * 2 * COUT + wbase + m * 32];
        if (tap + 1 < 27) {
          const int t1 = tap + 1, toff = ((t1 / 9) * HH + (t1 / 3) % 3) * HW + t1 % 3;
#pragma unroll
          for (int s = 0; s < S; ++s)
#pragma unroll
            for (int n = 0; n < NR; ++n) xfb[t1 & 1][n][s] = sx[(s * 2 + g) * HALO + xbase[n] + toff];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NR; ++n) acc[m][n] = mma<0>(wfr[tap % 3][m][0], xfb[tap & 1][n][0], acc[m][n]);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NR; ++n) cor[m][n] = mma<0>(wfr[tap % 3][m][0], xfb[tap & 1][n][1], cor[m][n]);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NR; ++n) cor[m][n] = mma<0>(wfr[tap % 3][m][1], xfb[tap & 1][n][0], cor[m][n]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float *o = out + (size_t)blockIdx.x * COUT * NVOX;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NR; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        const int vox = (NR * wave + n) * TW + l32;
        o[(size_t)co * NVOX + vox] = acc[m][n][r] + cor[m][n][r] * (1.f / 2048.f);
      }
}

static std::vector<float> g_x[5], g_w; // halo inputs per TD (index = TD), weights
static std::vector<double> g_truth[5];
static double g_rms[5];

template <int MODE, int WSRC, int TD, int OCC, int UNR, bool PIPE = false>
static void run(const char *name, int Cin, int n_cu) {
  constexpr int S = Split<MODE>::S;
  constexpr int HALO = halo_of(TD), NVOX = nvox_of(TD);
  const std::vector<float> &hx = g_x[TD], &hw = g_w;
  const int chunks = Cin / KC;
  float *dx, *dw, *dout;
  unsigned short *dxp, *dwp;
  const int G = 2 * n_cu * 4 * 2 / TD; // same number of output voxels for both tile sizes
  CHECK(hipMalloc(&dx, hx.size() * 4));
  CHECK(hipMalloc(&dw, hw.size() * 4));
  CHECK(hipMalloc(&dxp, (size_t)chunks * S * 2 * HALO * 16));
  CHECK(hipMalloc(&dwp, (size_t)chunks * 27 * S * 2 * COUT * 16));
  CHECK(hipMalloc(&dout, (size_t)G * COUT * NVOX * 4));
  CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  split_activations<MODE, HALO><<<(Cin * HALO + 255) / 256, 256>>>(dx, Cin, dxp);
  split_weights<MODE><<<(COUT * Cin * 27 + 255) / 256, 256>>>(dw, Cin, dwp);
  const size_t lds = (size_t)(S * 2 * HALO + (WSRC == 0 ? 2 * S * 2 * COUT : 0)) * 16;
  void (*kern)(const u4 *, const u4 *, int, int, float *);
  if constexpr (PIPE) kern = &conv_tile_pipe_kernel<TD, OCC>;
  else kern = &conv_tile_kernel<MODE, WSRC, TD, OCC, UNR>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const u4 *xp = reinterpret_cast<const u4 *>(dxp), *wp = reinterpret_cast<const u4 *>(dwp);
  // correctness: one workgroup, one pass
  kern<<<1, 256, lds>>>(xp, wp, chunks, 1, dout);
  CHECK(hipDeviceSynchronize());
  std::vector<float> ho((size_t)COUT * NVOX);
  CHECK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
  double emax = 0, e2 = 0;
  for (size_t i = 0; i < ho.size(); ++i) {
    const double e = (ho[i] - g_truth[TD][i]) / g_rms[TD];
    emax = fmax(emax, fabs(e));
    e2 += e * e;
  }
  // throughput: every CU busy, several rounds of workgroups, 4 passes each
  const int rep = 4, launches = 4;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  kern<<<G, 256, lds>>>(xp, wp, chunks, rep, dout);
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < launches; ++i) kern<<<G, 256, lds>>>(xp, wp, chunks, rep, dout);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 2.0 * COUT * NVOX * Cin * 27 * rep * G * launches;
  printf("%-7s W%s TD%d occ%d %s  lds %5.1f KB  rms err %.2e  max err %.2e  |  %7.1f us/launch  %6.1f TFLOP/s fp32-equivalent\n",
         name, WSRC ? "reg" : "lds", TD, OCC, PIPE ? "pipelined" : UNR ? "unrolled " : "loop     ", lds / 1024.0, sqrt(e2 / ho.size()), emax, ms * 1e3 / launches,
         flop / (ms * 1e-3) / 1e12);
  fflush(stdout);
  CHECK(hipFree(dx)); CHECK(hipFree(dw)); CHECK(hipFree(dxp)); CHECK(hipFree(dwp)); CHECK(hipFree(dout));
}

template <int TD> static void make_problem(int Cin) {
  constexpr int HALO = halo_of(TD), NVOX = nvox_of(TD);
  unsigned long long st = 88172645463325252ull + TD;
  auto rnd = [&]() { // xorshift -> roughly N(0,1) by summing 4 uniforms
    double s = 0;
    for (int i = 0; i < 4; ++i) {
      st ^= st << 13; st ^= st >> 7; st ^= st << 17;
      s += (double)(st >> 11) / 9007199254740992.0 - 0.5;
    }
    return s * 1.7320508;
  };
  g_x[TD].resize((size_t)Cin * HALO);
  for (auto &v : g_x[TD]) { const double t = rnd() * 1.5; v = (float)(t / (1.0 + exp(-t))); } // Swish outputs
  if (g_w.empty()) {
    g_w.resize((size_t)COUT * Cin * 27);
    for (auto &v : g_w) v = (float)(rnd() / sqrt(27.0 * Cin)); // ASYMMETRIC weights
  }
  g_truth[TD].resize((size_t)COUT * NVOX);
  double s2 = 0;
  for (int co = 0; co < COUT; ++co)
    for (int d = 0; d < TD; ++d)
      for (int h = 0; h < TH; ++h)
        for (int x = 0; x < TW; ++x) {
          double a = 0;
          for (int c = 0; c < Cin; ++c)
            for (int t = 0; t < 27; ++t) {
              const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
              a += (double)g_w[((size_t)co * Cin + c) * 27 + t] *
                   (double)g_x[TD][(size_t)c * HALO + ((d + dz) * HH + h + dy) * HW + x + dx];
            }
          g_truth[TD][(size_t)co * NVOX + (d * TH + h) * TW + x] = a;
          s2 += a * a;
        }
  g_rms[TD] = sqrt(s2 / g_truth[TD].size());
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int Cin = 64, n_cu = prop.multiProcessorCount;
  make_problem<2>(Cin);
  make_problem<4>(Cin);
  printf("%s, %d CUs; tiles TDx4x32 x %d channels, Cin %d\n", prop.name, n_cu, COUT, Cin);
  run<0, 1, 2, 2, 0>("fp16x2", Cin, n_cu);
  run<0, 1, 2, 2, 1>("fp16x2", Cin, n_cu);
  run<0, 1, 2, 2, 1, true>("fp16x2", Cin, n_cu);
  run<0, 1, 2, 1, 1, true>("fp16x2", Cin, n_cu);
  run<0, 0, 2, 2, 1>("fp16x2", Cin, n_cu);
  run<0, 1, 4, 1, 0>("fp16x2", Cin, n_cu);
  run<0, 1, 4, 1, 1>("fp16x2", Cin, n_cu);
  run<1, 1, 2, 2, 1>("bf16x3", Cin, n_cu);
  run<2, 1, 2, 2, 1>("fp16x1", Cin, n_cu);
  run<2, 1, 4, 2, 1>("fp16x1", Cin, n_cu);
  return 0;
}
