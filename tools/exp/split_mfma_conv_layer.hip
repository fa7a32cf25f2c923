// split_mfma_conv_layer.hip -- EXPERIMENT, step 2 (round-2 preparation, not part of liblion_hip.so).
//
// split_mfma_conv_tile.hip measured the inner product of the 3x3x3 convolution on the 16-bit MFMA pipe with fp16x2
// operand splitting (fp32-accurate, ~400 TF fp32-equivalent against 141 TF of the production fp32-MFMA kernel) with
// the split done beforehand.  This file is the whole LAYER the way the product would run it: fp32 NCDHW input read
// with zero padding, the optional AdaGN+Swish prologue, the split into fp16 hi / rescaled lo at LDS-staging time,
// weights pre-split once, bias, fp32 NCDHW output -- to find out what the staging + split cost leaves of the 400 TF.
//   y[b][co][v] = bias[co] + sum_{ci,tap} W[co][ci][tap] * act(x[b][ci][v + tap]),   act = identity or swish(x*pa+pb)
// Geometry as conv3d_k3_kernel: workgroup = 2x4x32 voxels x 64 output channels, 4 waves, grid batch-fastest;
// K in chunks of 16 input channels (one MFMA K).  LDS: operand planes [piece][k-half][832 halo positions][8 x fp16]
// (52 KiB) + weight double buffer (8 KiB): 2 workgroups per CU.  Staging is synchronous (load 8 channels of a halo
// position, activate, split, one 16-byte LDS write per piece); the other workgroup of the CU covers it with MFMAs.
//
// Build + run on an MI355X:
//   hipcc --offload-arch=gfx950 -O3 tools/exp/split_mfma_conv_layer.hip -o /tmp/split_layer && /tmp/split_layer
// prints the error against a float64 host convolution on sampled outputs (incl. borders) and us/launch + TFLOP/s
// for B=32, 64->64 channels, r=32 (production: 1280 us plain / 1620 us with prologue, dense).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int TD = 2, TH = 4, TW = 32, COT = 64, KC = 16;
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2, HALO = HD * HH * HW; // 816
constexpr int HP = 832;                                                   // plane stride: 13 waves x 64 positions

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__device__ __forceinline__ float pro_act(float v, float pa, float pb) {
  const float t = v * pa + pb;
  return t * __frcp_rn(1.0f + __expf(-t));
}
__device__ __forceinline__ void cut(float v, unsigned short &hi, unsigned short &lo) {
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)((v - (float)h) * 2048.f);
  hi = __builtin_bit_cast(unsigned short, h);
  lo = __builtin_bit_cast(unsigned short, l);
}

// __syncthreads() is fence + barrier: hipcc puts s_waitcnt vmcnt(0) in front of it, i.e. the weight load issued in a
// tap is waited for at the next tap's barrier, and the scheduler sinks that load to just before its use: L2 latency
// is exposed 27 times per chunk (seen in the ISA of the first version, 856 us).  TUNED: barrier = this wave's LDS
// writes have landed + s_barrier (register loads are tracked by the compiler's own waits), and the weight load is
// pinned right behind the barrier with a scheduling fence so that it has a whole tap of MFMAs to arrive.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// w f32[Cout][Cin][27] -> wp u16[chunk][tap][piece][g][Cout][8]   (channel ci = chunk*16 + g*8 + j)
__global__ void split_weights(const float *w, int Cin, int Cout, unsigned short *wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cout * Cin * 27) return;
  const int t = i % 27, c = (i / 27) % Cin, co = i / (27 * Cin), chunk = c / KC, g = (c % KC) / 8, j = c % 8;
  unsigned short hi, lo;
  cut(w[i], hi, lo);
  wp[(((((size_t)chunk * 27 + t) * 2 + 0) * 2 + g) * Cout + co) * 8 + j] = hi;
  wp[(((((size_t)chunk * 27 + t) * 2 + 1) * 2 + g) * Cout + co) * 8 + j] = lo;
}

template <bool PRO, int VAR>
__global__ __launch_bounds__(256, 2) void conv_layer_kernel(const float *__restrict__ x, const u4 *__restrict__ wp,
                                                            const float *__restrict__ bias,
                                                            const float *__restrict__ pro_a,
                                                            const float *__restrict__ pro_b, float *__restrict__ y,
                                                            int Cin, int Cout, int r) {
  constexpr int XPL = 2 * 2 * HP;               // u4 in the operand planes
  constexpr int WPL = 2 * 2 * COT;              // u4 per (chunk, tap) weight slice of this channel tile = 256
  constexpr int NI = (2 * HP + 255) / 256;      // staging items (k-half, position) per thread = 7 (last one half used)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u4 *sx = reinterpret_cast<u4 *>(smem);        // [piece][g][HP]
  u4 *sw = sx + XPL;                            // [2][piece][g][COT]
  float *spa = reinterpret_cast<float *>(sw + 2 * WPL), *spb = spa + 256; // prologue scalars, Cin <= 256
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, l32 = lane & 31;
  const int b = blockIdx.x, tile = blockIdx.y, co0 = blockIdx.z * COT;
  const int ntw = r / TW, nth = r / TH;
  const int d0 = (tile / (ntw * nth)) * TD, h0 = ((tile / ntw) % nth) * TH, w0 = (tile % ntw) * TW;
  const int r3 = r * r * r;

  if (PRO)
    for (int c = tid; c < Cin; c += 256) { spa[c] = pro_a[(size_t)b * Cin + c]; spb[c] = pro_b[(size_t)b * Cin + c]; }

  // staging items: item = tid + 256 i -> k-half ig = item / HP (wave uniform: HP is a multiple of 64), position
  // p = item % HP; positions >= HALO are padding, positions outside the grid read 0 through the buffer bounds check
  int goff[NI];
  bool gok[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int item = tid + 256 * i, p = item % HP;
    const int hd = p / (HH * HW), hh = (p / HW) % HH, hw = p % HW;
    const int gd = d0 - 1 + hd, gh = h0 - 1 + hh, gw = w0 - 1 + hw;
    gok[i] = item < 2 * HP && p < HALO && gd >= 0 && gd < r && gh >= 0 && gh < r && gw >= 0 && gw < r;
    goff[i] = gok[i] ? ((gd * r + gh) * r + gw) * 4 : 0x7fffff00;
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(x + (size_t)b * Cin * r3), 0, Cin * r3 * 4, 0x00020000);

  f16v acc[2][2], cor[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = cor[m][n][i] = 0.f;
  int xbase[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int row = 2 * wave + n, d = row / TH, h = row % TH;
    xbase[n] = (d * HH + h) * HW + l32;
  }
  // this thread's u4 of a weight slice: slice element e = (piece*2 + g)*COT + co -> global (piece*2 + g)*Cout + co0 + co
  const int we_g = (tid / COT) * Cout + co0 + (tid % COT);

  // VAR 0: first version.  VAR 1 (TUNED): LDS-only barriers + pinned weight prefetch.  VAR 2 (WREG): every wave loads
  // its weight fragments straight from L2 in operand order, one tap ahead -- no weight LDS, no per-tap barrier: the 4
  // waves (and the 2 workgroups of the CU) drift apart, so one's staging overlaps the others' MFMAs.
  constexpr bool TUNED = VAR == 1, WREG = VAR == 2;
  const int chunks = Cin / KC;
  u4 wreg = wp[we_g];
  u4 wfr[3][2][2]; // WREG: ring of weight fragments, slot = tap % 3 (27 taps per chunk keep the phase), 2 taps ahead
  const int wf_g = g * Cout + co0 + l32;
  if (WREG) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int pc = 0; pc < 2; ++pc)
#pragma unroll
        for (int m = 0; m < 2; ++m)
          wfr[a][m][pc] = wp[(size_t)min(a, chunks * 27 - 1) * 4 * Cout + pc * 2 * Cout + wf_g + m * 32];
  }
  for (int q = 0; q < chunks; ++q) {
    if (TUNED && q) lds_barrier(); else __syncthreads(); // the previous chunk's planes are no longer read (spa/spb visible)
    // all loads of the chunk first (56 per thread in flight; items past the planes carry an out-of-range offset and
    // read 0), then activate + split + write: one memory round trip per chunk instead of one per item
    float v[NI][8];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int ig = __builtin_amdgcn_readfirstlane(min((tid + 256 * i) / HP, 1));
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, goff[i], (q * KC + ig * 8 + j) * r3 * 4, 0));
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int item = tid + 256 * i;
      const int ig = __builtin_amdgcn_readfirstlane(min(item / HP, 1)), p = item - ig * HP;
      unsigned short hi[8], lo[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = v[i][j];
        if (PRO) t = gok[i] ? pro_act(t, spa[q * KC + ig * 8 + j], spb[q * KC + ig * 8 + j]) : 0.f;
        cut(t, hi[j], lo[j]);
      }
      u4 ph, pl;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ph[k] = (unsigned)hi[2 * k] | ((unsigned)hi[2 * k + 1] << 16);
        pl[k] = (unsigned)lo[2 * k] | ((unsigned)lo[2 * k + 1] << 16);
      }
      if (item < 2 * HP) { // wave uniform
        sx[(0 * 2 + ig) * HP + p] = ph;
        sx[(1 * 2 + ig) * HP + p] = pl;
      }
    }
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int s = q * 27 + tap;
      u4 *swb = sw + (s & 1) * WPL;
      u4 wf[2][2], xf[2][2];
      if (WREG) {
        if (tap == 0) __syncthreads(); // the planes of this chunk are written
        const int sn = min(s + 2, chunks * 27 - 1); // the last two requests repeat the last slice (never used)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            wfr[(tap + 2) % 3][m][pc] = wp[(size_t)sn * 4 * Cout + pc * 2 * Cout + wf_g + m * 32];
            wf[m][pc] = wfr[tap % 3][m][pc];
          }
        __builtin_amdgcn_sched_barrier(0); // the requests stay in front of this tap's MFMAs
      } else {
        swb[tid] = wreg;
        if (TUNED) lds_barrier(); else __syncthreads(); // weight slice s (and, for tap 0, the planes) are in LDS
        if (s + 1 < chunks * 27) wreg = wp[(size_t)(s + 1) * 4 * Cout + we_g];
        if (TUNED) __builtin_amdgcn_sched_barrier(0); // the load stays here, a whole tap ahead of its use
      }
      const int toff = ((tap / 9) * HH + (tap / 3) % 3) * HW + tap % 3;
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        if (!WREG) {
#pragma unroll
          for (int m = 0; m < 2; ++m) wf[m][pc] = swb[(pc * 2 + g) * COT + m * 32 + l32];
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) xf[n][pc] = sx[(pc * 2 + g) * HP + xbase[n] + toff];
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, wf[m][0]),
                                                             __builtin_bit_cast(h8, xf[n][0]), acc[m][n], 0, 0, 0);
          cor[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, wf[m][0]),
                                                             __builtin_bit_cast(h8, xf[n][1]), cor[m][n], 0, 0, 0);
          cor[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, wf[m][1]),
                                                             __builtin_bit_cast(h8, xf[n][0]), cor[m][n], 0, 0, 0);
        }
      if (TUNED || WREG) __builtin_amdgcn_sched_barrier(0); // the next tap's LDS write + barrier / loads stay behind these MFMAs
    }
  }
  // epilogue: D[row = channel][col = voxel], col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
  float *yb = y + ((size_t)b * Cout + co0) * r3;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int row = 2 * wave + n, d = row / TH, h = row % TH;
    const int gv = ((d0 + d) * r + (h0 + h)) * r + w0 + l32;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = m * 32 + (i & 3) + 8 * (i >> 2) + 4 * g;
        yb[(size_t)co * r3 + gv] = acc[m][n][i] + cor[m][n][i] * (1.f / 2048.f) + (bias ? bias[co0 + co] : 0.f);
      }
  }
}

template <bool PRO, int VAR>
static void run(const char *name, int B, int Cin, int Cout, int r, const std::vector<float> &hx,
                const std::vector<float> &hw, const std::vector<float> &hb, const std::vector<float> &hpa,
                const std::vector<float> &hpb, float *dx, unsigned short *dwp, float *db, float *dpa, float *dpb,
                float *dy) {
  const int r3 = r * r * r, tiles = (r / TD) * (r / TH) * (r / TW);
  const size_t lds = (size_t)(2 * 2 * HP + 2 * 2 * 2 * COT) * 16 + 2 * 256 * 4;
  auto kern = &conv_layer_kernel<PRO, VAR>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const dim3 grid(B, tiles, Cout / COT);
  const u4 *wp = reinterpret_cast<const u4 *>(dwp);
  kern<<<grid, 256, lds>>>(dx, wp, db, dpa, dpb, dy, Cin, Cout, r);
  CHECK(hipDeviceSynchronize());
  // sampled check (float64 host convolution), borders included
  std::vector<float> hy((size_t)B * Cout * r3);
  CHECK(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
  unsigned long long st = 1234567ull;
  auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  double e2 = 0, emax = 0, s2 = 0;
  const int NS = 1500;
  std::vector<double> want((size_t)NS * Cout), got((size_t)NS * Cout);
  for (int sidx = 0; sidx < NS; ++sidx) {
    const int b = next() % B;
    int d = next() % r, h = next() % r, w = next() % r;
    if (sidx % 5 == 0) d = (sidx & 8) ? 0 : r - 1; // faces / edges / corners
    if (sidx % 7 == 0) w = (sidx & 16) ? 0 : r - 1;
    if (sidx % 11 == 0) h = (sidx & 32) ? 0 : r - 1;
    std::vector<double> vals((size_t)Cin * 27, 0.0); // activated, zero-padded neighbourhood of this voxel
    for (int c = 0; c < Cin; ++c)
      for (int t = 0; t < 27; ++t) {
        const int zd = d + t / 9 - 1, zh = h + (t / 3) % 3 - 1, zw = w + t % 3 - 1;
        if (zd < 0 || zd >= r || zh < 0 || zh >= r || zw < 0 || zw >= r) continue;
        double v = hx[((size_t)b * Cin + c) * r3 + (zd * r + zh) * r + zw];
        if (PRO) {
          const double tt = v * hpa[(size_t)b * Cin + c] + hpb[(size_t)b * Cin + c];
          v = tt / (1.0 + exp(-tt));
        }
        vals[(size_t)c * 27 + t] = v;
      }
    for (int co = 0; co < Cout; ++co) {
      double a = hb[co];
      for (int k = 0; k < Cin * 27; ++k) a += (double)hw[(size_t)co * Cin * 27 + k] * vals[k];
      want[(size_t)sidx * Cout + co] = a;
      got[(size_t)sidx * Cout + co] = hy[((size_t)b * Cout + co) * r3 + (d * r + h) * r + w];
      s2 += a * a;
    }
  }
  const double rms = sqrt(s2 / want.size());
  for (size_t i = 0; i < want.size(); ++i) {
    const double e = (got[i] - want[i]) / rms;
    e2 += e * e;
    emax = fmax(emax, fabs(e));
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int launches = 5;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < launches; ++i) kern<<<grid, 256, lds>>>(dx, wp, db, dpa, dpb, dy, Cin, Cout, r);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 2.0 * 27 * Cin * Cout * (double)r3 * B;
  printf("%-26s B=%d %d->%d r=%d  rms err %.2e  max err %.2e  |  %7.1f us/launch  %6.1f TFLOP/s fp32-equivalent\n", name,
         B, Cin, Cout, r, sqrt(e2 / want.size()), emax, ms * 1e3 / launches, flop / (ms * 1e-3 / launches) / 1e12);
  fflush(stdout);
}

int main() {
  const int B = 32, Cin = 64, Cout = 64, r = 32, r3 = r * r * r;
  std::vector<float> hx((size_t)B * Cin * r3), hw((size_t)Cout * Cin * 27), hb(Cout), hpa((size_t)B * Cin),
      hpb((size_t)B * Cin);
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() {
    double s = 0;
    for (int i = 0; i < 4; ++i) {
      st ^= st << 13; st ^= st >> 7; st ^= st << 17;
      s += (double)(st >> 11) / 9007199254740992.0 - 0.5;
    }
    return s * 1.7320508;
  };
  for (auto &v : hx) v = (float)rnd();
  for (auto &v : hw) v = (float)(rnd() / sqrt(27.0 * Cin));
  for (auto &v : hb) v = (float)(0.1 * rnd());
  for (auto &v : hpa) v = (float)(1.0 + 0.2 * rnd());
  for (auto &v : hpb) v = (float)(0.3 * rnd());
  float *dx, *dw, *db, *dpa, *dpb, *dy;
  unsigned short *dwp;
  CHECK(hipMalloc(&dx, hx.size() * 4));
  CHECK(hipMalloc(&dw, hw.size() * 4));
  CHECK(hipMalloc(&db, hb.size() * 4));
  CHECK(hipMalloc(&dpa, hpa.size() * 4));
  CHECK(hipMalloc(&dpb, hpb.size() * 4));
  CHECK(hipMalloc(&dy, (size_t)B * Cout * r3 * 4));
  CHECK(hipMalloc(&dwp, (size_t)(Cin / KC) * 27 * 2 * 2 * Cout * 16));
  CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dpa, hpa.data(), hpa.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dpb, hpb.data(), hpb.size() * 4, hipMemcpyHostToDevice));
  split_weights<<<(Cout * Cin * 27 + 255) / 256, 256>>>(dw, Cin, Cout, dwp);
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("%s, %d CUs; whole layer, fp16x2 split at staging time\n", prop.name, prop.multiProcessorCount);
  run<false, 0>("plain", B, Cin, Cout, r, hx, hw, hb, hpa, hpb, dx, dwp, db, dpa, dpb, dy);
  run<false, 1>("plain, tuned barriers", B, Cin, Cout, r, hx, hw, hb, hpa, hpb, dx, dwp, db, dpa, dpb, dy);
  run<false, 2>("plain, weights from L2", B, Cin, Cout, r, hx, hw, hb, hpa, hpb, dx, dwp, db, dpa, dpb, dy);
  run<true, 0>("prologue", B, Cin, Cout, r, hx, hw, hb, hpa, hpb, dx, dwp, db, dpa, dpb, dy);
  run<true, 1>("prologue, tuned barriers", B, Cin, Cout, r, hx, hw, hb, hpa, hpb, dx, dwp, db, dpa, dpb, dy);
  run<true, 2>("prologue, weights from L2", B, Cin, Cout, r, hx, hw, hb, hpa, hpb, dx, dwp, db, dpa, dpb, dy);
  return 0;
}
