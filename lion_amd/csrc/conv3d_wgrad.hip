// conv3d_wgrad.hip -- weight gradient of the 3x3x3 / pad 1 / stride 1 Conv3d (training path of C3,
// models/pvcnn2_ada.py:211-222; the reference reaches cuDNN's backward-filter through autograd).
//
//   gw[co][ci][tap] = sum_{b, v} gy[b][co][v] * xpad[b][ci][v + tap]          (xpad: zero padded input)
//
// As a GEMM this is M = Cout, N = Cin * 27, K = B * r^3 (one million for 32 x 32^3): K is the long axis.  On
// v_mfma_f32_32x32x2_f32 the voxels therefore sit on the MFMA k axis: A = gy (32 output channels x 2 voxels, from an
// LDS tile [co][voxel] with an odd row stride -> conflict free), B = the haloed input tile shifted by the tap of the
// column (2 voxels x 32 (ci, tap) columns; the column order ci*27 + tap is exactly the memory order of a gw row, so
// the accumulator tile is written straight into the [Cout][Cin][27] layout).  A workgroup owns 32 output channels x
// CIT input channels (CIT*27 columns = 7 column blocks at CIT = 8) and walks the spatial tiles of ONE sample; its 4
// waves split each 256-voxel tile (combined through LDS in fixed order at the end), so there are B * splits partial
// results per (co, ci) tile, summed in a fixed order by a second kernel (deterministic, no atomics).  The next tile's
// loads travel through registers under the current tile's MFMAs.  Exact fp32 products, like the forward.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int TD, int TH, int TW, int CIT>
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                              int Cin, int Cout, int r, int TS,
                                                              float *__restrict__ partial) {
  constexpr int TV = TD * TH * TW; // 256 voxels per tile
  static_assert(TV == 256, "4 waves x 64 voxels");
  constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2, HALO = HD * HH * HW;
  constexpr int NCOL = CIT * 27, NB = (NCOL + 31) / 32;
  constexpr int GS = TV + 1; // row stride of the gy tile: odd -> the 32 channels of an A operand hit 32 banks
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *sgy = smem;            // [32][GS]
  float *sx = sgy + 32 * GS;    // [CIT][HALO]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / TS, ts = blockIdx.x % TS, cit = blockIdx.y, cot = blockIdx.z; // ts: every TS-th tile
  const int ci0 = cit * CIT, co0 = cot * 32;
  const int r2 = r * r, r3 = r2 * r;
  const int ntw = r / TW, nth = r / TH, ntiles = (r / TD) * nth * ntw;
  const int cl = lane & 31, kh = lane >> 5;

  // per-lane column constants: column c = nb*32 + cl -> (ci, tap) -> LDS offset of the shifted input
  int cofs[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int c = min(nb * 32 + cl, NCOL - 1); // padded columns recompute the last one; never stored
    const int ci = c / 27, tap = c - ci * 27;
    cofs[nb] = ci * HALO + ((tap / 9) * HH + (tap / 3) % 3) * HW + tap % 3;
  }
  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(x + ((size_t)b * Cin + ci0) * r3), 0, CIT * r3 * 4, 0x00020000);
  const float *gyb = gy + ((size_t)b * Cout + co0) * r3;

  // Staging through registers: the loads of tile t + TS are issued in front of the MFMAs of tile t (round 3; the kernel was
  // load -> barrier -> compute -> barrier with only the co-resident workgroup to hide the loads behind).
  constexpr int NXI = (HALO + 255) / 256; // halo positions per thread
  float rgy[32], rx[NXI][CIT];
  auto load_tile = [&](int t) {
    const int tw_i = t % ntw, th_i = (t / ntw) % nth, td_i = t / (ntw * nth);
    const int d0 = td_i * TD, h0 = th_i * TH, w0 = tw_i * TW;
    {
      const int v = tid, d = v / (TH * TW), h = (v / TW) % TH, w = v % TW;
      const int gv = ((d0 + d) * r + (h0 + h)) * r + (w0 + w);
#pragma unroll
      for (int c = 0; c < 32; ++c) rgy[c] = gyb[(size_t)c * r3 + gv];
    }
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int p = tid + 256 * i;
      const int hd = p / (HH * HW), hh = (p / HW) % HH, hw = p % HW;
      const int gd = d0 - 1 + hd, gh = h0 - 1 + hh, gw = w0 - 1 + hw;
      const bool ok = p < HALO && gd >= 0 && gd < r && gh >= 0 && gh < r && gw >= 0 && gw < r;
      const int off = ok ? ((gd * r + gh) * r + gw) * 4 : 0x7fffff00; // zero padding through out-of-range buffer offsets
#pragma unroll
      for (int c = 0; c < CIT; ++c)
        rx[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, off, c * r3 * 4, 0));
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int c = 0; c < 32; ++c) sgy[c * GS + tid] = rgy[c];
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int p = tid + 256 * i;
      if (p < HALO) {
#pragma unroll
        for (int c = 0; c < CIT; ++c) sx[c * HALO + p] = rx[i][c];
      }
    }
  };
  if (ts < ntiles) load_tile(ts);
  for (int t = ts; t < ntiles; t += TS) {
    __syncthreads(); // the previous tile's LDS reads are done
    store_tile();
    __syncthreads();
    if (t + TS < ntiles) load_tile(t + TS); // in flight under the MFMAs below
    // 32 k-steps of 2 voxels over this wave's 64 voxels
#pragma unroll 4
    for (int s = 0; s < 32; ++s) {
      const int v = wave * 64 + 2 * s + kh;
      const int d = v / (TH * TW), h = (v / TW) % TH, w = v % TW;
      const int vb = (d * HH + h) * HW + w;
      const float a = sgy[cl * GS + v];
      float bq[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bq[nb] = sx[cofs[nb] + vb];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[nb], acc[nb], 0, 0, 0);
    }
  }
  // The four waves hold partial sums over disjoint voxels of the same (co, column) tile: combine them through LDS in fixed
  // order (w0 + w1) + (w2 + w3), one column block at a time, and write ONE partial per workgroup (round 3: the reduce kernel
  // read 4x the bytes).  partial[blockIdx.x][co][ci*27 + tap]; acc register i of lane l: row (i&3) + 8*(i>>2) + 4*(l>>5),
  // column l&31
  float *pt = partial + (size_t)blockIdx.x * Cout * Cin * 27;
  float *red = smem; // [4 waves][16][64]
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[nb][i];
    __syncthreads();
    // 1024 values per column block, 4 per thread: value e = i * 64 + lane'
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = tid + 256 * k, i = e >> 6, ln = e & 63;
      const float v = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
      const int c = nb * 32 + (ln & 31);
      if (c < NCOL) {
        const int co = co0 + (i & 3) + 8 * (i >> 2) + 4 * (ln >> 5);
        pt[((size_t)co * Cin + ci0) * 27 + c] = v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void conv3d_wgrad_reduce_kernel(const float *__restrict__ partial, int nparts,
                                                                  size_t n, float *__restrict__ gw) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0.0; // the partials are sums of ~8k products each; their sum (up to 1024 of them) is taken in double
  for (int p = 0; p < nparts; ++p) s += (double)partial[(size_t)p * n + i]; // fixed order
  gw[i] = (float)s;
}

template <int TD, int TH, int TW, int CIT>
static int launch_wgrad(const float *x, const float *gy, int B, int Cin, int Cout, int r, int TS, float *partial,
                        hipStream_t st) {
  constexpr int HALO = (TD + 2) * (TH + 2) * (TW + 2);
  const size_t lds = (size_t)(32 * 257 + CIT * HALO) * 4;
  static LionLdsLimit cfg = {};
  if (int e = lion_dynamic_lds(&conv3d_wgrad_kernel<TD, TH, TW, CIT>, lds, cfg)) return e;
  conv3d_wgrad_kernel<TD, TH, TW, CIT><<<dim3(B * TS, Cin / CIT, Cout / 32), 256, lds, st>>>(x, gy, Cin, Cout, r, TS,
                                                                                           partial);
  LION_LAUNCH_CHECK();
  return 0;
}

} // namespace

extern "C" {

// spatial splits per sample: enough workgroups for ~2 per CU when the channel tiles alone are too few
static int wgrad_splits(int B, int Cin, int Cout, int r) {
  const int cit = Cin % 8 == 0 ? 8 : 4;
  const long wg = (long)B * (Cin / cit) * (Cout / 32);
  const int ntiles = r * r * r / 256;
  int ts = 1;
  while (ts < ntiles && ts < 8 && wg * ts < 512) ts *= 2;
  return ts;
}

// floats of scratch: B * splits partial copies of the [Cout,Cin,27] gradient (one per workgroup)
size_t lion_conv3d_wgrad_workspace_floats(int B, int Cin, int Cout, int r) {
  if (Cin % 4 != 0 || Cout % 32 != 0 || (r != 8 && r != 16 && r != 32)) return 0;
  return (size_t)B * wgrad_splits(B, Cin, Cout, r) * Cout * Cin * 27;
}

// x f32[B,Cin,r,r,r] (Cin % 4 == 0), gy f32[B,Cout,r,r,r] (Cout % 32 == 0), r in {8,16,32} -> gw f32[Cout,Cin,3,3,3]
int lion_conv3d_k3_wgrad(const float *x, const float *gy, int B, int Cin, int Cout, int r, float *gw, float *ws,
                         size_t ws_floats, lionStream_t stream) {
  if (!x || !gy || !gw || B <= 0 || Cin <= 0 || Cout <= 0) return LION_EINVAL;
  if (Cin % 4 != 0 || Cout % 32 != 0 || (r != 8 && r != 16 && r != 32)) return LION_EUNSUPPORTED;
  if (!ws || ws_floats < lion_conv3d_wgrad_workspace_floats(B, Cin, Cout, r)) return LION_EWORKSPACE;
  const int TS = wgrad_splits(B, Cin, Cout, r);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool c8 = Cin % 8 == 0;
  int rc;
  if (r == 32) rc = c8 ? launch_wgrad<2, 4, 32, 8>(x, gy, B, Cin, Cout, r, TS, ws, st) : launch_wgrad<2, 4, 32, 4>(x, gy, B, Cin, Cout, r, TS, ws, st);
  else if (r == 16) rc = c8 ? launch_wgrad<4, 4, 16, 8>(x, gy, B, Cin, Cout, r, TS, ws, st) : launch_wgrad<4, 4, 16, 4>(x, gy, B, Cin, Cout, r, TS, ws, st);
  else rc = c8 ? launch_wgrad<4, 8, 8, 8>(x, gy, B, Cin, Cout, r, TS, ws, st) : launch_wgrad<4, 8, 8, 4>(x, gy, B, Cin, Cout, r, TS, ws, st);
  if (rc) return rc;
  const size_t n = (size_t)Cout * Cin * 27;
  conv3d_wgrad_reduce_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(ws, B * TS, n, gw);
  LION_LAUNCH_CHECK();
  return 0;
}

} // extern "C"
