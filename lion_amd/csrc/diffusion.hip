// diffusion.hip -- D1: one fused elementwise kernel per denoiser update.
//
// Reference: utils/diffusion_pvd.py:451-467 (DDIM: ~6 ATen launches + a host->device noise copy
// per step) and :283-296 with get_q_posterior_mean :475-486 (DDPM ancestral step).
// The scalar coefficients are computed on the host exactly as the reference computes them
// (fp32 0-d tensor arithmetic); the kernel applies them in the reference's operation order with
// one rounding per operation, so the update is bit-exact versus the oracle for identical inputs.
// x is [B, 8192] or [B, 128] floats: launch-latency bound, hence a single launch, float4 lanes.
#include "common.h"

namespace {

__device__ __forceinline__ float ddim1(float x, float e, float z, float s, float c, float sigma) {
  // x = x_noisy*sqrt(a_next/a_t);  x += c*eps + sigma*randn
  return add_rn(mul_rn(x, s), add_rn(mul_rn(c, e), mul_rn(sigma, z)));
}

__global__ void ddim_kernel(const float *__restrict__ x, const float *__restrict__ eps,
                            const float *__restrict__ z, size_t numel, float s, float c,
                            float sigma, float *__restrict__ out) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < numel) {
    const float4 xv = *reinterpret_cast<const float4 *>(x + i);
    const float4 ev = *reinterpret_cast<const float4 *>(eps + i);
    float4 zv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (z) zv = *reinterpret_cast<const float4 *>(z + i);
    *reinterpret_cast<float4 *>(out + i) =
        make_float4(ddim1(xv.x, ev.x, zv.x, s, c, sigma), ddim1(xv.y, ev.y, zv.y, s, c, sigma),
                    ddim1(xv.z, ev.z, zv.z, s, c, sigma), ddim1(xv.w, ev.w, zv.w, s, c, sigma));
  } else {
    for (size_t j = i; j < numel; ++j) out[j] = ddim1(x[j], eps[j], z ? z[j] : 0.f, s, c, sigma);
  }
}

__global__ void ddpm_kernel(const float *__restrict__ x, const float *__restrict__ eps,
                            const float *__restrict__ z, size_t numel, int t_is_zero, float k_outer,
                            float k_a, float k_b, float scale, float temp, float *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= numel) return;
  if (t_is_zero) { // :479-480
    out[i] = mul_rn(k_outer, sub_rn(x[i], mul_rn(k_a, eps[i])));
  } else {         // :482-484, then :293-294
    const float mean = mul_rn(k_outer, sub_rn(x[i], div_rn(mul_rn(k_a, eps[i]), k_b)));
    out[i] = add_rn(mean, mul_rn(mul_rn(scale, z[i]), temp));
  }
}


// ---- whole-chain replay: device-resident step state + on-chip noise -----------------------------------------
// A sampling chain is 1000 replays of ONE captured graph [begin_step -> denoiser forward -> update_noise]; nothing
// a step needs may come from the host.  The schedule lives in a device table (8 floats per step:
// {t_model, a0..a5, -}); begin_step reads the step counter, broadcasts t to the denoiser's [B] timestep input,
// copies the step's coefficients to `cur` (cur[7] = the step index, as bits) and advances the counter.  The
// update kernel draws z ~ N(0,1) itself: Philox4x32-10 keyed by the chain's 64-bit seed, counter = (quad index,
// step, stream id) -- the seed is read from device memory so that a captured graph serves every chain --, four uniforms -> two Box-Muller pairs per float4 -- the reference draws the noise on the CPU
// and copies it to the device every step (diffusion_pvd.py:465-466).  Algorithmic bytes per step: x, eps in,
// x out (3*4*numel).
__device__ __forceinline__ float u01(uint32_t x) { // (0, 1]: x * 2^-32 + 2^-33, two roundings
  return add_rn(mul_rn((float)x, 2.3283064365386963e-10f), 1.1641532182693481e-10f);
}

__device__ __forceinline__ void normal4(uint64_t quad, uint32_t step, uint32_t stream_id, uint64_t seed, float z[4]) {
  uint32_t r[4];
  philox4x32_10((uint32_t)quad, (uint32_t)(quad >> 32), step, stream_id, (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float rad = sqrtf(mul_rn(-2.0f, logf(u01(r[2 * h]))));
    float sn, cs;
    sincospif(mul_rn(2.0f, u01(r[2 * h + 1])), &sn, &cs);
    z[2 * h] = mul_rn(rad, cs);
    z[2 * h + 1] = mul_rn(rad, sn);
  }
}

__global__ void begin_step_kernel(const float *__restrict__ table, int n_steps, int32_t *counter,
                                  float *__restrict__ t_out, int B, float *__restrict__ cur,
                                  const float *__restrict__ temb_table, int temb_width, float *__restrict__ temb_out) {
  __shared__ int step;
  if (threadIdx.x == 0) {
    int i = *counter;
    i = i < 0 ? 0 : (i >= n_steps ? n_steps - 1 : i);
    step = i;
  }
  __syncthreads();
  const int i = step;
  const float *row = table + (size_t)i * 8;
  for (int b = threadIdx.x; b < B; b += blockDim.x) t_out[b] = row[0];
  if (threadIdx.x < 7) cur[threadIdx.x] = row[threadIdx.x];
  if (threadIdx.x == 7) cur[7] = __int_as_float(i);
  if (threadIdx.x == 0) *counter = i + 1;
  if (temb_table)   // the step's row of the chain's time-embedding table (was an ATen index_select per step)
    for (int k = threadIdx.x; k < temb_width; k += blockDim.x) temb_out[k] = temb_table[(size_t)i * temb_width + k];
}

// MODE 0: DDIM  out = x*a0 + (a1*eps + a2*z)                     (a = {s, c, sigma})
// MODE 1: DDPM  a5 != 0 (t == 0): out = a0*(x - a1*eps);  else out = a0*(x - a1*eps/a2) + (a3*z)*a4
template <int MODE>
__global__ void update_noise_kernel(const float *__restrict__ x, const float *__restrict__ eps, size_t numel,
                                    const float *__restrict__ cur, const uint32_t *__restrict__ seed_words,
                                    uint32_t stream_id, float *__restrict__ out, float *__restrict__ z_out, int cm_points) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = q * 4;
  if (i >= numel) return;
  const float a0 = cur[1], a1 = cur[2], a2 = cur[3], a3 = cur[4], a4 = cur[5], a5 = cur[6];
  const uint32_t step = (uint32_t)__float_as_int(cur[7]);
  const uint64_t seed = (uint64_t)seed_words[0] | ((uint64_t)seed_words[1] << 32);
  float z[4];
  normal4(q, step, stream_id, seed, z);
  float xv[4], ev[4], o[4];
  const bool full = i + 3 < numel;
  if (cm_points) {
    // x is [B][N][4] (point-major latent), eps the denoiser's channel-major output [B][4][N]: quad q = (b, n) takes its four
    // channels from four rows -- the permute(0, 2, 1).contiguous() copy of the model's output folded into this read
    const float4 t = *reinterpret_cast<const float4 *>(x + i);
    xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
    const size_t b = q / (size_t)cm_points, n = q - b * (size_t)cm_points;
#pragma unroll
    for (int j = 0; j < 4; ++j) ev[j] = eps[(b * 4 + j) * (size_t)cm_points + n];
  } else if (full) {
    const float4 t = *reinterpret_cast<const float4 *>(x + i), u = *reinterpret_cast<const float4 *>(eps + i);
    xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
    ev[0] = u.x; ev[1] = u.y; ev[2] = u.z; ev[3] = u.w;
  } else {
    for (int j = 0; j < 4; ++j) { xv[j] = i + j < numel ? x[i + j] : 0.f; ev[j] = i + j < numel ? eps[i + j] : 0.f; }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (MODE == 0) {
      o[j] = ddim1(xv[j], ev[j], z[j], a0, a1, a2);
    } else if (a5 != 0.f) {
      o[j] = mul_rn(a0, sub_rn(xv[j], mul_rn(a1, ev[j])));
    } else {
      o[j] = add_rn(mul_rn(a0, sub_rn(xv[j], div_rn(mul_rn(a1, ev[j]), a2))), mul_rn(mul_rn(a3, z[j]), a4));
    }
  }
  if (full) {
    *reinterpret_cast<float4 *>(out + i) = make_float4(o[0], o[1], o[2], o[3]);
    if (z_out) *reinterpret_cast<float4 *>(z_out + i) = make_float4(z[0], z[1], z[2], z[3]);
  } else {
    for (int j = 0; j < 4 && i + j < numel; ++j) { out[i + j] = o[j]; if (z_out) z_out[i + j] = z[j]; }
  }
}

} // namespace

extern "C" {

int lion_ddim_update(const float *x, const float *eps, const float *z, size_t numel, float s,
                     float c, float sigma, float *out, lionStream_t stream) {
  if (!x || !eps || !out || numel == 0) return LION_EINVAL;
  if (!z && sigma != 0.f) return LION_EINVAL;
  const bool aligned = ((((uintptr_t)x) | ((uintptr_t)eps) | ((uintptr_t)out) | ((uintptr_t)z)) & 15) == 0;
  if (!aligned) return LION_EINVAL; // torch allocations are 256-byte aligned
  const size_t quads = (numel + 3) / 4;
  ddim_kernel<<<(unsigned)((quads + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      x, eps, z, numel, s, c, sigma, out);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_ddpm_update(const float *x, const float *eps, const float *z, size_t numel, int t_is_zero,
                     float k_outer, float k_a, float k_b, float scale, float temp, float *out,
                     lionStream_t stream) {
  if (!x || !eps || !out || numel == 0) return LION_EINVAL;
  if (!t_is_zero && !z) return LION_EINVAL;
  ddpm_kernel<<<(unsigned)((numel + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      x, eps, z, numel, t_is_zero, k_outer, k_a, k_b, scale, temp, out);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_chain_begin_step(const float *table, int n_steps, int32_t *counter, float *t_out, int B, float *cur,
                          lionStream_t stream) {
  if (!table || !counter || !t_out || !cur || n_steps <= 0 || B <= 0) return LION_EINVAL;
  begin_step_kernel<<<1, 64, 0, static_cast<hipStream_t>(stream)>>>(table, n_steps, counter, t_out, B, cur, nullptr, 0,
                                                                    nullptr);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_chain_begin_step_temb(const float *table, int n_steps, int32_t *counter, float *t_out, int B, float *cur,
                               const float *temb_table, int temb_width, float *temb_out, lionStream_t stream) {
  if (!table || !counter || !t_out || !cur || n_steps <= 0 || B <= 0) return LION_EINVAL;
  if (!temb_table || !temb_out || temb_width <= 0) return LION_EINVAL;
  begin_step_kernel<<<1, 64, 0, static_cast<hipStream_t>(stream)>>>(table, n_steps, counter, t_out, B, cur, temb_table,
                                                                    temb_width, temb_out);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_chain_update_noise(int mode, const float *x, const float *eps, size_t numel, const float *cur,
                            const uint32_t *seed, uint32_t stream_id, float *out, float *z_out,
                            lionStream_t stream) {
  if (!x || !eps || !cur || !seed || !out || numel == 0 || (mode != 0 && mode != 1)) return LION_EINVAL;
  if (((((uintptr_t)x) | ((uintptr_t)eps) | ((uintptr_t)out) | ((uintptr_t)z_out)) & 15) != 0) return LION_EINVAL;
  const size_t quads = (numel + 3) / 4;
  const unsigned blocks = (unsigned)((quads + 255) / 256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (mode == 0) update_noise_kernel<0><<<blocks, 256, 0, st>>>(x, eps, numel, cur, seed, stream_id, out, z_out, 0);
  else update_noise_kernel<1><<<blocks, 256, 0, st>>>(x, eps, numel, cur, seed, stream_id, out, z_out, 0);
  LION_LAUNCH_CHECK();
  return 0;
}

int lion_chain_update_noise_cm(int mode, const float *x, const float *eps_cm, int B, int N, const float *cur,
                               const uint32_t *seed, uint32_t stream_id, float *out, float *z_out, lionStream_t stream) {
  if (!x || !eps_cm || !cur || !seed || !out || B <= 0 || N <= 0 || (mode != 0 && mode != 1)) return LION_EINVAL;
  if (((((uintptr_t)x) | ((uintptr_t)out) | ((uintptr_t)z_out)) & 15) != 0) return LION_EINVAL;
  const size_t numel = (size_t)B * N * 4, quads = (size_t)B * N;
  const unsigned blocks = (unsigned)((quads + 255) / 256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (mode == 0) update_noise_kernel<0><<<blocks, 256, 0, st>>>(x, eps_cm, numel, cur, seed, stream_id, out, z_out, N);
  else update_noise_kernel<1><<<blocks, 256, 0, st>>>(x, eps_cm, numel, cur, seed, stream_id, out, z_out, N);
  LION_LAUNCH_CHECK();
  return 0;
}

} // extern "C"
