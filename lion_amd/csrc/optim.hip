// optim.hip -- the optimizer update of a training step (reference utils/utils.py:115-121: torch.optim.Adam with L2 weight decay;
// trainers/hvae_trainer.py:150-154, train_2prior.py:405-410 call optimizer.step() after the gradient averaging) as ONE launch over
// every parameter tensor of the model.
//
// ATen's multi-tensor Adam packs at most 36 tensors' pointers into a launch's arguments: the VAE's 867 parameter tensors
// (22.4 M floats) become 25 launches of ~39 us, the priors' 88 M floats 20 launches of ~66 us, where streaming p, g, m, v in and
// p, m, v out once is 0.16 ms / 0.6 ms at HBM rate.  Here the pointers live in a device table (one row per tensor) and a block map
// (one row per 4096-element chunk), both written once per gradient layout; a workgroup looks its chunk up and streams it.
// HBM bound: 28 algorithmic bytes per parameter (36 with the moving average of the weights, utils/ema.py, in the same pass).
//
// Arithmetic = torch.optim.Adam's single-tensor path, op for op in fp32 (the build has -ffp-contract=off):
//   g' = g + wd p;  m = m + (g' - m)(1 - b1);  v = v b2 + (1 - b2) g' g';  p = p - (lr / (1 - b1^t)) (m / (sqrt(v) / sqrt(1 - b2^t) + eps))
// with the bias corrections in double from the parameter's step count t, which lives in device memory (a captured step replays).
#include "common.h"

namespace {

constexpr int ADAM_CHUNK = 4096;   // elements per workgroup

// table row: {param, grad, exp_avg, exp_avg_sq, step, ema} -- torch.optim.Adam counts steps per parameter (one that got no
// gradient in a step is skipped and falls behind); ema = 0: no moving average of this tensor
constexpr int ADAM_ROW = 6;
__global__ void adam_tick_kernel(const unsigned long long *__restrict__ table, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < T) reinterpret_cast<float *>(table[ADAM_ROW * (size_t)t + 4])[0] += 1.0f;
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const unsigned long long *__restrict__ table,   // [T][ADAM_ROW]
                                                         const int *__restrict__ numel,                  // [T]
                                                         const int *__restrict__ blockmap,               // [blocks][2]: tensor, chunk
                                                         const float *__restrict__ lr_ptr, float beta1, float beta2, float eps,
                                                         float wd, float ema_decay) {
  __shared__ float sh[3];
  const int t = blockmap[2 * blockIdx.x], chunk = blockmap[2 * blockIdx.x + 1];
  if (threadIdx.x == 0) {
    const double s = (double)reinterpret_cast<const float *>(table[ADAM_ROW * (size_t)t + 4])[0];
    sh[2] = s == 1.0 ? 1.f : 0.f;               // the parameter's first step: its moving average starts from the updated value
    const double bc1 = 1.0 - pow((double)beta1, s), bc2 = 1.0 - pow((double)beta2, s);
    sh[0] = (float)((double)lr_ptr[0] / bc1);   // step size
    sh[1] = (float)sqrt(bc2);
  }
  __syncthreads();
  const float step_size = sh[0], bc2s = sh[1], omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
  const bool first = sh[2] != 0.f;
  const unsigned long long ap = table[ADAM_ROW * (size_t)t], ag = table[ADAM_ROW * (size_t)t + 1],
                           am = table[ADAM_ROW * (size_t)t + 2], av = table[ADAM_ROW * (size_t)t + 3],
                           ae = table[ADAM_ROW * (size_t)t + 5];
  float *ema = reinterpret_cast<float *>(ae);
  const float omd = 1.0f - ema_decay;
  float *p = reinterpret_cast<float *>(ap);
  const float *g = reinterpret_cast<const float *>(ag);
  float *m = reinterpret_cast<float *>(am);
  float *v = reinterpret_cast<float *>(av);
  const int n = numel[t], lo = chunk * ADAM_CHUNK, hi = min(n, lo + ADAM_CHUNK);
  auto upd = [&](float &pp, float gg, float &mm, float &vv) {
    if (wd != 0.f) gg = gg + wd * pp;
    mm = mm + (gg - mm) * omb1;
    vv = vv * beta2 + omb2 * gg * gg;
    const float denom = sqrtf(vv) / bc2s + eps;
    pp = pp - step_size * (mm / denom);
  };
  // utils/ema.py:60-75 (the reference wraps every optimizer in it): ema = ema * decay + (1 - decay) * param AFTER the update; a
  // parameter's first step starts it from the updated parameter
  auto avg = [&](float e, float pp) { return (first ? pp : e) * ema_decay + omd * pp; };
  const bool vec = ((ap | ag | am | av | ae) & 15ull) == 0;
  if (vec) {
    for (int i = lo + threadIdx.x * 4; i < hi; i += 1024) {
      if (i + 3 < hi) {
        float4 P = *reinterpret_cast<float4 *>(p + i), M = *reinterpret_cast<float4 *>(m + i), V = *reinterpret_cast<float4 *>(v + i);
        const float4 G = *reinterpret_cast<const float4 *>(g + i);
        upd(P.x, G.x, M.x, V.x); upd(P.y, G.y, M.y, V.y); upd(P.z, G.z, M.z, V.z); upd(P.w, G.w, M.w, V.w);
        *reinterpret_cast<float4 *>(p + i) = P;
        *reinterpret_cast<float4 *>(m + i) = M;
        *reinterpret_cast<float4 *>(v + i) = V;
        if (ema) {
          float4 E = first ? P : *reinterpret_cast<const float4 *>(ema + i);
          E = make_float4(avg(E.x, P.x), avg(E.y, P.y), avg(E.z, P.z), avg(E.w, P.w));
          *reinterpret_cast<float4 *>(ema + i) = E;
        }
      } else {
        for (int j = i; j < hi; ++j) {
          upd(p[j], g[j], m[j], v[j]);
          if (ema) ema[j] = avg(first ? 0.f : ema[j], p[j]);
        }
      }
    }
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
      upd(p[i], g[i], m[i], v[i]);
      if (ema) ema[i] = avg(first ? 0.f : ema[i], p[i]);
    }
  }
}

} // namespace

extern "C" {

int lion_adam_chunk(void) { return ADAM_CHUNK; }

int lion_adam_row(void) { return ADAM_ROW; }

int lion_adam_step(const uint64_t *table, const int32_t *numel, const int32_t *blockmap, int blocks, int tensors, const float *lr,
                   float beta1, float beta2, float eps, float weight_decay, float ema_decay, lionStream_t stream) {
  if (!table || !numel || !blockmap || !lr || blocks <= 0 || tensors <= 0) return LION_EINVAL;
  if (!(ema_decay >= 0.f && ema_decay <= 1.f)) return LION_EINVAL;
  if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f) || !(weight_decay >= 0.f)) return LION_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned long long *tb = reinterpret_cast<const unsigned long long *>(table);
  adam_tick_kernel<<<lion_cdiv(tensors, 256), 256, 0, st>>>(tb, tensors);
  adam_multi_kernel<<<blocks, 256, 0, st>>>(tb, numel, blockmap, lr, beta1, beta2, eps, weight_decay, ema_decay);
  LION_LAUNCH_CHECK();
  return 0;
}

} // extern "C"
