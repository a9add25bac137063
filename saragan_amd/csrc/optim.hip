// Fused TF-formulation Adam + EMA over a flat f32 parameter range, and per-segment sum of squares
// (gradient norms / global-norm clipping).  Reference: tf.train.AdamOptimizer as called at
// SURFGAN_3D/optimization.py:16,28; tf.train.ExponentialMovingAverage via ExtendedEMA.py:56-59;
// tf.norm / tf.clip_by_global_norm at optimization.py:66-71.  One pass: 5 reads + 4 writes of 4 B/param.
#include "common.h"

namespace {

__global__ void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, float* __restrict__ ema, int64_t numel, float lr_t, float b1,
                                float b2, float eps, float gscale, float ema_decay, const float* __restrict__ lr_dev) {
  if (lr_dev) lr_t = *lr_dev;      // captured step: the bias-corrected step size is written by the host before each replay
  const int64_t nv = numel / 4;
  const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  const float omd = 1.f - ema_decay;
  for (int64_t i = tid0; i < nv; i += stride) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
    if (g) {
      const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i] * gscale;
      f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
      mm = b1 * mm + (1.f - b1) * gg;
      vv = b2 * vv + (1.f - b2) * gg * gg;
#pragma unroll
      for (int e = 0; e < 4; ++e) pp[e] -= lr_t * mm[e] / (sqrtf(vv[e]) + eps);
      reinterpret_cast<f32x4*>(m)[i] = mm;
      SG_STORE16_GUARD(mm);
      reinterpret_cast<f32x4*>(v)[i] = vv;
      SG_STORE16_GUARD(vv);
      reinterpret_cast<f32x4*>(p)[i] = pp;
      SG_STORE16_GUARD(pp);
    }
    if (ema) {
      f32x4 ee = reinterpret_cast<f32x4*>(ema)[i];
      ee -= omd * (ee - pp);
      reinterpret_cast<f32x4*>(ema)[i] = ee;
      SG_STORE16_GUARD(ee);
    }
  }
  for (int64_t i = nv * 4 + tid0; i < numel; i += stride) {
    float pp = p[i];
    if (g) {
      const float gg = g[i] * gscale;
      const float mm = b1 * m[i] + (1.f - b1) * gg;
      const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
      pp -= lr_t * mm / (sqrtf(vv) + eps);
      m[i] = mm; v[i] = vv; p[i] = pp;
    }
    if (ema) ema[i] -= omd * (ema[i] - pp);
  }
}

// tf.train.GradientDescentOptimizer / MomentumOptimizer(use_nesterov) / AdadeltaOptimizer(rho, epsilon) as created
// at SURFGAN_3D/optimization.py:17-22,29-35, fused with the EMA update like Adam above.  TF's update rules
// (training_ops: ApplyGradientDescent, ApplyMomentum, ApplyAdadelta):
//   SGD       p -= lr * g
//   Momentum  a = h * a + g;  p -= nesterov ? lr * g + lr * h * a : lr * a
//   Adadelta  a = h * a + (1-h) g^2;  u = sqrt(a2 + eps) * rsqrt(a + eps) * g;  p -= lr * u;  a2 = h * a2 + (1-h) u^2
template <int KIND>
__device__ __forceinline__ void optim_rule(float& p, float g, float& s1, float& s2, float lr, float h, float eps,
                                           int nesterov) {
  if (KIND == SG_OPT_SGD) {
    p -= lr * g;
  } else if (KIND == SG_OPT_MOMENTUM) {
    s1 = h * s1 + g;
    p -= nesterov ? (lr * g + lr * h * s1) : lr * s1;
  } else {
    s1 = h * s1 + (1.f - h) * g * g;
    const float u = sqrtf(s2 + eps) * rsqrtf(s1 + eps) * g;
    p -= lr * u;
    s2 = h * s2 + (1.f - h) * u * u;
  }
}

template <int KIND>
__global__ void optim_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ s1,
                                  float* __restrict__ s2, float* __restrict__ ema, int64_t numel, float lr, float h,
                                  float eps, int nesterov, float gscale, float ema_decay, const float* __restrict__ lr_dev) {
  if (lr_dev) lr = *lr_dev;
  const int64_t nv = numel / 4;
  const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  const float omd = 1.f - ema_decay;
  for (int64_t i = tid0; i < nv; i += stride) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i] * gscale;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (KIND != SG_OPT_SGD) a = reinterpret_cast<f32x4*>(s1)[i];
    if (KIND == SG_OPT_ADADELTA) b = reinterpret_cast<f32x4*>(s2)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {   // (ext-vector elements cannot bind to references)
      float pe = pp[e], ae = a[e], be = b[e];
      optim_rule<KIND>(pe, gg[e], ae, be, lr, h, eps, nesterov);
      pp[e] = pe; a[e] = ae; b[e] = be;
    }
    if (KIND != SG_OPT_SGD) reinterpret_cast<f32x4*>(s1)[i] = a;
    SG_STORE16_GUARD(a);
    if (KIND == SG_OPT_ADADELTA) reinterpret_cast<f32x4*>(s2)[i] = b;
    SG_STORE16_GUARD(b);
    reinterpret_cast<f32x4*>(p)[i] = pp;
    SG_STORE16_GUARD(pp);
    if (ema) {
      f32x4 ee = reinterpret_cast<f32x4*>(ema)[i];
      ee -= omd * (ee - pp);
      reinterpret_cast<f32x4*>(ema)[i] = ee;
      SG_STORE16_GUARD(ee);
    }
  }
  for (int64_t i = nv * 4 + tid0; i < numel; i += stride) {
    float pp = p[i], a = KIND != SG_OPT_SGD ? s1[i] : 0.f, b = KIND == SG_OPT_ADADELTA ? s2[i] : 0.f;
    optim_rule<KIND>(pp, g[i] * gscale, a, b, lr, h, eps, nesterov);
    if (KIND != SG_OPT_SGD) s1[i] = a;
    if (KIND == SG_OPT_ADADELTA) s2[i] = b;
    p[i] = pp;
    if (ema) ema[i] -= omd * (ema[i] - pp);
  }
}

__global__ __launch_bounds__(256) void segment_sumsq_kernel(const float* __restrict__ flat,
                                                            const int64_t* __restrict__ offsets,
                                                            float* __restrict__ out) {
  __shared__ float red[256];
  const int seg = blockIdx.x;
  const int64_t b = offsets[seg], e = offsets[seg + 1];
  float s = 0.f;
  for (int64_t i = b + threadIdx.x; i < e; i += 256) s += flat[i] * flat[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[seg] = red[0];
}

}  // namespace

static int adam_ema_launch(float* p, const float* g, float* m, float* v, float* ema, int64_t numel, float lr_t, const float* lr_dev,
                           float b1, float b2, float eps, float gscale, float ema_decay, sg_stream_t st) {
  if (!p || numel < 1 || (g && (!m || !v)) || (!g && !ema)) return SG_EINVAL;
  if (!sg_aligned16(p) || (g && (!sg_aligned16(g) || !sg_aligned16(m) || !sg_aligned16(v))) ||
      (ema && !sg_aligned16(ema)))
    return SG_EALIGN;
  int64_t blocks = (numel / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_ema_kernel, dim3((unsigned)blocks), dim3(256), 0, sg_st(st), p, g, m, v, ema, numel, lr_t,
                     b1, b2, eps, gscale, ema_decay, lr_dev);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_adam_ema(float* p, const float* g, float* m, float* v, float* ema, int64_t numel, float lr_t,
                           float b1, float b2, float eps, float gscale, float ema_decay, sg_stream_t st) {
  return adam_ema_launch(p, g, m, v, ema, numel, lr_t, nullptr, b1, b2, eps, gscale, ema_decay, st);
}

extern "C" int sg_adam_ema_dev(float* p, const float* g, float* m, float* v, float* ema, int64_t numel, const float* lr_t,
                               float b1, float b2, float eps, float gscale, float ema_decay, sg_stream_t st) {
  if (!lr_t) return SG_EINVAL;
  return adam_ema_launch(p, g, m, v, ema, numel, 0.f, lr_t, b1, b2, eps, gscale, ema_decay, st);
}

static int optim_step_launch(int kind, float* p, const float* g, float* s1, float* s2, float* ema, int64_t numel,
                             float lr, const float* lr_dev, float h, float eps, int nesterov, float gscale, float ema_decay,
                             sg_stream_t st) {
  if (!p || !g || numel < 1) return SG_EINVAL;
  if (kind != SG_OPT_SGD && kind != SG_OPT_MOMENTUM && kind != SG_OPT_ADADELTA) return SG_EINVAL;
  if ((kind != SG_OPT_SGD && !s1) || (kind == SG_OPT_ADADELTA && !s2)) return SG_EINVAL;
  if (!sg_aligned16(p) || !sg_aligned16(g) || (s1 && !sg_aligned16(s1)) || (s2 && !sg_aligned16(s2)) ||
      (ema && !sg_aligned16(ema)))
    return SG_EALIGN;
  int64_t blocks = (numel / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  const dim3 grid((unsigned)blocks), blk(256);
  if (kind == SG_OPT_SGD)
    hipLaunchKernelGGL(optim_step_kernel<SG_OPT_SGD>, grid, blk, 0, sg_st(st), p, g, s1, s2, ema, numel, lr, h, eps,
                       nesterov, gscale, ema_decay, lr_dev);
  else if (kind == SG_OPT_MOMENTUM)
    hipLaunchKernelGGL(optim_step_kernel<SG_OPT_MOMENTUM>, grid, blk, 0, sg_st(st), p, g, s1, s2, ema, numel, lr, h,
                       eps, nesterov, gscale, ema_decay, lr_dev);
  else
    hipLaunchKernelGGL(optim_step_kernel<SG_OPT_ADADELTA>, grid, blk, 0, sg_st(st), p, g, s1, s2, ema, numel, lr, h,
                       eps, nesterov, gscale, ema_decay, lr_dev);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_optim_step(int kind, float* p, const float* g, float* s1, float* s2, float* ema, int64_t numel,
                             float lr, float h, float eps, int nesterov, float gscale, float ema_decay,
                             sg_stream_t st) {
  return optim_step_launch(kind, p, g, s1, s2, ema, numel, lr, nullptr, h, eps, nesterov, gscale, ema_decay, st);
}

extern "C" int sg_optim_step_dev(int kind, float* p, const float* g, float* s1, float* s2, float* ema, int64_t numel,
                                 const float* lr, float h, float eps, int nesterov, float gscale, float ema_decay,
                                 sg_stream_t st) {
  if (!lr) return SG_EINVAL;
  return optim_step_launch(kind, p, g, s1, s2, ema, numel, 0.f, lr, h, eps, nesterov, gscale, ema_decay, st);
}

extern "C" int sg_segment_sumsq(const float* flat, const int64_t* offsets, float* out, int32_t nseg,
                                sg_stream_t st) {
  if (!flat || !offsets || !out || nseg < 1) return SG_EINVAL;
  hipLaunchKernelGGL(segment_sumsq_kernel, dim3(nseg), dim3(256), 0, sg_st(st), flat, offsets, out);
  SG_LAUNCH_CHECK();
  return SG_OK;
}
