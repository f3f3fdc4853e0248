// Fused multi-tensor optimiser kernels over the flat parameter buffer.
//   Adam (K14): torch.optim.Adam(params, lr, betas) as configured at
//               d3f/train_denoiser/lit_module.py:95 and d3f/train_deep_fake/lit_module.py:116-120
//               (eps 1e-8, no weight decay, no amsgrad); same update order as torch's
//               single-tensor implementation (lerp, mul+addcmul, sqrt/bias-correction, addcdiv).
//   EMA  (K15): ema.lerp_(online, 1 - decay) -- ema_pytorch semantics used at
//               d3f/train_deep_fake/lit_module.py:62-70,185.
// One launch streams the whole 24.4 M-element flat buffer with 16-byte accesses.
#include "pointwise.h"

namespace d3f {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n,
                                                   float one_minus_b1, float b2, float one_minus_b2,
                                                   float step_size, float bc2_sqrt, float eps,
                                                   float grad_scale) {
  const long nvec = n / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* P = &pp.x; const float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = G[k] * grad_scale;
      M[k] = M[k] + one_minus_b1 * (gk - M[k]);
      V[k] = V[k] * b2 + one_minus_b2 * gk * gk;
      const float denom = sqrtf(V[k]) / bc2_sqrt + eps;
      P[k] = P[k] - step_size * (M[k] / denom);
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  // tail (n % 4 elements)
  const long t = nvec * 4 + (long)blockIdx.x * 256 + threadIdx.x;
  if (t < n) {
    const float gk = g[t] * grad_scale;
    const float mk = m[t] + one_minus_b1 * (gk - m[t]);
    const float vk = v[t] * b2 + one_minus_b2 * gk * gk;
    m[t] = mk;
    v[t] = vk;
    p[t] = p[t] - step_size * (mk / (sqrtf(vk) / bc2_sqrt + eps));
  }
}

// The same update with the seven coefficients read from DEVICE memory (the captured training step: a replayed graph
// cannot carry per-step kernel arguments).  coef = {1-b1, b2, 1-b2, lr/bc1, sqrt(bc2), eps, grad_scale}, written by the
// host with adam_coefficients() -- the values adam_step_launch passes as arguments, so both forms update identically.
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, long n,
                                                       const float* __restrict__ coef) {
  const float one_minus_b1 = coef[0], b2 = coef[1], one_minus_b2 = coef[2], step_size = coef[3], bc2_sqrt = coef[4],
              eps = coef[5], grad_scale = coef[6];
  const long nvec = n / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* P = &pp.x; const float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = G[k] * grad_scale;
      M[k] = M[k] + one_minus_b1 * (gk - M[k]);
      V[k] = V[k] * b2 + one_minus_b2 * gk * gk;
      const float denom = sqrtf(V[k]) / bc2_sqrt + eps;
      P[k] = P[k] - step_size * (M[k] / denom);
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  const long t = nvec * 4 + (long)blockIdx.x * 256 + threadIdx.x;
  if (t < n) {
    const float gk = g[t] * grad_scale;
    const float mk = m[t] + one_minus_b1 * (gk - m[t]);
    const float vk = v[t] * b2 + one_minus_b2 * gk * gk;
    m[t] = mk;
    v[t] = vk;
    p[t] = p[t] - step_size * (mk / (sqrtf(vk) / bc2_sqrt + eps));
  }
}

void adam_coefficients(float lr, float beta1, float beta2, float eps, int step, float grad_scale, float coef[8]) {
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  coef[0] = (float)(1.0 - (double)beta1);
  coef[1] = beta2;
  coef[2] = (float)(1.0 - (double)beta2);
  coef[3] = (float)((double)lr / bc1);
  coef[4] = (float)sqrt(bc2);
  coef[5] = eps;
  coef[6] = grad_scale;
  coef[7] = 0.f;
}

int adam_step_dev_launch(float* p, const float* g, float* m, float* v, long n, const float* coef_dev,
                         hipStream_t stream) {
  if (n == 0) return 0;
  D3F_CHECK((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
            "adam: params / grads / exp_avg / exp_avg_sq must be 16-byte aligned (slice offsets: multiples of 4 floats)");
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, g, m, v, n, coef_dev);
  D3F_HIP(hipGetLastError());
  return 0;
}

int adam_step_launch(float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                     float beta2, float eps, int step, float grad_scale, hipStream_t stream) {
  D3F_CHECK(step >= 1, "adam: step counts from 1");
  if (n == 0) return 0;
  // the kernel streams 16-byte vectors: a slice of the flat buffers (per-bucket updates) must start on a vector boundary
  D3F_CHECK((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
            "adam: params / grads / exp_avg / exp_avg_sq must be 16-byte aligned (slice offsets: multiples of 4 floats)");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, g, m, v, n,
                     (float)(1.0 - (double)beta1), beta2, (float)(1.0 - (double)beta2), step_size,
                     bc2_sqrt, eps, grad_scale);
  D3F_HIP(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256) void ema_lerp_kernel(float* __restrict__ ema,
                                                       const float* __restrict__ online, long n,
                                                       float w) {
  // torch.lerp: w < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w)
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float a = ema[i], b = online[i];
    const float d = b - a;
    ema[i] = (w < 0.5f) ? (a + w * d) : (b - d * (1.f - w));
  }
}

int ema_lerp_launch(float* ema, const float* online, long n, float weight, hipStream_t stream) {
  if (n == 0) return 0;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(ema_lerp_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, ema, online, n, weight);
  D3F_HIP(hipGetLastError());
  return 0;
}

}  // namespace d3f
