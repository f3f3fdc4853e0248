// Noise blend (K12) and the (MSE + 1 - SSIM)/2 criterion with its gradient (K13).
//
// noise blend  : d3f/train_denoiser/lit_module.py:128-153 == d3f/train_deep_fake/lit_module.py:208-233
//                r = (1/lam) * log(1 / (y*(1-c) + c)), c = exp(-lam);  out = sqrt(1-r)*x + sqrt(r)*noise
//                (same operation order and roundings as the torch expression: no FMA contraction).
// criterion    : d3f/loss_functions/structural_similarity_loss.py:14-26 with piqa.SSIM defaults
//                (11-tap sigma-1.5 separable Gaussian, valid filtering, k1=.01, k2=.03, value range 1).
// Tensors here are the NCHW fp32 boundary tensors (prediction / target images), 3 channels.
//
// SSIM forward is one LDS-tiled pass per (image, channel, 32x32 output tile): it filters
// x, y, x^2, y^2, xy (columns then rows), evaluates the SSIM map and at the same time the
// three derivative maps d ss / d(mu_x, E[x^2], E[xy]); the backward pass filters those three
// maps with the adjoint (zero-padded) Gaussian and adds the MSE gradient, so the loss and
// d loss / d prediction cost 2 stencil passes over the images instead of ~15 torch kernels.
#include "pointwise.h"

namespace d3f {

constexpr int SS_T = 32;            // tile edge (outputs)
constexpr int SS_K = 11;            // window
constexpr int SS_IN = SS_T + SS_K - 1;  // 42
constexpr int SS_LD = SS_IN + 1;        // 43: odd stride -> conflict-free column walks

struct GaussK {
  float g[SS_K];
};

static GaussK make_gauss() {
  GaussK k;
  // same arithmetic as the oracle (float32 throughout)
  float s = 0.f;
  for (int i = 0; i < SS_K; ++i) {
    const float d = (float)i - (float)(SS_K - 1) / 2.f;
    k.g[i] = expf(-(d * d) / (2.f * 1.5f * 1.5f));
    s += k.g[i];
  }
  for (int i = 0; i < SS_K; ++i) k.g[i] /= s;
  return k;
}

__device__ __forceinline__ float norm01(float v, float lo, float range) {
  const float t = (v - lo) / range;  // same op order as structural_similarity_loss.py:24
  return fminf(fmaxf(t, 0.f), 1.f);
}

__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float* __restrict__ pred,
                                                       const float* __restrict__ target, float lo,
                                                       float range, GaussK gk, int H, int W,
                                                       float* __restrict__ dA, float* __restrict__ dB,
                                                       float* __restrict__ dC,
                                                       float* __restrict__ partial) {
  chain_priority();
  __shared__ float tx[SS_IN * SS_LD], ty[SS_IN * SS_LD];
  __shared__ float vv[5][SS_T * SS_LD];
  __shared__ float wsum[4];
  const int Hv = H - (SS_K - 1), Wv = W - (SS_K - 1);
  const int plane = blockIdx.z;
  const int oy0 = blockIdx.y * SS_T, ox0 = blockIdx.x * SS_T;
  const float* __restrict__ px = pred + (long)plane * H * W;
  const float* __restrict__ py = target + (long)plane * H * W;
  const int tid = threadIdx.x;
  for (int e = tid; e < SS_IN * SS_IN; e += 256) {
    const int r = e / SS_IN, c = e - r * SS_IN;
    const int iy = oy0 + r, ix = ox0 + c;
    float a = 0.f, b = 0.f;
    if (iy < H && ix < W) {
      a = norm01(px[(long)iy * W + ix], lo, range);
      b = norm01(py[(long)iy * W + ix], lo, range);
    }
    tx[r * SS_LD + c] = a;
    ty[r * SS_LD + c] = b;
  }
  __syncthreads();
  // filter along H (piqa filters dim -2 first): vv[q][r][c], r < 32 output rows, c < 42 columns.  Register-blocked:
  // a thread owns column c and 8 consecutive output rows, reads its 18 input rows ONCE (the products a*a, b*b, a*b
  // are formed once per input instead of once per tap) -- 36 LDS reads instead of 176; every output is still summed
  // over the taps in the order i = 0 .. 10.
  constexpr int RB = 8;
  if (tid < SS_IN * (SS_T / RB)) {
    const int c = tid % SS_IN, r0 = (tid / SS_IN) * RB;
    float a[RB + SS_K - 1], b[RB + SS_K - 1], aa[RB + SS_K - 1], bb[RB + SS_K - 1], ab[RB + SS_K - 1];
#pragma unroll
    for (int i = 0; i < RB + SS_K - 1; ++i) {
      a[i] = tx[(r0 + i) * SS_LD + c];
      b[i] = ty[(r0 + i) * SS_LD + c];
      aa[i] = a[i] * a[i];
      bb[i] = b[i] * b[i];
      ab[i] = a[i] * b[i];
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
      for (int i = 0; i < SS_K; ++i) {
        const float g = gk.g[i];
        s0 += g * a[r + i];
        s1 += g * b[r + i];
        s2 += g * aa[r + i];
        s3 += g * bb[r + i];
        s4 += g * ab[r + i];
      }
      vv[0][(r0 + r) * SS_LD + c] = s0;
      vv[1][(r0 + r) * SS_LD + c] = s1;
      vv[2][(r0 + r) * SS_LD + c] = s2;
      vv[3][(r0 + r) * SS_LD + c] = s3;
      vv[4][(r0 + r) * SS_LD + c] = s4;
    }
  }
  __syncthreads();
  const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
  float local = 0.f;
  // row filter, register-blocked the same way: a thread owns row r and 4 consecutive output columns (14 reads per map
  // instead of 44)
  constexpr int CBk = 4;
  const int rr = tid / (SS_T / CBk), cq = (tid % (SS_T / CBk)) * CBk;
  float win[5][CBk + SS_K - 1];
#pragma unroll
  for (int q = 0; q < 5; ++q)
#pragma unroll
    for (int j = 0; j < CBk + SS_K - 1; ++j) win[q][j] = vv[q][rr * SS_LD + cq + j];
#pragma unroll
  for (int cc = 0; cc < CBk; ++cc) {
    const int r = rr, c = cq + cc;
    const int oy = oy0 + r, ox = ox0 + c;
    if (oy >= Hv || ox >= Wv) continue;
    float m[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < SS_K; ++j) s += gk.g[j] * win[q][cc + j];
      m[q] = s;
    }
    const float mux = m[0], muy = m[1];
    const float sxx = m[2] - mux * mux, syy = m[3] - muy * muy, sxy = m[4] - mux * muy;
    const float a1 = 2.f * mux * muy + c1, b1 = mux * mux + muy * muy + c1;
    const float a2 = 2.f * sxy + c2, b2 = sxx + syy + c2;
    const float l = a1 / b1, cs = a2 / b2;
    const float ss = l * cs;
    local += ss;
    const long o = ((long)plane * Hv + oy) * Wv + ox;
    const float dl_dmux = 2.f * (muy - l * mux) / b1;
    const float dcs_dmux = (2.f * mux * cs - 2.f * muy) / b2;
    dA[o] = cs * dl_dmux + l * dcs_dmux;   // d ss / d mu_x   (total)
    dB[o] = -l * cs / b2;                  // d ss / d E[x^2]
    dC[o] = 2.f * l / b2;                  // d ss / d E[xy]
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
  if ((tid & 63) == 0) wsum[tid >> 6] = local;
  __syncthreads();
  if (tid == 0)
    partial[((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] =
        (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(256) void ssim_mse_bwd_kernel(
    const float* __restrict__ pred, const float* __restrict__ target, float lo, float range, GaussK gk,
    int H, int W, const float* __restrict__ dA, const float* __restrict__ dB,
    const float* __restrict__ dC, float ssim_coef, float mse_coef, float* __restrict__ grad,
    float* __restrict__ partial_mse) {
  chain_priority();
  __shared__ float t3[3][SS_IN * SS_LD];
  __shared__ float vv[3][SS_T * SS_LD];
  __shared__ float wsum[4];
  const int Hv = H - (SS_K - 1), Wv = W - (SS_K - 1);
  const int plane = blockIdx.z;
  const int py0 = blockIdx.y * SS_T, px0 = blockIdx.x * SS_T;
  const int tid = threadIdx.x;
  // derivative maps at outputs o = p - k, k in [0,10]: tile origin (py0-10, px0-10)
  for (int e = tid; e < SS_IN * SS_IN; e += 256) {
    const int r = e / SS_IN, c = e - r * SS_IN;
    const int oy = py0 - (SS_K - 1) + r, ox = px0 - (SS_K - 1) + c;
    float a = 0.f, b = 0.f, d = 0.f;
    if (oy >= 0 && oy < Hv && ox >= 0 && ox < Wv) {
      const long o = ((long)plane * Hv + oy) * Wv + ox;
      a = dA[o];
      b = dB[o];
      d = dC[o];
    }
    t3[0][r * SS_LD + c] = a;
    t3[1][r * SS_LD + c] = b;
    t3[2][r * SS_LD + c] = d;
  }
  __syncthreads();
  // adjoint filter: G(p) = sum_k g[k] * D[p - k] = sum_i g[10 - i] * tile[r + i]; register-blocked like the forward
  // pass (column c, 8 consecutive output rows per thread; same summation order per output)
  constexpr int RB = 8;
  if (tid < SS_IN * (SS_T / RB)) {
    const int c = tid % SS_IN, r0 = (tid / SS_IN) * RB;
    float v0[RB + SS_K - 1], v1[RB + SS_K - 1], v2[RB + SS_K - 1];
#pragma unroll
    for (int i = 0; i < RB + SS_K - 1; ++i) {
      v0[i] = t3[0][(r0 + i) * SS_LD + c];
      v1[i] = t3[1][(r0 + i) * SS_LD + c];
      v2[i] = t3[2][(r0 + i) * SS_LD + c];
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < SS_K; ++i) {
        const float g = gk.g[SS_K - 1 - i];
        s0 += g * v0[r + i];
        s1 += g * v1[r + i];
        s2 += g * v2[r + i];
      }
      vv[0][(r0 + r) * SS_LD + c] = s0;
      vv[1][(r0 + r) * SS_LD + c] = s1;
      vv[2][(r0 + r) * SS_LD + c] = s2;
    }
  }
  __syncthreads();
  const float inv_range = 1.f / range;
  float local = 0.f;
  constexpr int CBk = 4;  // row r, 4 consecutive columns per thread
  const int rr = tid / (SS_T / CBk), cq = (tid % (SS_T / CBk)) * CBk;
  float win[3][CBk + SS_K - 1];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int j = 0; j < CBk + SS_K - 1; ++j) win[q][j] = vv[q][rr * SS_LD + cq + j];
#pragma unroll
  for (int cc = 0; cc < CBk; ++cc) {
    const int r = rr, c = cq + cc;
    const int iy = py0 + r, ix = px0 + c;
    if (iy >= H || ix >= W) continue;
    float ga = 0.f, gb = 0.f, gc = 0.f;
#pragma unroll
    for (int j = 0; j < SS_K; ++j) {
      const float g = gk.g[SS_K - 1 - j];
      ga += g * win[0][cc + j];
      gb += g * win[1][cc + j];
      gc += g * win[2][cc + j];
    }
    const long p = (long)plane * H * W + (long)iy * W + ix;
    const float pv = pred[p], tv = target[p];
    const float rawx = (pv - lo) / range;
    const float xh = fminf(fmaxf(rawx, 0.f), 1.f);
    const float yh = norm01(tv, lo, range);
    float gx = ga + 2.f * xh * gb + yh * gc;       // d(sum ss)/d xhat
    gx = (rawx >= 0.f && rawx <= 1.f) ? gx * inv_range : 0.f;  // through clip and affine
    const float diff = pv - tv;
    local += diff * diff;
    grad[p] = ssim_coef * gx + mse_coef * diff;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
  if ((tid & 63) == 0) wsum[tid >> 6] = local;
  __syncthreads();
  if (tid == 0)
    partial_mse[((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] =
        (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ part_ss, int n_ss,
                                                            const float* __restrict__ part_mse,
                                                            int n_mse, double n_ssim_elems,
                                                            double n_elems, float* __restrict__ out) {
  chain_priority();
  __shared__ double red[2][256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n_ss; i += 256) a += (double)part_ss[i];
  for (int i = threadIdx.x; i < n_mse; i += 256) b += (double)part_mse[i];
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float ssim = (float)(red[0][0] / n_ssim_elems);
    const float mse = (float)(red[1][0] / n_elems);
    out[0] = (mse + (1.0f - ssim)) / 2.0f;
    out[1] = mse;
    out[2] = ssim;
  }
}

static inline int tiles(int n) { return (n + SS_T - 1) / SS_T; }

size_t loss_workspace_floats(int B, int H, int W) {
  if (H < SS_K || W < SS_K) return 0;
  const long planes = (long)B * 3;
  const long maps = 3 * planes * (H - SS_K + 1) * (W - SS_K + 1);
  const long p1 = planes * tiles(H - SS_K + 1) * tiles(W - SS_K + 1);
  const long p2 = planes * tiles(H) * tiles(W);
  return (size_t)(maps + p1 + p2 + 16);
}

int mse_ssim_loss_launch(const float* pred, const float* target, float in_min, float in_max,
                         float* loss_out, float* grad_pred, float* workspace, int B, int H, int W,
                         hipStream_t stream) {
  D3F_CHECK(H >= SS_K && W >= SS_K, "loss: image %dx%d smaller than the 11x11 SSIM window", H, W);
  D3F_CHECK(in_max > in_min, "loss: input range");
  static const GaussK gk = make_gauss();
  const int Hv = H - SS_K + 1, Wv = W - SS_K + 1;
  const long planes = (long)B * 3;
  const long map = planes * Hv * Wv;
  float* dA = workspace;
  float* dB = dA + map;
  float* dC = dB + map;
  float* part_ss = dC + map;
  const int n_ss = (int)(planes * tiles(Hv) * tiles(Wv));
  float* part_mse = part_ss + n_ss;
  const int n_mse = (int)(planes * tiles(H) * tiles(W));
  const float range = in_max - in_min;
  hipLaunchKernelGGL(ssim_fwd_kernel, dim3(tiles(Wv), tiles(Hv), (unsigned)planes), dim3(256), 0, stream,
                     pred, target, in_min, range, gk, H, W, dA, dB, dC, part_ss);
  D3F_HIP(hipGetLastError());
  const double n_elems = (double)planes * H * W;
  const double n_ssim = (double)planes * Hv * Wv;
  // loss = (mse + 1 - mean(ss)) / 2
  const float ssim_coef = (float)(-0.5 / n_ssim);
  const float mse_coef = (float)(1.0 / n_elems);  // 0.5 * 2 * diff / N
  hipLaunchKernelGGL(ssim_mse_bwd_kernel, dim3(tiles(W), tiles(H), (unsigned)planes), dim3(256), 0,
                     stream, pred, target, in_min, range, gk, H, W, dA, dB, dC, ssim_coef, mse_coef,
                     grad_pred, part_mse);
  D3F_HIP(hipGetLastError());
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, stream, part_ss, n_ss, part_mse, n_mse,
                     n_ssim, n_elems, loss_out);
  D3F_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void noise_blend_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ noise,
                                                          const float* __restrict__ y_uniform,
                                                          float c, float one_minus_c, float inv_lam,
                                                          float* __restrict__ out,
                                                          float* __restrict__ r_out, long per_image) {
  const int b = blockIdx.y;
  // x = 1/lam * log(1 / (y*(1-c) + c))   -- every step rounded to f32 like the torch expression
  const float t = __fadd_rn(__fmul_rn(y_uniform[b], one_minus_c), c);
  // log / sqrt evaluated in double and rounded once: correctly rounded f32 results (the device's
  // f32 logf / sqrtf are 1-2 ulp off the host libm the reference's CPU path uses); 3 ops per image
  // inv_lam <= 0: fixed-ratio mode (balance_training_images), y_uniform holds r itself
  const float r = inv_lam > 0.f ? __fmul_rn(inv_lam, (float)log((double)__fdiv_rn(1.0f, t))) : y_uniform[b];
  // __fsqrt_rn: IEEE correctly rounded f32 square root
  const float sa = __fsqrt_rn(__fsub_rn(1.0f, r)), sb = __fsqrt_rn(r);
  if (r_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) r_out[b] = r;
  const long nvec = per_image / 4;
  const float4* __restrict__ xv = reinterpret_cast<const float4*>(x + (long)b * per_image);
  const float4* __restrict__ nv = reinterpret_cast<const float4*>(noise + (long)b * per_image);
  float4* __restrict__ ov = reinterpret_cast<float4*>(out + (long)b * per_image);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    const float4 a = xv[i], n = nv[i];
    float4 o;
    o.x = __fadd_rn(__fmul_rn(sa, a.x), __fmul_rn(sb, n.x));
    o.y = __fadd_rn(__fmul_rn(sa, a.y), __fmul_rn(sb, n.y));
    o.z = __fadd_rn(__fmul_rn(sa, a.z), __fmul_rn(sb, n.z));
    o.w = __fadd_rn(__fmul_rn(sa, a.w), __fmul_rn(sb, n.w));
    ov[i] = o;
  }
}

int noise_blend_launch(const float* x, const float* noise, const float* y_uniform, float lam,
                       float* out, float* r_out, int B, long per_image, hipStream_t stream) {
  D3F_CHECK(per_image % 4 == 0, "noise_blend: per-image element count %ld not a multiple of 4", per_image);
  D3F_CHECK(lam > 0.f, "noise_blend: lambda must be positive");
  if (B == 0 || per_image == 0) return 0;
  const double c = 1.0 / exp((double)lam);
  long bx = (per_image / 4 + 255) / 256;
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(noise_blend_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, stream, x, noise,
                     y_uniform, (float)c, (float)(1.0 - c), (float)(1.0 / (double)lam), out, r_out,
                     per_image);
  D3F_HIP(hipGetLastError());
  return 0;
}

// fixed-ratio blend (d3f/balance_training_images/lit_module.py:109-121): out = sqrt(1-r[b])*x + sqrt(r[b])*noise
int noise_blend_fixed_launch(const float* x, const float* noise, const float* r, float* out, int B, long per_image,
                             hipStream_t stream) {
  D3F_CHECK(per_image % 4 == 0, "noise_blend: per-image element count %ld not a multiple of 4", per_image);
  if (B == 0 || per_image == 0) return 0;
  long bx = (per_image / 4 + 255) / 256;
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(noise_blend_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, stream, x, noise, r, 0.f, 0.f,
                     -1.f, out, (float*)nullptr, per_image);
  D3F_HIP(hipGetLastError());
  return 0;
}

// per-image mean absolute error (compute_difficulty_loss, d3f/balance_training_images/lit_module.py:139-142):
// L1_PARTS partial sums per image, then one thread block per image adds them in a fixed order (f64)
constexpr int L1_PARTS = 64;
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                         double* __restrict__ partial, long per_image) {
  __shared__ double red[256];
  const int b = blockIdx.y;
  const float* pp = p + (long)b * per_image;
  const float* tt = t + (long)b * per_image;
  double s = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_image; i += (long)gridDim.x * 256)
    s += (double)fabsf(pp[i] - tt[i]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(long)b * L1_PARTS + blockIdx.x] = red[0];
}
__global__ void l1_finalize_kernel(const double* __restrict__ partial, float* __restrict__ out, int B, long per_image) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s = 0.0;
  for (int i = 0; i < L1_PARTS; ++i) s += partial[(long)b * L1_PARTS + i];
  out[b] = (float)(s / (double)per_image);
}
size_t l1_per_image_workspace_bytes(int B) { return (size_t)B * L1_PARTS * sizeof(double) + 64; }
int l1_per_image_launch(const float* pred, const float* target, float* out, void* workspace, int B, long per_image,
                        hipStream_t stream) {
  if (B == 0) return 0;
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(l1_partial_kernel, dim3(L1_PARTS, (unsigned)B), dim3(256), 0, stream, pred, target, partial,
                     per_image);
  hipLaunchKernelGGL(l1_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, partial, out, B, per_image);
  D3F_HIP(hipGetLastError());
  return 0;
}

}  // namespace d3f
