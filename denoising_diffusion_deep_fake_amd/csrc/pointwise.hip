// HBM-bound kernels of the U-Net hot path: BatchNorm (train / eval, forward / backward),
// ReLU + residual add, 3x3/s2 max-pool, 2x2 sum (backward of nearest up-sampling), layout
// conversion at the NCHW boundary and weight packing.  All activations NHWC, every access a
// 16-byte vector per lane, grid-stride over at most 2048 workgroups.
//
// These replace the ATen batch_norm / relu / add / max_pool2d / upsample_nearest2d / cat
// kernels (forward and autograd backward) the reference's U-Net dispatches
// (semantics: SURVEY.md Appendix A.1).
#include "pointwise.h"

namespace d3f {

template <typename T> struct V16;
template <> struct V16<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float (&o)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  }
};
template <> struct V16<bf16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const bf16_t* p, float (&o)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = __uint_as_float(w[i] << 16);
      o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&o)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] = (uint32_t)f32_to_bf16(o[2 * i]) | ((uint32_t)f32_to_bf16(o[2 * i + 1]) << 16);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

static inline int grid_for(long work_items, int per_block = 256, int cap = 2048) {
  long b = (work_items + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------------------------------
// BatchNorm forward
// ------------------------------------------------------------------------------------------
// one workgroup per channel: 256 lanes stride over the m-tile partials (up to 8192 of them for the
// 256x16 tiles), f64 accumulation, wave shuffle + LDS tree
__device__ __forceinline__ void block_sum2(double& s1, double& s2) {
  __shared__ double red[2][4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s1;
    red[1][threadIdx.x >> 6] = s2;
  }
  __syncthreads();
  s1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  s2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(
    const float* __restrict__ stats, int tiles, int C, int Cpad, double count,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
    float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ mean_o,
    float* __restrict__ invstd_o, float* __restrict__ scale_o, float* __restrict__ shift_o, NetSplit ns) {
  chain_priority();
  if (blockIdx.z != 0) {  // two networks in one launch (common.h, NetSplit): blockIdx.z = net
    net_shift(stats, ns.ws); net_shift(gamma, ns.par); net_shift(beta, ns.par);
    net_shift(running_mean, ns.bn); net_shift(running_var, ns.bn);
    net_shift(mean_o, ns.ws); net_shift(invstd_o, ns.ws); net_shift(scale_o, ns.ws); net_shift(shift_o, ns.ws);
  }
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int t = threadIdx.x; t < tiles; t += 256) {
    const float2 v = *reinterpret_cast<const float2*>(stats + ((long)t * Cpad + c) * 2);
    s1 += (double)v.x;
    s2 += (double)v.y;
  }
  block_sum2(s1, s2);
  if (threadIdx.x == 0) {
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float g = gamma[c], b = beta[c];
    mean_o[c] = (float)mean;
    invstd_o[c] = (float)invstd;
    scale_o[c] = (float)((double)g * invstd);
    shift_o[c] = (float)((double)b - mean * (double)g * invstd);
    if (running_mean != nullptr) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
      running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
    }
  }
}

int bn_finalize_launch(const float* stats, int tiles, int C, int Cpad, long count,
                       const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, float* mean, float* invstd,
                       float* scale, float* shift, hipStream_t stream, const NetSplit* ns) {
  const NetSplit nv = net_split_or_single(ns);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C, 1, nv.nets), dim3(256), 0, stream, stats, tiles, C,
                     Cpad, (double)count, gamma, beta, eps, momentum, running_mean, running_var, mean,
                     invstd, scale, shift, nv);
  D3F_HIP(hipGetLastError());
  return 0;
}

__global__ void bn_eval_coeff_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ rm, const float* __restrict__ rv,
                                     float eps, int C, float* __restrict__ scale,
                                     float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

// every BatchNorm of the network in one launch (eval forward): block = layer, table as kernel argument
__global__ __launch_bounds__(256) void bn_eval_coeff_all_kernel(const float* __restrict__ params,
                                                                const float* __restrict__ bnstats,
                                                                char* __restrict__ ws, float eps, BnEvalTable t) {
  const BnEvalEntry e = t.e[blockIdx.x];
  float* __restrict__ coef = reinterpret_cast<float*>(ws + (size_t)e.coef_off16 * 16);
  for (int c = threadIdx.x; c < e.C; c += 256) {
    const float invstd = 1.0f / sqrtf(bnstats[e.rv_off + c] + eps);
    const float sc = params[e.g_off + c] * invstd;
    coef[2 * e.C + c] = sc;
    coef[3 * e.C + c] = params[e.b_off + c] - bnstats[e.rm_off + c] * sc;
  }
}

int bn_eval_coeff_all_launch(const float* params, const float* bnstats, void* ws, float eps, const BnEvalTable& t,
                             hipStream_t stream) {
  if (t.n == 0) return 0;
  hipLaunchKernelGGL(bn_eval_coeff_all_kernel, dim3(t.n), dim3(256), 0, stream, params, bnstats, (char*)ws, eps, t);
  D3F_HIP(hipGetLastError());
  return 0;
}

int bn_eval_coeff_launch(const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, int C, float* scale, float* shift,
                         hipStream_t stream) {
  hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, gamma, beta,
                     running_mean, running_var, eps, C, scale, shift);
  D3F_HIP(hipGetLastError());
  return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(
    const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
    const T* __restrict__ res, const T* __restrict__ yr, const float* __restrict__ scale_r,
    const float* __restrict__ shift_r, int relu, T* __restrict__ out, long nvec, int C, long net_ws) {
  chain_priority();
  if (blockIdx.z != 0) {  // two networks in one launch (common.h, NetSplit): every operand lives in the workspace
    net_shift(y, net_ws); net_shift(scale, net_ws); net_shift(shift, net_ws); net_shift(res, net_ws); net_shift(yr, net_ws);
    net_shift(scale_r, net_ws); net_shift(shift_r, net_ws); net_shift(out, net_ws);
  }
  constexpr int N = V16<T>::N;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    const int c0 = (int)((i * N) % C);
    float v[N], sc[N], sf[N];
    V16<T>::load(y + i * N, v);
#pragma unroll
    for (int k = 0; k < N; k += 4) {
      const float4 a = *reinterpret_cast<const float4*>(scale + c0 + k);
      const float4 b = *reinterpret_cast<const float4*>(shift + c0 + k);
      sc[k] = a.x; sc[k + 1] = a.y; sc[k + 2] = a.z; sc[k + 3] = a.w;
      sf[k] = b.x; sf[k + 1] = b.y; sf[k + 2] = b.z; sf[k + 3] = b.w;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = v[k] * sc[k] + sf[k];
    if (res != nullptr) {
      float r[N];
      V16<T>::load(res + i * N, r);
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] += r[k];
    } else if (yr != nullptr) {
      float r[N];
      V16<T>::load(yr + i * N, r);
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] += r[k] * scale_r[c0 + k] + shift_r[c0 + k];
    }
    if (relu) {
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    V16<T>::store(out + i * N, v);
  }
}

int bn_apply_launch(int dtype, const void* y, const float* scale, const float* shift,
                    const void* res, const void* yr, const float* scale_r, const float* shift_r,
                    int relu, void* out, long rows, int C, hipStream_t stream, const NetSplit* ns) {
  const int ve = dtype == D3F_F32 ? 4 : 8;
  D3F_CHECK(C % ve == 0 && (256 * ve) % C == 0, "bn_apply: C=%d must divide %d", C, 256 * ve);
  const long nvec = rows * C / ve;
  if (nvec == 0) return 0;
  const NetSplit nv = net_split_or_single(ns);
  const dim3 grid(grid_for(nvec), 1, nv.nets);
  if (dtype == D3F_F32)
    hipLaunchKernelGGL(bn_apply_kernel<float>, grid, dim3(256), 0, stream, (const float*)y, scale,
                       shift, (const float*)res, (const float*)yr, scale_r, shift_r, relu, (float*)out,
                       nvec, C, nv.ws);
  else
    hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)y,
                       scale, shift, (const bf16_t*)res, (const bf16_t*)yr, scale_r, shift_r, relu,
                       (bf16_t*)out, nvec, C, nv.ws);
  D3F_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// BatchNorm backward.  dz = dA * [a > 0];  dbeta = sum dz;  dgamma = sum dz * xhat;
// dy = gamma*invstd * (dz - dbeta/N - xhat * dgamma/N)
// ------------------------------------------------------------------------------------------
int bn_bwd_reduce_blocks(long rows, int C, int dtype) {
  // a block covers at least one unrolled trip (4 passes of 256 threads) and 16 rows; small tensors then
  // still spread over enough CUs to hide the load latency
  const int ve = dtype == D3F_F32 ? 4 : 8;
  const long per_trip = 4L * (256 / std::max(1, std::min(256, C / ve)));
  const long min_rows = std::max(16L, per_trip);
  long b = (rows + min_rows - 1) / min_rows;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

// ReLU mask: from the saved activation `a` (a > 0), or -- for layers without a residual, mask_scale !=
// null -- recomputed from y with the forward's own arithmetic (y*scale + shift > 0), which saves reading `a`.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const T* __restrict__ dA, const T* __restrict__ a, const T* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ partial,
    long rows, int C, const float* __restrict__ mask_scale, const float* __restrict__ mask_shift, long net_ws) {
  chain_priority();
  if (blockIdx.z != 0) {  // two networks in one launch (common.h, NetSplit): every operand lives in the workspace
    net_shift(dA, net_ws); net_shift(a, net_ws); net_shift(y, net_ws); net_shift(mean, net_ws); net_shift(invstd, net_ws);
    net_shift(partial, net_ws); net_shift(mask_scale, net_ws); net_shift(mask_shift, net_ws);
  }
  constexpr int N = V16<T>::N;
  __shared__ float red[256 * N * 2];
  const int VC = C / N;        // vectors per row (power of two, <= 256)
  const int RP = 256 / VC;     // rows per pass
  const int cv = threadIdx.x % VC, r0 = threadIdx.x / VC;
  const long rows_per_block = (rows + gridDim.x - 1) / gridDim.x;
  const long rbeg = (long)blockIdx.x * rows_per_block;
  long rend = rbeg + rows_per_block;
  if (rend > rows) rend = rows;
  float mu[N], is[N], s1[N], s2[N], msc[N], msf[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    mu[k] = mean[cv * N + k];
    is[k] = invstd[cv * N + k];
    msc[k] = mask_scale ? mask_scale[cv * N + k] : 0.f;
    msf[k] = mask_scale ? mask_shift[cv * N + k] : 0.f;
    s1[k] = 0.f;
    s2[k] = 0.f;
  }
  // U rows per trip: all loads of a trip are issued before any arithmetic (memory-level parallelism)
  constexpr int U = 4;
  const bool from_y = mask_scale != nullptr, from_a = !from_y && a != nullptr;
  long r = rbeg + r0;
  for (; r + (U - 1) * RP < rend; r += U * RP) {
    float g[U][N], yy[U][N], aa[U][N];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long off = (r + u * RP) * C + cv * N;
      V16<T>::load(dA + off, g[u]);
      V16<T>::load(y + off, yy[u]);
      if (from_a) V16<T>::load(a + off, aa[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const float keep = from_y ? yy[u][k] * msc[k] + msf[k] : (from_a ? aa[u][k] : 1.f);
        const float gz = keep > 0.f ? g[u][k] : 0.f;
        s1[k] += gz;
        s2[k] += gz * ((yy[u][k] - mu[k]) * is[k]);
      }
    }
  }
  for (; r < rend; r += RP) {
    const long off = r * C + cv * N;
    float g[N], yy[N], aa[N];
    V16<T>::load(dA + off, g);
    V16<T>::load(y + off, yy);
    if (from_a) V16<T>::load(a + off, aa);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float keep = from_y ? yy[k] * msc[k] + msf[k] : (from_a ? aa[k] : 1.f);
      const float gz = keep > 0.f ? g[k] : 0.f;
      s1[k] += gz;
      s2[k] += gz * ((yy[k] - mu[k]) * is[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    red[(threadIdx.x * N + k) * 2 + 0] = s1[k];
    red[(threadIdx.x * N + k) * 2 + 1] = s2[k];
  }
  __syncthreads();
  // thread t < 2*C sums column t over the RP row groups
  for (int t = threadIdx.x; t < 2 * C; t += 256) {
    const int c = t >> 1, which = t & 1;
    const int v = c / N, k = c % N;
    float s = 0.f;
    for (int rr = 0; rr < RP; ++rr) s += red[((rr * VC + v) * N + k) * 2 + which];
    partial[((long)blockIdx.x * C + c) * 2 + which] = s;
  }
}

int bn_bwd_reduce_launch(int dtype, const void* dA, const void* a, const void* y,
                         const float* mean, const float* invstd, float* partial, int* nblocks,
                         long rows, int C, hipStream_t stream, const float* mask_scale,
                         const float* mask_shift, const NetSplit* ns) {
  const int ve = dtype == D3F_F32 ? 4 : 8;
  const int vc = C / ve;
  D3F_CHECK(C % ve == 0 && vc >= 1 && vc <= 256 && (256 % vc) == 0,
            "bn_bwd_reduce: unsupported channel count %d", C);
  const int blocks = bn_bwd_reduce_blocks(rows, C, dtype);
  *nblocks = blocks;
  const NetSplit nv = net_split_or_single(ns);
  const dim3 grid(blocks, 1, nv.nets);
  if (dtype == D3F_F32)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, grid, dim3(256), 0, stream,
                       (const float*)dA, (const float*)a, (const float*)y, mean, invstd, partial, rows, C,
                       mask_scale, mask_shift, nv.ws);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, grid, dim3(256), 0, stream,
                       (const bf16_t*)dA, (const bf16_t*)a, (const bf16_t*)y, mean, invstd, partial, rows, C,
                       mask_scale, mask_shift, nv.ws);
  D3F_HIP(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(
    const float* __restrict__ partial, int nblocks, int C, double count,
    const float* __restrict__ gamma, const float* __restrict__ invstd, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int accumulate, float* __restrict__ coef, NetSplit ns) {
  chain_priority();
  if (blockIdx.z != 0) {  // two networks in one launch (common.h, NetSplit)
    net_shift(partial, ns.ws); net_shift(gamma, ns.par); net_shift(invstd, ns.ws);
    net_shift(dgamma, ns.grad); net_shift(dbeta, ns.grad); net_shift(coef, ns.ws);
  }
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int t = threadIdx.x; t < nblocks; t += 256) {
    const float2 v = *reinterpret_cast<const float2*>(partial + ((long)t * C + c) * 2);
    s1 += (double)v.x;
    s2 += (double)v.y;
  }
  block_sum2(s1, s2);
  if (threadIdx.x == 0) {
    const float db = (float)s1, dg = (float)s2;
    if (dgamma != nullptr) {
      dgamma[c] = accumulate ? dgamma[c] + dg : dg;
      dbeta[c] = accumulate ? dbeta[c] + db : db;
    }
    coef[c] = gamma[c] * invstd[c];
    coef[C + c] = (float)(s1 / count);
    coef[2 * C + c] = (float)(s2 / count);
  }
}

int bn_bwd_finalize_launch(const float* partial, int nblocks, int C, long count,
                           const float* gamma, const float* invstd, float* dgamma, float* dbeta,
                           int accumulate, float* coef, hipStream_t stream, const NetSplit* ns) {
  const NetSplit nv = net_split_or_single(ns);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C, 1, nv.nets), dim3(256), 0, stream, partial, nblocks,
                     C, (double)count, gamma, invstd, dgamma, dbeta, accumulate, coef, nv);
  D3F_HIP(hipGetLastError());
  return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const T* __restrict__ dA, const T* __restrict__ a, const T* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ coef,
    T* __restrict__ dy, T* __restrict__ dres, int dres_acc, long nvec, int C,
    const float* __restrict__ mask_scale, const float* __restrict__ mask_shift, long net_ws) {
  chain_priority();
  if (blockIdx.z != 0) {  // two networks in one launch (common.h, NetSplit): every operand lives in the workspace
    net_shift(dA, net_ws); net_shift(a, net_ws); net_shift(y, net_ws); net_shift(mean, net_ws); net_shift(invstd, net_ws);
    net_shift(coef, net_ws); net_shift(dy, net_ws); net_shift(dres, net_ws); net_shift(mask_scale, net_ws);
    net_shift(mask_shift, net_ws);
  }
  constexpr int N = V16<T>::N;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    const int c0 = (int)((i * N) % C);
    float g[N], yy[N], o[N];
    V16<T>::load(dA + i * N, g);
    V16<T>::load(y + i * N, yy);
    if (mask_scale != nullptr) {
#pragma unroll
      for (int k = 0; k < N; ++k) g[k] = (yy[k] * mask_scale[c0 + k] + mask_shift[c0 + k]) > 0.f ? g[k] : 0.f;
    } else if (a != nullptr) {
      float aa[N];
      V16<T>::load(a + i * N, aa);
#pragma unroll
      for (int k = 0; k < N; ++k) g[k] = aa[k] > 0.f ? g[k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const int c = c0 + k;
      const float xhat = (yy[k] - mean[c]) * invstd[c];
      o[k] = coef[c] * (g[k] - coef[C + c] - xhat * coef[2 * C + c]);
    }
    V16<T>::store(dy + i * N, o);
    if (dres != nullptr) {
      if (dres_acc) {
        float d[N];
        V16<T>::load(dres + i * N, d);
#pragma unroll
        for (int k = 0; k < N; ++k) g[k] += d[k];
      }
      V16<T>::store(dres + i * N, g);
    }
  }
}

int bn_bwd_apply_launch(int dtype, const void* dA, const void* a, const void* y, const float* mean,
                        const float* invstd, const float* coef, void* dy, void* dres, int dres_acc,
                        long rows, int C, hipStream_t stream, const float* mask_scale,
                        const float* mask_shift, const NetSplit* ns) {
  const int ve = dtype == D3F_F32 ? 4 : 8;
  D3F_CHECK(C % ve == 0 && (256 * ve) % C == 0, "bn_bwd_apply: C=%d must divide %d", C, 256 * ve);
  const long nvec = rows * C / ve;
  if (nvec == 0) return 0;
  const NetSplit nv = net_split_or_single(ns);
  const dim3 grid(grid_for(nvec), 1, nv.nets);
  if (dtype == D3F_F32)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, grid, dim3(256), 0, stream, (const float*)dA,
                       (const float*)a, (const float*)y, mean, invstd, coef, (float*)dy, (float*)dres,
                       dres_acc, nvec, C, mask_scale, mask_shift, nv.ws);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)dA,
                       (const bf16_t*)a, (const bf16_t*)y, mean, invstd, coef, (bf16_t*)dy, (bf16_t*)dres,
                       dres_acc, nvec, C, mask_scale, mask_shift, nv.ws);
  D3F_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// MaxPool2d(3, stride 2, padding 1): first maximum in (kh, kw) scan order wins, like ATen.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                          uint8_t* __restrict__ idx, int B, int H,
                                                          int W, int C, long net_ws) {
  chain_priority();
  if (blockIdx.z != 0) {  // two networks in one launch (common.h, NetSplit)
    net_shift(in, net_ws); net_shift(out, net_ws); net_shift(idx, net_ws);
  }
  constexpr int N = V16<T>::N;
  const int Ho = H / 2, Wo = W / 2, VC = C / N;
  const long total = (long)B * Ho * Wo * VC;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cv = (int)(i % VC);
    long t = i / VC;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float best[N];
    int bi[N];
    bool first = true;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = 2 * oy - 1 + kh;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ix = 2 * ox - 1 + kw;
        if ((unsigned)ix >= (unsigned)W) continue;
        float v[N];
        V16<T>::load(in + (((long)b * H + iy) * W + ix) * C + cv * N, v);
        if (first) {
#pragma unroll
          for (int k = 0; k < N; ++k) { best[k] = v[k]; bi[k] = kh * 3 + kw; }
          first = false;
        } else {
#pragma unroll
          for (int k = 0; k < N; ++k)
            if (v[k] > best[k] || v[k] != v[k]) { best[k] = v[k]; bi[k] = kh * 3 + kw; }
        }
      }
    }
    const long o = (((long)b * Ho + oy) * Wo + ox) * C + cv * N;
    V16<T>::store(out + o, best);
#pragma unroll
    for (int k = 0; k < N; ++k) idx[o + k] = (uint8_t)bi[k];
  }
}

// grid = (blocks over one input row's W * C/N vectors, B * H rows): the row (b, iy) -- and with it which filter rows
// reach it -- is block-uniform, the column decode is one multiply-high; the argmax bytes of a vector are one load.
// (r03: the first form decoded a flat index with four 64-bit divisions and walked all 9 taps under branches: ~600
// vector instructions per 16 bytes written -- the pass was ALU-bound at 2.6 TB/s.)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dout,
                                                          const uint8_t* __restrict__ idx,
                                                          T* __restrict__ din, int accumulate, int B,
                                                          int H, int W, int C, unsigned vc_mul, unsigned vc_shr,
                                                          long net_ws) {
  chain_priority();
  if (blockIdx.z != 0) {  // two networks in one launch (common.h, NetSplit)
    net_shift(dout, net_ws); net_shift(idx, net_ws); net_shift(din, net_ws);
  }
  constexpr int N = V16<T>::N;
  const int Ho = H / 2, Wo = W / 2, VC = C / N;
  const int j = (int)blockIdx.x * 256 + (int)threadIdx.x;  // (ix, cv) in the row
  if (j >= W * VC) return;
  const int ix = fast_div(j, vc_mul, vc_shr), cv = j - ix * VC;
  const int row = (int)blockIdx.y;  // b * H + iy
  const int iy = row % H, b = row / H;  // (scalar: block-uniform)
  // windows containing (iy, ix): window oy covers rows 2*oy-1 .. 2*oy+1 -> an odd row lies in two windows (as their
  // filter row 0 / 2), an even row in one (filter row 1); the same for columns
  const int noy = (iy & 1) ? 2 : 1, nox = (ix & 1) ? 2 : 1;
  const int oy0 = (iy + 1) >> 1, kh0 = (iy & 1) ? 0 : 1;  // second candidate: oy0 - 1 with filter row 2
  const int ox0 = (ix + 1) >> 1, kw0 = (ix & 1) ? 0 : 1;
  float g[N];
#pragma unroll
  for (int k = 0; k < N; ++k) g[k] = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int oy = oy0 - a, kh = a ? 2 : kh0;
    if (a >= noy || oy >= Ho) continue;  // (block-uniform)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ox = ox0 - c, kw = c ? 2 : kw0;
      if (c < nox && ox < Wo) {
        const long o = (((long)b * Ho + oy) * Wo + ox) * C + cv * N;
        float d[N];
        V16<T>::load(dout + o, d);
        const unsigned tap = (unsigned)(kh * 3 + kw);
        uint32_t w[N / 4];
        if constexpr (N == 4) {
          w[0] = *reinterpret_cast<const uint32_t*>(idx + o);
        } else {
          const uint2 t = *reinterpret_cast<const uint2*>(idx + o);
          w[0] = t.x;
          w[1] = t.y;
        }
#pragma unroll
        for (int k = 0; k < N; ++k)
          if (((w[k >> 2] >> (8 * (k & 3))) & 0xffu) == tap) g[k] += d[k];
      }
    }
  }
  T* dst = din + ((long)row * W * VC + j) * N;
  if (accumulate) {
    float old[N];
    V16<T>::load(dst, old);
#pragma unroll
    for (int k = 0; k < N; ++k) g[k] += old[k];
  }
  V16<T>::store(dst, g);
}

int maxpool3x3s2_fwd_launch(int dtype, const void* in, void* out, uint8_t* idx, int B, int H, int W,
                            int C, hipStream_t stream, const NetSplit* ns) {
  const int ve = dtype == D3F_F32 ? 4 : 8;
  D3F_CHECK(C % ve == 0 && H % 2 == 0 && W % 2 == 0, "maxpool: shape (%d,%d,%d)", H, W, C);
  const long total = (long)B * (H / 2) * (W / 2) * (C / ve);
  if (total == 0) return 0;
  const NetSplit nv = net_split_or_single(ns);
  const dim3 grid(grid_for(total), 1, nv.nets);
  if (dtype == D3F_F32)
    hipLaunchKernelGGL(maxpool_fwd_kernel<float>, grid, dim3(256), 0, stream,
                       (const float*)in, (float*)out, idx, B, H, W, C, nv.ws);
  else
    hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, grid, dim3(256), 0, stream,
                       (const bf16_t*)in, (bf16_t*)out, idx, B, H, W, C, nv.ws);
  D3F_HIP(hipGetLastError());
  return 0;
}

int maxpool3x3s2_bwd_launch(int dtype, const void* dout, const uint8_t* idx, void* din, int accumulate,
                            int B, int H, int W, int C, hipStream_t stream, const NetSplit* ns) {
  const int ve = dtype == D3F_F32 ? 4 : 8;
  D3F_CHECK(C % ve == 0 && H % 2 == 0 && W % 2 == 0, "maxpool: shape (%d,%d,%d)", H, W, C);
  const NetSplit nv = net_split_or_single(ns);
  const long total = (long)B * H * W * (C / ve);
  if (total == 0) return 0;
  const int vc = C / ve;
  unsigned vc_mul, vc_shr;
  fast_div_setup((unsigned)vc, &vc_mul, &vc_shr);
  D3F_CHECK((long)B * H <= 65535, "maxpool backward: %d rows exceed the grid", B * H);
  const dim3 grid((unsigned)cdiv((long)W * vc, 256), (unsigned)(B * H), (unsigned)nv.nets);
  if (dtype == D3F_F32)
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, grid, dim3(256), 0, stream,
                       (const float*)dout, idx, (float*)din, accumulate, B, H, W, C, vc_mul, vc_shr, nv.ws);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, grid, dim3(256), 0, stream,
                       (const bf16_t*)dout, idx, (bf16_t*)din, accumulate, B, H, W, C, vc_mul, vc_shr, nv.ws);
  D3F_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// backward of F.interpolate(scale_factor=2, mode="nearest"): sum each 2x2 block
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sum2x2_kernel(const T* __restrict__ dfull, T* __restrict__ dlow,
                                                     int B, int Hl, int Wl, int C, long net_ws) {
  chain_priority();
  if (blockIdx.z != 0) {  // two networks in one launch (common.h, NetSplit)
    net_shift(dfull, net_ws); net_shift(dlow, net_ws);
  }
  constexpr int N = V16<T>::N;
  const int VC = C / N;
  const long total = (long)B * Hl * Wl * VC;
  const int Wf = 2 * Wl;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cv = (int)(i % VC);
    long t = i / VC;
    const int x = (int)(t % Wl);
    t /= Wl;  // t = b*Hl + y
    const long base = ((t * 2) * Wf + 2 * x) * (long)C + cv * N;
    float s[N], v[N];
    V16<T>::load(dfull + base, s);
    V16<T>::load(dfull + base + C, v);
#pragma unroll
    for (int k = 0; k < N; ++k) s[k] += v[k];
    V16<T>::load(dfull + base + (long)Wf * C, v);
#pragma unroll
    for (int k = 0; k < N; ++k) s[k] += v[k];
    V16<T>::load(dfull + base + (long)Wf * C + C, v);
#pragma unroll
    for (int k = 0; k < N; ++k) s[k] += v[k];
    V16<T>::store(dlow + i * N, s);
  }
}

int sum2x2_launch(int dtype, const void* dfull, void* dlow, int B, int Hl, int Wl, int C,
                  hipStream_t stream, const NetSplit* ns) {
  const int ve = dtype == D3F_F32 ? 4 : 8;
  D3F_CHECK(C % ve == 0, "sum2x2: C=%d", C);
  const long total = (long)B * Hl * Wl * (C / ve);
  if (total == 0) return 0;
  const NetSplit nv = net_split_or_single(ns);
  const dim3 grid(grid_for(total), 1, nv.nets);
  if (dtype == D3F_F32)
    hipLaunchKernelGGL(sum2x2_kernel<float>, grid, dim3(256), 0, stream,
                       (const float*)dfull, (float*)dlow, B, Hl, Wl, C, nv.ws);
  else
    hipLaunchKernelGGL(sum2x2_kernel<bf16_t>, grid, dim3(256), 0, stream,
                       (const bf16_t*)dfull, (bf16_t*)dlow, B, Hl, Wl, C, nv.ws);
  D3F_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// boundary layout conversion
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in,
                                                           T* __restrict__ out, int B, int C, long HW,
                                                           int Cpad, long net_in, long net_ws) {
  if (blockIdx.z != 0) {  // two networks in one launch (common.h, NetSplit): boundary tensor in, workspace tensor out
    net_shift(in, net_in); net_shift(out, net_ws);
  }
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / HW, pix = i - b * HW;
    for (int c = 0; c < Cpad; ++c) {
      const float v = c < C ? in[(b * C + c) * HW + pix] : 0.f;
      out[i * Cpad + c] = from_f32<T>(v);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ in,
                                                           float* __restrict__ out, int B, int C,
                                                           long HW, int Cpad) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / HW, pix = i - b * HW;
    for (int c = 0; c < C; ++c) out[(b * C + c) * HW + pix] = to_f32<T>(in[i * Cpad + c]);
  }
}

// K16 (inference boundary, d3f/train_deep_fake/lit_module.py:272-300): uint8 BGR frames [B][H][W][3] ->
// normalised NHWC activations in RGB order, (float(u8) - mean*255) / (std*255) in the reference's order of
// fp32 operations; and the way back: y*std*255 + mean*255, .int() truncation, clamp(0, 255), RGB -> BGR.
template <typename T>
__global__ __launch_bounds__(256) void u8bgr_to_nhwc_kernel(const uint8_t* __restrict__ in, T* __restrict__ out,
                                                            long npix, int Cpad, float m0, float m1, float m2,
                                                            float s0, float s1, float s2) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
    const uint8_t* px = in + i * 3;
    const float r = ((float)px[2] - m0) / s0, g = ((float)px[1] - m1) / s1, b = ((float)px[0] - m2) / s2;
    T* o = out + i * Cpad;
    o[0] = from_f32<T>(r);
    o[1] = from_f32<T>(g);
    o[2] = from_f32<T>(b);
    for (int c = 3; c < Cpad; ++c) o[c] = from_f32<T>(0.f);
  }
}

__global__ __launch_bounds__(256) void nchw_to_u8bgr_kernel(const float* __restrict__ in, uint8_t* __restrict__ out,
                                                            int B, long HW, float m0, float m1, float m2, float s0,
                                                            float s1, float s2) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / HW, pix = i - b * HW;
    const float* src = in + b * 3 * HW + pix;
    auto q = [](float y, float s, float m) {
      const float v = y * s + m;      // two roundings, like the reference's in-place mul then add
      int t = (int)v;                 // .int(): truncation toward zero
      t = t < 0 ? 0 : (t > 255 ? 255 : t);
      return (uint8_t)t;
    };
    uint8_t* o = out + i * 3;
    o[2] = q(src[0], s0, m0);
    o[1] = q(src[HW], s1, m1);
    o[0] = q(src[2 * HW], s2, m2);
  }
}

// Input pipeline: uint8 RGB frames [B][H][W][3] as PIL decodes them -> normalised NCHW fp32 training batch,
// ((float)u8 / 255 - mean) / std in the order of fp32 operations of albumentations.Normalize(max_pixel_value=255) +
// ToTensorV2 (d3f/train_deep_fake/lit_module.py:100-110) -- bit-identical to the host transform, but the batch crosses
// worker IPC and PCIe as 1 byte per value instead of 4.
__global__ __launch_bounds__(256) void u8rgb_to_nchw_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                            int B, long HW, float m0, float m1, float m2, float s0,
                                                            float s1, float s2) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / HW, pix = i - b * HW;
    const uint8_t* px = in + i * 3;
    float* o = out + b * 3 * HW + pix;
    o[0] = ((float)px[0] / 255.0f - m0) / s0;
    o[HW] = ((float)px[1] / 255.0f - m1) / s1;
    o[2 * HW] = ((float)px[2] / 255.0f - m2) / s2;
  }
}

// K17: batched affine warp of NCHW fp32 images -- affine_grid + grid_sample(bilinear, zeros, align_corners=False)
// in one pass (the GPU-side augmentation of train_denoiser, d3f/train_denoiser/lit_module.py:55-65,113).
// theta [B][2][3] maps normalised output coordinates to normalised input coordinates.
__global__ __launch_bounds__(256) void affine_warp_kernel(const float* __restrict__ in, const float* __restrict__ theta,
                                                          float* __restrict__ out, int B, int C, int H, int W) {
  const long HW = (long)H * W, total = (long)B * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int b = (int)(i / HW);
    const int pix = (int)(i - (long)b * HW);
    const int y = pix / W, x = pix - y * W;
    const float* t = theta + b * 6;
    const float xn = (2.0f * x + 1.0f) / W - 1.0f, yn = (2.0f * y + 1.0f) / H - 1.0f;
    const float xs = t[0] * xn + t[1] * yn + t[2], ys = t[3] * xn + t[4] * yn + t[5];
    const float fx = ((xs + 1.0f) * W - 1.0f) * 0.5f, fy = ((ys + 1.0f) * H - 1.0f) * 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float wx1 = fx - x0f, wy1 = fy - y0f, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
    const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
    const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
    const float* src = in + (long)b * C * HW;
    float* dst = out + (long)b * C * HW + pix;
    for (int c = 0; c < C; ++c) {
      const float* pl = src + (long)c * HW;
      const float v00 = (vx0 && vy0) ? pl[(long)y0 * W + x0] : 0.f;
      const float v01 = (vx1 && vy0) ? pl[(long)y0 * W + x0 + 1] : 0.f;
      const float v10 = (vx0 && vy1) ? pl[(long)(y0 + 1) * W + x0] : 0.f;
      const float v11 = (vx1 && vy1) ? pl[(long)(y0 + 1) * W + x0 + 1] : 0.f;
      dst[(long)c * HW] = v00 * (wx0 * wy0) + v01 * (wx1 * wy0) + v10 * (wx0 * wy1) + v11 * (wx1 * wy1);
    }
  }
}

int affine_warp_launch(const float* in, const float* theta, float* out, int B, int C, int H, int W,
                       hipStream_t stream) {
  const long total = (long)B * H * W;
  if (total == 0) return 0;
  hipLaunchKernelGGL(affine_warp_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, stream, in, theta, out, B, C,
                     H, W);
  D3F_HIP(hipGetLastError());
  return 0;
}

int u8bgr_to_nhwc_launch(int dtype, const uint8_t* in, void* out, long npix, int Cpad, const float mean255[3],
                         const float std255[3], hipStream_t stream) {
  if (npix == 0) return 0;
  if (dtype == D3F_F32)
    hipLaunchKernelGGL(u8bgr_to_nhwc_kernel<float>, dim3(grid_for(npix)), dim3(256), 0, stream, in, (float*)out, npix,
                       Cpad, mean255[0], mean255[1], mean255[2], std255[0], std255[1], std255[2]);
  else
    hipLaunchKernelGGL(u8bgr_to_nhwc_kernel<bf16_t>, dim3(grid_for(npix)), dim3(256), 0, stream, in, (bf16_t*)out,
                       npix, Cpad, mean255[0], mean255[1], mean255[2], std255[0], std255[1], std255[2]);
  D3F_HIP(hipGetLastError());
  return 0;
}

int u8rgb_to_nchw_launch(const uint8_t* in, float* out, int B, long HW, const float mean[3], const float stdv[3],
                         hipStream_t stream) {
  if ((long)B * HW == 0) return 0;
  hipLaunchKernelGGL(u8rgb_to_nchw_kernel, dim3(grid_for((long)B * HW)), dim3(256), 0, stream, in, out, B, HW, mean[0],
                     mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
  D3F_HIP(hipGetLastError());
  return 0;
}

int nchw_to_u8bgr_launch(const float* in, uint8_t* out, int B, long HW, const float mean255[3],
                         const float std255[3], hipStream_t stream) {
  if ((long)B * HW == 0) return 0;
  hipLaunchKernelGGL(nchw_to_u8bgr_kernel, dim3(grid_for((long)B * HW)), dim3(256), 0, stream, in, out, B, HW,
                     mean255[0], mean255[1], mean255[2], std255[0], std255[1], std255[2]);
  D3F_HIP(hipGetLastError());
  return 0;
}

int nchw_to_nhwc_launch(int dtype, const float* in, void* out, int B, int C, int H, int W, int Cpad,
                        hipStream_t stream, const NetSplit* ns) {
  const long total = (long)B * H * W;
  if (total == 0) return 0;
  const NetSplit nv = net_split_or_single(ns);
  const dim3 grid(grid_for(total), 1, nv.nets);
  if (dtype == D3F_F32)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, stream, in,
                       (float*)out, B, C, (long)H * W, Cpad, nv.in, nv.ws);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, dim3(256), 0, stream, in,
                       (bf16_t*)out, B, C, (long)H * W, Cpad, nv.in, nv.ws);
  D3F_HIP(hipGetLastError());
  return 0;
}

int nhwc_to_nchw_launch(int dtype, const void* in, float* out, int B, int C, int H, int W, int Cpad,
                        hipStream_t stream) {
  const long total = (long)B * H * W;
  if (total == 0) return 0;
  if (dtype == D3F_F32)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream,
                       (const float*)in, out, B, C, (long)H * W, Cpad);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, stream,
                       (const bf16_t*)in, out, B, C, (long)H * W, Cpad);
  D3F_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// weight packing (fp32 master in PyTorch layout -> kernel layouts)
// ------------------------------------------------------------------------------------------
// forward:  wf[n][(kh*KW + kw)*Cin + c]                     = w[n][c][kh][kw]
// dgrad:    wd[c][slot*CoutD + n]                            = w[n][c][kh][kw], flipped tap f = (KH-1-kh)*KW + (KW-1-kw)
//           stride 1: slot = f.  3x3 stride 2: the flipped taps are stored output-parity class by class,
//           (odd,odd) (odd,even) (even,odd) (even,even) = f in {0,2,6,8} {1,7} {3,5} {4}, so that each class of the
//           parity-decomposed data gradient (conv_igemm.hip) is one contiguous k-range of the same rows.
__device__ __forceinline__ int dgrad_tap_slot_to_flipped(int slot, int taps, int stride) {
  if (stride == 2 && taps == 9) return (int)((0x453718620ull >> (4 * slot)) & 15);  // 0,2,6,8,1,7,3,5,4
  return slot;
}
// store one packed weight: plain T, or (X3) its exact 3-way bf16 split into three planes `plane` elements apart
template <typename T, bool X3>
__device__ __forceinline__ void pack_store(T* __restrict__ base, long idx, long plane, float v) {
  if constexpr (X3) {
    const uint32_t xb = __float_as_uint(v), hb = xb & 0xffff0000u;
    const float r1 = v - __uint_as_float(hb);
    const uint32_t mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);
    base[idx] = (T)(hb >> 16);
    base[plane + idx] = (T)(mb >> 16);
    base[2 * plane + idx] = (T)(__float_as_uint(r2) >> 16);  // exact: r2 has at most 8 significant bits
  } else {
    base[idx] = from_f32<T>(v);
  }
}

template <typename T, bool X3>
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, int Cout,
                                                           int CinReal, int Cin, int KH, int KW,
                                                           T* __restrict__ wf, int CoutPad, int Kpad,
                                                           T* __restrict__ wd, int CinRows, int CoutD,
                                                           int KpadD, int stride) {
  const int taps = KH * KW;
  const long nf = wf ? (long)CoutPad * Kpad : 0;
  const long nd = wd ? (long)CinRows * KpadD : 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nf + nd; i += (long)gridDim.x * 256) {
    if (i < nf) {
      const int n = (int)(i / Kpad), k = (int)(i % Kpad);
      const int tap = k / Cin, c = k % Cin;
      float v = 0.f;
      if (n < Cout && tap < taps && c < CinReal) v = w[((long)n * CinReal + c) * taps + tap];
      pack_store<T, X3>(wf, i, nf, v);
    } else {
      const long j = i - nf;
      const int c = (int)(j / KpadD), k = (int)(j % KpadD);
      const int slot = k / CoutD, n = k % CoutD;
      float v = 0.f;
      if (c < CinReal && slot < taps && n < Cout)
        v = w[((long)n * CinReal + c) * taps + (taps - 1 - dgrad_tap_slot_to_flipped(slot, taps, stride))];
      pack_store<T, X3>(wd, j, nd, v);
    }
  }
}

int pack_weights_launch(int dtype, const float* w, int Cout, int CinReal, int Cin, int KH, int KW,
                        void* wf, int CoutPad, int Kpad, void* wd, int CinRows, int KpadD, int stride,
                        hipStream_t stream) {
  const int ve = dtype == D3F_BF16 ? 8 : 4;
  const int CoutD = (int)round_up(Cout, ve);
  const long total = (wf ? (long)CoutPad * Kpad : 0) + (wd ? (long)CinRows * KpadD : 0);
  if (total == 0) return 0;
  D3F_CHECK(!wf || Kpad >= KH * KW * Cin, "pack: Kpad");
  D3F_CHECK(!wd || KpadD >= KH * KW * CoutD, "pack: KpadD");
  if (dtype == D3F_F32)
    hipLaunchKernelGGL((pack_weights_kernel<float, false>), dim3(grid_for(total)), dim3(256), 0, stream, w, Cout,
                       CinReal, Cin, KH, KW, (float*)wf, CoutPad, Kpad, (float*)wd, CinRows, CoutD, KpadD, stride);
  else if (dtype == D3F_F32X3)
    hipLaunchKernelGGL((pack_weights_kernel<bf16_t, true>), dim3(grid_for(total)), dim3(256), 0, stream, w, Cout,
                       CinReal, Cin, KH, KW, (bf16_t*)wf, CoutPad, Kpad, (bf16_t*)wd, CinRows, CoutD, KpadD, stride);
  else
    hipLaunchKernelGGL((pack_weights_kernel<bf16_t, false>), dim3(grid_for(total)), dim3(256), 0, stream, w, Cout,
                       CinReal, Cin, KH, KW, (bf16_t*)wf, CoutPad, Kpad, (bf16_t*)wd, CinRows, CoutD, KpadD, stride);
  D3F_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// "Up-sample folded into the weights" layouts for the decoder's conv(cat(upsample2x(x), skip)), 3x3 pad 1:
// nearest x2 up-sampling repeats every low-resolution pixel 2x2 times, so inside an output-parity class (py, px)
// the 3x3 taps on the up-sampled operand touch only a 2x2 neighbourhood of x, each low-resolution pixel through
// 1, 2 or 4 taps whose weights can be summed ONCE per step instead of multiplied separately for every pixel:
// 4 MACs per (pixel, channel pair) instead of 9 on the up-sampled channels, forward and data gradient.
//   forward, per class z = 2*py + px:  wfc[z][n][(a*2+b)*C0 + c]      = sum_{kh in S(py,a), kw in S(px,b)} w[n][c][kh][kw]
//                                      wfc[z][n][4*C0 + (kh*3+kw)*C1 + c1] = w[n][C0 + c1][kh][kw]          (skip taps)
//     S(0,0) = {0}, S(0,1) = {1,2}, S(1,0) = {0,1}, S(1,1) = {2}; low-resolution row of tap a is j + a + py - 1
//   data gradient w.r.t. x (a 4x4 stride-2 pad-1 convolution over dY, written at low resolution):
//                                      wd4[c][(u*4+v)*CoutD + n]      = sum_{kh in T(u), kw in T(v)} w[n][c][kh][kw]
//     T(0) = {2}, T(1) = {1,2}, T(2) = {0,1}, T(3) = {0}
//   data gradient w.r.t. the skip tensor (an ordinary 3x3 data gradient with C1 outputs, flipped taps):
//                                      wds[c1][f*CoutD + n]            = w[n][C0 + c1][8 - f]
// The sums are formed in fp32 from the fp32 masters (one rounding per sum: the same order of error as the fp32
// accumulation order inside any conv kernel).  One thread per output element; ~9 M elements per step in total.
__device__ __forceinline__ unsigned up_fwd_set(int parity, int a) {  // bitmask over kh / kw
  return parity == 0 ? (a == 0 ? 1u : 6u) : (a == 0 ? 3u : 4u);
}
__device__ __forceinline__ unsigned up_bwd_set(int u) { return u == 0 ? 4u : u == 1 ? 6u : u == 2 ? 3u : 1u; }

// (the element decode divides by seven run-time constants: multiply-high pairs from the host, fast_div in common.h)
struct UpDiv {
  unsigned fplane[2], kc[2], c0[2], c1[2], k9[2], cd[2], k4[2];
};
template <typename T, bool X3>
__global__ __launch_bounds__(256) void pack_up_kernel(const float* __restrict__ w, int Cout, int C0, int C1,
                                                      T* __restrict__ wfc, int CoutPad, T* __restrict__ wd4,
                                                      int C0Rows, T* __restrict__ wds, int C1Rows, int CoutD,
                                                      int K9, UpDiv dv) {  // K9: wds row length, 9*CoutD padded to whole k-tiles
  const int Cin = C0 + C1;
  const int Kc = 4 * C0 + 9 * C1, K4 = 16 * CoutD;
  const int nf = 4 * CoutPad * Kc, nd = C0Rows * K4, ns = C1Rows * K9;  // (< 2^31 in all: pack_up_launch)
  const int fplane = CoutPad * Kc;  // x3: planes of one class
  for (int i = (int)blockIdx.x * 256 + (int)threadIdx.x; i < nf + nd + ns; i += (int)gridDim.x * 256) {
    if (i < nf) {
      const int z = fast_div(i, dv.fplane[0], dv.fplane[1]);
      const int r = i - z * fplane;
      const int n = fast_div(r, dv.kc[0], dv.kc[1]), k = r - n * Kc;
      float v = 0.f;
      if (n < Cout) {
        if (k < 4 * C0) {
          const int ab = fast_div(k, dv.c0[0], dv.c0[1]), c = k - ab * C0;
          const unsigned sh = up_fwd_set(z >> 1, ab >> 1), sw = up_fwd_set(z & 1, ab & 1);
          const float* __restrict__ wp = w + ((long)n * Cin + c) * 9;
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
              if (((sh >> kh) & 1u) && ((sw >> kw) & 1u)) v += wp[kh * 3 + kw];
        } else {
          const int kk = k - 4 * C0, tap = fast_div(kk, dv.c1[0], dv.c1[1]), c1 = kk - tap * C1;
          v = w[((long)n * Cin + C0 + c1) * 9 + tap];
        }
      }
      // class blocks are [z][plane][CoutPad][Kc] in x3 mode, [z][CoutPad][Kc] otherwise
      pack_store<T, X3>(wfc + (X3 ? 3L * z * fplane : (long)z * fplane), r, fplane, v);
    } else if (i >= nf + nd) {
      const int j = i - nf - nd;
      const int c1 = fast_div(j, dv.k9[0], dv.k9[1]), k = j - c1 * K9;
      const int f = fast_div(k, dv.cd[0], dv.cd[1]), n = k - f * CoutD;
      float v = 0.f;
      if (c1 < C1 && n < Cout && f < 9) v = w[((long)n * Cin + C0 + c1) * 9 + (8 - f)];
      pack_store<T, X3>(wds, j, ns, v);
    } else {
      const int j = i - nf;
      const int c = fast_div(j, dv.k4[0], dv.k4[1]), k = j - c * K4;
      const int uv = fast_div(k, dv.cd[0], dv.cd[1]), n = k - uv * CoutD;
      float v = 0.f;
      if (c < C0 && n < Cout) {
        const unsigned sh = up_bwd_set(uv >> 2), sw = up_bwd_set(uv & 3);
        const float* __restrict__ wp = w + ((long)n * Cin + c) * 9;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw)
            if (((sh >> kh) & 1u) && ((sw >> kw) & 1u)) v += wp[kh * 3 + kw];
      }
      pack_store<T, X3>(wd4, j, nd, v);
    }
  }
}

int pack_up_launch(int dtype, const float* w, int Cout, int C0, int C1, void* wfc, int CoutPad, void* wd4,
                   int C0Rows, void* wds, int C1Rows, hipStream_t stream) {
  const int ve = dtype == D3F_BF16 ? 8 : 4;
  const int CoutD = (int)round_up(Cout, ve);
  if (wd4 == nullptr) C0Rows = 0;
  if (wds == nullptr) C1Rows = 0;
  const int K9 = (int)round_up(9L * CoutD, dtype == D3F_BF16 ? 64 : 32);  // = the data gradient's KpadD
  const long total = 4L * CoutPad * (4 * C0 + 9 * C1) + (long)C0Rows * 16 * CoutD + (long)C1Rows * K9;
  if (total == 0) return 0;
  D3F_CHECK(total < (1L << 31) - 2048L * 256, "pack_up: %ld elements exceed the 32-bit element index", total);
  UpDiv dv;
  auto setup = [](long d, unsigned (&o)[2]) { fast_div_setup((unsigned)(d > 0 ? d : 1), &o[0], &o[1]); };
  setup((long)CoutPad * (4 * C0 + 9 * C1), dv.fplane);
  setup(4 * C0 + 9 * C1, dv.kc);
  setup(C0, dv.c0);
  setup(C1, dv.c1);
  setup(K9, dv.k9);
  setup(CoutD, dv.cd);
  setup(16 * CoutD, dv.k4);
  if (dtype == D3F_F32)
    hipLaunchKernelGGL((pack_up_kernel<float, false>), dim3(grid_for(total)), dim3(256), 0, stream, w, Cout, C0, C1,
                       (float*)wfc, CoutPad, (float*)wd4, C0Rows, (float*)wds, C1Rows, CoutD, K9, dv);
  else if (dtype == D3F_F32X3)
    hipLaunchKernelGGL((pack_up_kernel<bf16_t, true>), dim3(grid_for(total)), dim3(256), 0, stream, w, Cout, C0, C1,
                       (bf16_t*)wfc, CoutPad, (bf16_t*)wd4, C0Rows, (bf16_t*)wds, C1Rows, CoutD, K9, dv);
  else
    hipLaunchKernelGGL((pack_up_kernel<bf16_t, false>), dim3(grid_for(total)), dim3(256), 0, stream, w, Cout, C0, C1,
                       (bf16_t*)wfc, CoutPad, (bf16_t*)wd4, C0Rows, (bf16_t*)wds, C1Rows, CoutD, K9, dv);
  D3F_HIP(hipGetLastError());
  return 0;
}

// All layers of a network in one launch: the table travels as a kernel argument (no device-side state) and a
// block looks its layer up by its first block index.  A block transposes one [32 filters][CT channels][taps]
// tile through LDS: the torch layout [n][c][tap] is read in contiguous runs of CT*taps floats and both packed
// layouts ([n][tap][c] and [c][flipped tap][n]) are written in contiguous runs; padding rows/columns of the
// packed matrices are (re)written as zeros by the edge tiles, so the workspace needs no initialisation.
template <typename T, bool X3>
__global__ __launch_bounds__(256) void pack_all_kernel(const float* __restrict__ params, char* __restrict__ ws,
                                                       PackTable t) {
  __shared__ float tile[PACK_NT * (PACK_LDS_ROW + 1)];
  int l = 0;
  while (l + 1 < t.n && blockIdx.x >= t.e[l + 1].block0) ++l;
  const PackEntry e = t.e[l];
  const float* __restrict__ w = params + e.w_off;
  T* __restrict__ wf = reinterpret_cast<T*>(ws + (size_t)e.wf_off16 * 16);
  T* __restrict__ wd = reinterpret_cast<T*>(ws + (size_t)e.wd_off16 * 16);
  const int taps = e.taps, Cin = e.Cin, CinReal = e.CinReal, Cout = e.Cout, Kpad = e.Kpad, KpadD = e.KpadD,
            CoutD = e.CoutD, CT = e.CT, CoutPad = e.CoutPad, CinRows = e.CinRows;
  const int rel = (int)(blockIdx.x - e.block0);
  const int nt = rel / e.ctiles, ct = rel % e.ctiles;  // (block-uniform)
  const int n0 = nt * PACK_NT, c0 = ct * CT;
  const int run = CT * taps, stride = run + 1;  // LDS row: [c_l][tap]
  // per-element index arithmetic: CT and the filter tile are powers of two, the only real division is by the tap count
  // (multiply-high, PackEntry) -- with five integer divisions per element and phase the pass was ALU-bound (r03: 193 us
  // alone for 265 MB)
  const unsigned tmul = e.taps_mul, tshr = e.taps_shr, lgct = e.ct_log2;
  // ---- load: contiguous runs of the torch layout, zero outside the real filter ----
  for (int i = threadIdx.x; i < PACK_NT * run; i += 256) {
    const int nl = fast_div(i >> lgct, tmul, tshr), r = i - nl * run;
    const int n = n0 + nl, c = c0 + fast_div(r, tmul, tshr);
    float v = 0.f;
    if (n < Cout && c < CinReal) v = w[(unsigned)((n * CinReal + c0) * taps + r)];
    tile[nl * stride + r] = v;
  }
  __syncthreads();
  // ---- forward layout wf[n][tap*Cin + c] ----
  if (true)
  for (int i = threadIdx.x; i < PACK_NT * run; i += 256) {
    const int cl = i & (CT - 1), q = i >> lgct;
    const int nl = fast_div(q, tmul, tshr), tap = q - nl * taps;
    const int n = n0 + nl, c = c0 + cl;
    if (n < CoutPad && c < Cin)
      pack_store<T, X3>(wf, (long)(unsigned)(n * Kpad + tap * Cin + c), (long)CoutPad * Kpad, tile[nl * stride + cl * taps + tap]);
  }
  if (ct == 0) {  // zero tail of each row: k in [taps*Cin, Kpad)
    const int k0 = taps * Cin, tail = Kpad - k0;
    for (int i = threadIdx.x; i < PACK_NT * tail; i += 256) {
      const int n = n0 + i / tail;
      if (n < CoutPad) pack_store<T, X3>(wf, (long)n * Kpad + k0 + i % tail, (long)CoutPad * Kpad, 0.f);
    }
  }
  // ---- data-gradient layout wd[c][tapf*CoutD + n], taps flipped ----
  if (e.has_d) {
    for (int i = threadIdx.x; i < PACK_NT * run; i += 256) {
      const int nl = i & (PACK_NT - 1), q = i / PACK_NT;
      const int cl = fast_div(q, tmul, tshr), slot = q - cl * taps;
      const int n = n0 + nl, c = c0 + cl;
      if (c < CinRows && n < CoutD)
        pack_store<T, X3>(wd, (long)(unsigned)(c * KpadD + slot * CoutD + n), (long)CinRows * KpadD,
                          tile[nl * stride + cl * taps + (taps - 1 - dgrad_tap_slot_to_flipped(slot, taps, e.conv_stride))]);
    }
    if (nt == 0) {
      const int k0 = taps * CoutD, tail = KpadD - k0;
      for (int i = threadIdx.x; i < CT * tail; i += 256) {
        const int c = c0 + i / tail;
        if (c < CinRows) pack_store<T, X3>(wd, (long)c * KpadD + k0 + i % tail, (long)CinRows * KpadD, 0.f);
      }
    }
  }
}

int pack_all_launch(int dtype, const float* params, void* ws, const PackTable& t, int blocks,
                    hipStream_t stream) {
  if (t.n == 0 || blocks == 0) return 0;
  if (dtype == D3F_F32)
    hipLaunchKernelGGL((pack_all_kernel<float, false>), dim3(blocks), dim3(256), 0, stream, params, (char*)ws, t);
  else if (dtype == D3F_F32X3)
    hipLaunchKernelGGL((pack_all_kernel<bf16_t, true>), dim3(blocks), dim3(256), 0, stream, params, (char*)ws, t);
  else
    hipLaunchKernelGGL((pack_all_kernel<bf16_t, false>), dim3(blocks), dim3(256), 0, stream, params, (char*)ws, t);
  D3F_HIP(hipGetLastError());
  return 0;
}

}  // namespace d3f
