// BatchNorm (train mode) with the finalize step folded into the streaming pass that needs its result.
//
// pointwise.hip runs BatchNorm forward as   partial statistics (conv epilogue) -> bn_finalize (one tiny launch) ->
// bn_apply (streaming pass), and backward as   partial sums (data-gradient epilogue) -> bn_bwd_finalize ->
// bn_bwd_apply.  On the training step's critical path every one of those 92 finalize launches costs its dispatch
// latency plus 5-14 us of a kernel that keeps 64-512 workgroups busy for a few hundred loads each.  Here the streaming
// kernel's workgroups REDUNDANTLY reduce the partial rows of their own 32-channel slab first (f64, fixed order, 8-256
// KB of L2-resident partials per workgroup), derive the slab's coefficients in LDS and go straight on to the
// streaming pass: no second launch, no atomics, no fences, and every workgroup computes bit-identical coefficients.
// Row block 0 of each slab also writes the coefficients (the backward pass and the data-gradient epilogues read them)
// and updates the running statistics / dgamma, dbeta.  Replaces ATen's batch_norm / batch_norm_backward under
// segmentation_models_pytorch.Unet (d3f/train_denoiser/lit_module.py:46-52); fp32 or bf16 tensors (statistics,
// coefficients and arithmetic are fp32 / f64 in both; bf16 rows are 8-byte vectors per thread), channel counts that are
// a multiple of 32, at most BNF_MAX_ROWS partial rows.
#include "common.h"
#include "pointwise.h"

#include <algorithm>
#include <cstdlib>

namespace d3f {

constexpr int BNF_SC = 32;         // channels per slab = one 128-byte line per tensor row
constexpr int BNF_MAX_ROWS = 1024;  // partial rows a workgroup is asked to reduce (x 256 B): 512 / 1024 / 2048 within 0.2 % of each other; 2048 would take in the stem, whose 512 KB prologue per workgroup makes the pass 3x longer
#ifndef BNF_BWD_U
#define BNF_BWD_U 2  // rows in flight per thread in the backward streaming pass (4: +0.4 % step time -- 142 VGPRs leave one workgroup per CU next to the weight-gradient stream)
#endif  // partial rows a workgroup is asked to reduce (x 256 B)

bool bn_fused_finalize_ok(int dtype, int stat_rows, int C) {
  static const bool off = getenv("D3F_NO_BN_FUSED_FINALIZE") != nullptr;  // debugging knob: separate launches
  return !off && (dtype == D3F_F32 || dtype == D3F_BF16) && (C % BNF_SC) == 0 && stat_rows >= 1 && stat_rows <= BNF_MAX_ROWS;
}

// sums the partial rows [rows][ld][2] of channels [c0, c0 + 32) in f64: thread (rl = tid / 16, q = tid % 16) owns the
// float4 q of the slab (channels c0 + 2q, c0 + 2q + 1; sum, second sum each) of rows rl, rl + 16, ...; the 16 row
// lanes are then added in lane order.  tot[2 * ch + which] for ch < 32.
__device__ __forceinline__ void slab_reduce(const float* __restrict__ partial, int rows, int ld, int c0,
                                            double (&red)[16][64], double (&tot)[64]) {
  const int tid = threadIdx.x, q = tid & 15, rl = tid >> 4;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const float* base = partial + ((long)c0 * 2 + q * 4);
  int r = rl;
  for (; r + 112 < rows; r += 128) {  // eight rows in flight (layer1-type layers bring 512 partial rows)
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(base + (long)(r + 16 * j) * ld * 2);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a0 += (double)v[j].x; a1 += (double)v[j].y; a2 += (double)v[j].z; a3 += (double)v[j].w;
    }
  }
  for (; r + 48 < rows; r += 64) {  // four rows in flight
    const float4 v0 = *reinterpret_cast<const float4*>(base + (long)r * ld * 2);
    const float4 v1 = *reinterpret_cast<const float4*>(base + (long)(r + 16) * ld * 2);
    const float4 v2 = *reinterpret_cast<const float4*>(base + (long)(r + 32) * ld * 2);
    const float4 v3 = *reinterpret_cast<const float4*>(base + (long)(r + 48) * ld * 2);
    a0 += (double)v0.x; a1 += (double)v0.y; a2 += (double)v0.z; a3 += (double)v0.w;
    a0 += (double)v1.x; a1 += (double)v1.y; a2 += (double)v1.z; a3 += (double)v1.w;
    a0 += (double)v2.x; a1 += (double)v2.y; a2 += (double)v2.z; a3 += (double)v2.w;
    a0 += (double)v3.x; a1 += (double)v3.y; a2 += (double)v3.z; a3 += (double)v3.w;
  }
  for (; r < rows; r += 16) {
    const float4 v = *reinterpret_cast<const float4*>(base + (long)r * ld * 2);
    a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
  }
  red[rl][q * 4 + 0] = a0;
  red[rl][q * 4 + 1] = a1;
  red[rl][q * 4 + 2] = a2;
  red[rl][q * 4 + 3] = a3;
  __syncthreads();
  if (tid < 64) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][tid];
    tot[tid] = s;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------
// forward: statistics -> (mean, invstd, scale, shift, running stats) -> out = [relu](y * scale + shift [+ residual])
// grid = (row blocks, C / 32); residual forms as bn_apply_kernel (pointwise.hip)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_finalize_apply_kernel(
    const float* __restrict__ stats, int stat_rows, int C, int Cpad, double count,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
    float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ mean_o,
    float* __restrict__ invstd_o, float* __restrict__ scale_o, float* __restrict__ shift_o,
    const T* __restrict__ y, const T* __restrict__ res, const T* __restrict__ yr,
    const float* __restrict__ scale_r, const float* __restrict__ shift_r, int relu, T* __restrict__ out,
    long rows, long rows_per_block) {
  chain_priority();
  __shared__ double red[16][64];
  __shared__ double tot[64];
  __shared__ float cf[2][BNF_SC];
  const int tid = threadIdx.x, c0 = blockIdx.y * BNF_SC;
  slab_reduce(stats, stat_rows, Cpad, c0, red, tot);
  if (tid < BNF_SC) {  // same arithmetic as bn_finalize_kernel
    const int c = c0 + tid;
    const double mean = tot[2 * tid] / count;
    double var = tot[2 * tid + 1] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float g = gamma[c], b = beta[c];
    const float sc = (float)((double)g * invstd), sf = (float)((double)b - mean * (double)g * invstd);
    cf[0][tid] = sc;
    cf[1][tid] = sf;
    if (blockIdx.x == 0) {
      mean_o[c] = (float)mean;
      invstd_o[c] = (float)invstd;
      scale_o[c] = sc;
      shift_o[c] = sf;
      if (running_mean != nullptr) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
      }
    }
  }
  __syncthreads();
  // streaming pass: thread (rr = tid / 8, v = tid % 8) owns channels c0 + 4v .. + 3 of rows rr, rr + 32, ...
  const int v = tid & 7, rr = tid >> 3, cc = c0 + v * 4;
  float sc[4], sf[4], scr[4] = {0.f, 0.f, 0.f, 0.f}, sfr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sc[k] = cf[0][v * 4 + k];
    sf[k] = cf[1][v * 4 + k];
    if (yr != nullptr) {
      scr[k] = scale_r[cc + k];
      sfr[k] = shift_r[cc + k];
    }
  }
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  const T* __restrict__ second = res != nullptr ? res : yr;
  constexpr int U = 4;
  for (long r = r0 + rr; r < r1; r += 32 * U) {
    float4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long row = r + 32 * u;
      if (row < r1) {
        a[u] = ld4<T>(y + row * C + cc);
        if (second != nullptr) b[u] = ld4<T>(second + row * C + cc);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long row = r + 32 * u;
      if (row >= r1) continue;
      float x[4] = {a[u].x, a[u].y, a[u].z, a[u].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = x[k] * sc[k] + sf[k];
      if (res != nullptr) {
        x[0] += b[u].x; x[1] += b[u].y; x[2] += b[u].z; x[3] += b[u].w;
      } else if (yr != nullptr) {
        const float t[4] = {b[u].x, b[u].y, b[u].z, b[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] += t[k] * scr[k] + sfr[k];
      }
      if (relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = fmaxf(x[k], 0.f);
      }
      st4<T>(out + row * C + cc, make_float4(x[0], x[1], x[2], x[3]));
    }
  }
}

// rows per workgroup, whole passes of 32 rows.  Workgroups in all: ~256 for fp32 tensors, ~512 for bf16 (r03 sweep of
// 64 ... 1024 with the chain's kernels at wave priority 3: fp32 256x256 8.88 / 8.38 / 8.30 / 8.24 / 8.28 / 8.30 / 8.35 ms
// per step at 64 / 128 / 192 / 256 / 320 / 512 / 768 -- every workgroup repeats the slab reduce, fewer of them repeat it
// less; bf16 4.83 / 4.58 / 4.55 at 128 / 256 / 512: half the bytes per row, the streaming part wants the parallelism)
static long rows_per_block_for(long rows, int slabs, int dtype) {
  const long wgs = dtype == D3F_F32 ? 256 : 512;
  long rb = std::max(1L, wgs / slabs);
  long rpb = (rows + rb - 1) / rb;
  rpb = (rpb + 31) / 32 * 32;
  return std::max(32L, rpb);
}

int bn_finalize_apply_launch(int dtype, const float* stats, int stat_rows, int C, int Cpad, long count, const float* gamma,
                             const float* beta, float eps, float momentum, float* running_mean,
                             float* running_var, float* mean, float* invstd, float* scale, float* shift,
                             const void* y, const void* res, const void* yr, const float* scale_r,
                             const float* shift_r, int relu, void* out, long rows, hipStream_t stream) {
  D3F_CHECK(bn_fused_finalize_ok(dtype, stat_rows, C), "bn_finalize_apply: C=%d, %d partial rows", C, stat_rows);
  if (rows == 0) return 0;
  const int slabs = C / BNF_SC;
  const long rpb = rows_per_block_for(rows, slabs, dtype);
  const dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)slabs);
  if (dtype == D3F_F32)
    hipLaunchKernelGGL(bn_finalize_apply_kernel<float>, grid, dim3(256), 0, stream, stats, stat_rows, C, Cpad,
                       (double)count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift,
                       (const float*)y, (const float*)res, (const float*)yr, scale_r, shift_r, relu, (float*)out, rows,
                       rpb);
  else
    hipLaunchKernelGGL(bn_finalize_apply_kernel<bf16_t>, grid, dim3(256), 0, stream, stats, stat_rows, C, Cpad,
                       (double)count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift,
                       (const bf16_t*)y, (const bf16_t*)res, (const bf16_t*)yr, scale_r, shift_r, relu, (bf16_t*)out,
                       rows, rpb);
  D3F_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// backward: partial sums of dz, dz * xhat -> (dgamma, dbeta, coefficients) ->
//   dy = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)),  dz = dA * [a > 0]   (as bn_bwd_apply_kernel)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_finalize_apply_kernel(
    const float* __restrict__ partial, int nblocks, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int accumulate, float* __restrict__ coef, const T* __restrict__ dA,
    const T* __restrict__ a, const T* __restrict__ y, T* __restrict__ dy, T* __restrict__ dres,
    int dres_acc, long rows, long rows_per_block, const float* __restrict__ mask_scale,
    const float* __restrict__ mask_shift) {
  chain_priority();
  __shared__ double red[16][64];
  __shared__ double tot[64];
  __shared__ float cf[3][BNF_SC];
  const int tid = threadIdx.x, c0 = blockIdx.y * BNF_SC;
  slab_reduce(partial, nblocks, C, c0, red, tot);
  if (tid < BNF_SC) {  // same arithmetic as bn_bwd_finalize_kernel
    const int c = c0 + tid;
    const double s1 = tot[2 * tid], s2 = tot[2 * tid + 1];
    const float k0 = gamma[c] * invstd[c], k1 = (float)(s1 / count), k2 = (float)(s2 / count);
    cf[0][tid] = k0;
    cf[1][tid] = k1;
    cf[2][tid] = k2;
    if (blockIdx.x == 0) {
      const float db = (float)s1, dg = (float)s2;
      if (dgamma != nullptr) {
        dgamma[c] = accumulate ? dgamma[c] + dg : dg;
        dbeta[c] = accumulate ? dbeta[c] + db : db;
      }
      coef[c] = k0;
      coef[C + c] = k1;
      coef[2 * C + c] = k2;
    }
  }
  __syncthreads();
  const int v = tid & 7, rr = tid >> 3, cc = c0 + v * 4;
  float k0[4], k1[4], k2[4], mu[4], is[4], msc[4] = {0.f, 0.f, 0.f, 0.f}, msf[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    k0[k] = cf[0][v * 4 + k];
    k1[k] = cf[1][v * 4 + k];
    k2[k] = cf[2][v * 4 + k];
    mu[k] = mean[cc + k];
    is[k] = invstd[cc + k];
    if (mask_scale != nullptr) {
      msc[k] = mask_scale[cc + k];
      msf[k] = mask_shift[cc + k];
    }
  }
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  const bool from_a = mask_scale == nullptr && a != nullptr;
  const bool rd_dres = dres != nullptr && dres_acc;
  constexpr int U = BNF_BWD_U;
  for (long r = r0 + rr; r < r1; r += 32 * U) {
    float4 g4[U], y4[U], a4[U], d4[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long row = r + 32 * u;
      if (row < r1) {
        g4[u] = ld4<T>(dA + row * C + cc);
        y4[u] = ld4<T>(y + row * C + cc);
        if (from_a) a4[u] = ld4<T>(a + row * C + cc);
        if (rd_dres) d4[u] = ld4<T>(dres + row * C + cc);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long row = r + 32 * u;
      if (row >= r1) continue;
      float g[4] = {g4[u].x, g4[u].y, g4[u].z, g4[u].w};
      const float yy[4] = {y4[u].x, y4[u].y, y4[u].z, y4[u].w};
      if (mask_scale != nullptr) {
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = (yy[k] * msc[k] + msf[k]) > 0.f ? g[k] : 0.f;
      } else if (from_a) {
        const float aa[4] = {a4[u].x, a4[u].y, a4[u].z, a4[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = aa[k] > 0.f ? g[k] : 0.f;
      }
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xhat = (yy[k] - mu[k]) * is[k];
        o[k] = k0[k] * (g[k] - k1[k] - xhat * k2[k]);
      }
      st4<T>(dy + row * C + cc, make_float4(o[0], o[1], o[2], o[3]));
      if (dres != nullptr) {
        if (dres_acc) {
          g[0] += d4[u].x; g[1] += d4[u].y; g[2] += d4[u].z; g[3] += d4[u].w;
        }
        st4<T>(dres + row * C + cc, make_float4(g[0], g[1], g[2], g[3]));
      }
    }
  }
}

int bn_bwd_finalize_apply_launch(int dtype, const float* partial, int nblocks, int C, long count, const float* gamma,
                                 const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                 int accumulate, float* coef, const void* dA, const void* a, const void* y, void* dy,
                                 void* dres, int dres_acc, long rows, hipStream_t stream, const float* mask_scale,
                                 const float* mask_shift) {
  D3F_CHECK(bn_fused_finalize_ok(dtype, nblocks, C), "bn_bwd_finalize_apply: C=%d, %d partial rows", C, nblocks);
  if (rows == 0) return 0;
  const int slabs = C / BNF_SC;
  const long rpb = rows_per_block_for(rows, slabs, dtype);
  const dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)slabs);
  if (dtype == D3F_F32)
    hipLaunchKernelGGL(bn_bwd_finalize_apply_kernel<float>, grid, dim3(256), 0, stream, partial, nblocks, C,
                       (double)count, gamma, mean, invstd, dgamma, dbeta, accumulate, coef, (const float*)dA,
                       (const float*)a, (const float*)y, (float*)dy, (float*)dres, dres_acc, rows, rpb, mask_scale,
                       mask_shift);
  else
    hipLaunchKernelGGL(bn_bwd_finalize_apply_kernel<bf16_t>, grid, dim3(256), 0, stream, partial, nblocks, C,
                       (double)count, gamma, mean, invstd, dgamma, dbeta, accumulate, coef, (const bf16_t*)dA,
                       (const bf16_t*)a, (const bf16_t*)y, (bf16_t*)dy, (bf16_t*)dres, dres_acc, rows, rpb, mask_scale,
                       mask_shift);
  D3F_HIP(hipGetLastError());
  return 0;
}

}  // namespace d3f
