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
#include <type_traits>

namespace d3f {

constexpr int BNF_SC = 32;         // channels per slab = one 128-byte line per tensor row
constexpr int BNF_MAX_ROWS = 1024;  // partial rows a workgroup is asked to reduce (x 256 B): 512 / 1024 / 2048 within 0.2 % of each other; 2048 would take in the stem, whose 512 KB prologue per workgroup makes the pass 3x longer
#ifndef BNF_FWD_U_F32
#define BNF_FWD_U_F32 4  // fp32 rows per batch in the forward streaming pass (8: 7.855 / 7.831 / 7.856 ms against 7.866 / 7.860 / 7.843 with 4 -- no gain, 176 instead of 128 VGPRs)
#endif
#ifndef BNF_BWD_U
#define BNF_BWD_U 2  // rows in flight per thread in the backward streaming pass (4: +0.4 % step time -- 142 VGPRs leave one workgroup per CU next to the weight-gradient stream)
#endif  // partial rows a workgroup is asked to reduce (x 256 B)

bool bn_fused_finalize_ok(int dtype, int stat_rows, int C) {
  static const bool off = prof_knob("D3F_NO_BN_FUSED_FINALIZE") != nullptr;  // debugging knob: separate launches
  return !off && (dtype == D3F_F32 || dtype == D3F_BF16) && (C % BNF_SC) == 0 && stat_rows >= 1 && stat_rows <= BNF_MAX_ROWS;
}

// four channels of one tensor row as they sit in memory (fp32: 16 bytes, bf16: 8 bytes): what a prefetched batch keeps in
// registers until its turn
template <typename T> struct Raw4;
template <> struct Raw4<float> { typedef float4 type; };
template <> struct Raw4<bf16_t> { typedef uint2 type; };
__device__ __forceinline__ float4 ldraw(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ uint2 ldraw(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ float4 cvt4(const float4& v) { return v; }
__device__ __forceinline__ float4 cvt4(const uint2& v) {
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                     __uint_as_float(v.y & 0xffff0000u));
}

// sums the partial rows [rows][ld][2] of channels [c0, c0 + 32) in f64: thread (rl = tid / 16, q = tid % 16) owns the
// float4 q of the slab (channels c0 + 2q, c0 + 2q + 1; sum, second sum each) of rows rl, rl + 16, ...; the 16 row
// lanes are then added in lane order.  tot[2 * ch + which] for ch < 32.  `behind_first_loads()` runs once, right behind
// the issue of the first batch of partial-row loads: the streaming pass puts the loads of its first rows there, so that
// they travel while the reduce waits for its own (the coefficients do not depend on them).
template <typename F>
__device__ __forceinline__ void slab_reduce(const float* __restrict__ partial, int rows, int ld, int c0,
                                            double (&red)[16][64], double (&tot)[64], F behind_first_loads) {
  const int tid = threadIdx.x, q = tid & 15, rl = tid >> 4;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const float* base = partial + ((long)c0 * 2 + q * 4);
  int r = rl;
  bool hooked = false;
  for (; r + 112 < rows; r += 128) {  // eight rows in flight (layer1-type layers bring 512 partial rows)
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(base + (long)(r + 16 * j) * ld * 2);
    if (!hooked) { behind_first_loads(); hooked = true; }
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
    if (!hooked) { behind_first_loads(); hooked = true; }
    a0 += (double)v0.x; a1 += (double)v0.y; a2 += (double)v0.z; a3 += (double)v0.w;
    a0 += (double)v1.x; a1 += (double)v1.y; a2 += (double)v1.z; a3 += (double)v1.w;
    a0 += (double)v2.x; a1 += (double)v2.y; a2 += (double)v2.z; a3 += (double)v2.w;
    a0 += (double)v3.x; a1 += (double)v3.y; a2 += (double)v3.z; a3 += (double)v3.w;
  }
  if (!hooked) behind_first_loads();  // (fewer than 64 partial rows: in front of the tail's loads)
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
// The streaming pass keeps TWO batches of U rows per thread in flight: the loads of batch i + 1 are issued in front of the
// arithmetic and the stores of batch i, and batch 0 travels during the slab reduce -- the pass was latency-bound (a
// layer1-type fp32 launch: 16 rows per thread = four dependent load -> store rounds behind the reduce, 10-13 us for 34 MB).
template <typename T, bool HAS2>
__global__ __launch_bounds__(256) void bn_finalize_apply_kernel(
    const float* __restrict__ stats, int stat_rows, int C, int Cpad, double count,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
    float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ mean_o,
    float* __restrict__ invstd_o, float* __restrict__ scale_o, float* __restrict__ shift_o,
    const T* __restrict__ y, const T* __restrict__ res, const T* __restrict__ yr,
    const float* __restrict__ scale_r, const float* __restrict__ shift_r, int relu, T* __restrict__ out,
    long rows, long rows_per_block, NetSplit ns) {
  chain_priority();
  if (blockIdx.z != 0) {  // two networks in one launch (common.h, NetSplit): blockIdx.z = net
    net_shift(stats, ns.ws); net_shift(gamma, ns.par); net_shift(beta, ns.par);
    net_shift(running_mean, ns.bn); net_shift(running_var, ns.bn);
    net_shift(mean_o, ns.ws); net_shift(invstd_o, ns.ws); net_shift(scale_o, ns.ws); net_shift(shift_o, ns.ws);
    net_shift(y, ns.ws); net_shift(res, ns.ws); net_shift(yr, ns.ws); net_shift(scale_r, ns.ws); net_shift(shift_r, ns.ws);
    net_shift(out, ns.ws);
  }
  __shared__ double red[16][64];
  __shared__ double tot[64];
  __shared__ float cf[2][BNF_SC];
  typedef typename Raw4<T>::type raw_t;
  constexpr int U = sizeof(T) == 4 ? BNF_FWD_U_F32 : 8;  // rows per batch: 2 x U x 16 (8) bytes per thread in flight
  const int tid = threadIdx.x, c0 = blockIdx.y * BNF_SC;
  // streaming pass: thread (rr = tid / 8, v = tid % 8) owns channels c0 + 4v .. + 3 of rows rr, rr + 32, ...
  const int v = tid & 7, rr = tid >> 3, cc = c0 + v * 4;
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  const T* __restrict__ second = res != nullptr ? res : yr;
  raw_t a[U], b[U], an[U], bn[U];
  auto load = [&](raw_t (&ya)[U], raw_t (&yb)[U], long r) {
#pragma unroll
    for (int u = 0; u < U; ++u) {  // rows past the block's end re-read its last row (never stored): straight-line loads
      const long row = r + 32 * u < r1 ? r + 32 * u : r1 - 1;
      ya[u] = ldraw(y + row * C + cc);
      if (HAS2) yb[u] = ldraw(second + row * C + cc);
    }
  };
  slab_reduce(stats, stat_rows, Cpad, c0, red, tot, [&]() { load(a, b, r0 + rr); });
  if (tid < BNF_SC) {  // same arithmetic as bn_finalize_kernel
    const int c = c0 + tid;
    const double mean = tot[2 * tid] / count;
    double var = tot[2 * tid + 1] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float g = gamma[c], bt = beta[c];
    const float sc = (float)((double)g * invstd), sf = (float)((double)bt - mean * (double)g * invstd);
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
  float sc[4], sf[4], scr[4] = {0.f, 0.f, 0.f, 0.f}, sfr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sc[k] = cf[0][v * 4 + k];
    sf[k] = cf[1][v * 4 + k];
    if (HAS2 && res == nullptr) {
      scr[k] = scale_r[cc + k];
      sfr[k] = shift_r[cc + k];
    }
  }
  for (long r = r0 + rr; r < r1; r += 32 * U) {
    if (r + 32 * U < r1) load(an, bn, r + 32 * U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long row = r + 32 * u;
      if (row >= r1) continue;
      const float4 ya = cvt4(a[u]);
      float x[4] = {ya.x, ya.y, ya.z, ya.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = x[k] * sc[k] + sf[k];
      if (HAS2) {
        const float4 yb = cvt4(b[u]);
        if (res != nullptr) {
          x[0] += yb.x; x[1] += yb.y; x[2] += yb.z; x[3] += yb.w;
        } else {
          const float t[4] = {yb.x, yb.y, yb.z, yb.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) x[k] += t[k] * scr[k] + sfr[k];
        }
      }
      if (relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = fmaxf(x[k], 0.f);
      }
      st4<T>(out + row * C + cc, make_float4(x[0], x[1], x[2], x[3]));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a[u] = an[u];
      if (HAS2) b[u] = bn[u];
    }
  }
}

// rows per workgroup, whole passes of 32 rows.  Workgroups in all: ~256 for fp32 tensors, ~512 for bf16 (r03 sweep of
// 64 ... 1024 with the chain's kernels at wave priority 3: fp32 256x256 8.88 / 8.38 / 8.30 / 8.24 / 8.28 / 8.30 / 8.35 ms
// per step at 64 / 128 / 192 / 256 / 320 / 512 / 768 -- every workgroup repeats the slab reduce, fewer of them repeat it
// less; bf16 4.83 / 4.58 / 4.55 at 128 / 256 / 512: half the bytes per row, the streaming part wants the parallelism;
// re-swept in round 4 with two batches in flight: fp32 7.87 / 7.91 / 7.94 / 7.94 / 8.02 ms at 256 / 384 / 512 / 768 / 1024,
// bf16 4.09 / 4.09 / 4.12 / 4.18 at 384 / 512 / 768 / 1024 -- unchanged optimum; 512 / 768 / 1024 only for the tensors of 32 MB
// and more (stem, decoder block 3): 7.89 / 7.91 / 7.90 against 7.87 / 7.89 -- no gain either)
static long rows_per_block_for(long rows, int slabs, int dtype, int plan_nets) {
  const long wgs = (dtype == D3F_F32 ? 256 : 512) / plan_nets_for(plan_nets, 64);  // (two networks in one launch share the count)
  long rb = std::max(1L, wgs / slabs);
  long rpb = (rows + rb - 1) / rb;
  rpb = (rpb + 31) / 32 * 32;
  return std::max(32L, rpb);
}

int bn_finalize_apply_launch(int dtype, const float* stats, int stat_rows, int C, int Cpad, long count, const float* gamma,
                             const float* beta, float eps, float momentum, float* running_mean,
                             float* running_var, float* mean, float* invstd, float* scale, float* shift,
                             const void* y, const void* res, const void* yr, const float* scale_r,
                             const float* shift_r, int relu, void* out, long rows, hipStream_t stream,
                             const NetSplit* ns, int plan_nets) {
  D3F_CHECK(bn_fused_finalize_ok(dtype, stat_rows, C), "bn_finalize_apply: C=%d, %d partial rows", C, stat_rows);
  if (rows == 0) return 0;
  const int slabs = C / BNF_SC;
  const long rpb = rows_per_block_for(rows, slabs, dtype, plan_nets);
  const NetSplit nv = net_split_or_single(ns);
  const dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)slabs, (unsigned)nv.nets);
  const bool has2 = res != nullptr || yr != nullptr;
  auto go = [&](auto kernel, auto* typed) {
    typedef std::remove_pointer_t<decltype(typed)> T;
    hipLaunchKernelGGL(kernel, grid, dim3(256), 0, stream, stats, stat_rows, C, Cpad, (double)count, gamma, beta, eps,
                       momentum, running_mean, running_var, mean, invstd, scale, shift, (const T*)y, (const T*)res,
                       (const T*)yr, scale_r, shift_r, relu, (T*)out, rows, rpb, nv);
  };
  if (dtype == D3F_F32) {
    if (has2) go(bn_finalize_apply_kernel<float, true>, (float*)nullptr);
    else go(bn_finalize_apply_kernel<float, false>, (float*)nullptr);
  } else {
    if (has2) go(bn_finalize_apply_kernel<bf16_t, true>, (bf16_t*)nullptr);
    else go(bn_finalize_apply_kernel<bf16_t, false>, (bf16_t*)nullptr);
  }
  D3F_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// backward: partial sums of dz, dz * xhat -> (dgamma, dbeta, coefficients) ->
//   dy = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)),  dz = dA * [a > 0]   (as bn_bwd_apply_kernel)
// ------------------------------------------------------------------------------------------
// FROM_A: the ReLU mask is read from the stored activation (layers with a residual add; else it is recomputed from
// y * mask_scale + mask_shift, or there is no ReLU); RD_DRES: dres is accumulated into.  Template flags so that a launch
// keeps only the tensors it reads in registers: two batches of U rows per thread in flight as in the forward pass.
template <typename T, bool FROM_A, bool RD_DRES>
__global__ __launch_bounds__(256) void bn_bwd_finalize_apply_kernel(
    const float* __restrict__ partial, int nblocks, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int accumulate, float* __restrict__ coef, const T* __restrict__ dA,
    const T* __restrict__ a, const T* __restrict__ y, T* __restrict__ dy, T* __restrict__ dres,
    long rows, long rows_per_block, const float* __restrict__ mask_scale,
    const float* __restrict__ mask_shift, NetSplit ns) {
  chain_priority();
  if (blockIdx.z != 0) {  // two networks in one launch (common.h, NetSplit): blockIdx.z = net
    net_shift(partial, ns.ws); net_shift(gamma, ns.par); net_shift(mean, ns.ws); net_shift(invstd, ns.ws);
    net_shift(dgamma, ns.grad); net_shift(dbeta, ns.grad); net_shift(coef, ns.ws);
    net_shift(dA, ns.ws); net_shift(a, ns.ws); net_shift(y, ns.ws); net_shift(dy, ns.ws); net_shift(dres, ns.ws);
    net_shift(mask_scale, ns.ws); net_shift(mask_shift, ns.ws);
  }
  __shared__ double red[16][64];
  __shared__ double tot[64];
  __shared__ float cf[3][BNF_SC];
  typedef typename Raw4<T>::type raw_t;
  constexpr int U = BNF_BWD_U;
  const int tid = threadIdx.x, c0 = blockIdx.y * BNF_SC;
  const int v = tid & 7, rr = tid >> 3, cc = c0 + v * 4;
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  raw_t g4[U], y4[U], a4[U], d4[U], g4n[U], y4n[U], a4n[U], d4n[U];
  auto load = [&](raw_t (&gg)[U], raw_t (&yy)[U], raw_t (&aa)[U], raw_t (&dd)[U], long r) {
#pragma unroll
    for (int u = 0; u < U; ++u) {  // rows past the block's end re-read its last row (never stored): straight-line loads
      const long row = r + 32 * u < r1 ? r + 32 * u : r1 - 1;
      gg[u] = ldraw(dA + row * C + cc);
      yy[u] = ldraw(y + row * C + cc);
      if (FROM_A) aa[u] = ldraw(a + row * C + cc);
      if (RD_DRES) dd[u] = ldraw(dres + row * C + cc);
    }
  };
  slab_reduce(partial, nblocks, C, c0, red, tot, [&]() { load(g4, y4, a4, d4, r0 + rr); });
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
  float k0[4], k1[4], k2[4], mu[4], is[4], msc[4] = {0.f, 0.f, 0.f, 0.f}, msf[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    k0[k] = cf[0][v * 4 + k];
    k1[k] = cf[1][v * 4 + k];
    k2[k] = cf[2][v * 4 + k];
    mu[k] = mean[cc + k];
    is[k] = invstd[cc + k];
    if (!FROM_A && mask_scale != nullptr) {
      msc[k] = mask_scale[cc + k];
      msf[k] = mask_shift[cc + k];
    }
  }
  for (long r = r0 + rr; r < r1; r += 32 * U) {
    if (r + 32 * U < r1) load(g4n, y4n, a4n, d4n, r + 32 * U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long row = r + 32 * u;
      if (row >= r1) continue;
      const float4 gv = cvt4(g4[u]), yv = cvt4(y4[u]);
      float g[4] = {gv.x, gv.y, gv.z, gv.w};
      const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
      if (FROM_A) {
        const float4 av = cvt4(a4[u]);
        const float aa[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = aa[k] > 0.f ? g[k] : 0.f;
      } else if (mask_scale != nullptr) {
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = (yy[k] * msc[k] + msf[k]) > 0.f ? g[k] : 0.f;
      }
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xhat = (yy[k] - mu[k]) * is[k];
        o[k] = k0[k] * (g[k] - k1[k] - xhat * k2[k]);
      }
      st4<T>(dy + row * C + cc, make_float4(o[0], o[1], o[2], o[3]));
      if (dres != nullptr) {
        if (RD_DRES) {
          const float4 dv = cvt4(d4[u]);
          g[0] += dv.x; g[1] += dv.y; g[2] += dv.z; g[3] += dv.w;
        }
        st4<T>(dres + row * C + cc, make_float4(g[0], g[1], g[2], g[3]));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      g4[u] = g4n[u];
      y4[u] = y4n[u];
      if (FROM_A) a4[u] = a4n[u];
      if (RD_DRES) d4[u] = d4n[u];
    }
  }
}

int bn_bwd_finalize_apply_launch(int dtype, const float* partial, int nblocks, int C, long count, const float* gamma,
                                 const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                 int accumulate, float* coef, const void* dA, const void* a, const void* y, void* dy,
                                 void* dres, int dres_acc, long rows, hipStream_t stream, const float* mask_scale,
                                 const float* mask_shift, const NetSplit* ns, int plan_nets) {
  D3F_CHECK(bn_fused_finalize_ok(dtype, nblocks, C), "bn_bwd_finalize_apply: C=%d, %d partial rows", C, nblocks);
  if (rows == 0) return 0;
  const int slabs = C / BNF_SC;
  const long rpb = rows_per_block_for(rows, slabs, dtype, plan_nets);
  const NetSplit nv = net_split_or_single(ns);
  const dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)slabs, (unsigned)nv.nets);
  const bool from_a = mask_scale == nullptr && a != nullptr;
  const bool rd_dres = dres != nullptr && dres_acc;
  auto go = [&](auto kernel, auto* typed) {
    typedef std::remove_pointer_t<decltype(typed)> T;
    hipLaunchKernelGGL(kernel, grid, dim3(256), 0, stream, partial, nblocks, C, (double)count, gamma, mean, invstd, dgamma,
                       dbeta, accumulate, coef, (const T*)dA, (const T*)a, (const T*)y, (T*)dy, (T*)dres, rows, rpb,
                       mask_scale, mask_shift, nv);
  };
  auto pick = [&](auto* typed) {
    typedef std::remove_pointer_t<decltype(typed)> T;
    if (from_a && rd_dres) go(bn_bwd_finalize_apply_kernel<T, true, true>, typed);
    else if (from_a) go(bn_bwd_finalize_apply_kernel<T, true, false>, typed);
    else if (rd_dres) go(bn_bwd_finalize_apply_kernel<T, false, true>, typed);
    else go(bn_bwd_finalize_apply_kernel<T, false, false>, typed);
  };
  if (dtype == D3F_F32) pick((float*)nullptr);
  else pick((bf16_t*)nullptr);
  D3F_HIP(hipGetLastError());
  return 0;
}

}  // namespace d3f
