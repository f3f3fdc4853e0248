// Microbenchmark: fp32 contraction emulated on the bf16 matrix pipe by an exact 3-way split
//   x = h + m + l   (h, m, l bf16; truncation splits, so the sum is exact)
//   a*b ~= ah*bh + ah*bm + am*bh + ah*bl + am*bm + al*bh      (the three dropped terms are <= 2^-24 |ab|)
// Part A: accuracy of one 32x32 tile, K = 2304, against a float64 host result, next to the fp32 MFMA.
// Part B: throughput of a 128x128 block tile (4 waves, 64x64 per wave) with the whole per-k-tile pipeline:
//         global loads (L2 resident) -> split (VALU) -> LDS writes -> barrier -> fragment reads -> 6 MFMAs per
//         product tile, in fp32-equivalent TFLOP/s.
// hipcc --offload-arch=gfx950 -O3 -o split_bench split_bench.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ inline void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned xb = __float_as_uint(x);
  const unsigned hb = xb & 0xffff0000u;
  const float r1 = x - __uint_as_float(hb);
  const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(mb);
  h = hb >> 16;
  m = mb >> 16;
  l = __float_as_uint(r2) >> 16;  // exact: r2 has at most 8 significant bits
}
__device__ inline unsigned rne_bf16(float x) {
  const unsigned b = __float_as_uint(x);
  return (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
}

union Frag { bf16x8 v; unsigned short s[8]; uint4 u; };

// ---------------- Part A ----------------
// mode 0: fp32 MFMA 32x32x2; 1: 3-way split, 6 products; 2: 2-way split (RNE high, remainder), 3 products; 3: bf16
__global__ void acc_kernel(const float* A, const float* B, float* C, int K, int mode) {
  const int lane = threadIdx.x, r = lane & 31, q = lane >> 5;
  f32x16 acc = {0};
  if (mode == 0) {
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + q], B[r * K + k + q], acc, 0, 0, 0);
  } else {
    for (int k = 0; k < K; k += 16) {
      Frag a[3], b[3];
      for (int j = 0; j < 8; ++j) {
        const float x = A[r * K + k + q * 8 + j], y = B[r * K + k + q * 8 + j];
        unsigned h, m, l;
        if (mode == 1) {
          split3(x, h, m, l); a[0].s[j] = h; a[1].s[j] = m; a[2].s[j] = l;
          split3(y, h, m, l); b[0].s[j] = h; b[1].s[j] = m; b[2].s[j] = l;
        } else {
          h = rne_bf16(x); a[0].s[j] = h; a[1].s[j] = rne_bf16(x - __uint_as_float(h << 16)); a[2].s[j] = 0;
          h = rne_bf16(y); b[0].s[j] = h; b[1].s[j] = rne_bf16(y - __uint_as_float(h << 16)); b[2].s[j] = 0;
        }
      }
      if (mode == 1) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2].v, b[0].v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[2].v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[1].v, acc, 0, 0, 0);
      }
      if (mode == 1 || mode == 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[0].v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[1].v, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[0].v, acc, 0, 0, 0);
    }
  }
  for (int i = 0; i < 16; ++i) C[((i / 4) * 8 + q * 4 + (i % 4)) * 32 + r] = acc[i];
}

// ---------------- Part B ----------------
// SPLIT: 1 = do the split arithmetic, 0 = store the raw halves (no VALU work)   LOADS: global loads per k-tile
template <int BK, bool SPLIT, bool LOADS, int NPROD>
__global__ __launch_bounds__(256) void thr_kernel(float* out, int nk, const float* in) {
  constexpr int ROWB = BK * 2 + 16;           // bytes per row of one bf16 plane (padded)
  constexpr int PLANE = 128 * ROWB;           // one operand plane
  constexpr int STAGE = 6 * PLANE;            // A: 3 planes, B: 3 planes
  constexpr int PIECES = 128 * BK / 4 / 256;  // float4 pieces per thread per operand
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 31, fq = lane >> 5;
  for (int i = tid; i < 2 * STAGE / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u;
  __syncthreads();
  const float4* g = reinterpret_cast<const float4*>(in);
  float4 ld[2][PIECES];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int i = 0; i < PIECES; ++i) ld[o][i] = g[(o * 1024 + i * 256 + tid) & 4095];
  f32x16 acc[2][2] = {};
  for (int kt = 0; kt < nk; ++kt) {
    unsigned char* W = lds + ((kt + 1) & 1) * STAGE;
    const unsigned char* R = lds + (kt & 1) * STAGE;
    // split + LDS writes of the next stage
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        const int piece = i * 256 + tid, row = piece / (BK / 4), kq = piece % (BK / 4);
        const float x[4] = {ld[o][i].x, ld[o][i].y, ld[o][i].z, ld[o][i].w};
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (SPLIT) split3(x[e], h[e], m[e], l[e]);
          else { h[e] = __float_as_uint(x[e]) >> 16; m[e] = h[e]; l[e] = h[e]; }
        }
        unsigned char* dst = W + o * 3 * PLANE + row * ROWB + kq * 8;
        *reinterpret_cast<uint2*>(dst) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
        *reinterpret_cast<uint2*>(dst + PLANE) = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
        *reinterpret_cast<uint2*>(dst + 2 * PLANE) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
      }
    if (LOADS) {
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int i = 0; i < PIECES; ++i) ld[o][i] = g[((kt * 2 + o) * 1024 + i * 256 + tid) & 4095];
    }
    // fragments + MFMAs of the current stage
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      Frag a[3][2], b[3][2];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[p][i].u = *reinterpret_cast<const uint4*>(R + p * PLANE + (wm * 64 + i * 32 + fr) * ROWB + (s * 16 + fq * 8) * 2);
          b[p][i].u = *reinterpret_cast<const uint4*>(R + (3 + p) * PLANE + (wn * 64 + i * 32 + fr) * ROWB + (s * 16 + fq * 8) * 2);
        }
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 6 - NPROD; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t]][i].v, b[PB[t]][j].v, acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s + ld[0][0].x + ld[1][0].y;
}

template <int BK, bool SPLIT, bool LOADS, int NPROD> void run(const char* name, int blocks, int nk, float* out, float* in) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((thr_kernel<BK, SPLIT, LOADS, NPROD>), dim3(blocks), dim3(256), 0, 0, out, nk, in);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((thr_kernel<BK, SPLIT, LOADS, NPROD>), dim3(blocks), dim3(256), 0, 0, out, nk, in);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 100.0;
  const double flops = (double)blocks * nk * 128.0 * 128.0 * BK * 2.0;  // fp32-equivalent
  printf("%-58s blocks %5d nk %4d: %8.1f us  %6.1f TF fp32-equivalent\n", name, blocks, nk, us, flops / us / 1e6);
}

int main() {
  // ---- Part A ----
  const int K = 2304;
  std::vector<float> hA(32 * K), hB(32 * K), hC(32 * 32);
  unsigned st = 777;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffffff) / 8388608.0f - 1.0f; };
  for (int wide = 0; wide < 2; ++wide) {
    for (auto& v : hA) { v = rnd() * rnd(); if (wide) v = ldexpf(v, (int)(rnd() * 12)); }
    for (auto& v : hB) { v = rnd() * rnd(); if (wide) v = ldexpf(v, (int)(rnd() * 12)); }
    std::vector<double> ref(32 * 32, 0.0), mag(32 * 32, 0.0);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double s = 0, a = 0;
      for (int k = 0; k < K; ++k) { const double p = (double)hA[i * K + k] * hB[j * K + k]; s += p; a += fabs(p); }
      ref[i * 32 + j] = s; mag[i * 32 + j] = a;
    }
    float *dA, *dB, *dC; CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, 4096));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    const char* names[4] = {"fp32 MFMA 32x32x2", "bf16 3-way split, 6 products", "bf16 2-way split, 3 products", "plain bf16"};
    for (int mode = 0; mode < 4; ++mode) {
      hipLaunchKernelGGL(acc_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, mode);
      CK(hipMemcpy(hC.data(), dC, 4096, hipMemcpyDeviceToHost));
      double num = 0, den = 0, worst = 0;
      for (int i = 0; i < 1024; ++i) {
        const double d = hC[i] - ref[i];
        num += d * d; den += ref[i] * ref[i];
        worst = fmax(worst, fabs(d) / mag[i]);  // error relative to sum |a_k b_k| (the condition-free measure)
      }
      printf("accuracy %-6s %-30s rel-L2 %.3e   max |err| / sum|ab| %.3e\n", wide ? "wide" : "narrow", names[mode], sqrt(num / den), worst);
    }
    // fp32 sequential on the host for scale
    double num = 0, den = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      float s = 0.f;
      for (int k = 0; k < K; ++k) s += hA[i * K + k] * hB[j * K + k];
      const double d = s - ref[i * 32 + j]; num += d * d; den += ref[i * 32 + j] * ref[i * 32 + j];
    }
    printf("accuracy %-6s %-30s rel-L2 %.3e\n", wide ? "wide" : "narrow", "host fp32 sequential sum", sqrt(num / den));
    hipFree(dA); hipFree(dB); hipFree(dC);
  }
  // ---- Part B ----
  float *out, *in; CK(hipMalloc(&out, 8192 * 256 * 4)); CK(hipMalloc(&in, 4096 * 16 + 4096));
  {
    std::vector<float> h(4096 * 4 + 1024);
    for (auto& v : h) v = rnd();
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  const int blocks = 2048;
  run<32, true, true, 6>("BK32 split+loads, 6 products (full pipeline)", blocks, 72, out, in);
  run<32, false, true, 6>("BK32 no split VALU, loads, 6 products", blocks, 72, out, in);
  run<32, true, false, 6>("BK32 split, no loads, 6 products", blocks, 72, out, in);
  run<32, true, true, 3>("BK32 split+loads, 3 products", blocks, 72, out, in);
  run<32, true, true, 1>("BK32 split+loads, 1 product (pipeline floor)", blocks, 72, out, in);
  run<16, true, true, 6>("BK16 split+loads, 6 products (full pipeline)", blocks, 144, out, in);
  run<16, false, true, 6>("BK16 no split VALU, loads, 6 products", blocks, 144, out, in);
  run<16, true, true, 1>("BK16 split+loads, 1 product (pipeline floor)", blocks, 144, out, in);
  return 0;
}
