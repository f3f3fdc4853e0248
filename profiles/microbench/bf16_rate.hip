// pure v_mfma_f32_32x32x16_bf16 issue rate: W waves per SIMD, ACC independent accumulators, F filler VALU per MFMA
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int ACC, int FILL>
__global__ __launch_bounds__(256) void k(float* out, int n, const uint4* in) {
  union { uint4 u; bf16x8 v; } a, b;
  a.u = in[threadIdx.x]; b.u = in[threadIdx.x + 256];
  f32x16 acc[ACC] = {};
  unsigned va = threadIdx.x, vb = 3 * threadIdx.x;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int t = 0; t < 24; ++t) {
      acc[t % ACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[t % ACC], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < FILL; ++f) { va = va * 5 + vb; vb = vb ^ (va >> 3); }
    }
  }
  float s = 0;
  for (int j = 0; j < ACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(va + vb);
}
template <int ACC, int FILL> void run(int blocks, const char* name, float* out, uint4* in) {
  const int n = 200;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<ACC, FILL>), dim3(blocks), dim3(256), 0, 0, out, n, in);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<ACC, FILL>), dim3(blocks), dim3(256), 0, 0, out, n, in);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 200.0, flops = (double)blocks * 4 * n * 24 * 32768.0;
  // cycles per MFMA per SIMD at 2.4 GHz, given how many waves share a SIMD
  const double waves_per_simd = (double)blocks * 4 / (256.0 * 4);
  printf("%-40s blocks %5d: %8.1f us  %7.1f TF   %.1f ns per MFMA per SIMD\n", name, blocks, us, flops / us / 1e6,
         us * 1e3 / (n * 24 * (waves_per_simd < 1 ? 1 : waves_per_simd)));
}
int main() {
  float* out; uint4* in; CK(hipMalloc(&out, 8192 * 256 * 4)); CK(hipMalloc(&in, 8192)); CK(hipMemset(in, 0x3f, 8192));
  run<1, 0>(256, "1 wave/SIMD, 1 acc", out, in);
  run<4, 0>(256, "1 wave/SIMD, 4 acc", out, in);
  run<1, 0>(512, "2 waves/SIMD, 1 acc", out, in);
  run<4, 0>(512, "2 waves/SIMD, 4 acc", out, in);
  run<4, 0>(2048, "8 waves/SIMD-equiv (2048 blocks), 4 acc", out, in);
  run<4, 2>(512, "2 waves/SIMD, 4 acc, 4 VALU per MFMA", out, in);
  run<4, 4>(512, "2 waves/SIMD, 4 acc, 8 VALU per MFMA", out, in);
  run<4, 2>(256, "1 wave/SIMD, 4 acc, 4 VALU per MFMA", out, in);
  run<4, 4>(256, "1 wave/SIMD, 4 acc, 8 VALU per MFMA", out, in);
  return 0;
}
