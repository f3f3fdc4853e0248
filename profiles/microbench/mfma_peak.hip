// What fp32 MFMA rate does THIS chip hold, and at which shader clock?  (DESIGN quoted "131 TFLOP/s on
// real data" from mfma_bench.hip without its output; the guide quotes 155 TFLOP/s on random data.)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
// Every wave issues v_mfma_f32_32x32x2_f32 back to back from registers (no LDS, no memory): NACC independent accumulators,
// WPS waves per SIMD, operands all zero / random in [-1, 1) / random with a wide exponent spread.  Reports TFLOP/s from HIP
// events and the shader clock from s_memtime (shader cycles) against s_memrealtime (100 MHz) taken inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void peak(const float* __restrict__ in, float* out, unsigned long long* clk, int iters) {
  const int tid = threadIdx.x;
  float a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 4 + i) & 4095]; b[i] = in[(tid * 4 + i + 2048) & 4095]; }
  f32x16 acc[NACC];
#pragma unroll
  for (int q = 0; q < NACC; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();  // s_memtime: shader clock
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + q) & 3], acc[q], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < NACC; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
static void run(const char* what, const float* din, float* dout, unsigned long long* dclk, int wgs, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(peak<NACC>, dim3(wgs), dim3(256), 0, 0, din, dout, dclk, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
  }
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> clk(2 * wgs);
  CK(hipMemcpy(clk.data(), dclk, clk.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, real = 0;
  for (int i = 0; i < wgs; ++i) { cyc += (double)clk[2 * i]; real += (double)clk[2 * i + 1]; }
  const double flop = (double)wgs * 4 /*waves*/ * iters * 4.0 * NACC * (2.0 * 32 * 32 * 2);
  const double mhz = cyc / real * 100.0;  // s_memrealtime ticks at 100 MHz
  // cycles per MFMA per wave slot, from the shader clock: 16 passes x 4 cycles = 64 when one wave per SIMD issues back to back
  const double cyc_per_mfma = (cyc / wgs) / ((double)iters * 4 * NACC);
  printf("%-34s wgs %4d (%d wave/SIMD) acc %d : %7.1f TFLOP/s  kernel %.3f ms  shader clock %.0f MHz  %.1f cycles per MFMA per wave\n",
         what, wgs, wgs / 256, NACC, flop / (ms * 1e-3) * 1e-12, ms, mhz, cyc_per_mfma);
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("%s, %d CUs, clockRate %d kHz; nominal fp32 MFMA peak = CUs x 4 SIMD x 64 FLOP/clk x 2.4 GHz = %.1f TFLOP/s\n", prop.gcnArchName,
         prop.multiProcessorCount, prop.clockRate, prop.multiProcessorCount * 4 * 64 * 2.4e9 * 1e-12);
  std::vector<float> h(4096);
  float *dz, *dr, *dw, *dout;
  unsigned long long* dclk;
  CK(hipMalloc(&dz, 4096 * 4)); CK(hipMalloc(&dr, 4096 * 4)); CK(hipMalloc(&dw, 4096 * 4));
  CK(hipMalloc(&dout, 2048 * 256 * 4)); CK(hipMalloc(&dclk, 2048 * 16));
  CK(hipMemset(dz, 0, 4096 * 4));
  srand(1);
  for (auto& v : h) v = (float)(2.0 * rand() / (double)RAND_MAX - 1.0);
  CK(hipMemcpy(dr, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  for (auto& v : h) v = ((float)(2.0 * rand() / (double)RAND_MAX - 1.0)) * ldexpf(1.f, rand() % 16 - 8);
  CK(hipMemcpy(dw, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  const int iters = 20000;
  for (int wps = 1; wps <= 2; ++wps) {
    const int wgs = 256 * wps;
    run<1>("zeros", dz, dout, dclk, wgs, iters);
    run<2>("zeros", dz, dout, dclk, wgs, iters / 2);
    run<4>("zeros", dz, dout, dclk, wgs, iters / 4);
    run<1>("random [-1,1)", dr, dout, dclk, wgs, iters);
    run<2>("random [-1,1)", dr, dout, dclk, wgs, iters / 2);
    run<4>("random [-1,1)", dr, dout, dclk, wgs, iters / 4);
    run<4>("random, exponents 2^-8..2^7", dw, dout, dclk, wgs, iters / 4);
  }
  // a longer run: does the clock sag once the chip has been under load for a while?
  run<4>("random [-1,1), 10x longer", dr, dout, dclk, 512, 10 * iters / 4);
  return 0;
}
