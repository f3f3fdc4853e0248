// microbenchmark: what limits a 4-wave workgroup doing 16 x v_mfma_f32_32x32x2_f32 per "k-tile"?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// MODE bits: 1 = LDS fragment reads (8 x ds_read_b128 per k-tile), 2 = one barrier per k-tile,
// 4 = 4 x ds_write_b128 per k-tile, 8 = ~64 dummy VALU per k-tile, 16 = two independent accumulators,
// 32 = ~32 dependent SALU per k-tile, 64 = 4 buffer-style global loads per k-tile (L2 resident, consumed
// by the next k-tile's LDS writes), 128 = 12 cheap VALU per k-tile (the FAST address path)
template <int MODE>
__global__ __launch_bounds__(256) void bench(float* out, int nk, const float* in) {
  __shared__ __attribute__((aligned(16))) unsigned lds[2 * 128 * 36];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 128 * 36; i += 256) lds[i] = (unsigned)i;
  __syncthreads();
  f32x16 acc = {0}, acc2 = {0};
  float a0 = in[tid], b0 = in[tid + 256];
  uint4 stv = make_uint4(tid, tid + 1, tid + 2, tid + 3);
  int va = tid, vb = tid * 3;
  int sa = nk, sb = nk * 7;  // wave-uniform (SGPR) state
  const uint4* gsrc = reinterpret_cast<const uint4*>(in);
  uint4 ld[4] = {stv, stv, stv, stv};
  const int fr = lane & 31, fq = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned* As = lds + (kt & 1) * 128 * 36;
    const unsigned* Bs = As + 64 * 36;
    if (MODE & 4) {
      unsigned* Ws = lds + ((kt + 1) & 1) * 128 * 36;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<uint4*>(&Ws[((tid >> 3) + 32 * i) * 36 + (tid & 7) * 4]) = (MODE & 64) ? ld[i] : stv;
    }
    if (MODE & 64) {
#pragma unroll
      for (int i = 0; i < 4; ++i) ld[i] = gsrc[((kt * 4 + i) * 256 + tid) & 16383];
    }
    if (MODE & 32) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sa = sa * 3 + sb; sb = (sb ^ (sa >> 2)) + kt; sa = sa > sb ? sa - sb : sa + 5; sb += sa & 7;
      }
    }
    if (MODE & 128) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { va = (va >> (sa & 3)) & 0xffff; vb = va ? vb + sb : 0x7fffffff; va += vb; vb ^= 5; }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint4 a, b;
      if (MODE & 1) {
        const int ch = 2 * s + fq;
        a = *reinterpret_cast<const uint4*>(&As[(wm * 32 + fr) * 36 + ch * 4]);
        b = *reinterpret_cast<const uint4*>(&Bs[(wn * 32 + fr) * 36 + ch * 4]);
      } else {
        a = make_uint4(__float_as_uint(a0), __float_as_uint(b0), __float_as_uint(a0), __float_as_uint(b0));
        b = a;
      }
      const float av[4] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)};
      const float bv[4] = {__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if ((MODE & 16) && (q & 1)) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc2, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
        if (MODE & 8) {
#pragma unroll
          for (int v = 0; v < 4; ++v) { va = va * 5 + vb; vb = vb ^ (va >> 3); }
        }
      }
    }
    if (MODE & 2) __syncthreads();
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r];
  out[blockIdx.x * 256 + tid] = s + (float)(va + vb) + (float)(sa + sb) + __uint_as_float(ld[0].x ^ ld[1].y ^ ld[2].z ^ ld[3].w);
}

template <int MODE> void run(const char* name, int blocks, int nk, float* out, float* in) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(256), 0, 0, out, nk, in);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(256), 0, 0, out, nk, in);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 100.0;  // per launch
  const double flops = (double)blocks * 4 * nk * 16 * 4096.0;
  printf("%-44s blocks %5d nk %4d: %8.1f us  %6.1f TF\n", name, blocks, nk, us, flops / us / 1e6);
}

int main() {
  float *out, *in; CK(hipMalloc(&out, 4096 * 256 * 4)); CK(hipMalloc(&in, 16384 * 16 + 4096)); CK(hipMemset(in, 0, 16384 * 16 + 4096));
  for (int blocks : {512}) {
    const int nk = 36 * 512 / blocks * 4;  // long enough to amortise launch
    run<0>("mfma only (1 acc chain)", blocks, nk, out, in);
    run<16>("mfma only (2 acc chains)", blocks, nk, out, in);
    run<1>("+ lds reads", blocks, nk, out, in);
    run<1 | 2>("+ lds reads + barrier", blocks, nk, out, in);
    run<1 | 2 | 4>("+ reads + barrier + lds writes", blocks, nk, out, in);
    run<1 | 2 | 4 | 8>("+ reads + barrier + writes + 64 valu", blocks, nk, out, in);
    run<1 | 2 | 4 | 8 | 16>("all, 2 acc chains", blocks, nk, out, in);
    run<1 | 2 | 4 | 32>("reads+barrier+writes + 32 SALU", blocks, nk, out, in);
    run<1 | 2 | 4 | 128>("reads+barrier+writes + 12 cheap VALU", blocks, nk, out, in);
    run<1 | 2 | 4 | 64>("reads+barrier+writes + 4 global loads", blocks, nk, out, in);
    run<1 | 2 | 4 | 32 | 64 | 128>("reads+barrier+writes + SALU + VALU + loads", blocks, nk, out, in);
  }
  return 0;
}
