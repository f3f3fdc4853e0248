// microbenchmark: do v_mfma_f32_32x32x2_f32 and vector-ALU instructions of ANOTHER wave on the same SIMD share an
// execution resource?  (Within one wave VALU issue time adds to the fp32 MFMA's -- mfma_bench3.hip; the weight
// gradient runs ~3.6 waves per SIMD, so what matters for it is whether a co-resident wave's VALU work hides behind
// the MFMAs or competes with them.)
//
// One 512-thread workgroup per CU = 8 waves = 2 per SIMD.  Waves 0-3 ("matrix waves") run N x 16 MFMAs on two
// independent accumulators; waves 4-7 ("partner waves") run, for about as long:
//   mode 0: nothing (exit)             mode 1: dependent integer VALU (v_mad_u32_u24 / v_xor chains, 4 independent chains)
//   mode 2: fp32 v_fma_f32 chains      mode 3: the same MFMA loop (2 matrix waves per SIMD: the pipe is shared for sure)
//   mode 4: LDS reads only (ds_read_b128 of a private region)   mode 5: mode 1 at s_setprio 3
// Printed: time of the matrix waves' loop and of the partner waves' loop (wall clock, median over workgroups), each
// kind alone and together.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_corun mfma_valu_corun.hip && ./mfma_valu_corun
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// ROLE: which half of the workgroup's waves are the matrix waves (0: waves 0-3 = the older ones, 1: waves 4-7).  The
// issue arbiter favours the oldest wave, so both loops are timed (matrix: wave 0 or 4, partner: the other) and every
// combination is run alone and together: if the two kinds of work shared an execution resource, at least one of the two
// loops would take longer together than alone.
// BF: the matrix waves stream v_mfma_f32_32x32x16_bf16 (the XDL matrix core) instead of the fp32 MFMA
template <int MODE, int ROLE, bool BF = false>
__global__ __launch_bounds__(512) void bench(float* out, long long* ticks, int n, int partner_iters, const float* in) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int tid = threadIdx.x, wave = tid >> 6;
  for (int i = tid; i < 8192; i += 512) lds[i] = in[i & 4095];
  __syncthreads();
  float res = 0.f;
  const bool matrix = ROLE == 0 ? wave < 4 : wave >= 4;
  const long long t0 = wall_clock64();
  if (matrix) {
    if (n > 0) {
      f32x16 acc = {0}, acc2 = {0};
      const float a0 = in[tid], b0 = in[tid + 512];
      if constexpr (BF) {
        bf16x8 va, vb;
        for (int k = 0; k < 8; ++k) { va[k] = (__bf16)in[tid + k]; vb[k] = (__bf16)in[tid + 8 + k]; }
        for (int it = 0; it < n; ++it) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb, va, acc2, 0, 0, 0);
          }
        }
      } else {
        for (int it = 0; it < n; ++it) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a0, acc2, 0, 0, 0);
          }
        }
      }
      for (int r = 0; r < 16; ++r) res += acc[r] + acc2[r];
    }
  } else if (MODE == 1 || MODE == 5) {
    if (MODE == 5) __builtin_amdgcn_s_setprio(3);  // user wave priority: does the arbiter let this wave in between MFMAs?
    unsigned v0 = tid, v1 = tid * 3, v2 = tid * 5, v3 = tid * 7;
    for (int it = 0; it < partner_iters; ++it) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        v0 = v0 * 5u + v1; v1 ^= v0 >> 3; v2 = v2 * 9u + v3; v3 ^= v2 >> 5;
      }
    }
    res = (float)(v0 + v1 + v2 + v3);
  } else if (MODE == 2) {
    float f0 = in[tid], f1 = in[tid + 1], f2 = in[tid + 2], f3 = in[tid + 3];
    const float m = in[tid + 7];
    for (int it = 0; it < partner_iters; ++it) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        f0 = __builtin_fmaf(f0, m, f1); f1 = __builtin_fmaf(f1, m, f2); f2 = __builtin_fmaf(f2, m, f3); f3 = __builtin_fmaf(f3, m, f0);
      }
    }
    res = f0 + f1 + f2 + f3;
  } else if (MODE == 3) {
    if (partner_iters > 0) {
      f32x16 acc = {0}, acc2 = {0};
      const float a0 = in[tid], b0 = in[tid + 512];
      for (int it = 0; it < partner_iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a0, acc2, 0, 0, 0);
        }
      }
      for (int r = 0; r < 16; ++r) res += acc[r] + acc2[r];
    }
  } else if (MODE == 4) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < partner_iters; ++it) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(&lds[((tid * 4 + q * 64 + it) & 8188)]);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;  // 4 VALU per 16-byte read
      }
    }
    res = s.x + s.y + s.z + s.w;
  }
  const long long t1 = wall_clock64();
  if ((tid & 255) == 0) ticks[blockIdx.x * 2 + (matrix ? 0 : 1)] = t1 - t0;
  out[blockIdx.x * 512 + tid] = res;
}

template <int MODE, int ROLE, bool BF = false>
static void run(const char* what, int n, int partner_iters, float* out, long long* ticks, const float* in) {
  const int wgs = 256;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((bench<MODE, ROLE, BF>), dim3(wgs), dim3(512), 0, 0, out, ticks, n, partner_iters, in);
  CK(hipDeviceSynchronize());
  std::vector<long long> h(2 * wgs);
  CK(hipMemcpy(h.data(), ticks, 2 * wgs * sizeof(long long), hipMemcpyDeviceToHost));
  std::vector<long long> m(wgs), q(wgs);
  for (int i = 0; i < wgs; ++i) { m[i] = h[2 * i]; q[i] = h[2 * i + 1]; }
  std::sort(m.begin(), m.end());
  std::sort(q.begin(), q.end());
  printf("%-58s matrix waves %8.1f us   partner waves %8.1f us\n", what, m[wgs / 2] / 100.0, q[wgs / 2] / 100.0);
}

template <int MODE>
static void trio(const char* name, int n, int pi, float* out, long long* ticks, const float* in) {
  char buf[128];
  snprintf(buf, sizeof(buf), "%s: partner alone", name);
  run<MODE, 0>(buf, 0, pi, out, ticks, in);
  snprintf(buf, sizeof(buf), "%s: together, matrix waves older", name);
  run<MODE, 0>(buf, n, pi, out, ticks, in);
  snprintf(buf, sizeof(buf), "%s: together, matrix waves younger", name);
  run<MODE, 1>(buf, n, pi, out, ticks, in);
}

int main() {
  float *out, *in;
  long long* ticks;
  CK(hipMalloc(&out, 256 * 512 * sizeof(float)));
  CK(hipMalloc(&in, 8192 * sizeof(float)));
  CK(hipMalloc(&ticks, 512 * sizeof(long long)));
  std::vector<float> h(8192);
  for (int i = 0; i < 8192; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  CK(hipMemcpy(in, h.data(), 8192 * sizeof(float), hipMemcpyHostToDevice));
  const int n = 4000;  // 64 000 MFMAs per matrix wave: ~1.7 ms at 64 cycles each (= 157 TFLOP/s chip-wide)
  run<1, 0>("matrix waves alone", n, 0, out, ticks, in);
  run<1, 0>("matrix waves alone (again)", n, 0, out, ticks, in);
  trio<1>("integer VALU chains", n, 15000, out, ticks, in);
  trio<5>("integer VALU chains at s_setprio 3", n, 15000, out, ticks, in);
  trio<2>("fp32 v_fma chains", n, 15000, out, ticks, in);
  trio<4>("ds_read_b128 + 4 VALU each", n, 6000, out, ticks, in);
  trio<3>("second MFMA loop", n, 4000, out, ticks, in);
  // the same question for the bf16 matrix core (32x32x16: 8 passes = 32 cycles per instruction)
  const int nb = 8000;
  run<1, 0, true>("bf16 matrix waves alone", nb, 0, out, ticks, in);
  run<1, 0, true>("bf16 matrix + integer VALU chains, matrix older", nb, 15000, out, ticks, in);
  run<1, 1, true>("bf16 matrix + integer VALU chains, matrix younger", nb, 15000, out, ticks, in);
  run<2, 0, true>("bf16 matrix + fp32 v_fma chains, matrix older", nb, 15000, out, ticks, in);
  run<4, 0, true>("bf16 matrix + ds_read_b128 + 4 VALU, matrix older", nb, 6000, out, ticks, in);
  return 0;
}
