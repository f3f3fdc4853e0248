// Does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) let two consecutive kernels of ONE stream run side by side on
// gfx950 (AQL barrier bit cleared), and does a later ordinary launch still wait for both?
//   hipcc --offload-arch=gfx950 -O2 anyorder_launch.hip -o anyorder_launch && ./anyorder_launch
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(float* out, int iters, float seed) {
  float a = seed + threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; ++i) a = a * b + 0.5f;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
// writes late (after the spin), so a reader that did not wait sees the old value
__global__ void slow_write_kernel(float* buf, int n, int iters, float value) {
  float a = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] = value + (a > 1e30f ? 1.f : 0.f);
}
__global__ void copy_kernel(const float* in, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  hipStream_t s, s2;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  float *a, *b;
  const int wgs = 128, thr = 256;
  CK(hipMalloc(&a, wgs * thr * 4));
  CK(hipMalloc(&b, wgs * thr * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 400000;
  auto timeit = [&](int mode, const char* name) -> int {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s));
      for (int k = 0; k < 4; ++k) {
        hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(thr), 0, s, a, iters, 1.f);
        if (mode == 0) hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(thr), 0, s, b, iters, 2.f);
        else if (mode == 1) hipExtLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(thr), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, b, iters, 2.f);
        else hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(thr), 0, s2, b, iters, 2.f);
      }
      CK(hipEventRecord(e1, s));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-44s rep %d: %.3f ms for 4 x 2 kernels of %d workgroups\n", name, rep, ms, wgs);
    }
    return 0;
  };
  if (timeit(0, "same stream, ordinary launches")) return 1;
  if (timeit(1, "same stream, 2nd of each pair any-order")) return 1;
  if (timeit(2, "two streams")) return 1;

  // ordering: [A: slow write x := 1] [B any-order: slow write y := 2] [C ordinary: copy x -> ox] [D any-order: copy y -> oy]
  const int n = 1 << 16;
  float *x, *y, *ox, *oy;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&ox, n * 4)); CK(hipMalloc(&oy, n * 4));
  int bad = 0;
  for (int rep = 0; rep < 20; ++rep) {
    CK(hipMemsetAsync(x, 0, n * 4, s)); CK(hipMemsetAsync(y, 0, n * 4, s));
    CK(hipMemsetAsync(ox, 0, n * 4, s)); CK(hipMemsetAsync(oy, 0, n * 4, s));
    const float va = 1.f + rep, vb = 100.f + rep;
    hipLaunchKernelGGL(slow_write_kernel, dim3(n / 256), dim3(256), 0, s, x, n, 20000, va);
    hipExtLaunchKernelGGL(slow_write_kernel, dim3(n / 256), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, y, n, 60000, vb);
    hipLaunchKernelGGL(copy_kernel, dim3(n / 256), dim3(256), 0, s, x, ox, n);
    hipExtLaunchKernelGGL(copy_kernel, dim3(n / 256), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, y, oy, n);
    CK(hipStreamSynchronize(s));
    std::vector<float> hx(n), hy(n);
    CK(hipMemcpy(hx.data(), ox, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hy.data(), oy, n * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) bad += (hx[i] != va) + (hy[i] != vb);
  }
  printf("ordering check (ordinary launch behind an any-order pair waits for BOTH, any-order twin behind it too): %d wrong values\n", bad);
  return bad != 0;
}
