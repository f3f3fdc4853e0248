// cost of a software grid barrier (atomic counter + agent-scope fences) between the phases of a cooperative kernel:
// the price of fusing conv -> BatchNorm statistics -> normalise into one launch.
// hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip ; run under `timeout` (a barrier that never
// completes would hang the device).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned nblocks, unsigned& phase) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();                                   // release this block's writes (agent scope)
    const unsigned target = (phase + 1) * nblocks;
    atomicAdd(counter, 1u);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    __threadfence();                                   // acquire the others'
  }
  ++phase;
  __syncthreads();
}

__global__ __launch_bounds__(256) void k(unsigned* counter, float* data, int iters, int work) {
  unsigned phase = 0;
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    for (int w = 0; w < work; ++w) acc += data[(blockIdx.x * 256 + threadIdx.x + w * 131) & 65535];
    data[(blockIdx.x * 256 + threadIdx.x) & 65535] = acc;   // something for the fence to publish
    grid_barrier(counter, gridDim.x, phase);
  }
  if (acc == 12345.f) data[0] = acc;
}

int main() {
  unsigned* counter; float* data;
  CK(hipMalloc(&counter, 4)); CK(hipMalloc(&data, 65536 * 4)); CK(hipMemset(data, 0, 65536 * 4));
  for (int blocks : {256, 512, 1024}) {
    for (int work : {0, 16}) {
      int iters = 200;
      void* args[] = {&counter, &data, &iters, &work};
      CK(hipMemset(counter, 0, 4));
      CK(hipLaunchCooperativeKernel((void*)k, dim3(blocks), dim3(256), args, 0, 0));   // warm-up
      CK(hipDeviceSynchronize());
      CK(hipMemset(counter, 0, 4));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      CK(hipLaunchCooperativeKernel((void*)k, dim3(blocks), dim3(256), args, 0, 0));
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("blocks %4d, %2d loads between barriers: %.2f us per iteration\n", blocks, work, ms * 1e3 / iters);
    }
  }
  return 0;
}
