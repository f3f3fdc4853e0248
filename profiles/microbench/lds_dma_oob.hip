// Does a buffer load with an LDS destination (LDS-DMA, `buffer_load_dwordx4 ... offen lds`) write ZEROS for lanes whose
// offset is out of range, or leave the LDS bytes as they were?          hipcc --offload-arch=gfx950 -O2 lds_dma_oob.hip
// Result on MI355X (r03): ZEROS are written -- for offsets past num_records, for the 0x80000000 marker the conv kernels
// use, and the wave-uniform soffset takes part in the range check.  One more finding from the kernel that used this
// (profiles/README.md, negative results): the per-lane offset and soffset are added WITHOUT 32-bit wrap-around before
// the check, so "negative" per-lane offsets that a scalar offset brings back into range read as out of range.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__global__ void k(const uint32_t* src, unsigned bytes, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[512];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xDEADBEEFu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = make_rsrc(src, bytes);
  unsigned off = threadIdx.x * 16u;
  if (threadIdx.x == 5) off = 0x80000000u;          // explicit OOB marker as the conv kernels use it
  if (threadIdx.x == 9) off = 0x80000000u + 1024u;  // marker + a scalar delta
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
  // second piece at LDS + 1024 with soffset
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 256), 16, off, 64, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
  const int n = 512;
  std::vector<uint32_t> h(n);
  for (int i = 0; i < n; ++i) h[i] = 0x1000u + i;
  uint32_t *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 512 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  const unsigned bytes = 48 * 16;  // lanes 48..63 out of range
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, bytes, o);
  std::vector<uint32_t> r(512);
  hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
  for (int piece = 0; piece < 2; ++piece) {
    printf("piece %d:\n", piece);
    for (int l = 0; l < 64; ++l) printf("%s lane %2d: %08x %08x %08x %08x\n", (l == 5 || l == 9 || l >= 44) ? "*" : " ", l, r[piece * 256 + l * 4], r[piece * 256 + l * 4 + 1], r[piece * 256 + l * 4 + 2], r[piece * 256 + l * 4 + 3]);
  }
  return 0;
}
