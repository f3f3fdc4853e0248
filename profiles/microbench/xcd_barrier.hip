// Re-pricing of "conv + train-mode BatchNorm in ONE launch" (VERDICT r3 weak #5): what does the global dependency between
// "statistics complete" and "normalise" cost as an in-kernel grid barrier, against the kernel boundary it would replace?
//   hipcc --offload-arch=gfx950 -O3 -o xcd_barrier xcd_barrier.hip && timeout -k 5 120 ./xcd_barrier
// Three forms of the dependency, each behind a phase in which every workgroup writes `kb` KB (the conv epilogue's y tile;
// 32 KB x 512 workgroups = layer1's 16.8 MB) and followed by a phase that reads one 128-byte record per workgroup:
//   flat     one monotonic counter, release fence + arrive + poll + acquire fence per workgroup (microbench/grid_barrier.hip)
//   xcd      hierarchical, as MI355X_MICROARCH.md "barrier-xcd": per-XCC arrival counter; the LAST arriver of an XCC is its
//            leader: ONE release fence per XCD (the L2 is shared by the XCD's CUs) -> top counter -> poll -> acquire ->
//            publishes the XCC's generation word; the other workgroups poll that word and do an acquire fence
//   launch   the same two phases as two dependent launches on one stream (what the library does today)
// Every spin is bounded (a poll budget, then a failure flag: the kernel ends); grids of 1, 2 and 4 workgroups per CU of
// 256 threads and < 32 registers are co-resident by a wide margin (8 would fit).  The XCC census (workgroups per XCC) is
// taken once behind a flat barrier -- the block -> XCC map is not architecturally defined.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Bar {                     // every word on its own 128-byte line
  unsigned flat[32];
  unsigned top[32];
  unsigned arrive[8][32];
  unsigned gen[8][32];
  unsigned census[8][32];
  unsigned fail[32];
};
constexpr unsigned SPIN_BUDGET = 1u << 22;  // polls (with s_sleep): tens of ms, then give up

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }  // HW_REG_XCC_ID[3:0]
__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool wait_ge(const unsigned* p, unsigned target, unsigned* fail) {
  for (unsigned i = 0; i < SPIN_BUDGET; ++i) {
    if (ld_relaxed(p) >= target) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  atomicOr(fail, 1u);
  return false;
}

__device__ __forceinline__ void barrier_flat(Bar* b, unsigned nblocks, unsigned& phase) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(b->flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wait_ge(b->flat, (phase + 1) * nblocks, b->fail);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  ++phase;
  __syncthreads();
}

__device__ __forceinline__ void barrier_xcd(Bar* b, unsigned xcc, unsigned n_on_xcc, unsigned n_xcc, unsigned& phase) {
  __syncthreads();  // every wave's stores have left for the (XCD-shared) L2
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(&b->arrive[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (phase + 1) * n_on_xcc - 1) {  // last arriver of this XCC: its leader for this phase
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // one L2 write-back per XCD
      __hip_atomic_fetch_add(b->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wait_ge(b->top, (phase + 1) * n_xcc, b->fail);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(&b->gen[xcc][0], phase + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      wait_ge(&b->gen[xcc][0], phase + 1, b->fail);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  ++phase;
  __syncthreads();
}

__device__ __forceinline__ void write_phase(float* data, int kb, int it) {
  float4* dst = reinterpret_cast<float4*>(data) + (size_t)blockIdx.x * kb * 64;  // kb KB = kb * 64 float4
  const float v = (float)(it + (int)threadIdx.x);
  for (int i = threadIdx.x; i < kb * 64; i += 256) dst[i] = make_float4(v, v, v, v);
}
__device__ __forceinline__ float read_phase(const float* data, int kb) {
  // the 128-byte record of ANOTHER workgroup (what a statistics reduce reads): must be the value of this iteration
  const unsigned other = (blockIdx.x * 37u + 11u) % gridDim.x;
  return data[(size_t)other * kb * 256 + (threadIdx.x & 31)];
}

template <int FORM>  // 0 flat, 1 xcd
__global__ __launch_bounds__(256) void fused(Bar* b, float* data, int kb, int iters, unsigned* bad) {
  __shared__ unsigned s_xcc, s_non, s_nx;
  unsigned phase = 0;
  if (threadIdx.x == 0) {
    s_xcc = xcc_id() & 7;
    __hip_atomic_fetch_add(&b->census[s_xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  barrier_flat(b, gridDim.x, phase);  // census complete
  if (threadIdx.x == 0) {
    unsigned nx = 0;
    for (int x = 0; x < 8; ++x) nx += ld_relaxed(&b->census[x][0]) != 0;
    s_non = ld_relaxed(&b->census[s_xcc][0]);
    s_nx = nx;
  }
  __syncthreads();
  const unsigned xcc = s_xcc, n_on = s_non, nx = s_nx;
  unsigned xphase = 0;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    write_phase(data, kb, it);
    if (FORM == 0) barrier_flat(b, gridDim.x, phase);
    else barrier_xcd(b, xcc, n_on, nx, xphase);
    if (ld_relaxed(b->fail)) return;
    const float got = read_phase(data, kb);
    if (got != (float)(it + (int)((threadIdx.x & 31) >> 2)) && kb > 0) atomicAdd(bad, 1u);  // a stale read = a broken barrier (float4 i of the record was written by thread i)
    acc += got;
    // (the next iteration's writes must not overtake other workgroups' reads of this one: second barrier, not timed apart)
    if (FORM == 0) barrier_flat(b, gridDim.x, phase);
    else barrier_xcd(b, xcc, n_on, nx, xphase);
    if (ld_relaxed(b->fail)) return;
  }
  if (acc == 12345.f) data[0] = acc;
}

__global__ __launch_bounds__(256) void k_write(float* data, int kb, int it) { write_phase(data, kb, it); }
__global__ __launch_bounds__(256) void k_read(float* data, int kb, int it, unsigned* bad, float* sink) {
  const float got = read_phase(data, kb);
  if (got != (float)(it + (int)((threadIdx.x & 31) >> 2)) && kb > 0) atomicAdd(bad, 1u);
  if (got == 12345.f) sink[0] = got;
}

int main() {
  Bar* bar; float* data; unsigned* bad; float* sink;
  const size_t data_bytes = (size_t)1024 * 64 * 1024 + 4096;
  CK(hipMalloc(&bar, sizeof(Bar))); CK(hipMalloc(&data, data_bytes)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(data, 0, data_bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 200;
  printf("one iteration = write phase + dependency + read phase + dependency (two grid barriers, or two kernel boundaries):\n"
         "whole-iteration times, compare the forms at equal KB; at 0 KB half an iteration is the bare cost of one dependency\n");
  for (int blocks : {256, 512, 1024}) {
    for (int kb : {0, 4, 32, 64}) {
      float t[3] = {0, 0, 0};
      unsigned nbad[3] = {0, 0, 0}, nfail[2] = {0, 0};
      for (int form = 0; form < 2; ++form) {
        for (int rep = 0; rep < 2; ++rep) {  // rep 0 = warm-up
          CK(hipMemset(bar, 0, sizeof(Bar))); CK(hipMemset(bad, 0, 4));
          CK(hipEventRecord(e0));
          if (form == 0) hipLaunchKernelGGL(fused<0>, dim3(blocks), dim3(256), 0, 0, bar, data, kb, iters, bad);
          else hipLaunchKernelGGL(fused<1>, dim3(blocks), dim3(256), 0, 0, bar, data, kb, iters, bad);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          CK(hipEventElapsedTime(&t[form], e0, e1));
        }
        Bar h; CK(hipMemcpy(&h, bar, sizeof(Bar), hipMemcpyDeviceToHost));
        nfail[form] = h.fail[0];
        CK(hipMemcpy(&nbad[form], bad, 4, hipMemcpyDeviceToHost));
        if (blocks == 256 && kb == 0 && form == 1) {
          printf("census (workgroups per XCC at 256):");
          for (int x = 0; x < 8; ++x) printf(" %u", h.census[x][0]);
          printf("\n");
        }
      }
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(bad, 0, 4));
        CK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) {
          hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, data, kb, it);
          hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, data, kb, it, bad, sink);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t[2], e0, e1));
      }
      CK(hipMemcpy(&nbad[2], bad, 4, hipMemcpyDeviceToHost));
      printf("workgroups %4d, %2d KB written per workgroup (%5.1f MB): flat %6.2f us  xcd %6.2f us  two launches %6.2f us per iteration"
             "  [stale reads %u/%u/%u, spin budget exhausted %u/%u]\n",
             blocks, kb, blocks * kb / 1024.0, t[0] * 1e3 / iters, t[1] * 1e3 / iters, t[2] * 1e3 / iters, nbad[0], nbad[1], nbad[2],
             nfail[0], nfail[1]);
    }
  }
  return 0;
}
