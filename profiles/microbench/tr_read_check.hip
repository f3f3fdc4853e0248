// checks the ds_read_b64_tr_b16 addressing used for the k-strided MFMA operand of the f32x3 weight gradient
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef short v4s __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
// LDS image: [k = 32 pixels][32 channels] 16-bit, rows of 64 bytes.  value(k, c) = k * 100 + c
__global__ void k1(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short img[32 * 32];
  for (int i = threadIdx.x; i < 32 * 32; i += 64) img[i] = (unsigned short)((i / 32) * 100 + (i % 32));
  __syncthreads();
  const int l = threadIdx.x, grp = l >> 4, q = (l & 15) >> 2, p = l & 3;
  const int cbase = (grp & 1) * 16, kg = grp >> 1;
  for (int s = 0; s < 2; ++s) {
    unsigned short frag[8];
    for (int h = 0; h < 2; ++h) {
      const int row = s * 16 + kg * 8 + h * 4 + q;
      const unsigned short* addr = &img[row * 32 + cbase + 4 * p];
      v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)addr);
      for (int j = 0; j < 4; ++j) frag[h * 4 + j] = (unsigned short)r[j];
    }
    for (int j = 0; j < 8; ++j) out[(s * 64 + l) * 8 + j] = frag[j];
  }
}
int main() {
  unsigned short* d; CK(hipMalloc(&d, 2 * 64 * 8 * 2));
  hipLaunchKernelGGL(k1, dim3(1), dim3(64), 0, 0, d);
  std::vector<unsigned short> h(2 * 64 * 8);
  CK(hipMemcpy(h.data(), d, h.size() * 2, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int s = 0; s < 2; ++s) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
    // MFMA 32x32x16 operand: lane l holds row (channel) l % 32, k = (l / 32) * 8 + j  of this 16-wide k step
    const int c = l % 32, k = s * 16 + (l / 32) * 8 + j;
    const int want = k * 100 + c, got = h[(s * 64 + l) * 8 + j];
    if (want != got && bad++ < 10) printf("s %d lane %d j %d: want %d got %d\n", s, l, j, want, got);
  }
  printf("%s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
