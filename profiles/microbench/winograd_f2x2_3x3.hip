// Spike for the next round: fused Winograd F(2x2, 3x3) forward convolution, fp32 MFMA,
// NHWC, stride 1, pad 1 -- how fast can the layer1-type layers (64 -> 64 channels at 64 x 64, bs 16: 47 us as a direct
// implicit GEMM, 4.83 GFLOP) run with 2.25x fewer multiplies?          hipcc --offload-arch=gfx950 -O3 winograd_f2x2_3x3.hip
//
// One 256-thread workgroup = 64 2x2-output tiles (a 16 x 16 pixel block) x 64 filters, one workgroup per CU (all 512
// registers per lane, 148 KB of LDS).  Per chunk of 16 input channels:
//   stage the 18 x 18 pixel input patch in LDS; transform it to V[16][64 tiles][16] (thread = one tile x four channels,
//   packed adds, 16-byte chunks XOR-swizzled instead of padded); wave (mt, nt) runs its 32 tiles x 32 filters block for
//   ALL 16 Winograd positions: 16 accumulators = 256 registers, so that the output transform A^T M A is register-local.
//   V is double-buffered: chunk c + 1 is transformed piece by piece behind the MFMAs of chunk c.  The filter fragments
//   (U[16][C / 16][K][16], pre-transformed on the host) never pass through LDS: every lane loads its own 16 bytes per
//   MFMA group from L2, four positions ahead of their use; the A fragments run one position ahead.
//
// Results on MI355X (r03; every version checked against an fp64-accumulated direct convolution: rel. L2 error 2.2e-7 at
// 64 channels, 3.0e-7 at 128, 4.2e-7 at 256 -- the CPU study's figures):
//   64 -> 64 @ 64x64 bs 16 (layer1 type, 256 workgroups)   30.0 us = 161 direct-equivalent TFLOP/s  (direct kernel: 47 us)
//   64 -> 64 @ 128x128 bs 16 (1024 workgroups)            116.8 us = 165 TFLOP/s-equivalent
//   128 -> 128 @ 32x32 bs 16 (128 workgroups: half the CUs) 44.9 us (direct: 45 us) -- needs a 32-tile block variant
//   one workgroup alone (8 workgroups on the chip): 23.9 us for 4 chunks = ~19 us in the kernel against a 14-16 us
//   MFMA floor: the k-loop is at the floor, ~4.5 us are prologue (first loads, first transform) and epilogue.
//   History: v1 (U through LDS, 8-channel chunks, no overlap) 35.4 us; v2 (U from L2) 34.0; v3 (A fragments one position
//   ahead, packed adds) 30.4; v4 / v5 (double-buffered V, transform in the MFMAs' shadow; contiguous U fragments) 30.0 --
//   at 64 channels the fixed cost per workgroup, not the loop, is what is left.
//   Per 16-channel chunk a lone workgroup takes 4.85 us (256 channels: 77.6 us for 16 chunks) = 70-80 % of the MFMA rate;
//   NOPATCH=1 (timing only: the input loads return at once) 30.4 -> 27.6 us.
//   VARIANT=B (winograd_fwd32: 32 tiles x 64 filters, a wave owns HALF of the positions, two workgroups per CU, the
//   partial output transforms of the two halves added through LDS): 64 -> 64 @ 64x64 36.4 us (512 workgroups: worse, the
//   co-resident workgroups share the matrix pipe and repeat the halo), 128 -> 128 @ 32x32 37.5 us (256 workgroups; direct
//   45 us), 256 -> 256 @ 16x16 61.5 us on 128 workgroups (direct 45 us): the deep layers need the positions or the
//   channels spread over workgroups, not smaller tiles.
// Usage: ./a.out [C=64] [K=64] [H=64] [W=64] [B=16]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int CK = 16;           // channels per chunk
constexpr int PP = 18;           // patch edge
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load16(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}

// x [B][H][W][C], u [16][C / 16][K][16] (transformed filters), y [B][H][W][K]
constexpr int PB = 4;  // positions the filter-fragment loads run ahead
// V rows are 16 floats, unpadded; the 16-byte chunks of a row are XOR-swizzled with (row >> 2) & 3 so that the fragment
// reads of a 16-lane group (rows r0 .. r15 of one chunk column) cover all 64 banks once
__device__ __forceinline__ int vsw(int row, int ch) { return ch ^ ((row >> 2) & 3); }

__global__ __launch_bounds__(256) void winograd_fwd(const float* __restrict__ x, const float* __restrict__ u, float* __restrict__ y,
                                                    int B, int H, int W, int C, int K, int nopatch) {
  __shared__ __attribute__((aligned(16))) float P[PP * PP * CK];
  __shared__ __attribute__((aligned(16))) float V[2][16 * 64 * CK];  // two chunks: chunk c+1 is transformed while c is contracted
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mt = wave >> 1, nt = wave & 1;
  const int bx = W / 16, by = H / 16;
  const int blk = blockIdx.x;
  const int x0 = (blk % bx) * 16, y0 = ((blk / bx) % by) * 16, b = blk / (bx * by);
  const int n0 = blockIdx.y * 64;
  const __amdgpu_buffer_rsrc_t rx = rsrc(x, nopatch ? 0u : (unsigned)((size_t)B * H * W * C * 4));  // NOPATCH=1: timing only
  const __amdgpu_buffer_rsrc_t ru = rsrc(u, (unsigned)((size_t)16 * K * C * 4));

  f32x16 acc[16];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

  constexpr int NPX = (PP * PP * 4 + 255) / 256;
  uint4 px[NPX];
  unsigned poff[NPX];
#pragma unroll
  for (int i = 0; i < NPX; ++i) {
    const int id = tid + 256 * i, pix = id >> 2, h = id & 3;
    const int py = pix / PP, pxx = pix - py * PP;
    const int gy = y0 - 1 + py, gx = x0 - 1 + pxx;
    const bool ok = id < PP * PP * 4 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    poff[i] = ok ? (unsigned)((((b * H + gy) * W + gx) * C + 4 * h) * 4) : OOB;
  }
  auto issue_patch = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NPX; ++i) px[i] = load16(rx, poff[i] + (unsigned)c0 * 4u);
  };
  auto write_patch = [&]() {
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int id = tid + 256 * i;
      if (id < PP * PP * 4) *reinterpret_cast<uint4*>(&P[(id >> 2) * CK + 4 * (id & 3)]) = px[i];
    }
  };
  const int fr = lane & 31, fq = lane >> 5;
  const int arow = 32 * mt + fr;
  // byte-free form: float offsets of this lane's two A chunks (s2 = 0, 1) inside a position's [64][16] block
  const int aoff0 = arow * CK + 4 * vsw(arow, fq), aoff1 = arow * CK + 4 * vsw(arow, 2 + fq);
  // U is stored [pos][C / 16][K][16]: the 32 filters x 16 channels a wave contracts per position and chunk are 2 KB
  // contiguous (as [pos][K][C] every load instruction touched 32 cache lines for 1 KB of payload)
  const unsigned ubase = (unsigned)(((n0 + 32 * nt + fr) * CK + 4 * fq) * 4);
  const unsigned upos = (unsigned)(K * C * 4), uchunk = (unsigned)(K * CK * 4);
  uint4 bq[PB][2];
  auto issue_b = [&](int slot, int pos, int c0) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
      bq[slot][s2] = load16(ru, ubase + (unsigned)pos * upos + (unsigned)(c0 / CK) * uchunk + (unsigned)(8 * s2) * 4u);
  };
  const int tt = tid >> 2, cq = tid & 3;
  const int ty = tt >> 3, tx = tt & 7;
  const float* pbase = P + ((2 * ty) * PP + 2 * tx) * CK + 4 * cq;
  const int voff = tt * CK + 4 * vsw(tt, cq);

  auto sub = [](const float4& a, const float4& c) {
    const f32x2 lo = f32x2{a.x, a.y} - f32x2{c.x, c.y}, hi = f32x2{a.z, a.w} - f32x2{c.z, c.w};
    return make_float4(lo.x, lo.y, hi.x, hi.y);
  };
  auto add = [](const float4& a, const float4& c) {
    const f32x2 lo = f32x2{a.x, a.y} + f32x2{c.x, c.y}, hi = f32x2{a.z, a.w} + f32x2{c.z, c.w};
    return make_float4(lo.x, lo.y, hi.x, hi.y);
  };
  float4 t[4][4];
  auto tr_col = [&](int j) {  // patch column j of the 4 x 4 input tile -> t[.][j] = B^T d
    const float4 d0 = *reinterpret_cast<const float4*>(pbase + (0 * PP + j) * CK);
    const float4 d1 = *reinterpret_cast<const float4*>(pbase + (1 * PP + j) * CK);
    const float4 d2 = *reinterpret_cast<const float4*>(pbase + (2 * PP + j) * CK);
    const float4 d3 = *reinterpret_cast<const float4*>(pbase + (3 * PP + j) * CK);
    t[0][j] = sub(d0, d2);
    t[1][j] = add(d1, d2);
    t[2][j] = sub(d2, d1);
    t[3][j] = sub(d1, d3);
  };
  auto tr_row = [&](float* vb, int i) {  // row i of (B^T d) B -> positions 4 i .. 4 i + 3
    *reinterpret_cast<float4*>(vb + (4 * i + 0) * 64 * CK + voff) = sub(t[i][0], t[i][2]);
    *reinterpret_cast<float4*>(vb + (4 * i + 1) * 64 * CK + voff) = add(t[i][1], t[i][2]);
    *reinterpret_cast<float4*>(vb + (4 * i + 2) * 64 * CK + voff) = sub(t[i][2], t[i][1]);
    *reinterpret_cast<float4*>(vb + (4 * i + 3) * 64 * CK + voff) = sub(t[i][1], t[i][3]);
  };

  // prologue: chunk 0 transformed up front, chunk 1's patch in flight
  issue_patch(0);
#pragma unroll
  for (int q = 0; q < PB; ++q) issue_b(q, q, 0);
  write_patch();
  __syncthreads();
  issue_patch(CK);
#pragma unroll
  for (int j = 0; j < 4; ++j) tr_col(j);
#pragma unroll
  for (int i = 0; i < 4; ++i) tr_row(V[0], i);
  __syncthreads();

  const int nch = C / CK;
  for (int c = 0; c < nch; ++c) {
    const int c0 = c * CK;
    const float* va = V[c & 1];
    float* vn = V[(c + 1) & 1];
    write_patch();          // chunk c + 1 (junk behind the last chunk: never read)
    __syncthreads();        // the patch is visible
    issue_patch(c0 + 2 * CK);
    float4 av[2][2];
    av[0][0] = *reinterpret_cast<const float4*>(va + aoff0);
    av[0][1] = *reinterpret_cast<const float4*>(va + aoff1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const uint4 b0 = bq[p % PB][0], b1 = bq[p % PB][1];
      if (p + PB < 16) issue_b(p % PB, p + PB, c0);
      else issue_b(p % PB, p + PB - 16, c0 + CK);
      if (p + 1 < 16) {
        av[(p + 1) & 1][0] = *reinterpret_cast<const float4*>(va + (p + 1) * 64 * CK + aoff0);
        av[(p + 1) & 1][1] = *reinterpret_cast<const float4*>(va + (p + 1) * 64 * CK + aoff1);
      }
      // the next chunk's input transform, a piece per position, in the MFMAs' shadow
      if (p < 8 && (p & 1) == 0) tr_col(p >> 1);
      if (p >= 8 && (p & 1) == 0) tr_row(vn, (p - 8) >> 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const float4 a = av[p & 1][s2];
        const uint4 bb = s2 ? b1 : b0;
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, __uint_as_float(bb.x), acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, __uint_as_float(bb.y), acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, __uint_as_float(bb.z), acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, __uint_as_float(bb.w), acc[p], 0, 0, 0);
      }
    }
    __syncthreads();        // V[(c + 1) & 1] is complete, V[c & 1] and the patch are free
  }
  const int n = n0 + 32 * nt + fr;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int tl = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * fq;
    const int oy = y0 + 2 * (tl >> 3), ox = x0 + 2 * (tl & 7);
    float t0[4], t1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t0[j] = acc[0 + j][r] + acc[4 + j][r] + acc[8 + j][r];
      t1[j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
    }
    float* o = y + (((long)b * H + oy) * W + ox) * K + n;
    o[0] = t0[0] + t0[1] + t0[2];
    o[K] = t0[1] - t0[2] - t0[3];
    o[(long)W * K] = t1[0] + t1[1] + t1[2];
    o[(long)W * K + K] = t1[1] - t1[2] - t1[3];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Variant B: 32 output tiles (8 x 16 pixels) x 64 filters per workgroup, wave = (filter half nt, POSITION half ph): 8
// accumulators = 128 registers, 75.5 KB of LDS -> TWO workgroups per CU, so that one workgroup's prologue / epilogue
// runs under the other's MFMAs.  The output transform needs both position halves: it is linear, so every wave
// transforms its own half to four partial outputs per (tile, filter) and the ph = 1 waves hand theirs over through LDS.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PR2 = 10, PC2 = 18;  // patch of an 8 x 16 pixel block
__global__ __launch_bounds__(256, 2) void winograd_fwd32(const float* __restrict__ x, const float* __restrict__ u,
                                                         float* __restrict__ y, int B, int H, int W, int C, int K) {
  __shared__ __attribute__((aligned(16))) float P[PR2 * PC2 * CK];
  __shared__ __attribute__((aligned(16))) float V[2][16 * 32 * CK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = wave & 1, ph = wave >> 1;
  const int bx = W / 16, by = H / 8;
  const int blk = blockIdx.x;
  const int x0 = (blk % bx) * 16, y0 = ((blk / bx) % by) * 8, b = blk / (bx * by);
  const int n0 = blockIdx.y * 64;
  const __amdgpu_buffer_rsrc_t rx = rsrc(x, (unsigned)((size_t)B * H * W * C * 4));
  const __amdgpu_buffer_rsrc_t ru = rsrc(u, (unsigned)((size_t)16 * K * C * 4));

  f32x16 acc[8];
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

  constexpr int NPX = (PR2 * PC2 * 4 + 255) / 256;  // 3
  uint4 px[NPX];
  unsigned poff[NPX];
#pragma unroll
  for (int i = 0; i < NPX; ++i) {
    const int id = tid + 256 * i, pix = id >> 2, h = id & 3;
    const int py = pix / PC2, pxx = pix - py * PC2;
    const int gy = y0 - 1 + py, gx = x0 - 1 + pxx;
    const bool ok = id < PR2 * PC2 * 4 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    poff[i] = ok ? (unsigned)((((b * H + gy) * W + gx) * C + 4 * h) * 4) : OOB;
  }
  auto issue_patch = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NPX; ++i) px[i] = load16(rx, poff[i] + (unsigned)c0 * 4u);
  };
  auto write_patch = [&]() {
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int id = tid + 256 * i;
      if (id < PR2 * PC2 * 4) *reinterpret_cast<uint4*>(&P[(id >> 2) * CK + 4 * (id & 3)]) = px[i];
    }
  };
  const int fr = lane & 31, fq = lane >> 5;
  const int aoff0 = fr * CK + 4 * vsw(fr, fq), aoff1 = fr * CK + 4 * vsw(fr, 2 + fq);
  const unsigned ubase = (unsigned)(((n0 + 32 * nt + fr) * CK + 4 * fq) * 4);
  const unsigned upos = (unsigned)(K * C * 4), uchunk = (unsigned)(K * CK * 4);
  uint4 bq[PB][2];
  auto issue_b = [&](int slot, int pos, int c0) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
      bq[slot][s2] = load16(ru, ubase + (unsigned)pos * upos + (unsigned)(c0 / CK) * uchunk + (unsigned)(8 * s2) * 4u);
  };
  // transform role: tile tt (0..31), channel quad cq, output-row half hf (rows 2 hf, 2 hf + 1 of B^T d B)
  const int tt = tid >> 3, cq = (tid >> 1) & 3, hf = tid & 1;
  const int ty = tt >> 3, tx = tt & 7;
  const float* pbase = P + ((2 * ty) * PC2 + 2 * tx) * CK + 4 * cq;
  const int voff = tt * CK + 4 * vsw(tt, cq);
  auto sub = [](const float4& a, const float4& c) {
    const f32x2 lo = f32x2{a.x, a.y} - f32x2{c.x, c.y}, hi = f32x2{a.z, a.w} - f32x2{c.z, c.w};
    return make_float4(lo.x, lo.y, hi.x, hi.y);
  };
  auto add = [](const float4& a, const float4& c) {
    const f32x2 lo = f32x2{a.x, a.y} + f32x2{c.x, c.y}, hi = f32x2{a.z, a.w} + f32x2{c.z, c.w};
    return make_float4(lo.x, lo.y, hi.x, hi.y);
  };
  float4 t[2][4];
  auto tr_col = [&](int j) {  // t[0] = row 2 hf, t[1] = row 2 hf + 1 of B^T d
    const float4 d1 = *reinterpret_cast<const float4*>(pbase + (1 * PC2 + j) * CK);
    const float4 d2 = *reinterpret_cast<const float4*>(pbase + (2 * PC2 + j) * CK);
    const float4 de = *reinterpret_cast<const float4*>(pbase + ((hf ? 3 : 0) * PC2 + j) * CK);  // d0 or d3
    t[0][j] = hf ? sub(d2, d1) : sub(de, d2);
    t[1][j] = hf ? sub(d1, de) : add(d1, d2);
  };
  auto tr_row = [&](float* vb, int i) {  // i = 0, 1: global row 2 hf + i -> positions 4 (2 hf + i) ..
    float* dst = vb + (4 * (2 * hf + i)) * 32 * CK + voff;
    *reinterpret_cast<float4*>(dst + 0 * 32 * CK) = sub(t[i][0], t[i][2]);
    *reinterpret_cast<float4*>(dst + 1 * 32 * CK) = add(t[i][1], t[i][2]);
    *reinterpret_cast<float4*>(dst + 2 * 32 * CK) = sub(t[i][2], t[i][1]);
    *reinterpret_cast<float4*>(dst + 3 * 32 * CK) = sub(t[i][1], t[i][3]);
  };

  issue_patch(0);
#pragma unroll
  for (int q = 0; q < PB; ++q) issue_b(q, 8 * ph + q, 0);
  write_patch();
  __syncthreads();
  issue_patch(CK);
#pragma unroll
  for (int j = 0; j < 4; ++j) tr_col(j);
  tr_row(V[0], 0);
  tr_row(V[0], 1);
  __syncthreads();

  const int nch = C / CK;
  for (int c = 0; c < nch; ++c) {
    const int c0 = c * CK;
    const float* va = V[c & 1] + (8 * ph) * 32 * CK;
    float* vn = V[(c + 1) & 1];
    write_patch();
    __syncthreads();
    issue_patch(c0 + 2 * CK);
    float4 av[2][2];
    av[0][0] = *reinterpret_cast<const float4*>(va + aoff0);
    av[0][1] = *reinterpret_cast<const float4*>(va + aoff1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const uint4 b0 = bq[p % PB][0], b1 = bq[p % PB][1];
      if (p + PB < 8) issue_b(p % PB, 8 * ph + p + PB, c0);
      else issue_b(p % PB, 8 * ph + p + PB - 8, c0 + CK);
      if (p + 1 < 8) {
        av[(p + 1) & 1][0] = *reinterpret_cast<const float4*>(va + (p + 1) * 32 * CK + aoff0);
        av[(p + 1) & 1][1] = *reinterpret_cast<const float4*>(va + (p + 1) * 32 * CK + aoff1);
      }
      if (p < 4) tr_col(p);
      if (p == 5) tr_row(vn, 0);
      if (p == 7) tr_row(vn, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const float4 a = av[p & 1][s2];
        const uint4 bb = s2 ? b1 : b0;
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, __uint_as_float(bb.x), acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, __uint_as_float(bb.y), acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, __uint_as_float(bb.z), acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, __uint_as_float(bb.w), acc[p], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // partial output transform of this wave's position half: acc[4 a + j] = M[2 ph + a][j]
  //   ph 0: t0 = M0 + M1, t1 = M1;   ph 1: t0 = M2, t1 = -M2 - M3      (Y rows = t0 (+) / t1 (+) of both halves)
  float* X = &V[0][0];  // exchange area: [nt][r][4][64 lanes]
  const int n = n0 + 32 * nt + fr;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float t0[4], t1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t0[j] = ph ? acc[j][r] : acc[j][r] + acc[4 + j][r];
      t1[j] = ph ? -acc[j][r] - acc[4 + j][r] : acc[4 + j][r];
    }
    const float y00 = t0[0] + t0[1] + t0[2], y01 = t0[1] - t0[2] - t0[3];
    const float y10 = t1[0] + t1[1] + t1[2], y11 = t1[1] - t1[2] - t1[3];
    if (ph) {
      float* e = X + ((nt * 16 + r) * 4) * 64 + lane;
      e[0] = y00; e[64] = y01; e[128] = y10; e[192] = y11;
    }
    // keep ph 0's values in the accumulator registers it no longer needs
    if (!ph) { acc[0][r] = y00; acc[1][r] = y01; acc[2][r] = y10; acc[3][r] = y11; }
  }
  __syncthreads();
  if (!ph) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int tl = (r & 3) + 8 * (r >> 2) + 4 * fq;
      const int oy = y0 + 2 * (tl >> 3), ox = x0 + 2 * (tl & 7);
      const float* e = X + ((nt * 16 + r) * 4) * 64 + lane;
      float* o = y + (((long)b * H + oy) * W + ox) * K + n;
      o[0] = acc[0][r] + e[0];
      o[K] = acc[1][r] + e[64];
      o[(long)W * K] = acc[2][r] + e[128];
      o[(long)W * K + K] = acc[3][r] + e[192];
    }
  }
}

__global__ void direct_ref(const float* x, const float* w, float* y, int B, int H, int W, int C, int K) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * H * W * K) return;
  const int k = i % K;
  long t = i / K;
  const int ox = t % W; t /= W;
  const int oy = t % H;
  const int b = t / H;
  double s = 0.0;
  for (int kh = 0; kh < 3; ++kh)
    for (int kw = 0; kw < 3; ++kw) {
      const int iy = oy + kh - 1, ix = ox + kw - 1;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      const float* xp = x + (((long)b * H + iy) * W + ix) * C;
      const float* wp = w + ((long)k * 9 + kh * 3 + kw) * C;   // w [K][3][3][C]
      for (int c = 0; c < C; ++c) s += (double)xp[c] * wp[c];
    }
  y[i] = (float)s;
}

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 64, K = argc > 2 ? atoi(argv[2]) : 64;
  const int H = argc > 3 ? atoi(argv[3]) : 64, W = argc > 4 ? atoi(argv[4]) : 64, B = argc > 5 ? atoi(argv[5]) : 16;
  if (C % CK || K % 64 || H % 16 || W % 16) { printf("shape\n"); return 1; }
  const size_t nx = (size_t)B * H * W * C, nw = (size_t)K * 9 * C, ny = (size_t)B * H * W * K;
  std::vector<float> hx(nx), hw(nw), hu((size_t)16 * K * C);
  srand(1);
  for (auto& v : hx) { const float r = (float)rand() / RAND_MAX * 2.f - 1.f; v = r > 0 ? r * 1.7f : 0.f; }   // post-ReLU like
  const float ws = 1.f / sqrtf(9.f * C);
  for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 1.7f * ws;
  // U = G g G^T per (k, c); G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
  const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  for (int k = 0; k < K; ++k)
    for (int c = 0; c < C; ++c) {
      double g[3][3], t[4][3];
      for (int a = 0; a < 3; ++a) for (int bq = 0; bq < 3; ++bq) g[a][bq] = hw[((size_t)k * 9 + a * 3 + bq) * C + c];
      for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0][j] + G[i][1] * g[1][j] + G[i][2] * g[2][j];
      for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j)
        hu[(((size_t)(4 * i + j) * (C / CK) + c / CK) * K + k) * CK + c % CK] =
            (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
    }
  float *dx, *dw, *du, *dy, *dr;
  CHECK(hipMalloc(&dx, nx * 4)); CHECK(hipMalloc(&dw, nw * 4)); CHECK(hipMalloc(&du, hu.size() * 4));
  CHECK(hipMalloc(&dy, ny * 4)); CHECK(hipMalloc(&dr, ny * 4));
  CHECK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(du, hu.data(), hu.size() * 4, hipMemcpyHostToDevice));
  const bool vb = getenv("VARIANT") != nullptr && getenv("VARIANT")[0] == 'B';  // 32-tile workgroups, two per CU
  const dim3 grid((unsigned)(B * (H / (vb ? 8 : 16)) * (W / 16)), (unsigned)(K / 64));
  auto launch = [&]() {
    if (vb) hipLaunchKernelGGL(winograd_fwd32, grid, dim3(256), 0, 0, dx, du, dy, B, H, W, C, K);
    else hipLaunchKernelGGL(winograd_fwd, grid, dim3(256), 0, 0, dx, du, dy, B, H, W, C, K, getenv("NOPATCH") ? 1 : 0);
  };
  hipLaunchKernelGGL(direct_ref, dim3((unsigned)((ny + 255) / 256)), dim3(256), 0, 0, dx, dw, dr, B, H, W, C, K);
  launch();
  CHECK(hipDeviceSynchronize());
  std::vector<float> hy(ny), hr(ny);
  CHECK(hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hr.data(), dr, ny * 4, hipMemcpyDeviceToHost));
  double num = 0, den = 0, mx = 0;
  for (size_t i = 0; i < ny; ++i) { const double e = (double)hy[i] - hr[i]; num += e * e; den += (double)hr[i] * hr[i]; mx = fmax(mx, fabs(e)); }
  printf("C=%d K=%d %dx%d B=%d: rel L2 error vs fp64-accumulated direct conv %.3e, max abs %.3e\n", C, K, H, W, B, sqrt(num / den), mx);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) launch();
  CHECK(hipEventRecord(e0, 0));
  const int N = 50;
  for (int i = 0; i < N; ++i) launch();
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / N, gf = 2.0 * B * H * W * (double)K * 9 * C * 1e-9;
  printf("%s: %.1f us per launch (%u x %u workgroups) = %.1f direct-equivalent TFLOP/s (%.2f GFLOP direct, %.2f executed)\n",
         vb ? "winograd_fwd32" : "winograd_fwd", us, grid.x, grid.y, gf / us * 1e3, gf, gf / 2.25);
  return 0;
}
