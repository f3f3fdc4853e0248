// Winograd F(2x2, 3x3) forward convolution for the wide-enough stride-1 3x3 layers in train mode (fp32): the conv2d
// calls the reference dispatches inside torchvision's BasicBlock under segmentation_models_pytorch.Unet(resnet34)
// (d3f/train_denoiser/lit_module.py:46-52, :117) -- 2.25x fewer multiplies than the implicit GEMM of conv_igemm.hip.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A   per 2x2 output tile: 16 element-wise products, i.e. 16 independent
//   [tiles x Cin] x [Cin x Cout] contractions (the 16 Winograd POSITIONS) instead of 9 taps.
//
// One 256-thread workgroup = 64 output tiles (a 16 x 16 pixel block) x 64 filters, one workgroup per CU (all 512
// registers of a lane, 148 KB of LDS).  Per chunk of 16 input channels: the 18 x 18 pixel input patch is staged in LDS
// and transformed to V[16][64 tiles][16] (thread = one tile x four channels, packed adds; 64-byte rows, the 16-byte
// chunks XOR-swizzled); wave (mt, nt) contracts its 32 tiles x 32 filters block for ALL 16 positions -- 16 accumulators =
// 256 registers -- so that the output transform is register-local.  V is double-buffered: chunk c + 1 is transformed
// piece by piece behind the MFMAs of chunk c.  The filter fragments (U[16][Cin / 16][Cout][16], transformed from the fp32
// masters by winograd_pack_kernel in the weight-packing pass) never go through LDS: every lane loads its own 16 bytes per
// MFMA group from L2, four positions ahead of their use; the V fragments run one position ahead.
// Epilogue: y (NHWC fp32) and one (sum, sum of squares) partial row per workgroup for BatchNorm (fixed order), or --
// eval mode -- the folded BatchNorm (+ residual) (+ ReLU) of conv_igemm's fused epilogue.
// Measured stand-alone (profiles/microbench/winograd_f2x2_3x3.hip): 64 -> 64 @ 64x64 bs 16 in 30 us against the implicit
// GEMM's 47; relative error against float64 2.2e-7 (implicit GEMM 1.5e-7).  The plan takes it only where it gives at
// least one workgroup per CU; D3F_NO_WINOGRAD switches it off.
#include "common.h"

#include <cstdlib>

namespace d3f {

namespace {
constexpr int WCK = 16;   // channels per chunk
constexpr int WPP = 18;   // patch edge
constexpr int WPB = 4;    // positions the filter-fragment loads run ahead
typedef float wf32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int vsw(int row, int ch) { return ch ^ ((row >> 2) & 3); }
}  // namespace

// the shapes the kernel can run at all (the op-level entry d3f_conv_winograd_forward takes any of them)
bool conv_winograd_fits(const ConvParams& p, int dtype) {
  if (dtype != D3F_F32) return false;
  if (p.mode != CONV_RAW_STATS || p.par != 0 || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.C1 != 0 ||
      p.shift0 != 0 || p.zi != 0 || p.Ho != p.Hv || p.Wo != p.Wv || (p.Ho % 16) != 0 || (p.Wo % 16) != 0 ||
      (p.C0 % WCK) != 0 || (p.Cout % 64) != 0 || p.CoutPad < p.Cout)
    return false;
  // 32-bit byte offsets behind buffer descriptors whose out-of-bounds marker is 2^31 (as in wgrad_plan)
  if ((size_t)p.B * p.Hv * p.Wv * p.C0 * 4 >= (1ull << 31) || (size_t)16 * p.C0 * p.Cout * 4 >= (1ull << 31)) return false;
  return (long)p.B * (p.Ho / 16) * (p.Wo / 16) * (p.Cout / 64) <= 65535L * 4;
}
// the plan's selection rule: the shapes where it beats the implicit GEMM
bool conv_winograd_applies(const ConvParams& p, int dtype) {
  static const bool off = getenv("D3F_NO_WINOGRAD") != nullptr;  // debugging knob: the implicit GEMM instead
  if (off || !conv_winograd_fits(p, dtype)) return false;
  const long wgs = (long)p.B * (p.Ho / 16) * (p.Wo / 16) * (p.Cout / 64) * plan_nets_for(p.plan_nets, 8);
  // one workgroup per CU or more, and few enough chunks that the fixed cost per workgroup is what the tile form pays for
  // (128 channels on 128 workgroups: 44.9 us against the implicit GEMM's 45)
  return wgs >= 256;
}
int conv_winograd_stat_rows(const ConvParams& p) { return p.B * (p.Ho / 16) * (p.Wo / 16); }
size_t conv_winograd_filter_floats(const ConvParams& p) { return (size_t)16 * p.C0 * p.Cout; }

// U[pos = 4 i + j][c / 16][k][c % 16] = (G g G^T)[i][j],  g = w[k][c] (torch layout [K][C][3][3]),
// G = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]]
__global__ __launch_bounds__(256) void winograd_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int K, int C) {
  const int id = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (id >= K * C) return;
  const int k = id / C, c = id - k * C;
  const float* g = w + (long)id * 9;
  float t[4][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
    t[0][j] = g0;
    t[1][j] = 0.5f * (g0 + g1 + g2);
    t[2][j] = 0.5f * (g0 - g1 + g2);
    t[3][j] = g2;
  }
  const long cstride = (long)(C / WCK) * K * WCK;  // floats per position
  float* dst = u + ((long)(c / WCK) * K + k) * WCK + (c % WCK);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    dst[(4 * i + 0) * cstride] = t[i][0];
    dst[(4 * i + 1) * cstride] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
    dst[(4 * i + 2) * cstride] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
    dst[(4 * i + 3) * cstride] = t[i][2];
  }
}

int conv_winograd_pack_launch(const float* w, float* u, int K, int C, hipStream_t stream) {
  D3F_CHECK((C % WCK) == 0 && K > 0 && C > 0, "winograd pack: %d filters x %d channels", K, C);
  hipLaunchKernelGGL(winograd_pack_kernel, dim3((unsigned)cdiv((long)K * C, 256)), dim3(256), 0, stream, w, u, K, C);
  D3F_HIP(hipGetLastError());
  return 0;
}

// p.src0 [B][H][W][C] fp32, p.w = U (above), p.out0 = y [B][H][W][K] fp32, p.stats [workgroup rows][CoutPad][2] or null
__global__ __launch_bounds__(256) void conv_winograd_kernel(const ConvParams pin) {
  chain_priority();
  const ConvParams p = conv_params_of_net(pin, (int)blockIdx.z);  // two networks in one launch: blockIdx.z = net
  __shared__ __attribute__((aligned(16))) float P[WPP * WPP * WCK];
  __shared__ __attribute__((aligned(16))) float V[2][16 * 64 * WCK];
  const int H = p.Hv, W = p.Wv, C = p.C0, K = p.Cout;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mt = wave >> 1, nt = wave & 1;
  const int bx = W / 16, by = H / 16;
  const int blk = blockIdx.x;
  const int x0 = (blk % bx) * 16, y0 = ((blk / bx) % by) * 16, b = blk / (bx * by);
  const int n0 = blockIdx.y * 64;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.src0, p.src0_bytes);
  const __amdgpu_buffer_rsrc_t ru = make_rsrc(p.w, p.w_bytes);

  f32x16 acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  constexpr int NPX = (WPP * WPP * 4 + 255) / 256;
  uint4 px[NPX];
  unsigned poff[NPX];
#pragma unroll
  for (int i = 0; i < NPX; ++i) {
    const int id = tid + 256 * i, pix = id >> 2, h = id & 3;
    const int py = pix / WPP, pxx = pix - py * WPP;
    const int gy = y0 - 1 + py, gx = x0 - 1 + pxx;
    const bool ok = id < WPP * WPP * 4 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    poff[i] = ok ? (unsigned)((((b * H + gy) * W + gx) * C + 4 * h) * 4) : BUF_OOB;
  }
  // (a chunk past the last one reads the neighbouring pixel's channels or out of range: staged, never contracted)
  auto issue_patch = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NPX; ++i) px[i] = buf_load16(rx, poff[i] + (unsigned)c0 * 4u);
  };
  auto write_patch = [&]() {
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int id = tid + 256 * i;
      if (id < WPP * WPP * 4) *reinterpret_cast<uint4*>(&P[(id >> 2) * WCK + 4 * (id & 3)]) = px[i];
    }
  };
  const int fr = lane & 31, fq = lane >> 5;
  const int arow = 32 * mt + fr;
  const int aoff0 = arow * WCK + 4 * vsw(arow, fq), aoff1 = arow * WCK + 4 * vsw(arow, 2 + fq);
  const unsigned ubase = (unsigned)(((n0 + 32 * nt + fr) * WCK + 4 * fq) * 4);
  const unsigned upos = (unsigned)(K * C * 4), uchunk = (unsigned)(K * WCK * 4);
  const int nch = C / WCK;
  uint4 bq[WPB][2];
  auto issue_b = [&](int slot, int pos, int ch) {  // (ch == nch behind the last chunk: out of range -> zeros, unused)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
      bq[slot][s2] = buf_load16(ru, ch < nch ? ubase + (unsigned)pos * upos + (unsigned)ch * uchunk + (unsigned)(32 * s2) : BUF_OOB);
  };
  const int tt = tid >> 2, cq = tid & 3;
  const int ty = tt >> 3, tx = tt & 7;
  const float* pbase = P + ((2 * ty) * WPP + 2 * tx) * WCK + 4 * cq;
  const int voff = tt * WCK + 4 * vsw(tt, cq);

  auto sub = [](const float4& a, const float4& c) {  // two v_pk_add_f32 per float4
    const wf32x2 lo = wf32x2{a.x, a.y} - wf32x2{c.x, c.y}, hi = wf32x2{a.z, a.w} - wf32x2{c.z, c.w};
    return make_float4(lo.x, lo.y, hi.x, hi.y);
  };
  auto add = [](const float4& a, const float4& c) {
    const wf32x2 lo = wf32x2{a.x, a.y} + wf32x2{c.x, c.y}, hi = wf32x2{a.z, a.w} + wf32x2{c.z, c.w};
    return make_float4(lo.x, lo.y, hi.x, hi.y);
  };
  float4 t[4][4];
  auto tr_col = [&](int j) {  // patch column j of the 4 x 4 input tile -> t[.][j] = B^T d
    const float4 d0 = *reinterpret_cast<const float4*>(pbase + (0 * WPP + j) * WCK);
    const float4 d1 = *reinterpret_cast<const float4*>(pbase + (1 * WPP + j) * WCK);
    const float4 d2 = *reinterpret_cast<const float4*>(pbase + (2 * WPP + j) * WCK);
    const float4 d3 = *reinterpret_cast<const float4*>(pbase + (3 * WPP + j) * WCK);
    t[0][j] = sub(d0, d2);
    t[1][j] = add(d1, d2);
    t[2][j] = sub(d2, d1);
    t[3][j] = sub(d1, d3);
  };
  auto tr_row = [&](float* vb, int i) {  // row i of (B^T d) B -> positions 4 i .. 4 i + 3
    *reinterpret_cast<float4*>(vb + (4 * i + 0) * 64 * WCK + voff) = sub(t[i][0], t[i][2]);
    *reinterpret_cast<float4*>(vb + (4 * i + 1) * 64 * WCK + voff) = add(t[i][1], t[i][2]);
    *reinterpret_cast<float4*>(vb + (4 * i + 2) * 64 * WCK + voff) = sub(t[i][2], t[i][1]);
    *reinterpret_cast<float4*>(vb + (4 * i + 3) * 64 * WCK + voff) = sub(t[i][1], t[i][3]);
  };

  // prologue: chunk 0 transformed up front, chunk 1's patch in flight
  issue_patch(0);
#pragma unroll
  for (int q = 0; q < WPB; ++q) issue_b(q, q, 0);
  write_patch();
  __syncthreads();
  issue_patch(WCK);
#pragma unroll
  for (int j = 0; j < 4; ++j) tr_col(j);
#pragma unroll
  for (int i = 0; i < 4; ++i) tr_row(V[0], i);
  __syncthreads();

  for (int c = 0; c < nch; ++c) {
    const float* va = V[c & 1];
    float* vn = V[(c + 1) & 1];
    write_patch();          // chunk c + 1 (junk behind the last chunk: never contracted)
    __syncthreads();        // the patch is visible
    issue_patch((c + 2) * WCK);
    float4 av[2][2];
    av[0][0] = *reinterpret_cast<const float4*>(va + aoff0);
    av[0][1] = *reinterpret_cast<const float4*>(va + aoff1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const uint4 b0 = bq[q % WPB][0], b1 = bq[q % WPB][1];
      if (q + WPB < 16) issue_b(q % WPB, q + WPB, c);
      else issue_b(q % WPB, q + WPB - 16, c + 1);
      if (q + 1 < 16) {
        av[(q + 1) & 1][0] = *reinterpret_cast<const float4*>(va + (q + 1) * 64 * WCK + aoff0);
        av[(q + 1) & 1][1] = *reinterpret_cast<const float4*>(va + (q + 1) * 64 * WCK + aoff1);
      }
      // the next chunk's input transform, a piece per position, in the MFMAs' shadow
      if (q < 8 && (q & 1) == 0) tr_col(q >> 1);
      if (q >= 8 && (q & 1) == 0) tr_row(vn, (q - 8) >> 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const float4 a = av[q & 1][s2];
        const uint4 bb = s2 ? b1 : b0;
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, __uint_as_float(bb.x), acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, __uint_as_float(bb.y), acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, __uint_as_float(bb.z), acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, __uint_as_float(bb.w), acc[q], 0, 0, 0);
      }
    }
    __syncthreads();        // V[(c + 1) & 1] is complete, V[c & 1] and the patch are free
  }

  // output transform, register-local: register r of every accumulator = (tile 32 mt + m_l(r), filter 32 nt + fr)
  float* __restrict__ y = reinterpret_cast<float*>(p.out0);
  const int n = n0 + 32 * nt + fr;
  if (p.mode == CONV_EVAL_FUSED) {  // folded BatchNorm (+ residual) (+ ReLU), as conv_igemm's fused epilogue
    const float sc = p.scale[n], sf = p.shift[n];
    const float* __restrict__ res = reinterpret_cast<const float*>(p.res);
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
      float o[4] = {t0[0] + t0[1] + t0[2], t0[1] - t0[2] - t0[3], t1[0] + t1[1] + t1[2], t1[1] - t1[2] - t1[3]};
      const long base = (((long)b * H + oy) * W + ox) * K + n;
      const long off[4] = {0, K, (long)W * K, (long)W * K + K};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = o[e] * sc + sf;
        if (res != nullptr) v += res[base + off[e]];
        if (p.relu) v = fmaxf(v, 0.f);
        y[base + off[e]] = v;
      }
    }
    return;
  }
  float s1 = 0.f, s2 = 0.f;
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
    const float o00 = t0[0] + t0[1] + t0[2], o01 = t0[1] - t0[2] - t0[3];
    const float o10 = t1[0] + t1[1] + t1[2], o11 = t1[1] - t1[2] - t1[3];
    float* o = y + (((long)b * H + oy) * W + ox) * K + n;
    o[0] = o00;
    o[K] = o01;
    o[(long)W * K] = o10;
    o[(long)W * K + K] = o11;
    s1 += (o00 + o01) + (o10 + o11);
    s2 += (o00 * o00 + o01 * o01) + (o10 * o10 + o11 * o11);
  }
  if (p.stats != nullptr) {
    // per-channel sums of the workgroup's 256 pixels: the two k halves of a wave (lanes l, l + 32 hold the same filter),
    // then the two tile halves (waves mt = 0, 1) through LDS, in that order
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    float* red = P;  // [mt][64 filters][2]
    if (fq == 0) {
      red[((mt * 64) + 32 * nt + fr) * 2 + 0] = s1;
      red[((mt * 64) + 32 * nt + fr) * 2 + 1] = s2;
    }
    __syncthreads();
    if (tid < 64) {
      float* st = p.stats + ((long)blk * p.CoutPad + n0 + tid) * 2;
      st[0] = red[tid * 2 + 0] + red[(64 + tid) * 2 + 0];
      st[1] = red[tid * 2 + 1] + red[(64 + tid) * 2 + 1];
    }
  }
}

int conv_winograd_launch(const ConvParams& p, hipStream_t stream) {
  D3F_CHECK(p.src0 && p.w && p.out0 && (p.Hv % 16) == 0 && (p.Wv % 16) == 0 && (p.C0 % WCK) == 0 && (p.Cout % 64) == 0 &&
                (p.mode == CONV_RAW_STATS || (p.mode == CONV_EVAL_FUSED && p.scale && p.shift)),
            "winograd conv: bad description");
  D3F_CHECK((size_t)p.B * p.Hv * p.Wv * p.C0 * 4 < (1ull << 31) && (size_t)16 * p.C0 * p.Cout * 4 < (1ull << 31),
            "winograd conv: tensor beyond the 2 GiB reach of its 32-bit offsets");
  ConvParams q = p;
  q.src0_bytes = (unsigned)((size_t)p.B * p.Hv * p.Wv * p.C0 * 4);
  q.w_bytes = (unsigned)((size_t)16 * p.C0 * p.Cout * 4);
  const dim3 grid((unsigned)(p.B * (p.Hv / 16) * (p.Wv / 16)), (unsigned)(p.Cout / 64), (unsigned)nets_of(p.nets));
  const bool prof = prof_enabled(PROF_CONV_FWD);
  if (prof) prof_begin(PROF_CONV_FWD, q.flops, stream);  // the algorithmic (direct) FLOP count of the layer
  hipLaunchKernelGGL(conv_winograd_kernel, grid, dim3(256), 0, stream, q);
  if (prof) prof_end(stream);
  D3F_HIP(hipGetLastError());
  return 0;
}

}  // namespace d3f
