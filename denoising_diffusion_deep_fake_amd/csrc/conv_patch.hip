// Patch convolution for the full-resolution narrow layers (3x3, stride 1, pad 1, 16 input channels, <= 16 filters):
// forward and data gradient of the decoder's last block and of the segmentation head -- the conv2d calls the
// reference dispatches under segmentation_models_pytorch.Unet's decoder.blocks.4 / segmentation_head
// (d3f/train_denoiser/lit_module.py:46-52, :117).
//
// Why a second kernel: with 16 channels a 128-byte k-row of the implicit GEMM (conv_igemm.hip) spans two filter taps,
// so its loader decodes a tap per lane (~20 vector-ALU instructions per gathered 16-byte vector) and gathers every
// input pixel nine times; next to 0.4 us of MFMA work per k-tile that made those launches instruction-bound at
// ~40 TFLOP/s.  Here a workgroup stages its input patch ONCE -- a 4 x 64 pixel output tile + halo, 6 x 66 pixels x 16
// channels -- and the whole packed weight matrix (16 x 144) in LDS, and the k-loop is nothing but ds_read_b128 +
// v_mfma_f32_16x16x4_f32 with compile-time tap offsets: no address arithmetic, no barrier, no k padding (144, not 160).
//
//   wave w = output row w of the tile (64 pixels = 4 fragments of 16); per tap: 4 A reads + 1 B read, 16 MFMAs
//   LDS: patch [6][66][16 + 4] f32 (31.7 KB) + weights [16][144] f32 (9.2 KB) = 40 896 B -> 4 workgroups per CU; the
//   pixel pad of 4 dwords / the XOR swizzle of the weight chunks make both fragment reads (8 consecutive rows x 16
//   bytes per pass) bank-conflict free
//   epilogue: the C tile goes through LDS (aliasing the patch) and out as 16-byte vectors, as in conv_igemm.hip
#include "common.h"

#include <cstdlib>

namespace d3f {

constexpr int CP_PH = 4, CP_PW = 64;  // output tile (rows x columns); 256 rows of the implicit GEMM

bool conv_patch_applies(const ConvParams& p, int dtype) {
  static const bool off = getenv("D3F_NO_PATCH_CONV") != nullptr;  // debugging knob: the implicit-GEMM path instead
  if (off || dtype != D3F_F32) return false;
  const bool mode_ok = p.mode == CONV_RAW_STATS || p.mode == CONV_HEAD_NCHW ||
                       (p.mode == CONV_EVAL_FUSED && p.res == nullptr) ||
                       (p.mode == CONV_DGRAD && p.out_c0 == p.Cout && (p.Cout % 4) == 0);
  return mode_ok && p.par == 0 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.C0 == 16 && p.C1 == 0 &&
         p.shift0 == 0 && p.zi == 0 && p.Cout <= 16 && (p.mode == CONV_HEAD_NCHW || (p.Cout % 4) == 0) &&
         p.Hv == p.Ho && p.Wv == p.Wo && (p.Ho % CP_PH) == 0 &&
         (p.Wo % CP_PW) == 0 && p.Kpad >= 9 * 16;
}

template <int CIN, int BN>
__global__ __launch_bounds__(256) void conv_patch_kernel(const ConvParams p) {
  static_assert(CIN == 16 && BN == 16, "one configuration so far");
  constexpr int PH = CP_PH, PW = CP_PW, PR = PH + 2, PC = PW + 2;
  constexpr int CS = CIN + 4;        // dwords per staged pixel
  constexpr int KR = 9 * CIN;        // real k extent
  constexpr int WS = KR;             // dwords per staged weight row (no pad: 16-byte chunks XOR-swizzled by the row)
  constexpr int CV = CIN / 4;        // 16-byte vectors per pixel
  constexpr int NPV = PR * PC * CV;  // patch vectors
  constexpr int NWV = BN * (KR / 4); // weight vectors
  constexpr int NLP = (NPV + 255) / 256, NLW = (NWV + 255) / 256;
  constexpr int BM = PH * PW, LDC = BN + 4, FM = PW / 16;
  constexpr int PATCH_DW = PR * PC * CS;
  static_assert(BM * LDC <= PATCH_DW, "the C tile aliases the patch");
  __shared__ __attribute__((aligned(16))) float lds[PATCH_DW + BN * WS];
  float* P = lds;
  float* Wl = lds + PATCH_DW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = p.Wo / PW, tiles_y = p.Ho / PH;
  const int tile = (int)blockIdx.x;
  const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
  const int y0 = ty * PH, x0 = tx * PW;

  // ---- stage the patch and the weights (all loads in flight, then the LDS writes) ----------------------------------
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.src0, p.src0_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);
  uint4 pv[NLP], wv[NLW];
#pragma unroll
  for (int i = 0; i < NLP; ++i) {
    const int id = tid + 256 * i;
    const int pix = id / CV, cv = id - pix * CV;
    const int pr = pix / PC, pc = pix - pr * PC;
    const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
    const bool ok = id < NPV && (unsigned)gy < (unsigned)p.Hv && (unsigned)gx < (unsigned)p.Wv;
    pv[i] = buf_load16(rs, ok ? (unsigned)(((b * p.Hv + gy) * p.Wv + gx) * CIN + cv * 4) * 4u : BUF_OOB);
  }
#pragma unroll
  for (int i = 0; i < NLW; ++i) {
    const int id = tid + 256 * i;
    const int row = id / (KR / 4), ch = id - row * (KR / 4);
    const bool ok = id < NWV && row < p.CoutPad;
    wv[i] = buf_load16(rw, ok ? (unsigned)(row * p.w_ld + ch * 4) * 4u : BUF_OOB);
  }
#pragma unroll
  for (int i = 0; i < NLP; ++i) {
    const int id = tid + 256 * i;
    const int pix = id / CV, cv = id - pix * CV;
    if (NPV % 256 == 0 || id < NPV) *reinterpret_cast<uint4*>(&P[pix * CS + cv * 4]) = pv[i];
  }
#pragma unroll
  for (int i = 0; i < NLW; ++i) {
    const int id = tid + 256 * i;
    const int row = id / (KR / 4), ch = id - row * (KR / 4);
    if (NWV % 256 == 0 || id < NWV) *reinterpret_cast<uint4*>(&Wl[row * WS + (ch ^ ((row >> 1) & 3)) * 4]) = wv[i];
  }
  __syncthreads();

  // ---- k loop: 9 taps x 16 channels, everything from LDS -----------------------------------------------------------
  // v_mfma_f32_16x16x4_f32: lane holds A[m = lane & 15][k = lane >> 4] and B[k = lane >> 4][n = lane & 15].  One
  // 16-byte read gives a lane channels 4*fq .. 4*fq+3 of its pixel / filter: MFMA e of a tap contracts the channel
  // set {e, 4+e, 8+e, 12+e} -- the same permutation on both operands.
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* Abase = P + (wave * PC + fr) * CS + fq * 4;
  const float* Bbase = Wl + fr * WS + (fq ^ ((fr >> 1) & 3)) * 4;  // a tap's four chunks are permuted by the row
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int kh = tap / 3, kw = tap - kh * 3;
    const uint4 bb = *reinterpret_cast<const uint4*>(Bbase + tap * CIN);
    uint4 a[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i)
      a[i] = *reinterpret_cast<const uint4*>(Abase + ((kh * PC + kw) + i * 16) * CS);
    // element index outermost: consecutive MFMAs go to different accumulators
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i].x), __uint_as_float(bb.x), acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i].y), __uint_as_float(bb.y), acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i].z), __uint_as_float(bb.z), acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i].w), __uint_as_float(bb.w), acc[i], 0, 0, 0);
  }

  // ---- epilogue ----------------------------------------------------------------------------------------------------
  // accumulator register r of fragment i = out[pixel (y0 + wave, x0 + 16 i + 4 fq + r)][n = fr]
  const int HoWo = p.Ho * p.Wo;
  const long mrow0 = ((long)b * p.Ho + y0) * p.Wo + x0;  // output row (pixel index) of the tile's first pixel
  if (p.mode == CONV_HEAD_NCHW) {  // + bias, fp32 NCHW: a lane's four registers are four consecutive pixels
    if (fr < p.Cout) {
      float* __restrict__ out = reinterpret_cast<float*>(p.out0);
      const float bias = p.scale ? p.scale[fr] : 0.f;
      float* orow = out + ((long)b * p.Cout + fr) * HoWo + (long)(y0 + wave) * p.Wo + x0 + 4 * fq;
#pragma unroll
      for (int i = 0; i < FM; ++i)
        *reinterpret_cast<float4*>(orow + 16 * i) =
            make_float4(acc[i][0] + bias, acc[i][1] + bias, acc[i][2] + bias, acc[i][3] + bias);
    }
    return;
  }

  __syncthreads();  // every wave is done reading the patch: its space becomes the C tile [256][BN + 4]
  float* Cs = lds;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) Cs[(wave * PW + i * 16 + 4 * fq + r) * LDC + fr] = acc[i][r];
  __syncthreads();

  constexpr int VN = BN / 4, NVEC = BM * VN / 256, RSTEP = 256 / VN;
  const int cv = tid % VN, rv0 = tid / VN;
  const int n = cv * 4;
  const bool n_ok = n < p.Cout;
  // tile row -> output pixel index
  auto out_row = [&](int row) { return mrow0 + (long)(row / PW) * p.Wo + (row % PW); };

  if (p.mode == CONV_RAW_STATS) {
    float* __restrict__ out = reinterpret_cast<float*>(p.out0);
    if (n_ok) {
#pragma unroll
      for (int i = 0; i < NVEC; ++i) {
        const int row = rv0 + i * RSTEP;
        *reinterpret_cast<float4*>(out + out_row(row) * p.Cout + n) = *reinterpret_cast<const float4*>(&Cs[row * LDC + cv * 4]);
      }
    }
    if (p.stats != nullptr) {
      constexpr int NG = 256 / BN;
      const int col = tid % BN, rg = tid / BN;
      float s1 = 0.f, s2 = 0.f;
      for (int row = rg; row < BM; row += NG) {
        const float v = Cs[row * LDC + col];
        s1 += v;
        s2 += v * v;
      }
      __syncthreads();
      float* red = lds;
      red[(rg * BN + col) * 2 + 0] = s1;
      red[(rg * BN + col) * 2 + 1] = s2;
      __syncthreads();
      if (tid < BN) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          a1 += red[(g * BN + tid) * 2 + 0];
          a2 += red[(g * BN + tid) * 2 + 1];
        }
        if (tid < p.CoutPad) {
          p.stats[((long)tile * p.CoutPad + tid) * 2 + 0] = a1;
          p.stats[((long)tile * p.CoutPad + tid) * 2 + 1] = a2;
        }
      }
    }
  } else if (p.mode == CONV_EVAL_FUSED) {
    float* __restrict__ out = reinterpret_cast<float*>(p.out0);
    if (n_ok) {
      const float4 sc = *reinterpret_cast<const float4*>(p.scale + n), sf = *reinterpret_cast<const float4*>(p.shift + n);
#pragma unroll
      for (int i = 0; i < NVEC; ++i) {
        const int row = rv0 + i * RSTEP;
        float4 v = *reinterpret_cast<const float4*>(&Cs[row * LDC + cv * 4]);
        v.x = v.x * sc.x + sf.x; v.y = v.y * sc.y + sf.y; v.z = v.z * sc.z + sf.z; v.w = v.w * sc.w + sf.w;
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(out + out_row(row) * p.Cout + n) = v;
      }
    }
  } else {  // CONV_DGRAD, one destination
    float* __restrict__ o0 = reinterpret_cast<float*>(p.out0);
    if (n_ok) {
#pragma unroll
      for (int i = 0; i < NVEC; ++i) {
        const int row = rv0 + i * RSTEP;
        float4 v = *reinterpret_cast<const float4*>(&Cs[row * LDC + cv * 4]);
        float* dst = o0 + out_row(row) * p.Cout + n;
        if (p.acc0) {
          const float4 o = *reinterpret_cast<const float4*>(dst);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          if (p.bn_partial != nullptr) *reinterpret_cast<float4*>(&Cs[row * LDC + cv * 4]) = v;
        }
        *reinterpret_cast<float4*>(dst) = v;
      }
    }
    if (p.bn_partial != nullptr) {
      // fused BatchNorm-backward reduction of the consuming layer (conv_igemm.hip, same arithmetic and row order)
      if (p.acc0) __syncthreads();
      constexpr int NG = 256 / BN;
      const int col = tid % BN, rg = tid / BN;
      const int C = p.Cout;
      const bool cok = col < C;
      const float mu = cok ? p.bn_coef[col] : 0.f, is = cok ? p.bn_coef[C + col] : 0.f;
      const float sc = cok ? p.bn_coef[2 * C + col] : 0.f, sf = cok ? p.bn_coef[3 * C + col] : 0.f;
      const float* __restrict__ yb = reinterpret_cast<const float*>(p.bn_y);
      const float* __restrict__ ab = reinterpret_cast<const float*>(p.bn_a);
      float s1 = 0.f, s2 = 0.f;
      if (cok) {
        for (int row = rg; row < BM; row += NG) {
          const long m = out_row(row);
          const float yy = yb[m * C + col];
          const float keep = ab != nullptr ? ab[m * C + col] : yy * sc + sf;
          const float g = keep > 0.f ? Cs[row * LDC + col] : 0.f;
          s1 += g;
          s2 += g * ((yy - mu) * is);
        }
      }
      __syncthreads();
      float* red = lds;
      red[(rg * BN + col) * 2 + 0] = s1;
      red[(rg * BN + col) * 2 + 1] = s2;
      __syncthreads();
      if (tid < BN) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          a1 += red[(g * BN + tid) * 2 + 0];
          a2 += red[(g * BN + tid) * 2 + 1];
        }
        if (tid < C) {
          p.bn_partial[((long)tile * C + tid) * 2 + 0] = a1;
          p.bn_partial[((long)tile * C + tid) * 2 + 1] = a2;
        }
      }
    }
  }
}

// plan: one workgroup per 4 x 64 output tile, one statistics row per tile
void conv_patch_plan(ConvParams& p) {
  p.patch = 1;
  p.nz = 1;
  p.splitk = 1;
  p.xcd_swizzle = 0;
  p.w_ld = p.Kpad;
  p.tiles_m = p.B * (p.Ho / CP_PH) * (p.Wo / CP_PW);
  p.tiles_n = 1;
  p.stat_rows = p.tiles_m;
}

int conv_patch_launch(const ConvParams& p, hipStream_t stream) {
  D3F_CHECK(p.patch == 1 && p.tiles_m == p.B * (p.Ho / CP_PH) * (p.Wo / CP_PW) && p.C0 == 16 && p.Cout <= 16,
            "conv: patch params were not planned");
  hipLaunchKernelGGL((conv_patch_kernel<16, 16>), dim3((unsigned)p.tiles_m), dim3(256), 0, stream, p);
  D3F_HIP(hipGetLastError());
  return 0;
}

}  // namespace d3f
