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

static bool patch_conv_off() {
  static const bool off = getenv("D3F_NO_PATCH_CONV") != nullptr;  // debugging knob: the implicit-GEMM path instead
  return off;
}

bool conv_patch_applies(const ConvParams& p, int dtype) {
  if (patch_conv_off() || (dtype != D3F_F32 && dtype != D3F_BF16)) return false;
  const bool mode_ok = p.mode == CONV_RAW_STATS || p.mode == CONV_HEAD_NCHW ||
                       (p.mode == CONV_EVAL_FUSED && p.res == nullptr) ||
                       (p.mode == CONV_DGRAD && p.out_c0 == p.Cout && (p.Cout % 4) == 0);
  // C0 == 4: the data gradient of the segmentation head (dY has 3 channels padded to 4, 16 outputs): as an implicit
  // GEMM it ran 110 us in the traced step for 17 MB in / 67 MB out (profiles/r03_z_step_launches.txt)
  // bf16 storage (round 3): the 16-channel form on v_mfma_f32_16x16x16_bf16, one instruction per tap and fragment
  // bf16, 32 channels -> <= 32 filters (round 5: decoder block 3 conv2 forward + data gradient at 128x128): a pixel is
  // 64 bytes like fp32 x 16 channels, so the staged image is the fp32 one byte for byte; one v_mfma_f32_16x16x32_bf16
  // contracts a whole tap
  const bool wide = p.C0 == 32 && dtype == D3F_BF16 && p.mode != CONV_HEAD_NCHW;
  // C0 == 8, bf16: the head's data gradient in bf16 storage (3 channels in one 16-byte vector), staged as 16 channels
  const bool cin_ok = p.C0 == 16 || wide || (p.C0 == 4 && p.mode == CONV_DGRAD && dtype == D3F_F32) ||
                      (p.C0 == 8 && p.mode == CONV_DGRAD && dtype == D3F_BF16);
  // ... and behind a nearest x2 up-sampling (forward only): <= 16 filters, even extents
  const bool up = wide && p.shift0 == 1 && p.mode != CONV_DGRAD && p.Cout <= 16 && p.H0s * 2 == p.Hv && p.W0s * 2 == p.Wv;
  // a data gradient whose caller asked for the 2x2-summed (half-resolution) output: bf16 x 16 -> exactly 32 channels
  const bool sum2 = p.sum2 && dtype == D3F_BF16 && p.mode == CONV_DGRAD && p.C0 == 16 && p.Cout == 32 && p.shift0 == 0;
  if (p.sum2 && !sum2) return false;
  return mode_ok && p.par == 0 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && cin_ok && p.C1 == 0 &&
         (p.shift0 == 0 || up) && p.zi == 0 && p.Cout <= (wide && !up ? 64 : sum2 ? 32 : 16) && (p.mode == CONV_HEAD_NCHW || (p.Cout % 4) == 0) &&
         p.Hv == p.Ho && p.Wv == p.Wo && (p.Ho % CP_PH) == 0 &&
         (p.Wo % CP_PW) == 0 && p.Kpad >= 9 * p.C0;
}

// UP (bf16 x 32 channels -> <= 16 filters, round 5): the source is read through a nearest x2 up-sampling (decoder block 4
// conv1 in bf16 storage, whose 32 channels are half a k-tile of the implicit GEMM: it gathered through the up-sampling
// with the small-channel loader).  The patch is staged at the source's own LOW resolution -- 4 x 34 pixels for the 4 x 64
// output tile -- and tap (kh, kw) of output pixel (y, x) reads low-resolution pixel ((y + kh - 1) >> 1, (x + kw - 1) >> 1):
// the row is wave-uniform, the column one of three per-lane offsets made once.
// CSRC < CIN (bf16 x 8 source channels staged as 16, round 5): the data gradient of the segmentation head in bf16 storage
// (dY has 3 channels padded to one 16-byte vector of 8).  The source pixel and every tap of a weight row fill the first
// half of their 16-channel LDS slot, the second half is zero: the k-loop is the 16-channel one unchanged.
// SUM2 (bf16 x 16 -> 32, data gradient only, round 5): the data gradient of a convolution whose source was read through
// the nearest x2 up-sampling WITHOUT the folded weights (decoder block 4 conv1 in bf16 storage).  The gradient w.r.t. the
// low-resolution source is the 2x2 block sum of the full-resolution gradient: the epilogue adds the four pixels from the
// fp32 C tile and stores at half resolution -- the full-resolution scratch tensor (67 MB written + read), the sum2x2
// launch and the separate BatchNorm-backward reduce of the consumer (its partial sums ride in this epilogue) are gone.
template <typename T, int CIN, int BN, bool UP = false, int CSRC = CIN, bool SUM2 = false>
__global__ __launch_bounds__(256) void conv_patch_kernel(const ConvParams pin) {
  const ConvParams p = conv_params_of_net(pin, (int)blockIdx.z);  // two networks in one launch: blockIdx.z = net
  chain_priority();
  constexpr bool BF = sizeof(T) == 2;
  // W64: bf16 x 32 channels -- 64-byte pixels, 32 filters (decoder block 3 conv2)
  constexpr bool W64 = BF && CIN == 32;
  static_assert(!UP || (W64 && BN == 16), "up-sampled source: bf16 x 32 channels -> 16 filters");
  static_assert(CSRC == CIN || (BF && CIN == 16 && CSRC == 8 && !UP), "half-filled pixels: bf16, 8 of 16 channels");
  static_assert(!SUM2 || (BF && CIN == 16 && CSRC == 16 && BN == 32 && !UP), "2x2-summed data gradient: bf16 x 16 -> 32");
  static_assert(((CIN == 16 || (CIN == 4 && !BF)) && (BN == 16 || SUM2)) || (W64 && (BN == 32 || UP)),
                "16 channels (forward / data gradient of decoder block 4, head forward), fp32 x 4 (data gradient of the "
                "head) or bf16 x 32 -> 32 (decoder block 3 conv2)");
  constexpr int VEC = 16 / (int)sizeof(T);  // elements per 16-byte vector
  constexpr int PH = CP_PH, PW = CP_PW, PR = UP ? PH / 2 + 2 : PH + 2, PC = UP ? PW / 2 + 2 : PW + 2;
  // dwords per staged pixel: such that the fragment reads spread over the banks (fp32 CIN 16: +4 pad, 16-byte reads;
  // CIN 4: stride 12, scalar reads; bf16: 32 B of channels + 8 B pad = 10 dwords -- 10 * row covers the even banks
  // over 16 rows, the 8-byte fragment reads are conflict-free; the staging stores are 8-byte halves)
  constexpr int CS = W64 ? 20 : BF ? 10 : (CIN == 16 ? CIN + 4 : 12);
  constexpr int KR = 9 * CIN;        // real k extent
  // dwords per staged weight row (fp32 CIN 16: no pad, 16-byte chunks XOR-swizzled by the row; CIN 4: +4 pad, no
  // swizzle; bf16: 72 dwords of taps + 2 = 74: the same stride-10-mod-32 pattern over the filter rows)
  constexpr int WS = W64 ? KR / 2 : BF ? KR / 2 + 2 : (CIN == 16 ? KR : KR + 4);
  constexpr int CV = CSRC / VEC;       // 16-byte vectors per source pixel
  constexpr int KS = 9 * CSRC;         // k extent of a packed weight row in memory
  constexpr int NPV = PR * PC * CV;    // patch vectors
  constexpr int NWV = BN * (KS / VEC); // weight vectors
  constexpr int NLP = (NPV + 255) / 256, NLW = (NWV + 255) / 256;
  constexpr int BM = PH * PW, LDC = BN + 4, FM = PW / 16, NB = BN / 16;
  constexpr int PATCH_DW = PR * PC * CS > BM * LDC ? PR * PC * CS : BM * LDC;  // the C tile aliases the patch
  __shared__ __attribute__((aligned(16))) float lds[PATCH_DW + BN * WS];
  float* P = lds;
  float* Wl = lds + PATCH_DW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = p.Wo / PW, tiles_y = p.Ho / PH;
  const int tile = (int)blockIdx.x;
  const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
  const int y0 = ty * PH, x0 = tx * PW;
  const int n0 = (int)blockIdx.y * BN;  // first filter of this workgroup (grid.y > 1: the 64-filter bf16 launches)

  // ---- stage the patch and the weights (all loads in flight, then the LDS writes) ----------------------------------
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.src0, p.src0_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);
  uint4 pv[NLP], wv[NLW];
#pragma unroll
  for (int i = 0; i < NLP; ++i) {
    const int id = tid + 256 * i;
    const int pix = id / CV, cv = id - pix * CV;
    const int pr = pix / PC, pc = pix - pr * PC;
    // (UP: y0, x0 are even; low-resolution pixel -1 / H0s stands for the zero border of the up-sampled image)
    const int gy = (UP ? y0 / 2 : y0) - 1 + pr, gx = (UP ? x0 / 2 : x0) - 1 + pc;
    const int Hs = UP ? p.H0s : p.Hv, Ws = UP ? p.W0s : p.Wv;
    const bool ok = id < NPV && (unsigned)gy < (unsigned)Hs && (unsigned)gx < (unsigned)Ws;
    pv[i] = buf_load16(rs, ok ? (unsigned)(((b * Hs + gy) * Ws + gx) * CSRC + cv * VEC) * (unsigned)sizeof(T) : BUF_OOB);
  }
#pragma unroll
  for (int i = 0; i < NLW; ++i) {
    const int id = tid + 256 * i;
    const int row = id / (KS / VEC), ch = id - row * (KS / VEC);
    const bool ok = id < NWV && n0 + row < p.CoutPad;
    wv[i] = buf_load16(rw, ok ? (unsigned)((n0 + row) * p.w_ld + ch * VEC) * (unsigned)sizeof(T) : BUF_OOB);
  }
#pragma unroll
  for (int i = 0; i < NLP; ++i) {
    const int id = tid + 256 * i;
    const int pix = id / CV, cv = id - pix * CV;
    if (NPV % 256 == 0 || id < NPV) {
      if constexpr (BF && !W64) {  // 40-byte pixels: two 8-byte stores
        *reinterpret_cast<uint2*>(&P[pix * CS + cv * 4]) = make_uint2(pv[i].x, pv[i].y);
        *reinterpret_cast<uint2*>(&P[pix * CS + cv * 4 + 2]) = make_uint2(pv[i].z, pv[i].w);
        if constexpr (CSRC < CIN) {  // channels 8 .. 15 of the slot: zero
          *reinterpret_cast<uint2*>(&P[pix * CS + 4]) = make_uint2(0u, 0u);
          *reinterpret_cast<uint2*>(&P[pix * CS + 6]) = make_uint2(0u, 0u);
        }
      } else {
        *reinterpret_cast<uint4*>(&P[pix * CS + cv * 4]) = pv[i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NLW; ++i) {
    const int id = tid + 256 * i;
    const int row = id / (KS / VEC), ch = id - row * (KS / VEC);
    if (NWV % 256 == 0 || id < NWV) {
      if constexpr (CSRC < CIN) {  // vector `ch` = the 8 channels of tap `ch`: first half of the tap's 16-channel slot
        *reinterpret_cast<uint2*>(&Wl[row * WS + ch * 8]) = make_uint2(wv[i].x, wv[i].y);
        *reinterpret_cast<uint2*>(&Wl[row * WS + ch * 8 + 2]) = make_uint2(wv[i].z, wv[i].w);
        *reinterpret_cast<uint2*>(&Wl[row * WS + ch * 8 + 4]) = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(&Wl[row * WS + ch * 8 + 6]) = make_uint2(0u, 0u);
      } else if constexpr (BF && !W64) {
        *reinterpret_cast<uint2*>(&Wl[row * WS + ch * 4]) = make_uint2(wv[i].x, wv[i].y);
        *reinterpret_cast<uint2*>(&Wl[row * WS + ch * 4 + 2]) = make_uint2(wv[i].z, wv[i].w);
      } else {
        const int chs = (CIN == 16 || W64) ? (ch ^ ((row >> 1) & 3)) : ch;
        *reinterpret_cast<uint4*>(&Wl[row * WS + chs * 4]) = wv[i];
      }
    }
  }
  __syncthreads();

  // ---- k loop: 9 taps x 16 channels, everything from LDS -----------------------------------------------------------
  // v_mfma_f32_16x16x4_f32: lane holds A[m = lane & 15][k = lane >> 4] and B[k = lane >> 4][n = lane & 15].  One
  // 16-byte read gives a lane channels 4*fq .. 4*fq+3 of its pixel / filter: MFMA e of a tap contracts the channel
  // set {e, 4+e, 8+e, 12+e} -- the same permutation on both operands.
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc[FM * NB];  // [fragment i][16-filter group j] at i * NB + j
#pragma unroll
  for (int i = 0; i < FM * NB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (UP) {
    // staged column of (x0 + 16 i + fr + kw - 1) >> 1 = ((fr + kw - 1) >> 1) + 1 + 8 i; staged row of (y0 + wave + kh - 1) >> 1
    // = ((wave + kh - 1) >> 1) + 1
    const float* Bbase = Wl + fr * WS + (fq ^ ((fr >> 1) & 3)) * 4;
    const float* Acol[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) Acol[kw] = P + (((fr + kw - 1) >> 1) + 1) * CS + fq * 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - kh * 3;
      const int prow = ((wave + kh - 1) >> 1) + 1;
      const uint4 bb = *reinterpret_cast<const uint4*>(Bbase + tap * (CIN / 2));
      uint4 a[FM];
#pragma unroll
      for (int i = 0; i < FM; ++i) a[i] = *reinterpret_cast<const uint4*>(Acol[kw] + (prow * PC + 8 * i) * CS);
#pragma unroll
      for (int i = 0; i < FM; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a[i]),
                                                         *reinterpret_cast<const bf16x8*>(&bb), acc[i], 0, 0, 0);
    }
  } else if constexpr (W64) {
    // lane holds channels 8 fq .. 8 fq + 7 (16 bytes) of its pixel / filter: one MFMA contracts the tap's 32 channels
    const float* Abase = P + (wave * PC + fr) * CS + fq * 4;
    const float* Bbase = Wl + fr * WS + (fq ^ ((fr >> 1) & 3)) * 4;  // (rows fr and fr + 16 share the swizzle)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - kh * 3;
      uint4 bb[NB], a[FM];
#pragma unroll
      for (int j = 0; j < NB; ++j) bb[j] = *reinterpret_cast<const uint4*>(Bbase + j * 16 * WS + tap * (CIN / 2));
#pragma unroll
      for (int i = 0; i < FM; ++i) a[i] = *reinterpret_cast<const uint4*>(Abase + ((kh * PC + kw) + i * 16) * CS);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
          acc[i * NB + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a[i]),
                                                                    *reinterpret_cast<const bf16x8*>(&bb[j]), acc[i * NB + j], 0, 0, 0);
    }
  } else if constexpr (BF) {
    // bf16: v_mfma_f32_16x16x16_bf16 contracts all 16 channels of a tap at once: lane holds channels 4 fq .. 4 fq + 3
    // (8 bytes) of its pixel / filter
    typedef short v4s __attribute__((ext_vector_type(4)));
    const float* Abase = P + (wave * PC + fr) * CS + fq * 2;
    const float* Bbase = Wl + fr * WS + fq * 2;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - kh * 3;
      v4s bb[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) bb[j] = *reinterpret_cast<const v4s*>(Bbase + j * 16 * WS + tap * (CIN / 2));
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const v4s a = *reinterpret_cast<const v4s*>(Abase + ((kh * PC + kw) + i * 16) * CS);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i * NB + j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, bb[j], acc[i * NB + j], 0, 0, 0);
      }
    }
  } else if constexpr (CIN == 16) {
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
  } else {
    // 4 channels: one MFMA per tap and fragment, the lane's k index IS the channel (k = 3 is the zero pad channel)
    const float* Abase = P + (wave * PC + fr) * CS + fq;
    const float* Bbase = Wl + fr * WS + fq;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - kh * 3;
      const float bb = Bbase[tap * CIN];
      float a[FM];
#pragma unroll
      for (int i = 0; i < FM; ++i) a[i] = Abase[((kh * PC + kw) + i * 16) * CS];
#pragma unroll
      for (int i = 0; i < FM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bb, acc[i], 0, 0, 0);
    }
  }

  // ---- epilogue ----------------------------------------------------------------------------------------------------
  // accumulator register r of fragment i = out[pixel (y0 + wave, x0 + 16 i + 4 fq + r)][n = fr]
  const int HoWo = p.Ho * p.Wo;
  const long mrow0 = ((long)b * p.Ho + y0) * p.Wo + x0;  // output row (pixel index) of the tile's first pixel
  if (NB == 1 && p.mode == CONV_HEAD_NCHW) {  // + bias, fp32 NCHW: a lane's four registers are four consecutive pixels
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
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cs[(wave * PW + i * 16 + 4 * fq + r) * LDC + j * 16 + fr] = acc[i * NB + j][r];
  __syncthreads();

  if constexpr (SUM2) {
    // out[(b, y0/2 + lr, x0/2 + lc)][n] = sum of the 2x2 block of the C tile; thread -> channel vector cv of low-resolution
    // pixels lp = tid / 8 + 32 i (i = 0, 1: the tile's two low-resolution rows)
    constexpr int VN2 = BN / 4, LW = PW / 2, NLOW = (PH / 2) * LW, NV2 = NLOW * VN2 / 256, RS2 = 256 / VN2;
    static_assert(NV2 * 256 == NLOW * VN2, "whole vectors per thread");
    const int cv = tid % VN2, lp0 = tid / VN2;
    const int n = cv * 4, C = p.Cout;
    const int Hl = p.Ho / 2, Wl2 = p.Wo / 2;
    T* __restrict__ o0 = reinterpret_cast<T*>(p.out0);
    float4 v[NV2];
    long mlow[NV2];
#pragma unroll
    for (int i = 0; i < NV2; ++i) {
      const int lp = lp0 + i * RS2, lr = lp / LW, lc = lp - lr * LW;
      const float* c00 = &Cs[((2 * lr) * PW + 2 * lc) * LDC + n];
      const float4 a = *reinterpret_cast<const float4*>(c00), b2 = *reinterpret_cast<const float4*>(c00 + LDC);
      const float4 c = *reinterpret_cast<const float4*>(c00 + PW * LDC), d = *reinterpret_cast<const float4*>(c00 + PW * LDC + LDC);
      v[i] = make_float4((a.x + b2.x) + (c.x + d.x), (a.y + b2.y) + (c.y + d.y), (a.z + b2.z) + (c.z + d.z), (a.w + b2.w) + (c.w + d.w));
      mlow[i] = ((long)b * Hl + (y0 / 2 + lr)) * Wl2 + (x0 / 2 + lc);
      T* dst = o0 + mlow[i] * C + n;
      if (p.acc0) {
        const float4 o = ld4<T>(dst);
        v[i].x += o.x; v[i].y += o.y; v[i].z += o.z; v[i].w += o.w;
      }
      st4<T>(dst, v[i]);
    }
    if (p.bn_partial != nullptr) {
      // fused BatchNorm-backward reduction of the consuming layer over this tile's 64 low-resolution pixels
      const T* __restrict__ yb = reinterpret_cast<const T*>(p.bn_y);
      const T* __restrict__ ab = reinterpret_cast<const T*>(p.bn_a);
      const float4 mu = *reinterpret_cast<const float4*>(p.bn_coef + n), is = *reinterpret_cast<const float4*>(p.bn_coef + C + n);
      const float4 sc = *reinterpret_cast<const float4*>(p.bn_coef + 2 * C + n), sf = *reinterpret_cast<const float4*>(p.bn_coef + 3 * C + n);
      const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, i4[4] = {is.x, is.y, is.z, is.w};
      const float c4[4] = {sc.x, sc.y, sc.z, sc.w}, f4[4] = {sf.x, sf.y, sf.z, sf.w};
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NV2; ++i) {
        const float4 yv = ld4<T>(yb + mlow[i] * C + n);
        const float4 av = ab != nullptr ? ld4<T>(ab + mlow[i] * C + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float yy[4] = {yv.x, yv.y, yv.z, yv.w}, aa[4] = {av.x, av.y, av.z, av.w};
        const float gg[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float keep = ab != nullptr ? aa[k] : yy[k] * c4[k] + f4[k];
          const float g = keep > 0.f ? gg[k] : 0.f;
          s1[k] += g;
          s2[k] += g * ((yy[k] - m4[k]) * i4[k]);
        }
      }
      __syncthreads();  // all reads of the C tile are done
      float* red = lds;  // [RS2][BN][2]
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        red[(lp0 * BN + n + k) * 2 + 0] = s1[k];
        red[(lp0 * BN + n + k) * 2 + 1] = s2[k];
      }
      __syncthreads();
      if (tid < BN) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll 8
        for (int g = 0; g < RS2; ++g) {
          a1 += red[(g * BN + tid) * 2 + 0];
          a2 += red[(g * BN + tid) * 2 + 1];
        }
        if (tid < C) {
          p.bn_partial[((long)tile * C + tid) * 2 + 0] = a1;
          p.bn_partial[((long)tile * C + tid) * 2 + 1] = a2;
        }
      }
    }
    return;
  }

  constexpr int VN = BN / 4, NVEC = BM * VN / 256, RSTEP = 256 / VN;
  const int cv = tid % VN, rv0 = tid / VN;
  const int n = n0 + cv * 4;
  const bool n_ok = n < p.Cout;
  // tile row -> output pixel index
  auto out_row = [&](int row) { return mrow0 + (long)(row / PW) * p.Wo + (row % PW); };

  if (p.mode == CONV_RAW_STATS) {
    T* __restrict__ out = reinterpret_cast<T*>(p.out0);
    if (n_ok) {
#pragma unroll
      for (int i = 0; i < NVEC; ++i) {
        const int row = rv0 + i * RSTEP;
        st4<T>(out + out_row(row) * p.Cout + n, *reinterpret_cast<const float4*>(&Cs[row * LDC + cv * 4]));
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
        if (n0 + tid < p.CoutPad) {
          p.stats[((long)tile * p.CoutPad + n0 + tid) * 2 + 0] = a1;
          p.stats[((long)tile * p.CoutPad + n0 + tid) * 2 + 1] = a2;
        }
      }
    }
  } else if (p.mode == CONV_EVAL_FUSED) {
    T* __restrict__ out = reinterpret_cast<T*>(p.out0);
    if (n_ok) {
      const float4 sc = *reinterpret_cast<const float4*>(p.scale + n), sf = *reinterpret_cast<const float4*>(p.shift + n);
#pragma unroll
      for (int i = 0; i < NVEC; ++i) {
        const int row = rv0 + i * RSTEP;
        float4 v = *reinterpret_cast<const float4*>(&Cs[row * LDC + cv * 4]);
        v.x = v.x * sc.x + sf.x; v.y = v.y * sc.y + sf.y; v.z = v.z * sc.z + sf.z; v.w = v.w * sc.w + sf.w;
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        st4<T>(out + out_row(row) * p.Cout + n, v);
      }
    }
  } else {  // CONV_DGRAD, one destination
    T* __restrict__ o0 = reinterpret_cast<T*>(p.out0);
    if (n_ok) {
#pragma unroll
      for (int i = 0; i < NVEC; ++i) {
        const int row = rv0 + i * RSTEP;
        float4 v = *reinterpret_cast<const float4*>(&Cs[row * LDC + cv * 4]);
        T* dst = o0 + out_row(row) * p.Cout + n;
        if (p.acc0) {
          const float4 o = ld4<T>(dst);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          if (p.bn_partial != nullptr) *reinterpret_cast<float4*>(&Cs[row * LDC + cv * 4]) = v;
        }
        st4<T>(dst, v);
      }
    }
    if (p.bn_partial != nullptr) {
      // fused BatchNorm-backward reduction of the consuming layer (as conv_igemm.hip): 16-byte loads of y (and a) for
      // the thread's own rows, column sums in registers, thread rows added through LDS in order
      const int C = p.Cout;
      const T* __restrict__ yb = reinterpret_cast<const T*>(p.bn_y);
      const T* __restrict__ ab = reinterpret_cast<const T*>(p.bn_a);
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
      if (n_ok) {
        const float4 mu = *reinterpret_cast<const float4*>(p.bn_coef + n), is = *reinterpret_cast<const float4*>(p.bn_coef + C + n);
        const float4 sc = *reinterpret_cast<const float4*>(p.bn_coef + 2 * C + n), sf = *reinterpret_cast<const float4*>(p.bn_coef + 3 * C + n);
        const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, i4[4] = {is.x, is.y, is.z, is.w};
        const float c4[4] = {sc.x, sc.y, sc.z, sc.w}, f4[4] = {sf.x, sf.y, sf.z, sf.w};
        float4 yv[NVEC], av[NVEC];
#pragma unroll
        for (int i = 0; i < NVEC; ++i) {
          const long m = out_row(rv0 + i * RSTEP);
          yv[i] = ld4<T>(yb + m * C + n);
          av[i] = ab != nullptr ? ld4<T>(ab + m * C + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NVEC; ++i) {
          const float4 gv = *reinterpret_cast<const float4*>(&Cs[(rv0 + i * RSTEP) * LDC + cv * 4]);  // own slot
          const float yy[4] = {yv[i].x, yv[i].y, yv[i].z, yv[i].w}, aa[4] = {av[i].x, av[i].y, av[i].z, av[i].w};
          const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float keep = ab != nullptr ? aa[k] : yy[k] * c4[k] + f4[k];
            const float g = keep > 0.f ? gg[k] : 0.f;
            s1[k] += g;
            s2[k] += g * ((yy[k] - m4[k]) * i4[k]);
          }
        }
      }
      __syncthreads();  // all reads of the C tile are done
      float* red = lds;  // [RSTEP][BN][2]
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        red[(rv0 * BN + cv * 4 + k) * 2 + 0] = s1[k];
        red[(rv0 * BN + cv * 4 + k) * 2 + 1] = s2[k];
      }
      __syncthreads();
      if (tid < BN) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll 8
        for (int g = 0; g < RSTEP; ++g) {
          a1 += red[(g * BN + tid) * 2 + 0];
          a2 += red[(g * BN + tid) * 2 + 1];
        }
        if (n0 + tid < C) {
          p.bn_partial[((long)tile * C + n0 + tid) * 2 + 0] = a1;
          p.bn_partial[((long)tile * C + n0 + tid) * 2 + 1] = a2;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// encoder.conv1 (7x7, stride 2, pad 3, 3 input channels padded to 4, 64 filters) the same way.  As an implicit GEMM it
// has 7 k-tiles, so the prologue, the per-lane tap decode of the small-channel loader and the epilogue of its 2048
// workgroups were most of its 162 us.  Here a workgroup computes an 8 x 32 pixel output tile for 32 filters:
//   patch: 21 x 69 input pixels x 4 channels (23 KB), columns de-interleaved (even | odd) so that the stride-2 reads
//     of 32 consecutive output pixels are 32 consecutive 16-byte chunks (conflict-free);
//   weights: [32][7 rows][4 tap pairs][2][4 ch] (+4 dwords per filter: conflict-free), the 8th tap of a row is zero;
//   k-loop: v_mfma_f32_32x32x2_f32 contracts (tap 2j, tap 2j+1) x one channel per instruction -- the lane's k index
//     selects the tap of the pair -- and skips the zero pad channel: 28 x 3 instructions per 32 x 32 fragment
//     (168 k against the implicit GEMM's 224).
// wave w = output rows 2w, 2w+1 of the tile; 53 KB of LDS -> 3 workgroups per CU.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int ST_PH = 8, ST_PW = 32, ST_BN = 32;

bool conv_stem_applies(const ConvParams& p, int dtype) {
  if (patch_conv_off() || dtype != D3F_F32) return false;
  const bool mode_ok = (p.mode == CONV_RAW_STATS) || (p.mode == CONV_EVAL_FUSED && p.res == nullptr);
  return mode_ok && p.par == 0 && p.KH == 7 && p.KW == 7 && p.stride == 2 && p.pad == 3 && p.C0 == 4 && p.C1 == 0 &&
         p.shift0 == 0 && p.zi == 0 && (p.Cout % ST_BN) == 0 && p.Hv == 2 * p.Ho && p.Wv == 2 * p.Wo &&
         (p.Ho % ST_PH) == 0 && (p.Wo % ST_PW) == 0 && p.Kpad >= 49 * 4;
}

__global__ __launch_bounds__(256) void conv_stem_kernel(const ConvParams pin) {
  const ConvParams p = conv_params_of_net(pin, (int)blockIdx.z);  // two networks in one launch: blockIdx.z = net
  chain_priority();
  constexpr int PH = ST_PH, PW = ST_PW, BN = ST_BN;
  constexpr int PR = 2 * PH + 5, PC = 2 * PW + 5;  // 21 x 69 input pixels
  constexpr int NE = (PC + 1) / 2;                 // even columns come first in a staged row
  constexpr int PATCH_DW = PR * PC * 4 + 4;        // + one zeroed pixel: the pad tap of the last row reads it
  constexpr int WS = 7 * 4 * 2 * 4 + 4;            // dwords per staged filter
  constexpr int NPV = PR * PC, NWV = BN * 56;
  constexpr int NLP = (NPV + 255) / 256, NLW = (NWV + 255) / 256;
  constexpr int BM = PH * PW, LDC = BN + 4;
  static_assert(BM * LDC <= PATCH_DW + BN * WS, "the C tile aliases the staging area");
  __shared__ __attribute__((aligned(16))) float lds[PATCH_DW + BN * WS];
  float* P = lds;
  float* Wl = lds + PATCH_DW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = p.Wo / PW, tiles_y = p.Ho / PH;
  const int tile = (int)blockIdx.x, n0 = (int)blockIdx.y * BN;
  const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
  const int y0 = ty * PH, x0 = tx * PW;

  const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.src0, p.src0_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);
  uint4 pv[NLP], wv[NLW];
#pragma unroll
  for (int i = 0; i < NLP; ++i) {
    const int id = tid + 256 * i;
    const int pr = id / PC, pc = id - pr * PC;
    const int gy = 2 * y0 - 3 + pr, gx = 2 * x0 - 3 + pc;
    const bool ok = id < NPV && (unsigned)gy < (unsigned)p.Hv && (unsigned)gx < (unsigned)p.Wv;
    pv[i] = buf_load16(rs, ok ? (unsigned)(((b * p.Hv + gy) * p.Wv + gx) * 4) * 4u : BUF_OOB);
  }
#pragma unroll
  for (int i = 0; i < NLW; ++i) {
    const int id = tid + 256 * i;  // (filter, row kh, pair j, half h): tap kw = 2j + h
    const int n = id / 56, r = id - n * 56, kh = r >> 3, kw = r & 7;
    const bool ok = id < NWV && kw < 7 && (n0 + n) < p.CoutPad;
    wv[i] = buf_load16(rw, ok ? (unsigned)((n0 + n) * p.w_ld + (kh * 7 + kw) * 4) * 4u : BUF_OOB);
  }
#pragma unroll
  for (int i = 0; i < NLP; ++i) {
    const int id = tid + 256 * i;
    const int pr = id / PC, pc = id - pr * PC;
    if (id < NPV) *reinterpret_cast<uint4*>(&P[(pr * PC + (pc & 1) * NE + (pc >> 1)) * 4]) = pv[i];
  }
  if (tid == 0) *reinterpret_cast<uint4*>(&P[PR * PC * 4]) = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int i = 0; i < NLW; ++i) {
    const int id = tid + 256 * i;
    const int n = id / 56, r = id - n * 56;
    if (NWV % 256 == 0 || id < NWV) *reinterpret_cast<uint4*>(&Wl[n * WS + r * 4]) = wv[i];
  }
  __syncthreads();

  // lane: output pixel fr of its fragment, k index fq = which tap of the pair
  const int fr = lane & 31, fq = lane >> 5;
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // fragment i = tile row 2*wave + i; tap (kh, kw): input row 2*(2*wave + i) + kh, column 2*fr + kw -> staged column
  // (kw & 1) * NE + fr + (kw >> 1).  kw = 2j + fq: parity fq, offset j.
  const float* Abase = P + ((4 * wave) * PC + fq * NE + fr) * 4;
  const float* Bbase = Wl + fr * WS + fq * 4;
#pragma unroll
  for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint4 bb = *reinterpret_cast<const uint4*>(Bbase + (kh * 4 + j) * 8);
      uint4 a[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const uint4*>(Abase + ((2 * i + kh) * PC + j) * 4);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[i].x), __uint_as_float(bb.x), acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[i].y), __uint_as_float(bb.y), acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[i].z), __uint_as_float(bb.z), acc[i], 0, 0, 0);
    }
  }

  // epilogue: register r of fragment i = out[pixel (y0 + 2*wave + i, x0 + (r&3) + 8*(r>>2) + 4*fq)][n0 + fr]
  __syncthreads();
  float* Cs = lds;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      Cs[((2 * wave + i) * PW + (r & 3) + 8 * (r >> 2) + 4 * fq) * LDC + fr] = acc[i][r];
  __syncthreads();

  constexpr int VN = BN / 4, NVEC = BM * VN / 256, RSTEP = 256 / VN;
  const int cv = tid % VN, rv0 = tid / VN;
  const int n = n0 + cv * 4;
  const long mrow0 = ((long)b * p.Ho + y0) * p.Wo + x0;
  auto out_row = [&](int row) { return mrow0 + (long)(row / PW) * p.Wo + (row % PW); };
  float* __restrict__ out = reinterpret_cast<float*>(p.out0);
  if (p.mode == CONV_RAW_STATS) {
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
      const int row = rv0 + i * RSTEP;
      *reinterpret_cast<float4*>(out + out_row(row) * p.Cout + n) = *reinterpret_cast<const float4*>(&Cs[row * LDC + cv * 4]);
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
        p.stats[((long)tile * p.CoutPad + n0 + tid) * 2 + 0] = a1;
        p.stats[((long)tile * p.CoutPad + n0 + tid) * 2 + 1] = a2;
      }
    }
  } else {  // CONV_EVAL_FUSED: folded BatchNorm (+ ReLU)
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
}

// ---------------------------------------------------------------------------------------------------------------------
// encoder.conv1 in bf16 storage (round 5): the image's 3 channels arrive padded to one 16-byte vector of 8 bf16, i.e.
// a quarter of a k-tile per tap -- the implicit GEMM ran it with the small-channel loader at 69 us (99 us next to the
// weight packing) for 2 us of matrix work.  Same tiling as conv_stem_kernel (8 x 32 output pixels x 32 filters), but:
//   patch: 21 x 69 pixels x 4 channels (the first 8 bytes of every pixel; channel 3 is the zero pad), row-major, 8 bytes
//     per pixel -- output pixel ox reads input columns 2 ox + kw, so taps (kw, kw + 1) of lane ox are 16 contiguous bytes
//     and the 32 lanes of a fragment read 512 contiguous bytes (conflict-free);
//   weights: [32][7 rows][8 tap slots][4 ch] bf16 (+16 B per filter), slot 7 zero;
//   k-loop: v_mfma_f32_32x32x16_bf16 contracts FOUR taps x 4 channels per instruction (lane group g supplies taps
//     4 j + 2 g, 4 j + 2 g + 1): 7 x 2 instructions per 32 x 32 fragment.
// LDS 36.9 KB (the fp32 C tile aliases patch + weights) -> 4 workgroups per CU.
// ---------------------------------------------------------------------------------------------------------------------
bool conv_stem_bf16_applies(const ConvParams& p, int dtype) {
  if (patch_conv_off() || dtype != D3F_BF16) return false;
#ifdef D3F_NO_PATCH32
  return false;
#endif
  const bool mode_ok = (p.mode == CONV_RAW_STATS) || (p.mode == CONV_EVAL_FUSED && p.res == nullptr);
  // (cin_real in 1 .. 4: the packed weights of channels 4 .. 7 are zero, so only the first half of a pixel is staged)
  return mode_ok && p.par == 0 && p.KH == 7 && p.KW == 7 && p.stride == 2 && p.pad == 3 && p.C0 == 8 && p.C1 == 0 &&
         p.cin_real >= 1 && p.cin_real <= 4 &&
         p.shift0 == 0 && p.zi == 0 && (p.Cout % ST_BN) == 0 && p.Hv == 2 * p.Ho && p.Wv == 2 * p.Wo &&
         (p.Ho % ST_PH) == 0 && (p.Wo % ST_PW) == 0 && p.Kpad >= 49 * 8;
}

__global__ __launch_bounds__(256) void conv_stem_bf16_kernel(const ConvParams pin) {
  const ConvParams p = conv_params_of_net(pin, (int)blockIdx.z);  // two networks in one launch: blockIdx.z = net
  chain_priority();
  constexpr int PH = ST_PH, PW = ST_PW, BN = ST_BN;
  constexpr int PR = 2 * PH + 5, PC = 2 * PW + 5;   // 21 x 69 input pixels
  constexpr int PCS = PC + 3;                       // staged row: 72 pixels of 8 bytes (the last tap pair of a row reads one past)
  constexpr int PATCH_DW = PR * PCS * 2 + 4;
  constexpr int WS = 7 * 8 * 2 + 4;                 // dwords per staged filter: 56 tap slots x 8 bytes + 16 bytes
  constexpr int NPV = PR * PC, NWV = BN * 49;
  constexpr int NLP = (NPV + 255) / 256, NLW = (NWV + 255) / 256;
  constexpr int BM = PH * PW, LDC = BN + 4;
  constexpr int LDS_DW = BM * LDC > PATCH_DW + BN * WS ? BM * LDC : PATCH_DW + BN * WS;
  __shared__ __attribute__((aligned(16))) float lds[LDS_DW];
  float* P = lds;
  float* Wl = lds + PATCH_DW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = p.Wo / PW, tiles_y = p.Ho / PH;
  const int tile = (int)blockIdx.x, n0 = (int)blockIdx.y * BN;
  const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
  const int y0 = ty * PH, x0 = tx * PW;

  const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.src0, p.src0_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);
  uint4 pv[NLP], wv[NLW];
#pragma unroll
  for (int i = 0; i < NLP; ++i) {
    const int id = tid + 256 * i;
    const int pr = id / PC, pc = id - pr * PC;
    const int gy = 2 * y0 - 3 + pr, gx = 2 * x0 - 3 + pc;
    const bool ok = id < NPV && (unsigned)gy < (unsigned)p.Hv && (unsigned)gx < (unsigned)p.Wv;
    pv[i] = buf_load16(rs, ok ? (unsigned)(((b * p.Hv + gy) * p.Wv + gx) * 8) * 2u : BUF_OOB);
  }
#pragma unroll
  for (int i = 0; i < NLW; ++i) {
    const int id = tid + 256 * i;  // (filter, tap)
    const int n = id / 49, t = id - n * 49;
    const bool ok = id < NWV && (n0 + n) < p.CoutPad;
    wv[i] = buf_load16(rw, ok ? (unsigned)((n0 + n) * p.w_ld + t * 8) * 2u : BUF_OOB);
  }
  // zero what no load writes: pixels 69..71 of every staged row, tap slot 7 of every filter row
  for (int id = tid; id < PR * 3; id += 256) {
    const int pr = id / 3, pc = PC + id - pr * 3;
    *reinterpret_cast<uint2*>(&P[(pr * PCS + pc) * 2]) = make_uint2(0u, 0u);
  }
  for (int id = tid; id < BN * 7; id += 256) {
    const int n = id / 7, kh = id - n * 7;
    *reinterpret_cast<uint2*>(&Wl[n * WS + (kh * 8 + 7) * 2]) = make_uint2(0u, 0u);
  }
#pragma unroll
  for (int i = 0; i < NLP; ++i) {
    const int id = tid + 256 * i;
    const int pr = id / PC, pc = id - pr * PC;
    if (id < NPV) *reinterpret_cast<uint2*>(&P[(pr * PCS + pc) * 2]) = make_uint2(pv[i].x, pv[i].y);
  }
#pragma unroll
  for (int i = 0; i < NLW; ++i) {
    const int id = tid + 256 * i;
    const int n = id / 49, t = id - n * 49, kh = t / 7, kw = t - kh * 7;
    if (id < NWV) *reinterpret_cast<uint2*>(&Wl[n * WS + (kh * 8 + kw) * 2]) = make_uint2(wv[i].x, wv[i].y);
  }
  __syncthreads();

  // lane: output pixel fr of its fragment, k group fq = taps (4 j + 2 fq, 4 j + 2 fq + 1) x 4 channels
  const int fr = lane & 31, fq = lane >> 5;
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // fragment i = tile row 2 * wave + i: input row 2 * (2 * wave + i) + kh, input column 2 * fr + kw
  const float* Abase = P + ((4 * wave) * PCS + 2 * fr + 2 * fq) * 2;
  const float* Bbase = Wl + fr * WS + (2 * fq) * 2;
#pragma unroll
  for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint4 bb = *reinterpret_cast<const uint4*>(Bbase + (kh * 8 + 4 * j) * 2);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint4 a = *reinterpret_cast<const uint4*>(Abase + ((2 * i + kh) * PCS + 4 * j) * 2);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a),
                                                         *reinterpret_cast<const bf16x8*>(&bb), acc[i], 0, 0, 0);
      }
    }
  }

  // epilogue: register r of fragment i = out[pixel (y0 + 2*wave + i, x0 + (r&3) + 8*(r>>2) + 4*fq)][n0 + fr]
  __syncthreads();
  float* Cs = lds;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      Cs[((2 * wave + i) * PW + (r & 3) + 8 * (r >> 2) + 4 * fq) * LDC + fr] = acc[i][r];
  __syncthreads();

  constexpr int VN = BN / 4, NVEC = BM * VN / 256, RSTEP = 256 / VN;
  const int cv = tid % VN, rv0 = tid / VN;
  const int n = n0 + cv * 4;
  const long mrow0 = ((long)b * p.Ho + y0) * p.Wo + x0;
  auto out_row = [&](int row) { return mrow0 + (long)(row / PW) * p.Wo + (row % PW); };
  bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(p.out0);
  if (p.mode == CONV_RAW_STATS) {
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
      const int row = rv0 + i * RSTEP;
      st4<bf16_t>(out + out_row(row) * p.Cout + n, *reinterpret_cast<const float4*>(&Cs[row * LDC + cv * 4]));
    }
    if (p.stats != nullptr) {  // from the fp32 accumulators, like every other train-mode epilogue
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
        p.stats[((long)tile * p.CoutPad + n0 + tid) * 2 + 0] = a1;
        p.stats[((long)tile * p.CoutPad + n0 + tid) * 2 + 1] = a2;
      }
    }
  } else {  // CONV_EVAL_FUSED: folded BatchNorm (+ ReLU)
    const float4 sc = *reinterpret_cast<const float4*>(p.scale + n), sf = *reinterpret_cast<const float4*>(p.shift + n);
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
      const int row = rv0 + i * RSTEP;
      float4 v = *reinterpret_cast<const float4*>(&Cs[row * LDC + cv * 4]);
      v.x = v.x * sc.x + sf.x; v.y = v.y * sc.y + sf.y; v.z = v.z * sc.z + sf.z; v.w = v.w * sc.w + sf.w;
      if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      st4<bf16_t>(out + out_row(row) * p.Cout + n, v);
    }
  }
}

// plan: one workgroup per 4 x 64 output tile, one statistics row per tile
void conv_patch_plan(ConvParams& p, int dtype) {
  if (p.KH == 7) {  // encoder.conv1 form: 8 x 32 tiles, 32 filters per workgroup (8: the bf16-storage kernel)
    p.patch = dtype == D3F_BF16 ? 8 : 2;
    p.nz = 1;
    p.splitk = 1;
    p.w_ld = p.Kpad;
    p.tiles_m = p.B * (p.Ho / ST_PH) * (p.Wo / ST_PW);
    p.tiles_n = p.Cout / ST_BN;
    p.stat_rows = p.tiles_m;
    return;
  }
  // 3 / 4 / 5: the bf16-storage instantiations (16 channels / 32 channels / 32 channels behind an up-sampling)
  // 6: 8 source channels staged as 16 (the head's data gradient)
  // 7: 16 -> 32 channels with the 2x2-summed epilogue (ConvParams::sum2)
  p.patch = dtype == D3F_BF16 ? (p.sum2 ? 7 : p.C0 == 32 ? (p.shift0 ? 5 : 4) : p.C0 == 8 ? 6 : 3) : 1;
  p.nz = 1;
  p.splitk = 1;
  p.w_ld = p.Kpad;
  p.tiles_m = p.B * (p.Ho / CP_PH) * (p.Wo / CP_PW);
  p.tiles_n = p.patch == 4 ? cdiv(p.Cout, 32) : 1;  // 64 filters: two 32-filter workgroups per tile (grid.y), each stages the patch
  p.stat_rows = p.tiles_m;
}

int conv_patch_launch(const ConvParams& p, hipStream_t stream) {
  if (p.patch == 2 || p.patch == 8) {
    D3F_CHECK(p.tiles_m == p.B * (p.Ho / ST_PH) * (p.Wo / ST_PW) && p.tiles_n == p.Cout / ST_BN && p.C0 == (p.patch == 8 ? 8 : 4),
              "conv: stem patch params were not planned");
    const dim3 gs((unsigned)p.tiles_m, (unsigned)p.tiles_n, (unsigned)nets_of(p.nets));
    if (p.patch == 8) hipLaunchKernelGGL(conv_stem_bf16_kernel, gs, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(conv_stem_kernel, gs, dim3(256), 0, stream, p);
    D3F_HIP(hipGetLastError());
    return 0;
  }
  D3F_CHECK((p.patch == 1 || (p.patch >= 3 && p.patch <= 7)) && p.tiles_m == p.B * (p.Ho / CP_PH) * (p.Wo / CP_PW) &&
                (p.C0 == 16 || (p.C0 == 4 && p.patch == 1) || (p.C0 == 32 && (p.patch == 4 || p.patch == 5)) || (p.C0 == 8 && p.patch == 6)) &&
                p.Cout <= (p.patch == 4 ? 64 : p.patch == 7 ? 32 : 16) && (p.sum2 != 0) == (p.patch == 7) &&
                p.tiles_n == (p.patch == 4 ? cdiv(p.Cout, 32) : 1) &&
                (p.shift0 == 0) == (p.patch != 5),
            "conv: patch params were not planned");
  const dim3 g1((unsigned)p.tiles_m, 1, (unsigned)nets_of(p.nets)), g2((unsigned)p.tiles_m, (unsigned)p.tiles_n, (unsigned)nets_of(p.nets));
  if (p.patch == 7) hipLaunchKernelGGL((conv_patch_kernel<bf16_t, 16, 32, false, 16, true>), g1, dim3(256), 0, stream, p);
  else if (p.patch == 6) hipLaunchKernelGGL((conv_patch_kernel<bf16_t, 16, 16, false, 8>), g1, dim3(256), 0, stream, p);
  else if (p.patch == 5) hipLaunchKernelGGL((conv_patch_kernel<bf16_t, 32, 16, true>), g1, dim3(256), 0, stream, p);
  else if (p.patch == 4) hipLaunchKernelGGL((conv_patch_kernel<bf16_t, 32, 32>), g2, dim3(256), 0, stream, p);
  else if (p.patch == 3) hipLaunchKernelGGL((conv_patch_kernel<bf16_t, 16, 16>), g1, dim3(256), 0, stream, p);
  else if (p.C0 == 16) hipLaunchKernelGGL((conv_patch_kernel<float, 16, 16>), g1, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((conv_patch_kernel<float, 4, 16>), g1, dim3(256), 0, stream, p);
  D3F_HIP(hipGetLastError());
  return 0;
}

}  // namespace d3f
