// Patch-resident implicit GEMM for the wide 3x3 / stride-1 / pad-1 layers in bf16 storage (round 5): the conv2d calls
// under torchvision's BasicBlock (encoder.layer1-3) and smp's DecoderBlock.conv2 that the reference dispatches at
// d3f/train_denoiser/lit_module.py:117 (forward) and under loss.backward() (data gradient).
//
// Why: in bf16 these layers are 2-3 us of matrix work, but conv_igemm_kernel re-gathers every input pixel nine times
// (once per tap) from L2 into LDS -- its k-loop is bound by the L2 -> LDS fill (0.74 us per 128 x 64 k-tile, 16 KB of it
// the activation tile), DESIGN.md section 8.  Here a workgroup owns TR x TW output pixels of one image and stages their
// input patch, (TR + 2) x (TW + 2) pixels x ALL C channels, ONCE; the k-loop then streams only the weight tiles (one tap
// x 64 channels x BN filters = 4-8 KB per k-tile, double-buffered, one barrier per iteration) and reads the A fragments
// of every tap from the resident patch at compile-time offsets -- 2-2.3x fewer L2 -> LDS bytes per launch, no per-tap
// address arithmetic, no border tests in the loop (the patch's halo holds the zeros).
//
//   v_mfma_f32_32x32x16_bf16; waves = MW (pixels) x NW (filters) x KW (k-tiles of one iteration, summed through LDS in
//   wave order: reproducible); a wave's tile is (32 FM) x 32.
//   LDS: patch [(TR+2)(TW+2)][C/2 + 4] dwords (the +16 bytes per pixel make the fragment reads conflict-free) |
//        weight stages [2][KW][BN][32 + 4] dwords; the fp32 C tile(s) of the epilogue alias the front.
//   Tried and dropped (profiles/r05_negative_pres_weights_direct.patch): the B fragments straight from L2 into registers (16
//   bytes of one filter row per lane, no LDS stages, no barrier in the loop) -- 32 rows x 32 bytes per load instruction is an
//   uncoalesced gather for the texture path: 9.7 -> 20.7 us (128 ch), 11.4 -> 20.5 (256), 10.5 -> 14.8 (64), step 3.79 -> 4.19 ms.
//   Epilogues as conv_patch.hip: raw output + statistics row per workgroup (train), folded BatchNorm (+ residual)(+ ReLU)
//   (eval), data gradient with optional accumulate and the consumer's fused BatchNorm-backward partial sums.
#include "common.h"

#include <cstdlib>

namespace d3f {

static bool pres_off() {
  static const bool off = getenv("D3F_NO_PRES_CONV") != nullptr;  // debugging knob: the implicit GEMM for these layers
  return off;
}

// S1: ONE weight stage in LDS (two barriers per iteration instead of one; the tiles still arrive two iterations ahead in
// registers) -- 9-18 KB less LDS per workgroup, for the data gradients, which run next to the weight-gradient stream
template <int C, int TW, int MW, int NW, int KW, int FM, bool S1 = false> struct PresGeo {
  static_assert(MW * NW * KW == 4, "4 waves");
  static constexpr int BM = 32 * MW * FM, BN = 32 * NW;
  static_assert(BM % TW == 0, "whole tile rows");
  static constexpr int TR = BM / TW, PR = TR + 2, PC = TW + 2;
  static constexpr int CH = C / 64;            // 64-channel chunks per tap
  static constexpr int NKT = 9 * CH;           // k-tiles (tap x chunk)
  static_assert(NKT % KW == 0, "whole iterations");
  static constexpr int NIT = NKT / KW;
  static constexpr int PIXD = C / 2 + 4;       // dwords per staged pixel
  static constexpr int BROW = 36;              // dwords per staged weight row (64 bf16 + 16 bytes)
  static constexpr int CV = C / 8;             // 16-byte vectors per pixel
  static constexpr int NPV = PR * PC * CV;     // patch vectors
  static constexpr int PATCH_DW = PR * PC * PIXD;
  static constexpr int BST_DW = KW * BN * BROW;  // one weight stage
  static constexpr int LDC = BN + 4;
  static constexpr int CT_DW = KW * BM * LDC;    // C tile(s)
  static constexpr int NST = S1 ? 1 : 2;
  static constexpr int LDS_DW = (PATCH_DW + NST * BST_DW > CT_DW) ? PATCH_DW + NST * BST_DW : CT_DW;
  static constexpr int NBV = BN * KW * 8 / 256;  // weight vectors per thread and iteration
  static_assert(NBV >= 1 && NBV * 256 == BN * KW * 8, "whole weight vectors per thread");
};

template <int C, int TW, int MW, int NW, int KW, int FM, bool S1 = false>
__global__ __launch_bounds__(256) void conv_pres_kernel(const ConvParams pin) {
  chain_priority();
  const ConvParams p = conv_params_of_net(pin, (int)blockIdx.z);  // two networks in one launch: blockIdx.z = net
  using G = PresGeo<C, TW, MW, NW, KW, FM, S1>;
  constexpr int BM = G::BM, BN = G::BN, TR = G::TR, PC = G::PC, PIXD = G::PIXD, BROW = G::BROW, CV = G::CV;
  constexpr int CH = G::CH, NIT = G::NIT, NPV = G::NPV, NBV = G::NBV, LDC = G::LDC;
  __shared__ __attribute__((aligned(16))) float lds[G::LDS_DW];
  float* P = lds;
  float* Bs = lds + G::PATCH_DW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave % KW, wmn = wave / KW, wm = wmn / NW, wn = wmn % NW;
  const int tiles_x = p.Wo / TW, tiles_y = p.Ho / TR;
  const int tile = (int)blockIdx.x, n0 = (int)blockIdx.y * BN;
  const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
  const int y0 = ty * TR, x0 = tx * TW;

  const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.src0, p.src0_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);

  // ---- weight tiles: iteration `it` = k-tiles it * KW .. it * KW + KW - 1 = k [it * KW * 64, +KW * 64) of every row
  // Two register sets: the tile of iteration it + 2 is requested while iteration it computes, so a weight tile has two
  // iterations to arrive from L2 (with one set the loop waited for the L2 latency every iteration: 4-8 MFMAs cover ~150
  // cycles of it).  The loop is fully unrolled, so the set index is a compile-time constant.
  uint4 wv[2][NBV];
  unsigned woff[NBV];  // byte offset of this thread's vectors at iteration 0 (or BUF_OOB)
  int wdst[NBV];
#pragma unroll
  for (int j = 0; j < NBV; ++j) {
    const int id = tid + 256 * j;
    const int row = id / (KW * 8), v = id - row * (KW * 8), g = v >> 3, vec = v & 7;
    woff[j] = (n0 + row) < p.CoutPad ? (unsigned)((n0 + row) * p.w_ld + g * 64 + vec * 8) * 2u : BUF_OOB;
    wdst[j] = (g * BN + row) * BROW + vec * 4;
  }
  auto load_w = [&](int it, int set) {
#pragma unroll
    for (int j = 0; j < NBV; ++j) wv[set][j] = buf_load16s(rw, woff[j], (unsigned)(it * KW * 128));
  };
  auto store_w = [&](int stage, int set) {
#pragma unroll
    for (int j = 0; j < NBV; ++j) *reinterpret_cast<uint4*>(&Bs[stage * G::BST_DW + wdst[j]]) = wv[set][j];
  };
  load_w(0, 0);
  if (NIT > 1) load_w(1, 1);

  // ---- the patch: (TR + 2) x (TW + 2) pixels x C channels, zeros outside the image -----------------------------------
  constexpr int PB = (NPV + 255) / 256;  // every vector of the patch in flight at once (9 / 13 / 14 per thread)
  for (int base = 0; base < NPV; base += 256 * PB) {
    uint4 pv[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int id = base + tid + 256 * i;
      const int pix = id / CV, cv = id - pix * CV;
      const int pr = pix / PC, pc = pix - pr * PC;
      const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
      const bool ok = id < NPV && (unsigned)gy < (unsigned)p.Hv && (unsigned)gx < (unsigned)p.Wv;
      pv[i] = buf_load16(rs, ok ? (unsigned)(((b * p.Hv + gy) * p.Wv + gx) * C + cv * 8) * 2u : BUF_OOB);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int id = base + tid + 256 * i;
      const int pix = id / CV, cv = id - pix * CV;
      if (id < NPV) *reinterpret_cast<uint4*>(&P[pix * PIXD + cv * 4]) = pv[i];
    }
  }
  store_w(0, 0);
  __syncthreads();

  // ---- k-loop ---------------------------------------------------------------------------------------------------------
  // A: lane holds channels 8 fq .. 8 fq + 7 of the 16-channel step for pixel row m = fr of its fragment; B alike for filter fr
  const int fr = lane & 31, fq = lane >> 5;
  int abase[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = (wm * FM + i) * 32 + fr;
    abase[i] = ((m / TW) * PC + (m % TW)) * PIXD + fq * 4;
  }
  const int bbase = (wk * BN + wn * 32 + fr) * BROW + fq * 4;
  f32x16 acc[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    if (it + 2 < NIT) load_w(it + 2, it & 1);  // (set it & 1 was stored to LDS at the end of iteration it - 1)
    const int t = KW == 1 ? it : __builtin_amdgcn_readfirstlane(it * KW + wk);  // this wave's k-tile: tap t / CH, chunk t % CH
    const int tap = t / CH, ch = t - tap * CH;
    const int kh = (tap * 11) >> 5, kw = tap - 3 * kh;
    const float* Ap = P + (kh * PC + kw) * PIXD + ch * 32;
    const float* Bp = Bs + (S1 ? 0 : (it & 1)) * G::BST_DW + bbase;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint4 bb = *reinterpret_cast<const uint4*>(Bp + s * 8);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const uint4 a = *reinterpret_cast<const uint4*>(Ap + abase[i] + s * 8);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a),
                                                         *reinterpret_cast<const bf16x8*>(&bb), acc[i], 0, 0, 0);
      }
    }
    if constexpr (S1) {
      __syncthreads();  // every wave is done reading the stage
      if (it + 1 < NIT) store_w(0, (it + 1) & 1);
    } else {
      if (it + 1 < NIT) store_w((it + 1) & 1, (it + 1) & 1);  // (that stage was last read in iteration it - 1, behind a barrier)
    }
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------------------------------------
  // register r of fragment i = out[tile row (wm FM + i) 32 + (r & 3) + 8 (r >> 2) + 4 fq][n0 + wn 32 + fr]
  float* Cs = lds;  // (the loop ended with a barrier: patch and stages are free)
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      Cs[wk * (BM * LDC) + ((wm * FM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fq) * LDC + wn * 32 + fr] = acc[i][r];
  __syncthreads();
  if constexpr (KW > 1) {  // the k-groups' partial tiles, summed in wave order into tile 0
    for (int e = tid * 4; e < BM * LDC; e += 256 * 4) {
      float4 v = *reinterpret_cast<const float4*>(&Cs[e]);
#pragma unroll
      for (int k = 1; k < KW; ++k) {
        const float4 w = *reinterpret_cast<const float4*>(&Cs[k * (BM * LDC) + e]);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      }
      *reinterpret_cast<float4*>(&Cs[e]) = v;
    }
    __syncthreads();
  }

  constexpr int VN = BN / 4, NVEC = BM * VN / 256, RSTEP = 256 / VN;
  static_assert(NVEC >= 1 && NVEC * 256 == BM * VN, "whole output vectors per thread");
  const int cv = tid % VN, rv0 = tid / VN;
  const int n = n0 + cv * 4;
  const bool n_ok = n < p.Cout;
  const long mrow0 = ((long)b * p.Ho + y0) * p.Wo + x0;
  auto out_row = [&](int row) { return mrow0 + (long)(row / TW) * p.Wo + (row % TW); };
  using T = bf16_t;

  if (p.mode == CONV_RAW_STATS) {
    T* __restrict__ out = reinterpret_cast<T*>(p.out0);
    if (n_ok) {
#pragma unroll
      for (int i = 0; i < NVEC; ++i) {
        const int row = rv0 + i * RSTEP;
        st4<T>(out + out_row(row) * p.Cout + n, *reinterpret_cast<const float4*>(&Cs[row * LDC + cv * 4]));
      }
    }
    if (p.stats != nullptr) {  // from the fp32 accumulators
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
    const T* __restrict__ res = reinterpret_cast<const T*>(p.res);
    if (n_ok) {
      const float4 sc = *reinterpret_cast<const float4*>(p.scale + n), sf = *reinterpret_cast<const float4*>(p.shift + n);
#pragma unroll
      for (int i = 0; i < NVEC; ++i) {
        const int row = rv0 + i * RSTEP;
        const long orow = out_row(row);
        float4 v = *reinterpret_cast<const float4*>(&Cs[row * LDC + cv * 4]);
        v.x = v.x * sc.x + sf.x; v.y = v.y * sc.y + sf.y; v.z = v.z * sc.z + sf.z; v.w = v.w * sc.w + sf.w;
        if (res != nullptr) {
          const float4 rr = ld4<T>(res + orow * p.Cout + n);
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        st4<T>(out + orow * p.Cout + n, v);
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
      // fused BatchNorm-backward reduction of the consuming layer (as conv_igemm.hip / conv_patch.hip)
      const int Cc = p.Cout;
      const T* __restrict__ yb = reinterpret_cast<const T*>(p.bn_y);
      const T* __restrict__ ab = reinterpret_cast<const T*>(p.bn_a);
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
      if (n_ok) {
        const float4 mu = *reinterpret_cast<const float4*>(p.bn_coef + n), is = *reinterpret_cast<const float4*>(p.bn_coef + Cc + n);
        const float4 sc = *reinterpret_cast<const float4*>(p.bn_coef + 2 * Cc + n), sf = *reinterpret_cast<const float4*>(p.bn_coef + 3 * Cc + n);
        const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, i4[4] = {is.x, is.y, is.z, is.w};
        const float c4[4] = {sc.x, sc.y, sc.z, sc.w}, f4[4] = {sf.x, sf.y, sf.z, sf.w};
        float4 yv[NVEC], av[NVEC];
#pragma unroll
        for (int i = 0; i < NVEC; ++i) {
          const long m = out_row(rv0 + i * RSTEP);
          yv[i] = ld4<T>(yb + m * Cc + n);
          av[i] = ab != nullptr ? ld4<T>(ab + m * Cc + n) : make_float4(0.f, 0.f, 0.f, 0.f);
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
        if (n0 + tid < Cc) {
          p.bn_partial[((long)tile * Cc + n0 + tid) * 2 + 0] = a1;
          p.bn_partial[((long)tile * Cc + n0 + tid) * 2 + 1] = a2;
        }
      }
    }
  }
}

// one configuration per channel count, sized for the 256-wide network input (layer1 at 64 x 64, layer2 at 32 x 32, layer3
// at 16 x 16, layer4 at 8 x 8 with 16 images: 512 workgroups each); other extents keep the implicit GEMM
struct PresPick {
  int id, TW, BM, BN;
};
static PresPick pres_pick(const ConvParams& p) {
  if (p.C0 == 64 && p.Wo % 64 == 0) return {1, 64, 128, 64};
  if (p.C0 == 128 && p.Wo % 32 == 0) return {2, 32, 128, 32};
  if (p.C0 == 256 && p.Wo % 16 == 0) return {3, 16, 64, 32};
  if (p.C0 == 512 && p.Wo % 8 == 0) return {4, 8, 64, 32};  // a whole 8 x 8 image per workgroup: every weight tile serves 64 pixels
  return {0, 0, 0, 0};
}

bool conv_pres_applies(const ConvParams& p, int dtype) {
  if (pres_off() || dtype != D3F_BF16) return false;
  const bool mode_ok = p.mode == CONV_RAW_STATS || p.mode == CONV_EVAL_FUSED || (p.mode == CONV_DGRAD && p.out_c0 == p.Cout);
  if (!(mode_ok && p.par == 0 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.C1 == 0 && p.shift0 == 0 &&
        p.zi == 0 && p.sum2 == 0 && p.Hv == p.Ho && p.Wv == p.Wo && p.Kpad == 9 * p.C0))
    return false;
  const PresPick k = pres_pick(p);
  if (!k.id || (p.Cout % k.BN) != 0 || (p.Ho % (k.BM / k.TW)) != 0) return false;
  const long wgs = (long)p.B * (p.Ho / (k.BM / k.TW)) * (p.Wo / k.TW) * (p.Cout / k.BN) * plan_nets_for(p.plan_nets, 4);
  // fewer workgroups: the implicit GEMM's split tiles fill the chip better (the 512-channel form holds 122 KB of LDS, one
  // workgroup per CU: 256 of them are one full round)
  // ... and many more (B = 64 eval batches: 2048): every workgroup streams the full weight matrix of its filter group, so the
  // weight traffic grows with the workgroup count and the implicit GEMM's larger tiles win (50 eval forwards of B = 64 in
  // bf16: 105.6 ms with the implicit GEMM, 114.5 ms with this kernel)
  // (round 6: the lower limit was 384 for the 64 / 128 / 256-channel forms; at 8 images per launch -- train_deep_fake's swap
  // mode, 256 workgroups each -- the implicit GEMM's split tiles are the slower choice too: 8.09 -> 7.43 ms per combined
  // swap-mode batch with this limit alone, profiles/r06_plan_mask_8_image_launches.txt)
  // (the 512-channel form, a whole 8 x 8 image x 32 filters per workgroup: from 128 -- layer4 at 8 images; 7.68 -> 7.43)
  return wgs >= (k.id == 4 ? 128 : 256) && wgs <= 1024;
}

void conv_pres_plan(ConvParams& p) {
  const PresPick k = pres_pick(p);
  p.patch = 8 + k.id;  // 9 / 10 / 11 / 12
  p.nz = 1;
  p.splitk = 1;
  p.w_ld = p.Kpad;
  p.tiles_m = p.B * (p.Ho / (k.BM / k.TW)) * (p.Wo / k.TW);
  p.tiles_n = p.Cout / k.BN;
  p.stat_rows = p.tiles_m;
}

int conv_pres_launch(const ConvParams& p, hipStream_t stream) {
  const PresPick k = pres_pick(p);
  D3F_CHECK(k.id && p.patch == 8 + k.id && p.tiles_m == p.B * (p.Ho / (k.BM / k.TW)) * (p.Wo / k.TW) && p.tiles_n == p.Cout / k.BN,
            "conv: patch-resident params were not planned");
  const dim3 grid((unsigned)p.tiles_m, (unsigned)p.tiles_n, (unsigned)nets_of(p.nets)), block(256);
  // data gradients: one weight stage (47 / 60 / 66 / 122 KB instead of 56 / 65 / 75 / 141): in the backward pass the launches
  // share the CUs with the weight-gradient stream's workgroups, and the LDS they leave free is co-residency
  // (same box: 3.388 / 3.389 / 3.407 -> 3.373 / 3.370 / 3.377 ms per bf16 step)
  if (p.mode == CONV_DGRAD) {
    if (k.id == 1) hipLaunchKernelGGL((conv_pres_kernel<64, 64, 2, 2, 1, 2, true>), grid, block, 0, stream, p);
    else if (k.id == 2) hipLaunchKernelGGL((conv_pres_kernel<128, 32, 4, 1, 1, 1, true>), grid, block, 0, stream, p);
    else if (k.id == 3) hipLaunchKernelGGL((conv_pres_kernel<256, 16, 2, 1, 2, 1, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((conv_pres_kernel<512, 8, 1, 1, 4, 2, true>), grid, block, 0, stream, p);
    D3F_HIP(hipGetLastError());
    return 0;
  }
  if (k.id == 1) hipLaunchKernelGGL((conv_pres_kernel<64, 64, 2, 2, 1, 2>), grid, block, 0, stream, p);
  else if (k.id == 2) hipLaunchKernelGGL((conv_pres_kernel<128, 32, 4, 1, 1, 1>), grid, block, 0, stream, p);
  else if (k.id == 3) hipLaunchKernelGGL((conv_pres_kernel<256, 16, 2, 1, 2, 1>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((conv_pres_kernel<512, 8, 1, 1, 4, 2>), grid, block, 0, stream, p);  // (141 KB of LDS: one workgroup per CU)
  D3F_HIP(hipGetLastError());
  return 0;
}

}  // namespace d3f
