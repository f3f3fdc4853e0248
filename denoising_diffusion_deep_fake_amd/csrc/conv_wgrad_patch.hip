// Weight gradient for the high-resolution, few-channel layers (decoder blocks 3/4, the head, the
// 7x7 stem): "patch" formulation.
//
// The tap-parallel kernel (conv_wgrad.hip) re-reads dY and X once per filter tap -- 9x (49x for the
// stem) the algorithmic bytes, which is what bounds it on these layers (1 M pixels, <= 32 channels).
// Here a workgroup stages ONE spatial tile of dY (TH x TW output pixels) and the matching halo patch
// of X ((TH-1)*S+KS) x ((TW-1)*S+KS) input pixels, gathered through the same up-sample / two-source
// description as the forward pass) in LDS, then runs every tap from LDS: the B operand of
//     dW[co][(tap, ci)] += sum_pixels dY[pixel][co] * X[pixel*S + tap][ci]
// is the patch read at a compile-time tap offset, so the inner loop is ds_read_b32 (immediate offsets)
// + MFMA only.  Each wave owns the accumulators of ALL (tap, ci) column tiles for the rows it sweeps,
// keeps them in registers across the tiles of a persistent loop, and the four waves are summed through
// LDS once at the end.  Output: one fp32 slab per workgroup in the [split][Cout][tap][Cin] layout of
// conv_wgrad.hip, reduced by the same fixed-order wgrad_reduce_kernel (bitwise reproducible).
#include "common.h"

#include <cstdlib>

namespace d3f {

constexpr int PT_TH = 8;  // tile rows: 2 per wave

template <int MT> struct PAcc;
template <> struct PAcc<32> { using T = f32x16; static constexpr int N = 16; };
template <> struct PAcc<16> { using T = f32x4; static constexpr int N = 4; };

template <int MT> __device__ __forceinline__ void pmma(typename PAcc<MT>::T& c, float a, float b);
template <> __device__ __forceinline__ void pmma<32>(f32x16& c, float a, float b) {
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ void pmma<16>(f32x4& c, float a, float b) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// MT: MFMA tile edge (32: 32x32x2, 16: 16x16x4); CO_T = MT output channels per workgroup slice;
// CI_T input channels per slice; KS x KS taps, stride S; TW tile width.
template <typename T> __device__ __forceinline__ void patch_store_f32(float* dst, const uint4& v);
template <> __device__ __forceinline__ void patch_store_f32<float>(float* dst, const uint4& v) {
  *reinterpret_cast<uint4*>(dst) = v;
}
template <> __device__ __forceinline__ void patch_store_f32<bf16_t>(float* dst, const uint4& v) {
  *reinterpret_cast<uint4*>(dst) = make_uint4(v.x << 16, v.x & 0xffff0000u, v.y << 16, v.y & 0xffff0000u);
  *reinterpret_cast<uint4*>(dst + 4) = make_uint4(v.z << 16, v.z & 0xffff0000u, v.w << 16, v.w & 0xffff0000u);
}

// T: activation type in memory; bf16 operands are widened to f32 when staged (f32 MFMA, f32 accumulate)
// CR (small-channel layers, CI_T < MT): REAL channels per tap that get an output column.  The stem stages 4 channels of which
// 3 are the image's; with CR = 3 the 49 taps are 147 columns = 5 column tiles instead of 196 -> 7 (a quarter of the MFMAs
// computed the gradient of the zero pad channel); the pad channels' slab entries are written as zeros.
template <typename T, int MT, int CI_T, int KS, int S, int TW, int CR = CI_T>
__global__ __launch_bounds__(256) void conv_wgrad_patch_kernel(const WgradParams pin) {
  const WgradParams p = wgrad_params_of_net(pin, (int)blockIdx.z);  // two networks in one launch: blockIdx.z = net
  constexpr int VE = Elem<T>::VE;
  constexpr int CO_T = MT;
  // HALFV (the stem in bf16 storage: 3 image channels padded to one 16-byte vector of 8): only the first CI_T = 4
  // channels of a pixel's vector are staged -- with 8 staged channels the kernel ran 13 column tiles instead of the fp32
  // stem's 7 and, last on its stream, 150 us of the bf16 step's tail (profiles/r03_end_bf16_step_timeline.txt)
  constexpr bool HALFV = CI_T < VE;
  static_assert(!HALFV || (CI_T == 4 && VE == 8), "half vectors: 4 of 8 bf16 channels");
  constexpr int PH = (PT_TH - 1) * S + KS, PW = (TW - 1) * S + KS;
  constexpr int KSTEP = (MT == 32) ? 2 : 4;            // pixels per MFMA
  static_assert(CR == CI_T || (CR < CI_T && CI_T < MT), "fewer columns than staged channels: small-channel path only");
  constexpr int NCOL = KS * KS * CR;                   // (tap, ci) columns
  constexpr int NNT = (NCOL + MT - 1) / MT;            // column tiles per wave
  // LDS row strides (floats).  X rows: consecutive pixels (the 2 or 4 lane groups of one MFMA operand)
  // must land on disjoint banks: stride*S == MT (mod 32) for MT = 16; any 16-byte multiple for MT = 32.
  constexpr int LXS = (CI_T >= 32) ? (MT == 16 ? CI_T + 16 : CI_T + 4) : CI_T;
  constexpr int LYS = (MT == 32) ? CO_T + 4 : CO_T;
  constexpr int XF = PH * PW * LXS, YF = PT_TH * TW * LYS;
  constexpr int RED = 4 * MT * MT;
  constexpr int LDS_F = (XF + YF > RED) ? XF + YF : RED;
  __shared__ __attribute__((aligned(16))) float lds[LDS_F];
  float* Xs = lds;
  float* Ys = lds + XF;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & (MT - 1);       // column / row inside the MFMA tile
  const int lg = lane / MT;             // pixel group inside a k-step: 0..KSTEP-1
  // channels this launch covers: all of them, or one source's (WgradParams::part: WG_SKIP = source 1 only); its slabs
  // are slab_cin wide and hold launch-local channel indices
  const int Cin = p.slab_cin;
  const int ci_slices = HALFV ? 1 : Cin / CI_T;
  const int slice = blockIdx.y;
  const int cil0 = (slice % ci_slices) * CI_T;           // launch-local
  const int ci0 = p.ci_base + cil0, co0 = (slice / ci_slices) * CO_T;
  const bool from0 = ci0 < p.C0;
  const int Cs = from0 ? p.C0 : p.C1;
  const int sh = from0 ? p.shift0 : 0;
  const int Hs = from0 ? p.H0s : p.Hv;
  const int Ws = from0 ? p.W0s : p.Wv;
  const int cl0 = from0 ? ci0 : ci0 - p.C0;
  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy, p.dy_bytes);
  const __amdgpu_buffer_rsrc_t rx = from0 ? make_rsrc(p.src0, p.src0_bytes) : make_rsrc(p.src1, p.src1_bytes);

  // per-lane column offsets into the patch for the small-channel (stem) case
  int loff[(CI_T >= MT) ? 1 : NNT];
  if (CI_T < MT) {
#pragma unroll
    for (int j = 0; j < NNT; ++j) {
      const int n = j * MT + lc;
      int tap = n / CR;
      const int c = n - tap * CR;
      if (tap >= KS * KS) tap = KS * KS - 1;  // padding columns: any valid address, never stored
      const int kh = tap / KS, kw = tap - kh * KS;
      loff[(CI_T >= MT) ? 0 : j] = (kh * PW + kw) * LXS + c;
    }
  }

  typename PAcc<MT>::T acc[NNT];
#pragma unroll
  for (int j = 0; j < NNT; ++j)
#pragma unroll
    for (int r = 0; r < PAcc<MT>::N; ++r) acc[j][r] = 0.f;

  const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + PT_TH - 1) / PT_TH;
  const int ntiles = p.B * tiles_y * tiles_x;
  constexpr int VX = HALFV ? 1 : CI_T / VE, VY = CO_T / VE;  // 16-byte global vectors per pixel row

  // staging: every load of a tile is issued into registers first (no load -> wait -> store chains),
  // and the NEXT tile's loads are issued before the MFMA sweep of the current one.
  constexpr int NVX = (PH * PW * VX + 255) / 256, NVY = (PT_TH * TW * VY + 255) / 256;
  uint4 rxv[NVX], ryv[NVY];
  auto issue_loads = [&](int t) {
    const int b = t / (tiles_y * tiles_x);
    const int tr = t - b * tiles_y * tiles_x;
    const int oy0 = (tr / tiles_x) * PT_TH, ox0 = (tr % tiles_x) * TW;
#pragma unroll
    for (int i = 0; i < NVX; ++i) {
      const int v = tid + i * 256;
      const int pix = v / VX, cv = v - pix * VX;
      const int py = pix / PW, px = pix - py * PW;
      const int iy = oy0 * S - p.pad + py, ix = ox0 * S - p.pad + px;
      const bool ok = v < PH * PW * VX && (unsigned)iy < (unsigned)p.Hv && (unsigned)ix < (unsigned)p.Wv;
      const int gp = (b * Hs + (iy >> sh)) * Ws + (ix >> sh);
      rxv[i] = buf_load16(rx, ok ? (unsigned)(gp * Cs + cl0 + cv * VE) * (unsigned)sizeof(T) : BUF_OOB);
    }
#pragma unroll
    for (int i = 0; i < NVY; ++i) {
      const int v = tid + i * 256;
      const int pix = v / VY, cv = v - pix * VY;
      const int y = pix / TW, x = pix - y * TW;
      const int oy = oy0 + y, ox = ox0 + x;
      const int co = co0 + cv * VE;
      const bool ok = v < PT_TH * TW * VY && oy < p.Ho && ox < p.Wo && co < p.Cout;
      ryv[i] = buf_load16(rdy, ok ? (unsigned)(((b * p.Ho + oy) * p.Wo + ox) * p.Cout + co) * (unsigned)sizeof(T) : BUF_OOB);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NVX; ++i) {
      const int v = tid + i * 256;
      const int pix = v / VX, cv = v - pix * VX;
      if (v < PH * PW * VX) {
        if constexpr (HALFV)
          *reinterpret_cast<uint4*>(&Xs[pix * LXS]) =
              make_uint4(rxv[i].x << 16, rxv[i].x & 0xffff0000u, rxv[i].y << 16, rxv[i].y & 0xffff0000u);
        else
          patch_store_f32<T>(&Xs[pix * LXS + cv * VE], rxv[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < NVY; ++i) {
      const int v = tid + i * 256;
      const int pix = v / VY, cv = v - pix * VY;
      if (v < PT_TH * TW * VY) patch_store_f32<T>(&Ys[pix * LYS + cv * VE], ryv[i]);
    }
  };

  if ((int)blockIdx.x < ntiles) issue_loads(blockIdx.x);
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    __syncthreads();  // previous tile's MFMA reads are done
    store_tile();
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) issue_loads(t + gridDim.x);
    // ---- all taps from LDS -----------------------------------------------------------------------
#pragma unroll
    for (int yy = 0; yy < PT_TH / 4; ++yy) {
      const int y = wave * (PT_TH / 4) + yy;
      const float* __restrict__ yrow = Ys + (y * TW + lg) * LYS + lc;
      const float* __restrict__ xrow = Xs + ((y * S) * PW + lg * S) * LXS + ((CI_T >= MT) ? lc : 0);
#pragma unroll
      for (int xs = 0; xs < TW / KSTEP; ++xs) {
        const float a = yrow[xs * KSTEP * LYS];
        const float* __restrict__ xb = xrow + xs * KSTEP * S * LXS;
#pragma unroll
        for (int j = 0; j < NNT; ++j) {
          float bv;
          if (CI_T >= MT) {
            constexpr int per_tap = CI_T / MT;   // column tiles per tap
            const int tap = j / per_tap, cb = (j % per_tap) * MT;
            const int kh = tap / KS, kw = tap % KS;
            bv = xb[(kh * PW + kw) * LXS + cb];   // compile-time offset -> ds_read immediate
          } else {
            bv = xb[loff[(CI_T >= MT) ? 0 : j]];
          }
          pmma<MT>(acc[j], a, bv);
        }
      }
    }
  }

  // ---- sum the four waves' partial accumulators, one column tile at a time -------------------------
  const int taps = KS * KS;
  float* __restrict__ slab = p.partial + (long)blockIdx.x * p.Cout * taps * Cin;
  float* red = lds;
#pragma unroll
  for (int j = 0; j < NNT; ++j) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PAcc<MT>::N; ++r) {
      // D[co][col]: col = lane % MT; MT=32: co = (r&3) + 8*(r>>2) + 4*(lane>>5); MT=16: co = 4*(lane>>4) + r
      const int co_l = (MT == 32) ? ((r & 3) + 8 * (r >> 2) + 4 * lg) : (4 * lg + r);
      red[(wave * MT + co_l) * MT + lc] = acc[j][r];
    }
    __syncthreads();
    for (int e = tid; e < MT * MT; e += 256) {
      const float s = (red[e] + red[MT * MT + e]) + (red[2 * MT * MT + e] + red[3 * MT * MT + e]);
      const int co = co0 + e / MT;
      const int n = j * MT + (e % MT);
      const int tap = n / CR, ci = cil0 + (n - tap * CR);
      if (co < p.Cout && tap < taps) {
        slab[((long)co * taps + tap) * Cin + ci] = s;
        if constexpr (HALFV && CR == CI_T) slab[((long)co * taps + tap) * Cin + ci + CI_T] = 0.f;  // the unstaged padding channels
      }
    }
  }
  if constexpr (CR < CI_T) {  // channels CR .. Cin - 1 of every (filter, tap): no column computes them -> zeros
    const int npad = Cin - CR;
    for (int e = tid; e < MT * taps * npad; e += 256) {
      const int c = CR + e % npad, q = e / npad, tap = q % taps, co = co0 + q / taps;
      if (co < p.Cout) slab[((long)co * taps + tap) * Cin + c] = 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Class form (WgradParams::cls) of the same kernel for conv(cat(upsample2x(x), skip)): the gradient for the channels of
// the up-sampled source x.  Inside output-parity class (py, px) the nine taps on the up-sampled operand touch a 2x2
// neighbourhood of the LOW-resolution x (conv_wgrad.hip, WG_CLASS), so instead of 9 taps over every pixel a workgroup
// runs 2 classes (its py, both px) x 2x2 folded taps over the pixels of those classes: 8 accumulator column groups for
// half of the tile's pixels -- 4/9 of the MACs.  blockIdx.y = (py, co slice, ci slice): the workgroup stages only the
// tile rows of parity py (half the dY tile) and the (TH/2 + 2) x (TW/2 + 2) low-resolution patch; wave w sweeps tile
// row 2w + py.  Slabs: [split][Cout][16 folded taps][slab_cin], tap = ((py*2 + px)*2 + a)*2 + b; the two py workgroups
// of a slab write disjoint taps; wgrad_reduce_kernel<true> folds them into the 3x3 gradient.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int MT, int CI_T, int TW>
__global__ __launch_bounds__(256) void conv_wgrad_patch_cls_kernel(const WgradParams pin) {
  const WgradParams p = wgrad_params_of_net(pin, (int)blockIdx.z);  // two networks in one launch: blockIdx.z = net
  constexpr int VE = Elem<T>::VE;
  constexpr int CO_T = MT;
  constexpr int RH = PT_TH / 2;                   // tile rows of one parity: one per wave
  static_assert(RH == 4, "one tile row of the parity per wave");
  constexpr int PHL = PT_TH / 2 + 2, PWL = TW / 2 + 2;  // low-resolution patch with its halo
  constexpr int KSTEP = (MT == 32) ? 2 : 4;       // pixels per MFMA
  constexpr int NCT = CI_T / MT;                  // column tiles per folded tap
  constexpr int NNT = 8 * NCT;                    // (px, a, b) x column tiles
  static_assert(CI_T % MT == 0, "whole column tiles per folded tap");
  constexpr int LXS = (MT == 16 ? CI_T + 16 : CI_T + 4);
  constexpr int LYS = (MT == 32) ? CO_T + 4 : CO_T;
  constexpr int XF = PHL * PWL * LXS, YF = RH * TW * LYS;
  constexpr int RED = 4 * MT * MT;
  constexpr int LDS_F = (XF + YF > RED) ? XF + YF : RED;
  __shared__ __attribute__((aligned(16))) float lds[LDS_F];
  float* Xs = lds;
  float* Ys = lds + XF;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & (MT - 1);
  const int lg = lane / MT;
  const int Cin = p.slab_cin;                     // = C0: the class form covers source 0
  const int ci_slices = Cin / CI_T;
  int slice = blockIdx.y;
  const int py = slice & 1;
  slice >>= 1;
  const int ci0 = (slice % ci_slices) * CI_T, co0 = (slice / ci_slices) * CO_T;
  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy, p.dy_bytes);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.src0, p.src0_bytes);

  typename PAcc<MT>::T acc[NNT];
#pragma unroll
  for (int j = 0; j < NNT; ++j)
#pragma unroll
    for (int r = 0; r < PAcc<MT>::N; ++r) acc[j][r] = 0.f;

  const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + PT_TH - 1) / PT_TH;
  const int ntiles = p.B * tiles_y * tiles_x;
  constexpr int VX = CI_T / VE, VY = CO_T / VE;
  constexpr int NVX = (PHL * PWL * VX + 255) / 256, NVY = (RH * TW * VY + 255) / 256;
  uint4 rxv[NVX], ryv[NVY];
  auto issue_loads = [&](int t) {
    const int b = t / (tiles_y * tiles_x);
    const int tr = t - b * tiles_y * tiles_x;
    const int oy0 = (tr / tiles_x) * PT_TH, ox0 = (tr % tiles_x) * TW;
    const int ly0 = oy0 / 2 - 1, lx0 = ox0 / 2 - 1;  // low-resolution patch origin
#pragma unroll
    for (int i = 0; i < NVX; ++i) {
      const int v = tid + i * 256;
      const int pix = v / VX, cv = v - pix * VX;
      const int yy = pix / PWL, xx = pix - yy * PWL;
      const int iy = ly0 + yy, ix = lx0 + xx;
      const bool ok = v < PHL * PWL * VX && (unsigned)iy < (unsigned)p.H0s && (unsigned)ix < (unsigned)p.W0s;
      const int gp = (b * p.H0s + iy) * p.W0s + ix;
      rxv[i] = buf_load16(rx, ok ? (unsigned)(gp * p.C0 + ci0 + cv * VE) * (unsigned)sizeof(T) : BUF_OOB);
    }
#pragma unroll
    for (int i = 0; i < NVY; ++i) {
      const int v = tid + i * 256;
      const int pix = v / VY, cv = v - pix * VY;
      const int r = pix / TW, x = pix - r * TW;
      const int oy = oy0 + 2 * r + py, ox = ox0 + x;
      const int co = co0 + cv * VE;
      const bool ok = v < RH * TW * VY && oy < p.Ho && ox < p.Wo && co < p.Cout;
      ryv[i] = buf_load16(rdy, ok ? (unsigned)(((b * p.Ho + oy) * p.Wo + ox) * p.Cout + co) * (unsigned)sizeof(T) : BUF_OOB);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NVX; ++i) {
      const int v = tid + i * 256;
      const int pix = v / VX, cv = v - pix * VX;
      if (v < PHL * PWL * VX) patch_store_f32<T>(&Xs[pix * LXS + cv * VE], rxv[i]);
    }
#pragma unroll
    for (int i = 0; i < NVY; ++i) {
      const int v = tid + i * 256;
      const int pix = v / VY, cv = v - pix * VY;
      if (v < RH * TW * VY) patch_store_f32<T>(&Ys[pix * LYS + cv * VE], ryv[i]);
    }
  };

  if ((int)blockIdx.x < ntiles) issue_loads(blockIdx.x);
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) issue_loads(t + gridDim.x);
    // wave w: tile row 2w + py -> low-resolution row j = w (tile-local); source row of folded tap a: j + a - 1 + py,
    // i.e. patch row w + a + py; source column of (px, b) for the pixel at low-resolution column i: i + b - 1 + px,
    // i.e. patch column i + b + px
    const float* __restrict__ yrow = Ys + (wave * TW) * LYS + lc;
#pragma unroll
    for (int px = 0; px < 2; ++px) {
#pragma unroll
      for (int xs = 0; xs < (TW / 2) / KSTEP; ++xs) {
        const int il = lg + xs * KSTEP;                      // low-resolution column of this lane's pixel (tile-local)
        const float a = yrow[(2 * il + px) * LYS];
#pragma unroll
        for (int fa = 0; fa < 2; ++fa)
#pragma unroll
          for (int fb = 0; fb < 2; ++fb) {
            const float* __restrict__ xb = Xs + ((wave + fa + py) * PWL + il + fb + px) * LXS + lc;
#pragma unroll
            for (int jc = 0; jc < NCT; ++jc) pmma<MT>(acc[((px * 2 + fa) * 2 + fb) * NCT + jc], a, xb[jc * MT]);
          }
      }
    }
  }

  // ---- sum the four waves' partial accumulators, one column tile at a time -------------------------
  float* __restrict__ slab = p.partial + (long)blockIdx.x * p.Cout * 16 * Cin;
  float* red = lds;
#pragma unroll
  for (int j = 0; j < NNT; ++j) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PAcc<MT>::N; ++r) {
      const int co_l = (MT == 32) ? ((r & 3) + 8 * (r >> 2) + 4 * lg) : (4 * lg + r);
      red[(wave * MT + co_l) * MT + lc] = acc[j][r];
    }
    __syncthreads();
    const int f8 = j / NCT, jc = j - f8 * NCT;           // f8 = (px*2 + a)*2 + b
    const int tap = py * 8 + f8;                          // ((py*2 + px)*2 + a)*2 + b
    for (int e = tid; e < MT * MT; e += 256) {
      const float s = (red[e] + red[MT * MT + e]) + (red[2 * MT * MT + e] + red[3 * MT * MT + e]);
      const int co = co0 + e / MT;
      const int ci = ci0 + jc * MT + (e % MT);
      if (co < p.Cout) slab[((long)co * 16 + tap) * Cin + ci] = s;
    }
  }
}

// variants: 1: 3x3 s1, <=16 out, 32-channel slices (decoder 4 conv1)   2: 3x3 s1, <=16 out, 16-channel slices
//           3: 3x3 s1, 32-out slices, 32-channel slices (decoder 3)     4: 7x7 s2 stem, 4 (3+pad) channels
//           6: as 5 with at most 4 real channels: 4 of the 8 are staged      5: 7x7 s2 stem in bf16 storage, 8 (3+pad)
//              channels (the tap-parallel kernel took 725 us for this
//              layer -- 49 taps x 2 filter tiles re-reading the 128x128 dY 49 times -- and sat at the very end of the
//              weight-gradient stream: profiles/r03_bf16_step_timeline.txt)

// ---------------------------------------------------------------------------------------------------------------------
// bf16 storage, native bf16 matrix cores (round 5).  The kernels above widen bf16 operands to fp32 in LDS and run the fp32
// MFMA: in bf16 storage the narrow layers' weight gradients were ~0.5 ms of a weight-gradient stream that is as long as
// the dependent chain in the backward window (profiles/README.md, round 5).  Here both operands stay bf16: the staged
// images are [pixel][32 channels] rows of 64 bytes (16-channel sources fill the first half of a row, the rest is zero),
// and the MFMA operands -- 8 consecutive PIXELS of one channel per lane, k = pixel -- are gathered by the transposing LDS
// read ds_read_b64_tr_b16 exactly as in conv_wgrad_kernel's bf16 loop (two reads per 32 x 16 fragment); a tap is a
// constant byte offset (kh * PW + kw) * 64 on the patch.  One v_mfma_f32_32x32x16_bf16 per (tile row, tap): 16 pixels x
// 32 output channels x 32 input channels.  Every 3x3 / stride-1 layer with <= 32 filters and 16 or 32 k input channels
// runs this ONE shape (narrower sources / filters are zero-padded in LDS; an up-sampled source is gathered at (y >> 1,
// x >> 1) while staging, so the class-form kernels are not needed: the 9 taps are 9 cheap MFMAs).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PB_TW = 16;
__global__ __launch_bounds__(256) void conv_wgrad_patch_bf16_kernel(const WgradParams pin) {
  const WgradParams p = wgrad_params_of_net(pin, (int)blockIdx.z);  // two networks in one launch: blockIdx.z = net
  typedef short v4s __attribute__((ext_vector_type(4)));
  constexpr int TW = PB_TW, KS = 3, PH = PT_TH + 2, PW = TW + 2;
  constexpr int XB = PH * PW * 64, YB = PT_TH * TW * 64;   // bytes
  constexpr int RED = 4 * 32 * 32;                           // floats: the four waves' accumulators of one tap
  constexpr int LDS_B = (XB + YB > RED * 4) ? XB + YB : RED * 4;
  __shared__ __attribute__((aligned(16))) unsigned char ldsb[LDS_B];
  unsigned char* Xs = ldsb;
  unsigned char* Ys = ldsb + XB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Cin = p.slab_cin;
  const int ci_slices = (Cin + 31) / 32;
  const int slice = blockIdx.y;
  const int cil0 = (slice % ci_slices) * 32;             // launch-local first input channel of this slice
  const int ci0 = p.ci_base + cil0, co0 = (slice / ci_slices) * 32;
  const bool from0 = ci0 < p.C0;
  const int Cs = from0 ? p.C0 : p.C1;
  const int sh = from0 ? p.shift0 : 0;
  const int Hs = from0 ? p.H0s : p.Hv;
  const int Ws = from0 ? p.W0s : p.Wv;
  const int cl0 = from0 ? ci0 : ci0 - p.C0;
  const int xreal = min(32, Cs - cl0), yreal = min(32, p.Cout - co0);   // real channels of this slice (multiples of 8)
  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy, p.dy_bytes);
  const __amdgpu_buffer_rsrc_t rx = from0 ? make_rsrc(p.src0, p.src0_bytes) : make_rsrc(p.src1, p.src1_bytes);

  f32x16 acc[KS * KS];
#pragma unroll
  for (int j = 0; j < KS * KS; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + PT_TH - 1) / PT_TH;
  const int ntiles = p.B * tiles_y * tiles_x;
  constexpr int NVX = (PH * PW * 4 + 255) / 256, NVY = (PT_TH * TW * 4 + 255) / 256;  // 16-byte vectors per thread
  uint4 rxv[NVX], ryv[NVY];
  auto issue_loads = [&](int t) {
    const int b = t / (tiles_y * tiles_x);
    const int tr = t - b * tiles_y * tiles_x;
    const int oy0 = (tr / tiles_x) * PT_TH, ox0 = (tr % tiles_x) * TW;
#pragma unroll
    for (int i = 0; i < NVX; ++i) {
      const int v = tid + i * 256;
      const int pix = v >> 2, cv = v & 3;
      const int py = pix / PW, px = pix - py * PW;
      const int iy = oy0 - p.pad + py, ix = ox0 - p.pad + px;
      const bool ok = v < PH * PW * 4 && cv * 8 < xreal && (unsigned)iy < (unsigned)p.Hv && (unsigned)ix < (unsigned)p.Wv;
      const int gp = (b * Hs + (iy >> sh)) * Ws + (ix >> sh);
      rxv[i] = buf_load16(rx, ok ? (unsigned)(gp * Cs + cl0 + cv * 8) * 2u : BUF_OOB);
    }
#pragma unroll
    for (int i = 0; i < NVY; ++i) {
      const int v = tid + i * 256;
      const int pix = v >> 2, cv = v & 3;
      const int y = pix / TW, x = pix - y * TW;
      const int oy = oy0 + y, ox = ox0 + x;
      const bool ok = v < PT_TH * TW * 4 && cv * 8 < yreal && oy < p.Ho && ox < p.Wo;
      ryv[i] = buf_load16(rdy, ok ? (unsigned)(((b * p.Ho + oy) * p.Wo + ox) * p.Cout + co0 + cv * 8) * 2u : BUF_OOB);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NVX; ++i) {
      const int v = tid + i * 256;
      if (v < PH * PW * 4) *reinterpret_cast<uint4*>(Xs + v * 16) = rxv[i];   // [pixel][4 vectors]: 64-byte rows
    }
#pragma unroll
    for (int i = 0; i < NVY; ++i) {
      const int v = tid + i * 256;
      if (v < PT_TH * TW * 4) *reinterpret_cast<uint4*>(Ys + v * 16) = ryv[i];
    }
  };
  // transposing fragment gather (conv_wgrad.hip): the 16-lane group g reads 4 pixel rows x 16 channels, lane 4 q + pq
  // supplies row q, channels 4 pq ..; two reads = 8 consecutive pixels of the lane's channel
  const int lg = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
  const int rd_lane = ((lg >> 1) * 8 + q) * 64 + ((lg & 1) * 16 + 4 * pq) * 2;
  auto frag = [&](const unsigned char* base) {
    union { uint4 u; v4s h[2]; } f;
#pragma unroll
    for (int h = 0; h < 2; ++h)
      f.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(base + rd_lane + h * 4 * 64));
    return f.u;
  };

  if ((int)blockIdx.x < ntiles) issue_loads(blockIdx.x);
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    __syncthreads();  // the previous tile's fragment reads are done
    store_tile();
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) issue_loads(t + gridDim.x);
#pragma unroll
    for (int yy = 0; yy < PT_TH / 4; ++yy) {
      const int y = wave * (PT_TH / 4) + yy;
      const uint4 a = frag(Ys + (y * TW) * 64);
#pragma unroll
      for (int tap = 0; tap < KS * KS; ++tap) {
        const int kh = tap / KS, kw = tap - kh * KS;
        const uint4 b = frag(Xs + ((y + kh) * PW + kw) * 64);
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a),
                                                           *reinterpret_cast<const bf16x8*>(&b), acc[tap], 0, 0, 0);
      }
    }
  }

  // ---- sum the four waves' partial accumulators, one tap at a time; D[co][ci]: ci = lane & 31, co = (r&3) + 8 (r>>2) + 4 (lane>>5)
  const int taps = KS * KS;
  float* __restrict__ slab = p.partial + (long)blockIdx.x * p.Cout * taps * Cin;
  float* red = reinterpret_cast<float*>(ldsb);
  const int lc = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int j = 0; j < KS * KS; ++j) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lc] = acc[j][r];
    __syncthreads();
    for (int e = tid; e < 32 * 32; e += 256) {
      const float sum = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
      const int co = co0 + (e >> 5), cil = e & 31;
      if (co < p.Cout && cil < xreal) slab[((long)co * taps + j) * Cin + cil0 + cil] = sum;
    }
  }
}

int wgrad_patch_variant(const WgradParams& p, int dtype) {
  const int cin = p.C0 + p.C1;
  if (p.KH != p.KW) return 0;
  // bf16 storage: every narrow 3x3 stride-1 layer on the native bf16 kernel (7), whole layers only (no class form)
  // ... up to 128 filters and 384 input channels (layer1, layer2, decoder blocks 1-4): a workgroup owns one (32 filters x 32
  // channels) pair, so dY / X are read Cin/32 / Cout/32 times instead of the tap-parallel kernel's 9; round-5 sweep of the
  // limits (filters, channels) = (32,128) / (64,128) / (128,128) / (128,384) / (256,768) / (512,768): 3.457 / 3.427 / 3.409 and,
  // on another box, 3.465 (128,128) / 3.435 / 3.432 / 3.450 ms per bf16 step
  constexpr int maxco = 128;
  constexpr int maxci = 384;
  if (dtype == D3F_BF16 && p.KH == 3 && p.stride == 1 && p.pad == 1 && p.Cout <= maxco && (cin == 16 || (cin % 32 == 0 && cin <= maxci)) &&
      (p.C1 == 0 || p.C0 % 32 == 0) && (p.C0 % 8) == 0 && (p.C1 % 8) == 0 && (p.Cout % 8) == 0)
    return 7;
  if (p.KH == 3 && p.stride == 1 && p.pad == 1) {
    if (p.Cout <= 16 && cin % 32 == 0 && cin <= 64 && (p.C1 == 0 || p.C0 % 32 == 0)) return 1;
    if (p.Cout <= 16 && cin == 16) return 2;
    if (p.Cout == 32 && cin % 32 == 0 && cin <= 128 && (p.C1 == 0 || p.C0 % 32 == 0)) return 3;
  }
  if (dtype == D3F_F32 && p.KH == 7 && p.stride == 2 && p.pad == 3 && cin == 4 && p.C1 == 0 &&
      p.Cout % 32 == 0 && p.Cout <= 64)
    return 4;  // f32: 3 channels padded to 4
  if (dtype == D3F_BF16 && p.KH == 7 && p.stride == 2 && p.pad == 3 && cin == 8 && p.C1 == 0 &&
      p.Cout % 32 == 0 && p.Cout <= 64)  // bf16 pads the stem input to 8 channels (one 16-byte vector per pixel)
    return (p.cin_real > 0 && p.cin_real <= 4) ? 6 : 5;  // 6: only the first 4 are staged (RGB: 3 real channels)
  return 0;
}

void wgrad_patch_grid(const WgradParams& p, int variant, int* gx, int* gy) {
  const int cin = p.part == WG_WHOLE ? p.C0 + p.C1 : p.part == WG_CLASS ? p.C0 : p.C1;  // channels of this launch
  const int ci_t = variant == 1 ? 32 : variant == 2 ? 16 : (variant == 3 || variant == 7) ? 32 : (variant == 5 || variant == 6) ? 8 : 4;
  const int co_t = (variant == 3 || variant == 4 || variant == 5 || variant == 6 || variant == 7) ? 32 : 16;
  const int slices = cdiv(cin, ci_t) * cdiv(p.Cout, co_t) * (p.part == WG_CLASS ? 2 : 1);  // class form: x 2 row parities
  const int tiles = p.B * cdiv(p.Ho, PT_TH) * cdiv(p.Wo, 16);
  // ~3 workgroups per CU in total; the native bf16 kernel (7): ONE per CU -- its tiles are 18 MFMAs per wave, so fewer,
  // longer-lived workgroups amortise the nine-tap epilogue and write a third of the slabs (sweep 128 / 192 / 256 / 320 / 384 /
  // 512 / 768 / 1024 workgroups: 3.64 / 3.55 / 3.46 / 3.53 / 3.48 / 3.49-3.54 / 3.52 / 3.55 ms per bf16 step)
  constexpr int g7 = 256;
  // the fp32-MFMA kernels: 2 per CU (round 5 sweep 384 / 448 / 512 / 576 / 640 / 768 / 1024 workgroups: 7.85 / 7.81 / 7.80 /
  // 7.94 / 7.91 / 7.875 / 7.91 ms per fp32 step; rounds 1-4 ran 768)
  static const int gall = prof_knob("D3F_WGRAD_PATCH_WGS") ? atoi(prof_knob("D3F_WGRAD_PATCH_WGS")) : 512;  // sweep knob
  int g = (variant == 7 ? g7 : gall) / plan_nets_for(p.plan_nets, 32) / slices;  // (two networks in one launch share the count)
  if (g > tiles) g = tiles;
  if (g < 1) g = 1;
  *gx = g;
  *gy = slices;
}

template <typename T> static int patch_launch_t(const WgradParams& p, int variant, dim3 grid, hipStream_t stream) {
  const dim3 block(256);
  switch (variant) {
    case 1: hipLaunchKernelGGL((conv_wgrad_patch_kernel<T, 16, 32, 3, 1, 16>), grid, block, 0, stream, p); break;
    case 2: hipLaunchKernelGGL((conv_wgrad_patch_kernel<T, 16, 16, 3, 1, 16>), grid, block, 0, stream, p); break;
    case 3: hipLaunchKernelGGL((conv_wgrad_patch_kernel<T, 32, 32, 3, 1, 16>), grid, block, 0, stream, p); break;
    default: return set_error(-1, "wgrad patch: bad variant %d", variant);
  }
  return 0;
}

int wgrad_patch_launch(const WgradParams& p, int variant, int dtype, hipStream_t stream) {
  int gx, gy;
  wgrad_patch_grid(p, variant, &gx, &gy);
  D3F_CHECK(p.splits == gx, "wgrad patch: params were not planned (splits %d vs %d)", p.splits, gx);
  const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)nets_of(p.nets)), block(256);
  if (p.cls) {
    D3F_CHECK((variant == 1 || variant == 3) && p.slab_cin == p.C0 && (p.C0 % 32) == 0 && (p.Ho % 2) == 0 && (p.Wo % 2) == 0,
              "wgrad patch: class form needs variant 1 or 3 and whole 32-channel slices of source 0");
    if (dtype == D3F_F32) {
      if (variant == 1) hipLaunchKernelGGL((conv_wgrad_patch_cls_kernel<float, 16, 32, 16>), grid, block, 0, stream, p);
      else hipLaunchKernelGGL((conv_wgrad_patch_cls_kernel<float, 32, 32, 16>), grid, block, 0, stream, p);
    } else {
      if (variant == 1) hipLaunchKernelGGL((conv_wgrad_patch_cls_kernel<bf16_t, 16, 32, 16>), grid, block, 0, stream, p);
      else hipLaunchKernelGGL((conv_wgrad_patch_cls_kernel<bf16_t, 32, 32, 16>), grid, block, 0, stream, p);
    }
    D3F_HIP(hipGetLastError());
    return 0;
  }
  if (variant == 7) {
    D3F_CHECK(dtype == D3F_BF16 && p.part == WG_WHOLE, "wgrad patch: variant 7 is the native bf16 kernel of whole layers");
    hipLaunchKernelGGL(conv_wgrad_patch_bf16_kernel, grid, block, 0, stream, p);
  } else if (variant == 4) {
    D3F_CHECK(dtype == D3F_F32, "wgrad patch: stem variant 4 is the f32 one");
    if (p.cin_real == 3) hipLaunchKernelGGL((conv_wgrad_patch_kernel<float, 32, 4, 7, 2, 16, 3>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_patch_kernel<float, 32, 4, 7, 2, 16>), grid, block, 0, stream, p);
  } else if (variant == 5) {
    D3F_CHECK(dtype == D3F_BF16, "wgrad patch: stem variant 5 is the bf16 one");
    hipLaunchKernelGGL((conv_wgrad_patch_kernel<bf16_t, 32, 8, 7, 2, 16>), grid, block, 0, stream, p);
  } else if (variant == 6) {
    D3F_CHECK(dtype == D3F_BF16, "wgrad patch: stem variant 6 is a bf16 one");
    if (p.cin_real == 3) hipLaunchKernelGGL((conv_wgrad_patch_kernel<bf16_t, 32, 4, 7, 2, 16, 3>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_patch_kernel<bf16_t, 32, 4, 7, 2, 16>), grid, block, 0, stream, p);
  } else if (dtype == D3F_F32) {
    if (int rc = patch_launch_t<float>(p, variant, grid, stream)) return rc;
  } else {
    if (int rc = patch_launch_t<bf16_t>(p, variant, grid, stream)) return rc;
  }
  D3F_HIP(hipGetLastError());
  return 0;
}

}  // namespace d3f
