// Implicit-GEMM convolution on the CDNA4 matrix cores: forward and data-gradient.
//
//   out[m][n] = sum_k A[m][k] * W[n][k],   m = (b, oy, ox),  n = cout,  k = (kh, kw, cin)
//
// A is never materialised: each 16-byte vector of an A row is gathered straight from the
// NHWC activation(s) -- optionally from two channel-concatenated tensors, the first one
// read through a nearest x2 up-sample (decoder "upsample + cat" fused into the loader) or
// through zero insertion (data-gradient of a stride-2 conv).  Replaces the ATen/cuDNN
// conv2d + F.interpolate + torch.cat the reference dispatches under
// segmentation_models_pytorch.Unet (d3f/train_denoiser/lit_module.py:46-52, :117).
//
// Tiling: 256 threads = 4 waves; block tile BM x BN, k-tile = one 128-byte row per m / n
// (32 f32 or 64 bf16).  LDS rows are padded to 144 bytes so that the ds_read_b128 fragment
// reads (16 rows x one 16-byte chunk per lane group) are bank-conflict free.  Fragments are
// fed to v_mfma_f32_32x32x2_f32 / 16x16x4_f32 (exact f32) or the bf16 32x32x16 / 16x16x32
// forms; the per-lane k order is permuted identically for A and W, which leaves the sum
// unchanged.  Global loads for k-tile t+1 are issued before the MFMAs of tile t
// (register-staged prefetch, one LDS buffer, two barriers per k-tile).
#include "common.h"

#include <cstdlib>

namespace d3f {

template <typename T, int MT> struct Mma;
template <> struct Mma<float, 32> {
  using Acc = f32x16;
  static constexpr int NREG = 16;
  static __device__ __forceinline__ void run(Acc& c, const uint4& a, const uint4& b) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};
template <> struct Mma<float, 16> {
  using Acc = f32x4;
  static constexpr int NREG = 4;
  static __device__ __forceinline__ void run(Acc& c, const uint4& a, const uint4& b) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};
template <> struct Mma<bf16_t, 32> {
  using Acc = f32x16;
  static constexpr int NREG = 16;
  static __device__ __forceinline__ void run(Acc& c, const uint4& a, const uint4& b) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a),
                                                *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
  }
};
template <> struct Mma<bf16_t, 16> {
  using Acc = f32x4;
  static constexpr int NREG = 4;
  static __device__ __forceinline__ void run(Acc& c, const uint4& a, const uint4& b) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a),
                                                *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
  }
};

constexpr int LDS_ROW = 36;  // dwords per LDS row: 128 B of data + 16 B pad

template <typename T, int BM, int BN, int WGM, int WGN, int MT, bool SMALLC>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
  constexpr int VE = Elem<T>::VE, BKE = Elem<T>::BKE;
  constexpr int TM = BM / WGM, TN = BN / WGN, FM = TM / MT, FN = TN / MT;
  constexpr int NVA = BM * 8 / 256;
  constexpr int NVB = (BN * 8 + 255) / 256;
  constexpr int NS = (MT == 32) ? 4 : 2;  // fragment reads per k-tile
  static_assert(WGM * WGN == 4, "4 waves");
  static_assert(TM % MT == 0 && TN % MT == 0, "wave tile");
  using M_ = Mma<T, MT>;
  using Acc = typename M_::Acc;

  __shared__ __attribute__((aligned(16))) uint32_t lds[(BM + BN) * LDS_ROW];
  uint32_t* As = lds;
  uint32_t* Bs = lds + BM * LDS_ROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int tile_n = blockIdx.x % p.tiles_n, tile_m = blockIdx.x / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int chunk = tid & 7, rbase = tid >> 3;

  const int Cin = p.C0 + p.C1;

  // ---- per-row output pixel -> input origin -----------------------------------------
  int iy0[NVA], ix0[NVA], bidx[NVA];
  {
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int m = m0 + rbase + 32 * i;
      if (m < p.M) {
        const int b = m / HoWo;
        const int r = m - b * HoWo;
        const int oy = r / p.Wo;
        const int ox = r - oy * p.Wo;
        bidx[i] = b;
        iy0[i] = oy * p.stride - p.pad;
        ix0[i] = ox * p.stride - p.pad;
      } else {
        bidx[i] = 0;
        iy0[i] = -(1 << 24);  // fails every bounds test
        ix0[i] = 0;
      }
    }
  }

  Acc acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < M_::NREG; ++r) acc[i][j][r] = 0.f;

  const int nk_total = p.Kpad / BKE;
  // split-K: grid.y cuts the k-tile range; each slice writes raw accumulators to its slab
  const int kt_begin = (int)((long)nk_total * blockIdx.y / p.splitk);
  const int kt_end = (int)((long)nk_total * (blockIdx.y + 1) / p.splitk);
  // running (tap, channel) position of the k-tile (regular mode: wave-uniform)
  int t_kh = 0, t_kw = 0, t_c = 0;
  if (!SMALLC && kt_begin > 0) {
    const int cpos = kt_begin * BKE;
    const int tap = cpos / Cin;
    t_c = cpos - tap * Cin;
    t_kh = tap / p.KW;
    t_kw = tap - t_kh * p.KW;
  }

  // Buffer descriptors (wave-uniform): lanes whose tap falls outside the image, whose row is past M
  // or whose weight row is past CoutPad get the offset BUF_OOB and read zeros in hardware -- no
  // divergent branch, no select after the load, so all loads of a k-tile are in flight together.
  const __amdgpu_buffer_rsrc_t r0 = make_rsrc(p.src0, p.src0_bytes);
  const __amdgpu_buffer_rsrc_t r1 = make_rsrc(p.src1 != nullptr ? p.src1 : p.src0, p.src1_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);
  const int w_row_bytes = p.Kpad * (int)sizeof(T);
  unsigned woff[NVB];
#pragma unroll
  for (int j = 0; j < NVB; ++j) {
    const int row = rbase + 32 * j;
    const int n = n0 + row;
    woff[j] = (row < BN && n < p.CoutPad) ? (unsigned)(n * w_row_bytes + chunk * 16) : BUF_OOB;
  }

  uint4 ra[NVA], rb[NVB];
  auto load_tile = [&](int kt) {
    int kh, kw, c;
    bool tapvalid = true;
    bool from0 = true;  // wave-uniform by construction (plan checks C0 % BKE == 0 when C1 > 0)
    if (SMALLC) {
      const int kk = kt * BKE + chunk * VE;
      const int tap = kk / Cin;
      c = kk - tap * Cin;
      kh = tap / p.KW;
      kw = tap - kh * p.KW;
      tapvalid = tap < p.KH * p.KW;
    } else {
      kh = t_kh;
      kw = t_kw;
      from0 = t_c < p.C0;
      c = (from0 ? t_c : t_c - p.C0) + chunk * VE;
      t_c += BKE;
      if (t_c >= Cin) {
        t_c = 0;
        if (++t_kw == p.KW) { t_kw = 0; ++t_kh; }
      }
    }
    const int Cs = from0 ? p.C0 : p.C1;
    const int sh = from0 ? p.shift0 : 0;
    const int Hs = from0 ? p.H0s : p.Hv;
    const int Ws = from0 ? p.W0s : p.Wv;
    const bool parity = from0 && p.zi;
    unsigned off[NVA];
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int iy = iy0[i] + kh, ix = ix0[i] + kw;
      bool v = tapvalid && (unsigned)iy < (unsigned)p.Hv && (unsigned)ix < (unsigned)p.Wv;
      if (parity) v = v && (((iy | ix) & 1) == 0);
      const int pix = (bidx[i] * Hs + (iy >> sh)) * Ws + (ix >> sh);
      off[i] = v ? (unsigned)(pix * Cs + c) * (unsigned)sizeof(T) : BUF_OOB;
    }
    if (from0) {
#pragma unroll
      for (int i = 0; i < NVA; ++i) ra[i] = buf_load16(r0, off[i]);
    } else {
#pragma unroll
      for (int i = 0; i < NVA; ++i) ra[i] = buf_load16(r1, off[i]);
    }
#pragma unroll
    for (int j = 0; j < NVB; ++j)
      rb[j] = buf_load16(rw, woff[j] == BUF_OOB ? BUF_OOB : woff[j] + (unsigned)(kt * BKE) * (unsigned)sizeof(T));
  };

  load_tile(kt_begin);
  const int fr = (MT == 32) ? (lane & 31) : (lane & 15);
  const int fq = (MT == 32) ? (lane >> 5) : (lane >> 4);

  for (int kt = kt_begin; kt < kt_end; ++kt) {
    // stage the prefetched tile
#pragma unroll
    for (int i = 0; i < NVA; ++i)
      *reinterpret_cast<uint4*>(&As[(rbase + 32 * i) * LDS_ROW + chunk * 4]) = ra[i];
#pragma unroll
    for (int j = 0; j < NVB; ++j) {
      const int row = rbase + 32 * j;
      if (row < BN) *reinterpret_cast<uint4*>(&Bs[row * LDS_ROW + chunk * 4]) = rb[j];
    }
    __syncthreads();
    if (kt + 1 < kt_end) load_tile(kt + 1);

#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int ch = (MT == 32) ? (2 * s + fq) : (4 * s + fq);
      uint4 a[FM], b[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
        a[i] = *reinterpret_cast<const uint4*>(&As[(wm * TM + i * MT + fr) * LDS_ROW + ch * 4]);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        b[j] = *reinterpret_cast<const uint4*>(&Bs[(wn * TN + j * MT + fr) * LDS_ROW + ch * 4]);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) M_::run(acc[i][j], a[i], b[j]);
    }
    __syncthreads();
  }

  // ---- epilogue ----------------------------------------------------------------------
  // accumulator element (i, j, r) of this lane = out[m][n] with
  //   MT=32: n_l = lane&31, m_l = (r&3) + 8*(r>>2) + 4*(lane>>5)
  //   MT=16: n_l = lane&15, m_l = 4*(lane>>4) + r
  const int n_l = fr;
  auto m_local = [&](int r) { return (MT == 32) ? ((r & 3) + 8 * (r >> 2) + 4 * fq) : (4 * fq + r); };

  if (p.splitk > 1) {
    float* __restrict__ slab = p.partial + (long)blockIdx.y * p.M * p.Cout;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn * TN + j * MT + n_l;
#pragma unroll
        for (int r = 0; r < M_::NREG; ++r) {
          const int m = m0 + wm * TM + i * MT + m_local(r);
          if (m < p.M && n < p.Cout) slab[(long)m * p.Cout + n] = acc[i][j][r];
        }
      }
  } else if (p.mode == CONV_RAW_STATS) {
    T* __restrict__ out = reinterpret_cast<T*>(p.out0);
    float s1[FN], s2[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn * TN + j * MT + n_l;
#pragma unroll
        for (int r = 0; r < M_::NREG; ++r) {
          const int m = m0 + wm * TM + i * MT + m_local(r);
          const float v = acc[i][j][r];
          s1[j] += v;
          s2[j] += v * v;
          if (m < p.M && n < p.Cout) out[(long)m * p.Cout + n] = from_f32<T>(v);
        }
      }
    if (p.stats != nullptr) {
      // rows beyond M were gathered as zeros (no bias) -> contribute 0 to both sums
      float* red = reinterpret_cast<float*>(lds);  // [WGM][BN][2]; k-loop ended with a barrier
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        float a1 = s1[j], a2 = s2[j];
        a1 += __shfl_xor(a1, 32);
        a2 += __shfl_xor(a2, 32);
        if (MT == 16) {
          a1 += __shfl_xor(a1, 16);
          a2 += __shfl_xor(a2, 16);
        }
        if (lane < MT) {
          const int col = wn * TN + j * MT + lane;
          red[(wm * BN + col) * 2 + 0] = a1;
          red[(wm * BN + col) * 2 + 1] = a2;
        }
      }
      __syncthreads();
      if (tid < BN) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int w = 0; w < WGM; ++w) {
          a1 += red[(w * BN + tid) * 2 + 0];
          a2 += red[(w * BN + tid) * 2 + 1];
        }
        const int n = n0 + tid;
        if (n < p.CoutPad) {
          p.stats[((long)tile_m * p.CoutPad + n) * 2 + 0] = a1;
          p.stats[((long)tile_m * p.CoutPad + n) * 2 + 1] = a2;
        }
      }
    }
  } else if (p.mode == CONV_EVAL_FUSED) {
    T* __restrict__ out = reinterpret_cast<T*>(p.out0);
    const T* __restrict__ res = reinterpret_cast<const T*>(p.res);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn * TN + j * MT + n_l;
        if (n >= p.Cout) continue;
        const float sc = p.scale[n], sf = p.shift[n];
#pragma unroll
        for (int r = 0; r < M_::NREG; ++r) {
          const int m = m0 + wm * TM + i * MT + m_local(r);
          if (m >= p.M) continue;
          float v = acc[i][j][r] * sc + sf;
          if (res != nullptr) v += to_f32<T>(res[(long)m * p.Cout + n]);
          if (p.relu) v = fmaxf(v, 0.f);
          out[(long)m * p.Cout + n] = from_f32<T>(v);
        }
      }
  } else if (p.mode == CONV_DGRAD) {
    T* __restrict__ o0 = reinterpret_cast<T*>(p.out0);
    T* __restrict__ o1 = reinterpret_cast<T*>(p.out1);
    const int c0 = p.out_c0, c1 = p.Cout - p.out_c0;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn * TN + j * MT + n_l;
        if (n >= p.Cout) continue;
        const bool first = n < c0;
        T* __restrict__ dst = first ? o0 + n : o1 + (n - c0);
        const int ld = first ? c0 : c1;
        const bool accum = first ? p.acc0 : p.acc1;
#pragma unroll
        for (int r = 0; r < M_::NREG; ++r) {
          const int m = m0 + wm * TM + i * MT + m_local(r);
          if (m >= p.M) continue;
          float v = acc[i][j][r];
          if (accum) v += to_f32<T>(dst[(long)m * ld]);
          dst[(long)m * ld] = from_f32<T>(v);
        }
      }
  } else {  // CONV_HEAD_NCHW: + bias, fp32 NCHW
    float* __restrict__ out = reinterpret_cast<float*>(p.out0);
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn * TN + j * MT + n_l;
        if (n >= p.Cout) continue;
        const float bias = p.scale ? p.scale[n] : 0.f;
#pragma unroll
        for (int r = 0; r < M_::NREG; ++r) {
          const int m = m0 + wm * TM + i * MT + m_local(r);
          if (m >= p.M) continue;
          const int b = m / HoWo;
          const int pix = m - b * HoWo;
          out[((long)b * p.Cout + n) * HoWo + pix] = acc[i][j][r] + bias;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------
// split-K epilogue: sums the slabs in a fixed order and applies the mode's store
//   CONV_RAW_STATS : out0[m][n] = sum (+ per-channel sum / sumsq partial per row block)
//   CONV_DGRAD     : dual-destination store with optional accumulate
// ---------------------------------------------------------------------------------------
constexpr int SK_ROWS = 16;  // rows per reduce workgroup

template <typename T>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvParams p) {
  __shared__ float red[256 * 4 * 2];
  const int VC = p.Cout / 4;   // 16-byte vectors per row; 256 % VC == 0 (plan)
  const int RP = 256 / VC;     // rows per pass
  const int cv = threadIdx.x % VC, r0 = threadIdx.x / VC;
  const int m_begin = blockIdx.x * SK_ROWS;
  const int m_end = min(m_begin + SK_ROWS, p.M);
  const long MN = (long)p.M * p.Cout;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  T* __restrict__ o0 = reinterpret_cast<T*>(p.out0);
  T* __restrict__ o1 = reinterpret_cast<T*>(p.out1);
  for (int m = m_begin + r0; m < m_end; m += RP) {
    const long e = (long)m * p.Cout + cv * 4;
    float4 v = *reinterpret_cast<const float4*>(p.partial + e);
    for (int z = 1; z < p.splitk; ++z) {
      const float4 w = *reinterpret_cast<const float4*>(p.partial + z * MN + e);
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    float vv[4] = {v.x, v.y, v.z, v.w};
    if (p.mode == CONV_RAW_STATS) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s1[k] += vv[k];
        s2[k] += vv[k] * vv[k];
        o0[e + k] = from_f32<T>(vv[k]);
      }
    } else {  // CONV_DGRAD
      const int n = cv * 4;
      const bool first = n < p.out_c0;
      T* __restrict__ dst = first ? o0 + (long)m * p.out_c0 + n : o1 + (long)m * (p.Cout - p.out_c0) + (n - p.out_c0);
      const bool accum = first ? p.acc0 : p.acc1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float x = vv[k];
        if (accum) x += to_f32<T>(dst[k]);
        dst[k] = from_f32<T>(x);
      }
    }
  }
  if (p.mode == CONV_RAW_STATS && p.stats != nullptr) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      red[(threadIdx.x * 4 + k) * 2 + 0] = s1[k];
      red[(threadIdx.x * 4 + k) * 2 + 1] = s2[k];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * p.Cout; t += 256) {
      const int c = t >> 1, which = t & 1;
      const int v = c >> 2, k = c & 3;
      float s = 0.f;
      for (int rr = 0; rr < RP; ++rr) s += red[(((rr * VC + v) * 4) + k) * 2 + which];
      p.stats[((long)blockIdx.x * p.CoutPad + c) * 2 + which] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side: tile selection + launch
// ---------------------------------------------------------------------------------------
static bool is_small_c(const ConvParams& p, int dtype) {
  const int bke = dtype == D3F_F32 ? 32 : 64;
  return ((p.C0 + p.C1) % bke) != 0;
}

static ConvTile pick_tile(const ConvParams& p) {
  const int co = p.Cout;
  if (co <= 16) return {256, 16};
  if (co <= 32) return {256, 32};
  // prefer the biggest tile that still gives >= 2 blocks per CU; small problems fall to 64x64
  const long M = p.M;
  auto blocks = [&](int bm, int bn) { return (long)cdiv(M, bm) * cdiv(co, bn); };
  if (co % 128 == 0 && blocks(128, 128) >= 512) return {128, 128};
  if (blocks(128, 64) >= 512) return {128, 64};
  return {64, 64};
}

size_t conv_splitk_floats(const ConvParams& p) {
  return p.splitk > 1 ? (size_t)p.splitk * p.M * p.Cout : 0;
}

int conv_igemm_plan(ConvParams& p, int dtype, bool allow_splitk) {
  const int ve = dtype == D3F_F32 ? 4 : 8;
  const int bke = dtype == D3F_F32 ? 32 : 64;
  D3F_CHECK(dtype == D3F_F32 || dtype == D3F_BF16, "conv: bad dtype %d", dtype);
  D3F_CHECK((p.C0 % ve) == 0 && (p.C1 % ve) == 0, "conv: channels (%d,%d) not a multiple of %d",
            p.C0, p.C1, ve);
  D3F_CHECK(p.Kpad % bke == 0 && p.Kpad >= p.KH * p.KW * (p.C0 + p.C1),
            "conv: Kpad %d inconsistent with K=%d", p.Kpad, p.KH * p.KW * (p.C0 + p.C1));
  if (is_small_c(p, dtype)) {
    D3F_CHECK(p.C1 == 0, "conv: small-channel mode takes one source");
  } else {
    D3F_CHECK(p.Kpad == p.KH * p.KW * (p.C0 + p.C1), "conv: regular mode needs Kpad == K");
  }
  D3F_CHECK(p.shift0 == 0 || p.shift0 == 1, "conv: shift0");
  D3F_CHECK(!p.zi || p.shift0 == 1, "conv: zero insertion needs shift0");
  D3F_CHECK(p.H0s == (p.Hv >> p.shift0) && p.W0s == (p.Wv >> p.shift0), "conv: src0 extent");
  D3F_CHECK(p.M == p.B * p.Ho * p.Wo, "conv: M");
  D3F_CHECK(p.CoutPad >= p.Cout, "conv: CoutPad");
  D3F_CHECK(p.C1 == 0 || (p.C0 % bke) == 0, "conv: C0=%d must be a multiple of %d when a second source is concatenated", p.C0, bke);
  const long es = dtype == D3F_F32 ? 4 : 2;
  const long b0 = (long)p.B * p.H0s * p.W0s * p.C0 * es, b1 = (long)p.B * p.Hv * p.Wv * p.C1 * es;
  const long bw = (long)p.CoutPad * p.Kpad * es;
  D3F_CHECK(b0 < (1L << 31) && b1 < (1L << 31) && bw < (1L << 31), "conv: operand larger than 2 GiB");
  p.src0_bytes = (unsigned)b0; p.src1_bytes = (unsigned)b1; p.w_bytes = (unsigned)bw;
  const ConvTile t = pick_tile(p);
  p.tiles_m = cdiv(p.M, t.BM);
  p.tiles_n = cdiv(p.Cout, t.BN);
  p.splitk = 1;
  p.stat_rows = p.tiles_m;
  // deep layers: M x Cout gives too few workgroups to fill 256 CUs -> cut the K loop
  const long base = (long)p.tiles_m * p.tiles_n;
  const int nk = p.Kpad / bke;
  const int vc = p.Cout / 4;
  static const bool no_splitk = getenv("D3F_NO_SPLITK") != nullptr;  // debugging knob
  if (allow_splitk && !no_splitk && !is_small_c(p, dtype) && base < 384 && (p.Cout % 4) == 0 && vc <= 256 &&
      (256 % vc) == 0 && (p.mode == CONV_RAW_STATS || p.mode == CONV_DGRAD) &&
      (p.mode != CONV_DGRAD || (p.out_c0 % 4) == 0)) {
    int sk = (int)((640 + base - 1) / base);
    while (sk > 1 && nk / sk < 6) --sk;  // keep >= 6 k-tiles per slice
    if (sk > 8) sk = 8;
    if (sk > 1) {
      p.splitk = sk;
      p.stat_rows = cdiv(p.M, SK_ROWS);
    }
  }
  return 0;
}

template <typename T, int BM, int BN, int WGM, int WGN, int MT>
static int launch_cfg(const ConvParams& p, bool smallc, hipStream_t stream) {
  const dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)p.splitk), block(256);
  if (smallc)
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WGM, WGN, MT, true>), grid, block, 0, stream, p);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WGM, WGN, MT, false>), grid, block, 0, stream, p);
  D3F_HIP(hipGetLastError());
  return 0;
}

template <typename T> static int launch_t(const ConvParams& p, bool smallc, hipStream_t stream) {
  const ConvTile t = pick_tile(p);
  D3F_CHECK(p.tiles_m == cdiv(p.M, t.BM) && p.tiles_n == cdiv(p.Cout, t.BN),
            "conv: params were not planned (tiles %d,%d)", p.tiles_m, p.tiles_n);
  if (t.BM == 256 && t.BN == 16) return launch_cfg<T, 256, 16, 4, 1, 16>(p, smallc, stream);
  if (t.BM == 256 && t.BN == 32) return launch_cfg<T, 256, 32, 4, 1, 32>(p, smallc, stream);
  if (t.BM == 128 && t.BN == 128) return launch_cfg<T, 128, 128, 2, 2, 32>(p, smallc, stream);
  if (t.BM == 128 && t.BN == 64) return launch_cfg<T, 128, 64, 2, 2, 32>(p, smallc, stream);
  return launch_cfg<T, 64, 64, 2, 2, 32>(p, smallc, stream);
}

int conv_igemm_launch(const ConvParams& p, int dtype, hipStream_t stream) {
  if (p.M == 0) return 0;
  const bool smallc = is_small_c(p, dtype);
  ConvParams q = p;
  if (q.partial == nullptr) q.splitk = 1;
  D3F_CHECK(q.splitk == 1 || q.stat_rows == cdiv(q.M, SK_ROWS), "conv: split-K params were not planned");
  const bool prof = prof_enabled();
  if (prof) prof_begin(q.mode == CONV_DGRAD ? PROF_CONV_DGRAD : PROF_CONV_FWD, q.flops, stream);
  int rc = dtype == D3F_F32 ? launch_t<float>(q, smallc, stream) : launch_t<bf16_t>(q, smallc, stream);
  if (rc == 0 && q.splitk > 1) {
    const dim3 grid((unsigned)cdiv(q.M, SK_ROWS)), block(256);
    if (dtype == D3F_F32)
      hipLaunchKernelGGL(conv_splitk_reduce_kernel<float>, grid, block, 0, stream, q);
    else
      hipLaunchKernelGGL(conv_splitk_reduce_kernel<bf16_t>, grid, block, 0, stream, q);
    if (hipGetLastError() != hipSuccess) rc = set_error(-2, "conv split-K reduce launch failed");
  }
  if (prof) prof_end(stream);
  return rc;
}

}  // namespace d3f
