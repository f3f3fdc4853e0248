// Convolution weight gradient on the matrix cores.
//
//   dW[co][tap][ci] = sum_m dY[m][co] * V[m, tap][ci]       m = (b, oy, ox)
//
// i.e. per filter tap a GEMM whose reduction runs over output pixels.  In NHWC both
// operands are stored pixel-major, which is exactly the k-major image the f32 MFMA wants
// (lane l supplies element [k = l>>5][i = l&31]): tiles are staged in LDS as [pixel][channel]
// rows and fragments are read with conflict-free ds_read_b32.  V is gathered like the
// forward A operand (two concatenated sources, optional nearest x2 up-sample), so the
// decoder's upsample+cat never exists in memory in the backward pass either.
//
// The pixel range is cut into `splits` slabs (one grid.y each) that write fp32 partial
// slabs; wgrad_reduce sums them in a fixed order (bitwise reproducible, no float atomics)
// and scatters into the PyTorch [Cout][Cin][KH][KW] gradient layout.
// Replaces the conv weight-gradient ATen/cuDNN kernels autograd runs under
// loss.backward() for the reference's U-Net.
#include "common.h"

#include <cstdlib>
#include <cstring>

namespace d3f {

constexpr int KP = 32;  // pixels per k-chunk

// store one 16-byte global vector (4 f32 or 8 bf16) as f32 into LDS
template <typename T> __device__ __forceinline__ void lds_store_as_f32(float* dst, const uint4& v);
template <> __device__ __forceinline__ void lds_store_as_f32<float>(float* dst, const uint4& v) {
  *reinterpret_cast<uint4*>(dst) = v;
}
template <> __device__ __forceinline__ void lds_store_as_f32<bf16_t>(float* dst, const uint4& v) {
  *reinterpret_cast<uint4*>(dst) = make_uint4(v.x << 16, v.x & 0xffff0000u, v.y << 16, v.y & 0xffff0000u);
  *reinterpret_cast<uint4*>(dst + 4) = make_uint4(v.z << 16, v.z & 0xffff0000u, v.w << 16, v.w & 0xffff0000u);
}

// 4 fp32 -> three 8-byte groups of bf16: the exact 3-way truncation split of conv_igemm.hip's x3 mode
__device__ __forceinline__ void wg_split3x4(const uint4& v, uint2& h, uint2& m, uint2& l) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const uint32_t x[4] = {v.x, v.y, v.z, v.w};
  uint32_t r1[4], r2[4];
#pragma unroll
  for (int e = 0; e < 4; e += 2) {
    const f32x2 f = {__uint_as_float(x[e]), __uint_as_float(x[e + 1])};
    const f32x2 hi = {__uint_as_float(x[e] & 0xffff0000u), __uint_as_float(x[e + 1] & 0xffff0000u)};
    const f32x2 a = f - hi;
    r1[e] = __float_as_uint(a.x);
    r1[e + 1] = __float_as_uint(a.y);
    const f32x2 mid = {__uint_as_float(r1[e] & 0xffff0000u), __uint_as_float(r1[e + 1] & 0xffff0000u)};
    const f32x2 b = a - mid;
    r2[e] = __float_as_uint(b.x);
    r2[e + 1] = __float_as_uint(b.y);
  }
  constexpr uint32_t SEL = 0x07060302u;
  h = make_uint2(__builtin_amdgcn_perm(x[1], x[0], SEL), __builtin_amdgcn_perm(x[3], x[2], SEL));
  m = make_uint2(__builtin_amdgcn_perm(r1[1], r1[0], SEL), __builtin_amdgcn_perm(r1[3], r1[2], SEL));
  l = make_uint2(__builtin_amdgcn_perm(r2[1], r2[0], SEL), __builtin_amdgcn_perm(r2[3], r2[2], SEL));
}

// T = activation type in memory (f32 or bf16).  The contraction itself runs on the f32 MFMA for both:
// bf16 operands are widened when they are staged into LDS (exact), accumulation is f32.
//
// X3 (64x64 tile, fp32 operands): the f32x3 contraction of conv_igemm.hip for the weight gradient.  Both operands
// are k-strided here (k = pixel, the image is [pixel][channel]), so the bf16 planes are staged as
// [32-channel subtile][32 pixels][32 channels] with 64-byte rows and the MFMA operands (8 consecutive k of one
// channel per lane) are gathered with ds_read_b64_tr_b16, the hardware transposing read of gfx950: two reads
// per fragment, conflict-free because the 4 rows x 64 B a half-wave touches are contiguous.
//
// CLS (WgradParams::cls): the class form of a layer behind a nearest x2 up-sampling, for the channels of the up-sampled
// source.  blockIdx "tap" = one of 16 folded taps ((py, px) output-parity class x (a, b) position in the 2x2
// neighbourhood); the k-loop runs over the LOW-resolution pixel grid (b, j, i): dY is read at (2j + py, 2i + px), the
// source at (j + a - 1 + py, i + b - 1 + px) -- the low-resolution pixel under up-sampled row 2j + py + kh - 1 for the
// taps kh that share `a` (kh = 0 | 1,2 for py = 0; kh = 0,1 | 2 for py = 1; columns alike).
template <typename T, int BMW, int BNW, int WGM, int WGN, int KSPLIT, bool X3, bool CLS = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams pin) {
  const WgradParams p = wgrad_params_of_net(pin, (int)blockIdx.z);  // two networks in one launch: blockIdx.z = net
  constexpr int VE = Elem<T>::VE;

  static_assert(!X3 || (BMW == 64 && BNW == 64 && WGM == 2 && WGN == 2 && KSPLIT == 1), "x3 weight gradient: 64x64 tile");
  // bf16 storage: the same k-major staging with ONE plane (the operands are bf16 already) = native bf16 MFMA
  constexpr int NPL = sizeof(T) == 4 ? 3 : 1;
  constexpr int TM = BMW / WGM, TN = BNW / WGN, FM = TM / 32, FN = TN / 32;
  constexpr int LY = BMW + 4, LX = BNW + 4;    // LDS row strides (floats), 16-B aligned rows
  constexpr int VY = BMW / VE, VX = BNW / VE;  // 16-byte global vectors per row
  constexpr int NVY = (VY * KP + 255) / 256, NVX = (VX * KP + 255) / 256;  // 16-byte vectors per thread: its row's columns lcol + j * (256 / KP)
  static_assert(WGM * WGN * KSPLIT == 4, "4 waves");
  constexpr int RED = (KSPLIT > 1) ? KSPLIT * 32 * 32 : 1;
  // x3: per operand 3 planes x 2 subtiles x [32 pixels][32 channels] bf16
  constexpr int X3_SUB = KP * 64, X3_PLANE = 2 * X3_SUB, X3_OP = (sizeof(T) == 4 ? 3 : 1) * X3_PLANE;  // bytes
  constexpr int F32_FLOATS = (KP * (LY + LX) > RED) ? KP * (LY + LX) : RED;
  // bf16 storage: UB chunks of 32 pixels are staged per barrier pair (the matrix work of a chunk is two MFMAs -- with one
  // chunk per pair the loop was two barriers and a global-load latency per 2 MFMAs)
  // (same-box sweep of chunks per pair, round 5: 1: 3.817, 2: 3.746, 4: 3.745, 8: 3.97 ms per bf16 step; 2 holds 16 KB of LDS, 4 holds 32)
  constexpr int UB = (X3 && sizeof(T) == 2) ? 2 : 1;
  constexpr int LDS_FLOATS = X3 ? UB * 2 * X3_OP / 4 : F32_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  float* Ys = lds;
  float* Xs = lds + KP * LY;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave % KSPLIT;
  const int wmn = wave / KSPLIT;
  const int wm = wmn / WGN, wn = wmn % WGN;

  // XCD-aware workgroup order (speed only, never correctness): the (tap, co, ci) workgroups of one pixel slab read
  // the SAME dY rows and nearly the same X rows, but consecutive block ids are dealt round-robin over the 8 XCDs, each
  // with its own L2 -- every slab was fetched by up to 8 L2s (layer1: 256 MB per launch against 34 MB of operands).
  // The bijective remap of cdna_hip_programming.md T1 gives every XCD a contiguous range of the logical order
  // (x fastest, then slab), so a slab's workgroups share one L2 (its L2-miss traffic 4.13 -> 1.29 GB per step, r02_t).
  int bid, split;
  {
    const unsigned gx = gridDim.x, nwg = gridDim.x * gridDim.y;
    const unsigned orig = blockIdx.y * gx + blockIdx.x;
    const unsigned xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;
    const unsigned wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    bid = (int)(wgid % gx);
    split = (int)(wgid / gx);
  }
  const int tile_ci = bid % p.tiles_ci;
  bid /= p.tiles_ci;
  const int tile_co = bid % p.tiles_co;
  const int tap = bid / p.tiles_co;
  // plain: (kh, kw); class form: folded tap = ((py*2 + px)*2 + a)*2 + b -> dY at (2j + py, 2i + px), source pixel
  // offset (a - 1 + py, b - 1 + px) on the low-resolution grid
  const int kh = CLS ? ((tap >> 1) & 1) - 1 + (tap >> 3) : tap / p.KW;
  const int kw = CLS ? (tap & 1) - 1 + ((tap >> 2) & 1) : tap - (tap / p.KW) * p.KW;
  const int cpy = tap >> 3, cpx = (tap >> 2) & 1;  // class form only
  const int co0 = tile_co * BMW, ci0 = tile_ci * BNW;  // ci0: channel offset inside this launch's channel range
  const int Cin = p.slab_cin;

  // buffer descriptors: out-of-range lanes read zeros in hardware (see conv_igemm.hip)
  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy, p.dy_bytes);
  const __amdgpu_buffer_rsrc_t rx0 = make_rsrc(p.src0, p.src0_bytes);
  const __amdgpu_buffer_rsrc_t rx1 = make_rsrc(p.src1 != nullptr ? p.src1 : p.src0, p.src1_bytes);

  // loader roles: every thread owns ONE pixel row of the chunk (lrow) and the 16-byte vectors lcol + j * TPR of it, in
  // both operands -- one pixel decode and one validity test per thread and chunk serve all of its loads.  (On gfx950
  // the fp32 MFMA and the vector ALU of a SIMD exclude each other ACROSS waves, profiles/microbench/mfma_valu_corun.hip:
  // every VALU cycle of the loader is a cycle no wave of the SIMD can spend in the matrix pipe.)
  constexpr int TPR = 256 / KP;  // threads per pixel row
  static_assert(256 % KP == 0, "whole threads per pixel row");
  const int lrow = tid / TPR, lcol = tid % TPR;
  const int yco = co0 + lcol * VE;                 // first of this thread's output channels (vector j: + j * TPR * VE)
  const int xci = p.ci_base + ci0 + lcol * VE;     // concatenated input channel of vector 0
  const bool from0 = CLS || (p.ci_base + ci0 < p.C0);  // block-uniform: plan keeps ci tiles inside one source
  const int xcl = from0 ? xci : xci - p.C0;
  const int Cs = from0 ? p.C0 : p.C1;
  const int sh = CLS ? 0 : (from0 ? p.shift0 : 0);  // class form reads the low-resolution source directly
  const int Hs = from0 ? p.H0s : p.Hv;
  const int Ws = from0 ? p.W0s : p.Wv;
  const int HcWc = p.Hc * p.Wc;
  // LIN: the source pixel of output pixel m at this tap is m + a constant (stride 1, 'same' extent, source read at its
  // own resolution -- every layer but the stride-2 ones and the class form by construction): the byte offset then
  // advances by a constant per chunk, the pixel decode is only needed for the border test
  const bool lin = CLS || (p.stride == 1 && p.Ho == p.Hv && p.Wo == p.Wv && sh == 0);

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int chunk_begin = split * p.chunks_per_split;
  int chunk_end = chunk_begin + p.chunks_per_split;
  const int total_chunks = (p.Mi + KP - 1) / KP;
  if (chunk_end > total_chunks) chunk_end = total_chunks;

  // Incremental pixel decode: (b, oy, ox) of this thread's row on the iterated grid (Hc x Wc per image), advanced by
  // exactly one chunk = step_img images + step_row rows + step_col columns (computed on the host; at most one carry
  // each since step_col < Wc and step_row < Hc) -- two integer divisions per chunk would cost ~100 VALU.
  int xb, xoy, xox;
  {
    const int m = chunk_begin * KP + lrow;
    xb = m / HcWc;
    const int r = m - xb * HcWc;
    xoy = r / p.Wc;
    xox = r - xoy * p.Wc;
  }
  // taps as pixel offsets on the grid the source is addressed on
  const int dkh = CLS ? kh : kh - p.pad, dkw = CLS ? kw : kw - p.pad;
  // running byte offsets of vector 0 of this thread's row (valid or not; selected against BUF_OOB per chunk)
  unsigned yoff = CLS ? (unsigned)((4 * (chunk_begin * KP + lrow) + 2 * cpy * p.Wc + cpx) * p.Cout + yco) * (unsigned)sizeof(T)
                      : (unsigned)((chunk_begin * KP + lrow) * p.Cout + yco) * (unsigned)sizeof(T);
  const unsigned ystep = (unsigned)((CLS ? 4 : 1) * KP * p.Cout) * (unsigned)sizeof(T);
  const unsigned yox2 = (unsigned)(2 * p.Cout) * (unsigned)sizeof(T);  // class form: dY pixel = 4 m - 2 ox + const
  unsigned xoff = (unsigned)((chunk_begin * KP + lrow + dkh * Ws + dkw) * Cs + xcl) * (unsigned)sizeof(T);  // LIN only
  const unsigned xstep = (unsigned)(KP * Cs) * (unsigned)sizeof(T);
  constexpr unsigned VSTEP = TPR * VE * sizeof(T);  // bytes between a thread's vectors of one row
  bool ycol[NVY], xcol[NVX];
#pragma unroll
  for (int j = 0; j < NVY; ++j) ycol[j] = lcol + j * TPR < VY && yco + j * TPR * VE < p.Cout;
#pragma unroll
  for (int j = 0; j < NVX; ++j) xcol[j] = lcol + j * TPR < VX && ci0 + (lcol + j * TPR) * VE < Cin;
  uint4 ry[NVY], rx[NVX];
  auto load_chunk_to = [&](int ch, uint4 (&ry)[NVY], uint4 (&rx)[NVX], bool live) {
    const bool rowok = live && lrow < p.Mi - ch * KP;  // the last chunk may be partial
    const int b = xb, oy = xoy, ox = xox;
    {
      int nx = ox + p.step_col;
      const int w = nx >= p.Wc ? 1 : 0;
      nx -= w ? p.Wc : 0;
      int ny = oy + p.step_row + w;
      const int h = ny >= p.Hc ? 1 : 0;
      ny -= h ? p.Hc : 0;
      xox = nx; xoy = ny;
      if (!lin) xb = b + p.step_img + h;
    }
    unsigned ox_off, oy_off;
    if (lin) {
      const bool inb = (unsigned)(oy + dkh) < (unsigned)Hs && (unsigned)(ox + dkw) < (unsigned)Ws;
      ox_off = (rowok && inb) ? xoff : BUF_OOB;
      xoff += xstep;
    } else {
      const int iy = oy * p.stride + dkh, ix = ox * p.stride + dkw;
      const bool inb = (unsigned)iy < (unsigned)p.Hv && (unsigned)ix < (unsigned)p.Wv;
      const int pix = (b * Hs + (iy >> sh)) * Ws + (ix >> sh);
      ox_off = (rowok && inb) ? (unsigned)(pix * Cs + xcl) * (unsigned)sizeof(T) : BUF_OOB;
    }
    if constexpr (CLS) oy_off = rowok ? yoff - __umul24((unsigned)ox, yox2) : BUF_OOB;
    else oy_off = rowok ? yoff : BUF_OOB;
    yoff += ystep;
#pragma unroll
    for (int j = 0; j < NVY; ++j) ry[j] = buf_load16s(rdy, ycol[j] ? oy_off : BUF_OOB, j * VSTEP);
    if (from0) {
#pragma unroll
      for (int j = 0; j < NVX; ++j) rx[j] = buf_load16s(rx0, xcol[j] ? ox_off : BUF_OOB, j * VSTEP);
    } else {
#pragma unroll
      for (int j = 0; j < NVX; ++j) rx[j] = buf_load16s(rx1, xcol[j] ? ox_off : BUF_OOB, j * VSTEP);
    }
  };
  auto load_chunk = [&](int ch) { load_chunk_to(ch, ry, rx, true); };

  if (UB == 1 && chunk_begin < chunk_end) load_chunk(chunk_begin);
  const int fr = lane & 31, fh = lane >> 5;
  if constexpr (X3) {
    typedef short v4s __attribute__((ext_vector_type(4)));
    unsigned char* lb = reinterpret_cast<unsigned char*>(lds);
    // staging: this thread's VE channels of one pixel -> 8 bytes per plane (fp32) / 16 bytes (bf16)
    constexpr int VPS = 32 / VE;  // 16-byte global vectors per 32-channel subtile
    auto woff = [&](int cv) { return (cv / VPS) * X3_SUB + (cv % VPS) * (VE * 2); };
    // fragment gather: 16-lane group g reads 4 pixel rows x 16 channels; lane 4q+p supplies row q, channels 4p..
    const int grp = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
    const int rd_lane = ((grp >> 1) * 8 + q) * 64 + ((grp & 1) * 16 + 4 * pq) * 2;  // + s*16*64 + h*4*64
    const unsigned char* ya = lb + wm * X3_SUB + rd_lane;
    const unsigned char* xa = lb + X3_OP + wn * X3_SUB + rd_lane;
    auto frag = [&](const unsigned char* base, int pl, int s) {
      union { uint4 u; v4s h[2]; } f;
#pragma unroll
      for (int h = 0; h < 2; ++h)
        f.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s*)(base + pl * X3_PLANE + s * 16 * 64 + h * 4 * 64));
      return f.u;
    };
    if constexpr (UB > 1) {
      // bf16 storage: groups of UB chunks; chunks past this slab's range load zeros (load_chunk_to(.., live = false))
      constexpr int SLOT = 2 * X3_OP;  // bytes per staged chunk (dY | X)
      uint4 qy[UB][NVY], qx[UB][NVX];
      auto load_group = [&](int ch0) {
#pragma unroll
        for (int u = 0; u < UB; ++u) load_chunk_to(ch0 + u, qy[u], qx[u], ch0 + u < chunk_end);
      };
      if (chunk_begin < chunk_end) load_group(chunk_begin);
      for (int ch = chunk_begin; ch < chunk_end; ch += UB) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
#pragma unroll
          for (int i = 0; i < NVY; ++i)
            if (lcol + i * TPR < VY) *reinterpret_cast<uint4*>(lb + u * SLOT + lrow * 64 + woff(lcol + i * TPR)) = qy[u][i];
#pragma unroll
          for (int i = 0; i < NVX; ++i)
            if (lcol + i * TPR < VX) *reinterpret_cast<uint4*>(lb + u * SLOT + X3_OP + lrow * 64 + woff(lcol + i * TPR)) = qx[u][i];
        }
        __syncthreads();
        if (ch + UB < chunk_end) load_group(ch + UB);
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
          for (int s = 0; s < KP / 16; ++s) {
            const uint4 a = frag(ya + u * SLOT, 0, s), b = frag(xa + u * SLOT, 0, s);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a),
                                                                *reinterpret_cast<const bf16x8*>(&b), acc[0][0], 0, 0, 0);
          }
        __syncthreads();
      }
    } else
    for (int ch = chunk_begin; ch < chunk_end; ++ch) {
#pragma unroll
      for (int i = 0; i < NVY; ++i) {
        if (lcol + i * TPR < VY) {
          unsigned char* d = lb + lrow * 64 + woff(lcol + i * TPR);
          if constexpr (sizeof(T) == 4) {
            uint2 h, m, l;
            wg_split3x4(ry[i], h, m, l);
            *reinterpret_cast<uint2*>(d) = h;
            *reinterpret_cast<uint2*>(d + X3_PLANE) = m;
            *reinterpret_cast<uint2*>(d + 2 * X3_PLANE) = l;
          } else {
            *reinterpret_cast<uint4*>(d) = ry[i];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NVX; ++i) {
        if (lcol + i * TPR < VX) {
          unsigned char* d = lb + X3_OP + lrow * 64 + woff(lcol + i * TPR);
          if constexpr (sizeof(T) == 4) {
            uint2 h, m, l;
            wg_split3x4(rx[i], h, m, l);
            *reinterpret_cast<uint2*>(d) = h;
            *reinterpret_cast<uint2*>(d + X3_PLANE) = m;
            *reinterpret_cast<uint2*>(d + 2 * X3_PLANE) = l;
          } else {
            *reinterpret_cast<uint4*>(d) = rx[i];
          }
        }
      }
      __syncthreads();
      if (ch + 1 < chunk_end) load_chunk(ch + 1);
#pragma unroll
      for (int s = 0; s < KP / 16; ++s) {
        uint4 a[NPL], b[NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          a[pl] = frag(ya, pl, s);
          b[pl] = frag(xa, pl, s);
        }
        if constexpr (NPL == 3) {
          constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // small terms first
#pragma unroll
          for (int t = 0; t < 6; ++t)
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a[PA[t]]),
                                                                *reinterpret_cast<const bf16x8*>(&b[PB[t]]), acc[0][0], 0, 0, 0);
        } else {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a[0]),
                                                              *reinterpret_cast<const bf16x8*>(&b[0]), acc[0][0], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  } else {
  for (int ch = chunk_begin; ch < chunk_end; ++ch) {
  #pragma unroll
      for (int i = 0; i < NVY; ++i)
        if (lcol + i * TPR < VY) lds_store_as_f32<T>(&Ys[lrow * LY + (lcol + i * TPR) * VE], ry[i]);
  #pragma unroll
      for (int i = 0; i < NVX; ++i)
        if (lcol + i * TPR < VX) lds_store_as_f32<T>(&Xs[lrow * LX + (lcol + i * TPR) * VE], rx[i]);
      __syncthreads();
      if (ch + 1 < chunk_end) load_chunk(ch + 1);
  #pragma unroll
      for (int kk = wk; kk < KP / 2; kk += KSPLIT) {
        const int k = 2 * kk + fh;
        float a[FM], b[FN];
  #pragma unroll
        for (int i = 0; i < FM; ++i) a[i] = Ys[k * LY + wm * TM + i * 32 + fr];
  #pragma unroll
        for (int j = 0; j < FN; ++j) b[j] = Xs[k * LX + wn * TN + j * 32 + fr];
  #pragma unroll
        for (int i = 0; i < FM; ++i)
  #pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
  }

  // D[co][ci]: ci_l = lane&31, co_l = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int taps = p.slab_taps;
  float* __restrict__ slab = p.partial + (long)split * p.Cout * taps * Cin;
  if (KSPLIT > 1) {
    // the 4 waves hold partial sums of the same 32x32 tile: reduce through LDS
    float* red = lds;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co_l = (r & 3) + 8 * (r >> 2) + 4 * fh;
      red[(wk * 32 + co_l) * 32 + fr] = acc[0][0][r];
    }
    __syncthreads();
    for (int e = tid; e < 32 * 32; e += 256) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < KSPLIT; ++w) s += red[w * 1024 + e];
      const int co = co0 + (e >> 5), ci = ci0 + (e & 31);
      if (co < p.Cout && ci < Cin) slab[((long)co * taps + tap) * Cin + ci] = s;
    }
  } else {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int ci = ci0 + wn * TN + j * 32 + fr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
          if (co < p.Cout && ci < Cin) slab[((long)co * taps + tap) * Cin + ci] = acc[i][j][r];
        }
      }
  }
}

// Slab reduce + layout change.  One workgroup per (output filter, group member, channel chunk of CB inputs): the
// chunk's [tap][CB] segments of every slab are summed with 16-byte loads and written out as the PyTorch [ci][tap]
// run -- both global sides coalesced.  The 256 threads are nsg slab groups x VB vector columns: group g sums slabs
// g, g + nsg, ... (8 loads in flight each) and the groups are combined through LDS in group order -- a fixed
// summation tree, bitwise reproducible.  (Round 1's one-element-per-thread form read 4 bytes per lane, scattered
// 4-byte stores `taps` floats apart and ran at 0.8 TB/s; a 64-filter layer with 114 slabs needs both the channel
// chunks -- enough workgroups -- and the slab groups -- enough loads in flight.)
//
// Part form: the slabs cover `Cin` channels, `CinReal` of them real, that land at channel c_off of a gradient with
// CinTot input channels.  FOLD: the slabs hold 16 folded taps per filter (WgradParams::cls); 3x3 tap (kh, kw) is the
// sum over the four output-parity classes (py, px) of folded tap ((py*2 + px)*2 + a(py, kh))*2 + a(px, kw),
// a(0, k) = k > 0, a(1, k) = k > 1 -- added in class order, a fixed tree.
template <bool FOLD>
__device__ __forceinline__ void wgrad_reduce_body(float* row, const float* __restrict__ partial, int splits, int CoutP,
                                                  int Cin, int CinReal, int taps, int CB, int nsg, int VB,
                                                  float* __restrict__ dw, int CinTot, int c_off, int co, int c0) {
  const int rowlen = taps * Cin;
  const long n = (long)CoutP * rowlen;
  const float* __restrict__ src = partial + (long)co * rowlen + c0;
  const int LR = CB + 1, plane = taps * LR, nvec = taps * CB / 4;
  const int sg = threadIdx.x / VB, vb = threadIdx.x - sg * VB;
  if (sg < nsg) {
    for (int v = vb; v < nvec; v += VB) {
      const int e = v * 4, tap = e / CB, cl = e - tap * CB;  // CB % 4 == 0: the four elements share a tap
      const float* __restrict__ s0 = src + tap * Cin + cl;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      int k = sg;
      for (; k + 7 * nsg < splits; k += 8 * nsg) {  // eight loads in flight
        float4 w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = *reinterpret_cast<const float4*>(s0 + (long)(k + u * nsg) * n);
        float4 t;
        t.x = ((w[0].x + w[1].x) + (w[2].x + w[3].x)) + ((w[4].x + w[5].x) + (w[6].x + w[7].x));
        t.y = ((w[0].y + w[1].y) + (w[2].y + w[3].y)) + ((w[4].y + w[5].y) + (w[6].y + w[7].y));
        t.z = ((w[0].z + w[1].z) + (w[2].z + w[3].z)) + ((w[4].z + w[5].z) + (w[6].z + w[7].z));
        t.w = ((w[0].w + w[1].w) + (w[2].w + w[3].w)) + ((w[4].w + w[5].w) + (w[6].w + w[7].w));
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
      }
      for (; k + nsg < splits; k += 2 * nsg) {
        const float4 a = *reinterpret_cast<const float4*>(s0 + (long)k * n);
        const float4 b = *reinterpret_cast<const float4*>(s0 + (long)(k + nsg) * n);
        acc.x += a.x + b.x; acc.y += a.y + b.y; acc.z += a.z + b.z; acc.w += a.w + b.w;
      }
      if (k < splits) {
        const float4 a = *reinterpret_cast<const float4*>(s0 + (long)k * n);
        acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
      }
      float* d = row + sg * plane + tap * LR + cl;
      d[0] = acc.x; d[1] = acc.y; d[2] = acc.z; d[3] = acc.w;
    }
  }
  __syncthreads();
  const int creal = min(CB, CinReal - c0);  // <= 0 for a chunk of padding channels only
  if constexpr (FOLD) {
    float* __restrict__ out = dw + ((long)co * CinTot + c_off + c0) * 9;
    for (int o = threadIdx.x; o < creal * 9; o += 256) {
      const int ci = o / 9, t9 = o - ci * 9, kh = t9 / 3, kw = t9 - kh * 3;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int py = c >> 1, px = c & 1;
        const int a = py ? (kh > 1) : (kh > 0), b = px ? (kw > 1) : (kw > 0);
        const int tap = ((c * 2) + a) * 2 + b;
        float sc = row[tap * LR + ci];
        for (int g = 1; g < nsg; ++g) sc += row[g * plane + tap * LR + ci];
        s += sc;
      }
      out[o] = s;
    }
  } else {
    float* __restrict__ out = dw + ((long)co * CinTot + c_off + c0) * taps;
    for (int o = threadIdx.x; o < creal * taps; o += 256) {
      const int ci = o / taps, tap = o - ci * taps;
      float s = row[tap * LR + ci];
      for (int g = 1; g < nsg; ++g) s += row[g * plane + tap * LR + ci];
      out[o] = s;
    }
  }
}

// The slab reduces of one layer (one job per pass: whole layer, or class-form part + skip part) as ONE launch.  A
// workgroup finds its job in the table by its block index.  (One launch per gradient BUCKET -- every layer keeping its
// slabs until then -- was measured equal or slightly slower per step, profiles/README.md round 3.)
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const WgradReduceBatch tb) {
  extern __shared__ __attribute__((aligned(16))) float row[];
  int j = 0;
#pragma unroll 1
  for (int i = 1; i < tb.n; ++i) j = ((int)blockIdx.x >= tb.job[i].block0) ? i : j;
  const WgradReduceJob& q = tb.job[j];
  const int local = (int)blockIdx.x - q.block0;
  const int co = local % q.Cout, cz = local / q.Cout;
  const float* partial = q.partial;
  float* dw = q.dw;
  if (blockIdx.y != 0) {  // two networks in one launch (common.h, NetSplit): blockIdx.y = net
    net_shift(partial, tb.net_ws);
    net_shift(dw, tb.net_grad);
  }
  if (q.fold)
    wgrad_reduce_body<true>(row, partial, q.splits, q.CoutP, q.Cin, q.CinReal, q.taps, q.CB, q.nsg, q.VB, dw, q.CinTot,
                            q.c_off, co, cz * q.CB);
  else
    wgrad_reduce_body<false>(row, partial, q.splits, q.CoutP, q.Cin, q.CinReal, q.taps, q.CB, q.nsg, q.VB, dw, q.CinTot,
                             q.c_off, co, cz * q.CB);
}

// tile edge (co x ci) of the tap-parallel kernel.  Measured per layer of Unet(resnet34) at B=16, 256x256
// (profiles/README.md): with ~1024 workgroups the 64x64 tile beats 128x128 on every wide layer by 15-20 % (fewer split
// slabs to write and re-read, r01_h); a 32x32 tile over 128-pixel chunks (a quarter of the slab bytes) is much slower
// (twice the L2 -> LDS staging per MFMA, r03)
static int pick_wtile(const WgradParams& p) { return (p.Cout > 32 && p.C0 + p.C1 > 32) ? 64 : 32; }
static bool patch_wgrad_off() {
  static const bool off = prof_knob("D3F_NO_PATCH_WGRAD") != nullptr;  // debugging knob: the tap-parallel kernel everywhere
  return off;
}

int wgrad_patch_variant(const WgradParams& p, int dtype);
void wgrad_patch_grid(const WgradParams& p, int variant, int* gx, int* gy);
int wgrad_patch_launch(const WgradParams& p, int variant, int dtype, hipStream_t stream);

// Does the weight gradient of conv(cat(upsample2x(src0), src1)) run as WG_CLASS + WG_SKIP passes?  The tap-parallel
// kernel's 64x64 tile only (the narrow decoder layers keep the persistent patch kernel); C0 a whole number of ci tiles.
bool wgrad_class_applies(const WgradParams& p, int dtype) {
  static const bool off = prof_knob("D3F_NO_WGRAD_CLASS") != nullptr;  // debugging knob: nine taps through the up-sampling
  if (off || !p.shift0 || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1) return false;
  if (p.Ho != p.Hv || p.Wo != p.Wv || p.H0s * 2 != p.Hv || p.W0s * 2 != p.Wv) return false;
  if (!patch_wgrad_off()) {
    // the narrow decoder layers (persistent patch kernel, conv_wgrad_patch.hip): its class form takes variants 1 and 3
    // with whole 32-channel slices per source
    const int v = wgrad_patch_variant(p, dtype);
    if (v) return (v == 1 || v == 3) && (p.C0 % 32) == 0 && (p.C1 % 32) == 0 && (p.Ho % 2) == 0 && (p.Wo % 2) == 0;
  }
  return pick_wtile(p) == 64 && (p.C0 % 64) == 0 && (p.C1 % 64) == 0;
}

int wgrad_plan(WgradParams& p, int dtype) {
  D3F_CHECK(dtype == D3F_F32 || dtype == D3F_BF16, "wgrad: bad dtype %d", dtype);
  const int ve = dtype == D3F_F32 ? 4 : 8;
  const long es = dtype == D3F_F32 ? 4 : 2;
  D3F_CHECK((p.C0 % ve) == 0 && (p.C1 % ve) == 0 && (p.Cout % ve) == 0,
            "wgrad: channels (%d,%d,%d) must be multiples of %d", p.C0, p.C1, p.Cout, ve);
  D3F_CHECK(p.H0s == (p.Hv >> p.shift0) && p.W0s == (p.Wv >> p.shift0), "wgrad: src0 extent");
  D3F_CHECK(p.M == p.B * p.Ho * p.Wo, "wgrad: M");
  D3F_CHECK(p.part == WG_WHOLE || p.part == WG_CLASS || p.part == WG_SKIP, "wgrad: part %d", p.part);
  const int t = pick_wtile(p);
  D3F_CHECK(p.C1 == 0 || (p.C0 % t) == 0, "wgrad: C0=%d must be a multiple of the ci tile %d", p.C0, t);
  const long bdy = (long)p.M * p.Cout * es, b0 = (long)p.B * p.H0s * p.W0s * p.C0 * es,
             b1 = (long)p.B * p.Hv * p.Wv * p.C1 * es;
  D3F_CHECK(bdy < (1L << 31) && b0 < (1L << 31) && b1 < (1L << 31), "wgrad: operand larger than 2 GiB");
  p.dy_bytes = (unsigned)bdy; p.src0_bytes = (unsigned)b0; p.src1_bytes = (unsigned)b1;
  // the part of the layer this launch covers
  p.cls = p.part == WG_CLASS ? 1 : 0;
  p.ci_base = p.part == WG_SKIP ? p.C0 : 0;
  p.slab_cin = p.part == WG_CLASS ? p.C0 : p.part == WG_SKIP ? p.C1 : p.C0 + p.C1;
  p.slab_taps = p.cls ? 16 : p.KH * p.KW;
  p.Hc = p.cls ? p.H0s : p.Ho;
  p.Wc = p.cls ? p.W0s : p.Wo;
  p.Mi = p.B * p.Hc * p.Wc;
  p.patch = patch_wgrad_off() ? 0 : wgrad_patch_variant(p, dtype);
  if (p.part != WG_WHOLE) D3F_CHECK(wgrad_class_applies(p, dtype), "wgrad: class form does not apply to this layer");
  if (p.patch) {  // persistent patch kernel: one slab per workgroup column
    int gx, gy;
    wgrad_patch_grid(p, p.patch, &gx, &gy);
    p.splits = gx;
    p.chunks_per_split = 0;
    p.tiles_co = p.tiles_ci = 0;
    return 0;
  }
  {  // one k-chunk (KP pixels) in units of the iterated grid
    const int hw = p.Hc * p.Wc, rem = KP % hw;
    p.step_img = KP / hw;
    p.step_row = rem / p.Wc;
    p.step_col = rem % p.Wc;
  }
  p.tiles_co = cdiv(p.Cout, t);
  p.tiles_ci = cdiv(p.slab_cin, t);
  const long base = (long)p.tiles_co * p.tiles_ci * p.slab_taps;
  const int total_chunks = cdiv(p.Mi, KP);
  static const long target_env = prof_knob("D3F_WGRAD_TARGET") ? atol(prof_knob("D3F_WGRAD_TARGET")) : 0;  // sweep knob
  // fp32: ~3.6 workgroups per CU (r02_ao/ap/aq sweep of 640 ... 1280: 896-960 best, 1024 +1 %, 1280 +2 %).  bf16 storage
  // (round 5 sweep, profiles/README.md: 232 ... 1856): the launches are latency-bound, not MFMA-bound, and the stream is as
  // long as the chain in the backward window -- 1152 is 1.2 % faster per step than 928, 640 1.4 % slower, 1856 1.8 % slower
  // (two networks in one launch share the target: half the slabs per net)
  const long target = (target_env > 0 ? target_env : (dtype == D3F_BF16 ? 1152 : 928)) / plan_nets_for(p.plan_nets, 16);
  long splits = (target + base - 1) / base;
  const long max_splits = (total_chunks + 3) / 4;  // keep >= 4 chunks per slab
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  p.chunks_per_split = cdiv(total_chunks, splits);
  p.splits = cdiv(total_chunks, p.chunks_per_split);
  return 0;
}

size_t wgrad_partial_floats(const WgradParams& p) {
  return (size_t)p.splits * p.Cout * p.slab_taps * p.slab_cin;
}

// (Wave slots, not priorities: these launches run at the lowest stream priority NEXT to the dependent BatchNorm ->
// data-gradient chain; an occupancy cap through unused dynamic LDS was measured slower, profiles/README.md round 2.)
template <typename T>
static void wgrad_launch_t(const WgradParams& p, int bm, dim3 grid, hipStream_t stream, bool x3) {
  const dim3 block(256);
  if (bm == 64) {
    if (x3 || sizeof(T) == 2) {  // f32x3, and bf16 storage on the native bf16 MFMA
      if (p.cls) hipLaunchKernelGGL((conv_wgrad_kernel<T, 64, 64, 2, 2, 1, true, true>), grid, block, 0, stream, p);
      else hipLaunchKernelGGL((conv_wgrad_kernel<T, 64, 64, 2, 2, 1, true>), grid, block, 0, stream, p);
      return;
    }
    if (p.cls) hipLaunchKernelGGL((conv_wgrad_kernel<T, 64, 64, 2, 2, 1, false, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<T, 64, 64, 2, 2, 1, false>), grid, block, 0, stream, p);
  } else {
    hipLaunchKernelGGL((conv_wgrad_kernel<T, 32, 32, 1, 1, 4, false>), grid, block, 0, stream, p);
  }
}

int wgrad_launch(const WgradParams& p, int dtype, hipStream_t stream) {
  if (p.M == 0) return 0;
  if (p.patch) {
    const bool prof = prof_enabled(PROF_WGRAD);
    if (prof) prof_begin(PROF_WGRAD, p.flops, stream);
    // (the persistent patch kernel of the narrow layers stays on the fp32 MFMA in f32x3 mode)
    const int rc = wgrad_patch_launch(p, p.patch, dtype == D3F_F32X3 ? D3F_F32 : dtype, stream);
    if (prof) prof_end(stream);
    return rc;
  }
  const int t = pick_wtile(p);
  D3F_CHECK(p.tiles_co == cdiv(p.Cout, t) && p.splits >= 1 && p.slab_taps >= 1 && p.Mi >= 1, "wgrad: params were not planned");
  D3F_CHECK(!p.cls || t == 64, "wgrad: class form needs the 64x64 tile");
  const dim3 grid((unsigned)(p.tiles_ci * p.tiles_co * p.slab_taps), (unsigned)p.splits, (unsigned)nets_of(p.nets));
  const bool prof = prof_enabled(PROF_WGRAD);
  if (prof) prof_begin(PROF_WGRAD, p.flops, stream);
  if (dtype == D3F_BF16) wgrad_launch_t<bf16_t>(p, t, grid, stream, false);
  else wgrad_launch_t<float>(p, t, grid, stream, dtype == D3F_F32X3);
  if (prof) prof_end(stream);
  D3F_HIP(hipGetLastError());
  return 0;
}

// work split of one reduce: channel chunks (cz of CB channels), slab groups (nsg) and vector columns (VB) per workgroup
static int wgrad_reduce_job(WgradReduceJob& q, const float* partial, int splits, int CoutP, int Cout, int Cin,
                            int CinRealPart, int CinRealTotal, int c_off, int KH, int KW, int fold, float* dw) {
  D3F_CHECK(Cin % 4 == 0 && CinRealPart <= Cin && Cout <= CoutP && c_off + CinRealPart <= CinRealTotal &&
                (!fold || (KH == 3 && KW == 3)),
            "wgrad reduce: arguments");
  const int taps = fold ? 16 : KH * KW;
  int cz = 1;  // channel chunks: enough workgroups to fill the chip, chunks of at least 16 channels (64-byte segments)
  while ((long)Cout * cz < 512 && Cin % (cz * 2) == 0 && Cin / (cz * 2) >= 16 && (Cin / (cz * 2)) % 4 == 0) cz *= 2;
  const int CB = Cin / cz, nvec = taps * CB / 4;
  int nsg = nvec >= 256 ? 1 : 256 / nvec;  // slab groups: fill the 256 threads, every group gets >= 2 slabs
  if (nsg > 8) nsg = 8;
  while (nsg > 1 && 2 * nsg > splits) --nsg;
  const int VB = nsg == 1 ? 256 : (nvec < 256 / nsg ? nvec : 256 / nsg);
  const size_t lds = (size_t)nsg * taps * (CB + 1) * sizeof(float);
  D3F_CHECK(lds <= 64 * 1024, "wgrad reduce: a %d-tap x %d-channel filter chunk exceeds the LDS tile", taps, CB);
  q.partial = partial; q.dw = dw;
  q.splits = splits; q.CoutP = CoutP; q.Cin = Cin; q.CinReal = CinRealPart; q.taps = taps; q.CB = CB; q.nsg = nsg;
  q.VB = VB; q.CinTot = CinRealTotal; q.c_off = c_off; q.fold = fold ? 1 : 0; q.Cout = Cout; q.cz = cz; q.block0 = 0;
  q.lds = (int)lds;
  return 0;
}

int wgrad_reduce_batch_add(WgradReduceBatch& tb, const float* partial, int splits, int CoutP, int Cout, int Cin,
                           int CinRealPart, int CinRealTotal, int c_off, int KH, int KW, int fold, float* dw) {
  D3F_CHECK(tb.n < WG_BATCH, "wgrad reduce batch: table full");
  WgradReduceJob& q = tb.job[tb.n];
  if (int rc = wgrad_reduce_job(q, partial, splits, CoutP, Cout, Cin, CinRealPart, CinRealTotal, c_off, KH, KW, fold, dw))
    return rc;
  q.block0 = tb.blocks;
  tb.blocks += q.Cout * q.cz;
  tb.lds = std::max(tb.lds, q.lds);
  ++tb.n;
  return 0;
}

int wgrad_reduce_batch_launch(const WgradReduceBatch& tb, hipStream_t stream) {
  if (tb.n == 0) return 0;
  hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)tb.blocks, (unsigned)nets_of(tb.nets)), dim3(256), (size_t)tb.lds, stream, tb);
  D3F_HIP(hipGetLastError());
  return 0;
}

// ---- a layer's weight gradient as 1 or 2 passes -----------------------------------------------------------------
int wgrad_layer_plan(WgradLayer& L, const WgradParams& base, int dtype) {
  L.nparts = 1;
  L.part[0] = base;
  L.part[0].part = WG_WHOLE;
  if (wgrad_class_applies(base, dtype)) {
    L.part[0].part = WG_CLASS;
    L.part[0].flops = base.flops * base.C0 / (base.C0 + base.C1);  // algorithmic credit: the original nine taps
    if (base.C1 > 0) {
      L.nparts = 2;
      L.part[1] = base;
      L.part[1].part = WG_SKIP;
      L.part[1].flops = base.flops * base.C1 / (base.C0 + base.C1);
    }
  }
  for (int i = 0; i < L.nparts; ++i)
    if (int rc = wgrad_plan(L.part[i], dtype)) return rc;
  return 0;
}

size_t wgrad_layer_partial_floats(const WgradLayer& L) {
  size_t n = 0;
  for (int i = 0; i < L.nparts; ++i) n += (wgrad_partial_floats(L.part[i]) + 63) / 64 * 64;
  return n;
}

// the passes' launches only: every pass writes its own slab region of `partial` (wgrad_layer_partial_floats floats),
// and its reduce is appended to `tb` for a later wgrad_reduce_batch_launch on the same stream
int wgrad_layer_launch_deferred(const WgradLayer& L, const void* dy, const void* src0, const void* src1, float* partial,
                                float* dw, int CoutReal, int CinReal, int dtype, WgradReduceBatch& tb,
                                hipStream_t stream, const NetSplit* ns) {
  if (ns != nullptr && ns->nets > 1) {  // two networks: every launch and the reduce carry both (common.h, NetSplit)
    tb.nets = ns->nets;
    tb.net_ws = ns->ws;
    tb.net_grad = ns->grad;
  }
  for (int i = 0; i < L.nparts; ++i) {
    WgradParams w = L.part[i];
    w.dy = dy; w.src0 = src0; w.src1 = src1; w.partial = partial;
    w.nets = ns != nullptr ? ns->nets : 1;
    w.net_ws = ns != nullptr ? ns->ws : 0;
    if (int rc = wgrad_launch(w, dtype, stream)) return rc;
    const int c_off = w.ci_base;
    const int creal = std::max(0, std::min(w.slab_cin, CinReal - c_off));
    if (int rc = wgrad_reduce_batch_add(tb, partial, w.splits, w.Cout, CoutReal, w.slab_cin, creal, CinReal, c_off, w.KH,
                                        w.KW, w.cls, dw))
      return rc;
    partial += (wgrad_partial_floats(w) + 63) / 64 * 64;
  }
  return 0;
}

int wgrad_layer_launch(const WgradLayer& L, const void* dy, const void* src0, const void* src1, float* partial,
                       float* dw, int CoutReal, int CinReal, int dtype, hipStream_t stream) {
  WgradReduceBatch tb;
  if (int rc = wgrad_layer_launch_deferred(L, dy, src0, src1, partial, dw, CoutReal, CinReal, dtype, tb, stream)) return rc;
  return wgrad_reduce_batch_launch(tb, stream);
}

}  // namespace d3f
