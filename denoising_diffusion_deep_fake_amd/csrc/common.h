// Shared declarations for libd3f_hip.so (gfx950 / CDNA4 only).
//
// Internal activation layout: NHWC ("pixel-major"): [B][H][W][C], C a multiple of 4 (f32)
// or 8 (bf16) so every pixel's channel run is a whole number of 16-byte vectors.  The
// drop-in boundary (include/d3f_hip.h) is NCHW fp32, converted at the network's first and
// last kernel only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

namespace d3f {

typedef uint16_t bf16_t;  // raw bf16 bits

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

}  // namespace d3f
#ifndef D3F_F32
#define D3F_F32 0  // same values as include/d3f_hip.h
#define D3F_BF16 1
#define D3F_F32X3 2  // fp32 storage; contractions on the bf16 matrix pipe from an exact 3-way split
#endif
namespace d3f {

// Debugging / sweep knobs whose verdict is recorded (profiles/README.md) exist only in profiling builds
// (make EXTRA=-DD3F_PROFILING, loaded through D3F_LIB): the shipped library does not read them.
static inline const char* prof_knob(const char* name) {
#ifdef D3F_PROFILING
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// thread-local error string (c_api.cpp)
int set_error(int code, const char* fmt, ...);
#define D3F_CHECK(cond, ...)                              \
  do {                                                    \
    if (!(cond)) return ::d3f::set_error(-1, __VA_ARGS__); \
  } while (0)
#define D3F_HIP(expr)                                                                 \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess)                                                             \
      return ::d3f::set_error(-2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                              __FILE__, __LINE__);                                    \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN)
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&b);
}
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }

// four consecutive elements of a tensor row as fp32: 16 bytes (f32) / 8 bytes (bf16)
template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                     __uint_as_float(v.y & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ void st4(T* p, const float4& v);
template <> __device__ __forceinline__ void st4<float>(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, const float4& v) {
  const uint32_t a = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
  const uint32_t b = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
  *reinterpret_cast<uint2*>(p) = make_uint2(a, b);
}

// ---- raw buffer loads: out-of-range lanes read zeros in hardware (no branch, no select) --------
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned BUF_OOB = 0x80000000u;  // every descriptor here covers < 2 GiB
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}

// the same with a wave-uniform byte offset added by the instruction (soffset): one address register serves several loads
__device__ __forceinline__ uint4 buf_load16s(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned soff) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, (int)soff, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}

// Wave priority of the kernels on the training step's DEPENDENT chain (forward / data-gradient convolutions, BatchNorm,
// pooling): the weight-gradient kernels of the second stream run at the default priority 0 on the same SIMDs, and the
// issue arbiter otherwise serves whichever wave issues matrix instructions more densely -- a faster weight-gradient
// kernel then SLOWED the data gradients it runs next to by as much as it gained (r03: 3.40 -> 3.75 ms per step), and
// the chain's 5 us BatchNorm kernels ran 3x longer than alone.  Stream priorities only order the dispatch of workgroups;
// this orders the issue of instructions.  -DD3F_CHAIN_PRIO=0 builds the A/B variant.
#ifndef D3F_CHAIN_PRIO
#define D3F_CHAIN_PRIO 3
#endif
__device__ __forceinline__ void chain_priority() {
#if D3F_CHAIN_PRIO > 0
  __builtin_amdgcn_s_setprio(D3F_CHAIN_PRIO);
#endif
}

// ---- two networks, one launch -----------------------------------------------------------------------------------
// train_deep_fake's denoise mode steps two INDEPENDENT networks of identical shape per batch (model_a on domain a, model_b
// on domain b: d3f/train_deep_fake/lit_module.py:142-181) with 8 images each -- half of what fills the chip.  A launch of
// the pair engine carries both: gridDim.z = nets x (the kernel's own z extent), and a workgroup of net 1 adds the byte
// offsets below to its pointers (the second network's copy of the workspace, its parameters, gradients, running
// statistics and boundary tensors).  Nothing else differs between the two halves of the grid, so the results are bit for
// bit those of the same plan run net by net.
struct NetSplit {
  int nets;   // 1 (the offsets are unused) or 2
  long ws;    // workspace tensors: activations, packed weights, statistics rows, coefficients, slabs, scratch
  long par;   // flat parameter buffer
  long grad;  // flat gradient buffer
  long bn;    // flat BatchNorm running statistics
  long in;    // NCHW boundary input of the launch (the forward's x, the backward's output gradient)
  long out;   // NCHW boundary output (the forward's prediction)
};
static inline int nets_of(int n) { return n > 1 ? n : 1; }
// plan_nets as one heuristic sees it.  Profiling builds only: D3F_PLAN_MASK=<bits> applies plan_nets to the heuristics whose
// bit is set and 1 to the others (1 tile choice, 2 split-K, 4 patch-resident conv, 8 Winograd, 16 weight-gradient slabs,
// 32 persistent weight-gradient grids, 64 BatchNorm row blocks) -- which of the pair's choices pays on 8-image launches
static inline int plan_nets_for(int plan_nets, int bit) {
  static const int mask = prof_knob("D3F_PLAN_MASK") ? atoi(prof_knob("D3F_PLAN_MASK")) : -1;
  return (mask & bit) ? nets_of(plan_nets) : 1;
}
static inline NetSplit net_split_or_single(const NetSplit* ns) {
  NetSplit one{};
  one.nets = 1;
  return (ns != nullptr && ns->nets > 1) ? *ns : one;
}
template <typename P> __device__ __forceinline__ void net_shift(P& p, long bytes) {
  if (p != nullptr) p = reinterpret_cast<P>(reinterpret_cast<uintptr_t>(p) + bytes);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VE = 4;    // elements per 16-byte vector
  static constexpr int BKE = 32;  // elements per 128-byte k-row
};
template <> struct Elem<bf16_t> {
  static constexpr int VE = 8;
  static constexpr int BKE = 64;
};

// division by a run-time constant as multiply-high + shift (exact for dividends < 2^31): mul == 0 encodes divisor 1
static inline void fast_div_setup(unsigned d, unsigned* mul, unsigned* shr) {
  if (d <= 1) { *mul = 0; *shr = 0; return; }
  unsigned lg = 0;
  while ((1ull << lg) < d) ++lg;  // ceil(log2 d)
  const unsigned p = 31 + lg;
  *mul = (unsigned)(((1ull << p) + d - 1) / d);
  *shr = p - 32;
}
__device__ __forceinline__ int fast_div(int n, unsigned mul, unsigned shr) {
  return mul ? (int)(__umulhi((unsigned)n, mul) >> shr) : n;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline long round_up(long a, long b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------------------------------
// Optional per-launch timing with HIP events on the launch stream (c_api.hip); used by bench.py
// to report achieved FLOP/s of the contraction kernels against the MFMA roofline.
// ---------------------------------------------------------------------------------------
enum ProfClass : int { PROF_CONV_FWD = 0, PROF_CONV_DGRAD = 1, PROF_WGRAD = 2, PROF_NUM = 3 };
bool prof_enabled(int cls);
void prof_begin(int cls, double flops, hipStream_t s);
void prof_end(hipStream_t s);

// ---------------------------------------------------------------------------------------
// Implicit-GEMM convolution (forward and data-gradient): conv_igemm.hip
// ---------------------------------------------------------------------------------------
enum ConvMode : int {
  CONV_RAW_STATS = 0,   // store raw NHWC conv output (+ per-channel partial sum / sumsq)
  CONV_HEAD_NCHW = 1,   // + bias, store NCHW fp32 (network boundary)
  CONV_EVAL_FUSED = 2,  // y*scale + shift (+res) (relu) -> NHWC   (eval-mode BN folded)
  CONV_DGRAD = 3,       // dual-destination NHWC store with optional accumulate
};

struct ConvParams {
  const void* src0;  // NHWC [B][H0s][W0s][C0]
  const void* src1;  // NHWC [B][Hv][Wv][C1] (channel-concatenated after src0) or null
  const void* w;     // packed weights [CoutPad][Kpad], k = (kh*KW + kw)*Cin + c
  void* out0;
  void* out1;
  float* stats;        // [tiles_m][CoutPad][2] or null           (CONV_RAW_STATS)
  const float* scale;  // per-out-channel scale (EVAL) / bias (HEAD)
  const float* shift;  // per-out-channel shift (EVAL)
  const void* res;     // residual NHWC (EVAL) or null
  unsigned src0_bytes, src1_bytes, w_bytes;  // extents for the buffer descriptors (filled by plan)
  int B, Hv, Wv;       // virtual input extent (after upsample / zero insertion)
  int C0, C1;          // C0 + C1 = Cin (padded to a vector multiple)
  int H0s, W0s;        // physical extent of src0 = (Hv, Wv) >> shift0
  int shift0;          // 1: src0 is read at (iy>>1, ix>>1)   (nearest x2 upsample)
  int zi;              // 1 (with shift0): only even (iy, ix) exist (zero insertion = dgrad of stride 2)
  int Ho, Wo, Cout;    // output extent, real output channels
  int CoutPad, Kpad;
  int w_ld;            // elements between packed weight rows (= Kpad; filled by plan)
  int par;             // output-parity classes.  CONV_DGRAD of a stride-2 conv: 1 = 3x3 pad 1, 2 = 1x1 pad 0;
                       // forward of conv(cat(upsample2x(src0), src1)) with the up-sampling folded into pre-summed
                       // weights: 3 (src0 is then described at its own low resolution H0s x W0s, shift0 = 0)
  int nz;              // classes in the launch (grid.z), filled by plan: 4 for par 1 / 3, else 1
  int KH, KW, stride, pad;
  int M;               // B*Ho*Wo
  int tiles_m, tiles_n;
  float* partial;      // split-K slabs [splitk][M][Cout] f32 (null: no split)
  int splitk;          // k-tile ranges (grid.y); > 1 only when `partial` is given
  int stat_rows;       // rows of the stats partial array: tiles_m, or the reduce kernel's row blocks
  int out_c0;          // CONV_DGRAD: channels [0,out_c0) -> out0, the rest -> out1
  int acc0, acc1;      // CONV_DGRAD: read-modify-write destinations
  int mode;
  int relu;
  double flops;        // algorithmic FLOPs (2*MAC, unpadded channels) of this launch, for profiling
  // CONV_DGRAD, optional: the BatchNorm-backward reduction of the layer that CONSUMES this gradient, fused
  // into the epilogue (single destination, no accumulate): per row block the column sums of
  //   dz = dX * (y*scale + shift > 0)   and   dz * (y - mean) * invstd
  // go to bn_partial[stat_rows][Cout][2]; y = that layer's raw conv output, bn_coef = its mean | invstd |
  // scale | shift arrays (Cout floats each, contiguous).
  const void* bn_y;
  const float* bn_coef;
  float* bn_partial;
  const void* bn_a;    // that layer's activation when its ReLU mask cannot be recomputed from y (residual add) or null
  // m -> (image, row, column) without integer divisions (filled by plan): q = umulhi(n, mul) >> shr for n < 2^31,
  // mul == 0 stands for a divisor of 1
  unsigned div_howo_mul, div_howo_shr, div_wo_mul, div_wo_shr;
  int cin_real;        // forward launches: real (unpadded) input channels when the caller knows them (the packed weights of
                       // the channels beyond are zero), else 0; lets the bf16 stem kernel stage 4 of its 8 channels
  int sum2;            // CONV_DGRAD request (set before the plan): store the 2x2 block sums of the gradient at HALF resolution
                       // (the gradient w.r.t. a source that was read through the nearest x2 up-sampling); kept only when
                       // a patch kernel takes the launch -- the plan clears it otherwise and the caller reduces itself
  // which kernel takes the launch (filled by the plan; THE table -- d3f_unet_plan_counts and the tests refer to it):
  //   0 conv_igemm_kernel;  conv_patch.hip: 1 conv_patch_kernel fp32 (16 or 4 staged channels), 2 conv_stem_kernel,
  //   3 conv_patch_kernel bf16 x 16 channels, 4 bf16 x 32 channels (32 / 64 filters), 5 bf16 x 32 behind the up-sampling,
  //   6 bf16 x 8 source channels staged as 16 (the head's data gradient), 7 bf16 16 -> 32 with the 2x2-summed epilogue,
  //   8 conv_stem_bf16_kernel;  conv_pres.hip: 9 / 10 / 11 / 12 conv_pres_kernel for 64 / 128 / 256 / 512 channels
  int patch;
  // two networks in one launch (NetSplit): grid.z = nets * nz; a workgroup of net 1 shifts every pointer above by net_ws,
  // except out0 (net_out0: the NCHW prediction of CONV_HEAD_NCHW lives outside the workspace) and scale (net_scale: the
  // head's bias is a parameter).  plan_nets (set before the plan): the tile / split-K / patch-kernel choices count the
  // workgroups of plan_nets networks, so that a single engine planned with plan_nets = 2 runs exactly the pair's kernels.
  int nets, plan_nets;
  long net_ws, net_out0, net_scale;
};
// the launch description as net `net` (blockIdx.z / nz) sees it
__device__ __forceinline__ ConvParams conv_params_of_net(const ConvParams& pin, int net) {
  ConvParams p = pin;
  if (net != 0) {
    net_shift(p.src0, p.net_ws); net_shift(p.src1, p.net_ws); net_shift(p.w, p.net_ws);
    net_shift(p.out0, p.net_out0); net_shift(p.out1, p.net_ws);
    net_shift(p.stats, p.net_ws); net_shift(p.scale, p.net_scale); net_shift(p.shift, p.net_ws);
    net_shift(p.res, p.net_ws); net_shift(p.partial, p.net_ws);
    net_shift(p.bn_y, p.net_ws); net_shift(p.bn_coef, p.net_ws); net_shift(p.bn_partial, p.net_ws); net_shift(p.bn_a, p.net_ws);
  }
  return p;
}

struct ConvTile {
  int BM, BN;
};
// Does conv(cat(upsample2x(x0), x1)) run with the up-sampling folded into pre-summed weights (ConvParams::par == 3
// forward, 4x4 stride-2 data gradient)?  Needs whole k-tiles per tap in both sources.
static inline bool upfold_applies(int dtype, int upsample0, int k, int stride, int pad, int C0, int C1) {
  static const bool off = prof_knob("D3F_NO_UPFOLD") != nullptr;  // debugging knob: gather through the up-sampling
  const int bke = dtype == D3F_BF16 ? 64 : 32;
  return !off && upsample0 && k == 3 && stride == 1 && pad == 1 && C0 > 0 && (C0 % bke) == 0 && (C1 % bke) == 0;
}
// Does the data gradient of this convolution run as the parity-class decomposition (ConvParams::par)?  The weight
// packers (pointwise.hip) store the flipped taps class by class exactly when it does, so both ask this one function.
static inline bool parity_dgrad_applies(int dtype, int stride, int k, int pad, int CoutD, int C1) {
  const int bke = dtype == D3F_BF16 ? 64 : 32;
  return stride == 2 && C1 == 0 && (CoutD % bke) == 0 && ((k == 3 && pad == 1) || (k == 1 && pad == 0));
}
// chooses the tile configuration for a problem; tiles_m/tiles_n are filled in p.
// allow_splitk: the caller can provide `partial` (conv_splitk_floats(p) floats) -- deep layers
// whose M x Cout yields too few workgroups then split the K loop over grid.y.
int conv_igemm_plan(ConvParams& p, int dtype, bool allow_splitk = false);
size_t conv_splitk_floats(const ConvParams& p);
int conv_igemm_launch(const ConvParams& p, int dtype, hipStream_t stream);
// conv_patch.hip: LDS-patch form of the full-resolution 16-channel 3x3 layers (chosen inside conv_igemm_plan)
// conv_winograd.hip: Winograd F(2x2, 3x3) forward of the wide-enough stride-1 3x3 layers in train mode (fp32)
bool conv_winograd_fits(const ConvParams& p, int dtype);     // shapes the kernel can run
bool conv_winograd_applies(const ConvParams& p, int dtype);  // shapes the plan takes it for (>= 256 workgroups)
int conv_winograd_stat_rows(const ConvParams& p);           // one statistics row per workgroup
size_t conv_winograd_filter_floats(const ConvParams& p);    // U[16][Cin / 16][Cout][16]
int conv_winograd_pack_launch(const float* w /*[Cout][Cin][3][3]*/, float* u, int Cout, int Cin, hipStream_t stream);
int conv_winograd_launch(const ConvParams& p /*w = U*/, hipStream_t stream);
// conv_pres.hip: patch-resident implicit GEMM of the wide 3x3 stride-1 layers in bf16 storage (chosen inside conv_igemm_plan)
bool conv_pres_applies(const ConvParams& p, int dtype);
void conv_pres_plan(ConvParams& p);
int conv_pres_launch(const ConvParams& p, hipStream_t stream);
bool conv_patch_applies(const ConvParams& p, int dtype);
bool conv_stem_applies(const ConvParams& p, int dtype);  // encoder.conv1 (7x7 stride 2, 4 staged channels)
bool conv_stem_bf16_applies(const ConvParams& p, int dtype);  // ... in bf16 storage (one 16-byte vector per pixel)
void conv_patch_plan(ConvParams& p, int dtype);
int conv_patch_launch(const ConvParams& p, hipStream_t stream);

// ---------------------------------------------------------------------------------------
// Weight gradient: conv_wgrad.hip
// ---------------------------------------------------------------------------------------
struct WgradParams {
  const void* dy;    // NHWC [B][Ho][Wo][Cout] (T)
  const void* src0;  // conv input, same gather description as ConvParams
  const void* src1;
  float* partial;    // [splits][Cout][KH*KW][Cin] fp32 slabs
  unsigned dy_bytes, src0_bytes, src1_bytes;  // extents for the buffer descriptors (filled by plan)
  int B, Hv, Wv, C0, C1, H0s, W0s, shift0;
  int Ho, Wo, Cout;
  int KH, KW, stride, pad;
  int M;        // B*Ho*Wo
  int cin_real; // real (unpadded) input channels when the caller knows them, else 0 = every channel counts
  int splits;   // pixel slabs
  int chunks_per_split;  // 32-pixel chunks per slab
  int tiles_co, tiles_ci;
  int patch;     // 0: tap-parallel kernel; >0: variant of the persistent patch kernel (conv_wgrad_patch.hip)
  double flops;  // algorithmic FLOPs of this launch, for profiling
  // Which part of the layer's gradient this launch computes (set before wgrad_plan; everything below is filled by it):
  //   WG_WHOLE  every input channel, KH*KW taps (the ordinary launch);
  //   WG_CLASS  conv(cat(upsample2x(src0), src1)), channels of src0 only, in OUTPUT-PARITY-CLASS form: inside class
  //             (py, px) the nine taps on the up-sampled operand touch a 2x2 neighbourhood of the low-resolution
  //             src0, so the launch runs 4 classes x 2x2 = 16 "folded taps" over M/4 pixels each (4*M*C0 MACs per
  //             filter instead of 9*M*C0) and the slab reduce adds the four folded taps every 3x3 tap belongs to;
  //   WG_SKIP   the same layer's src1 channels, plain 3x3 (a launch over the ci tiles of the second source).
  int part;
  int cls;        // 1: class form (WG_CLASS)
  int ci_base;    // first concatenated input channel of the launch (0, or C0 for WG_SKIP)
  int slab_cin;   // input channels the launch covers = row width of its slabs
  int slab_taps;  // taps per slab row block: KH*KW, or 16 folded taps
  int Mi, Hc, Wc; // pixel grid the k-loop iterates: (M, Ho, Wo), or one parity class (M/4, H0s, W0s)
  int step_img, step_row, step_col;  // one k-chunk (32 pixels) as whole images + rows + columns of that grid
  // two networks in one launch (NetSplit): grid.z = nets; net 1 shifts dy / src0 / src1 / partial by net_ws.  plan_nets
  // (set before the plan): the slab count aims at the workgroup target of the WHOLE launch of plan_nets networks.
  int nets, plan_nets;
  long net_ws;
};
__device__ __forceinline__ WgradParams wgrad_params_of_net(const WgradParams& pin, int net) {
  WgradParams p = pin;
  if (net != 0) {
    net_shift(p.dy, p.net_ws); net_shift(p.src0, p.net_ws); net_shift(p.src1, p.net_ws); net_shift(p.partial, p.net_ws);
  }
  return p;
}
enum WgradPart : int { WG_WHOLE = 0, WG_CLASS = 1, WG_SKIP = 2 };
// fills splits / tiles / extents for one launch; returns 0
int wgrad_plan(WgradParams& p, int dtype);
size_t wgrad_partial_floats(const WgradParams& p);
int wgrad_launch(const WgradParams& p, int dtype, hipStream_t stream);  // p.dy / p.src0 / p.src1 -> slabs in p.partial

// Slab reduces: the slabs [splits][CoutP][taps][Cin] are summed in a fixed order and written as the PyTorch-layout
// gradient [Cout][CinReal][KH][KW] (fp32), dropping padded channels.  A job's slabs cover `Cin` channels (`CinReal` of
// them real) that land at channel `c_off` of a gradient with `CinTot` input channels; fold != 0: the slabs hold 16
// folded taps per filter (WG_CLASS), summed into the 3x3 taps they belong to.  The jobs of one layer (1 or 2 passes)
// run as ONE launch.
constexpr int WG_BATCH = 20;
struct WgradReduceJob {
  const float* partial;
  float* dw;
  int splits, CoutP, Cin, CinReal, taps, CB, nsg, VB, CinTot, c_off, fold, Cout, cz, block0, lds;
};
struct WgradReduceBatch {
  WgradReduceJob job[WG_BATCH];
  int n = 0, blocks = 0, lds = 0;
  int nets = 1;               // two networks (NetSplit): grid.y = nets; net 1 reads its slabs at + net_ws, writes dw at + net_grad
  long net_ws = 0, net_grad = 0;
};
int wgrad_reduce_batch_add(WgradReduceBatch& tb, const float* partial, int splits, int CoutP, int Cout, int Cin,
                           int CinRealPart, int CinRealTotal, int c_off, int KH, int KW, int fold, float* dw);
int wgrad_reduce_batch_launch(const WgradReduceBatch& tb, hipStream_t stream);

// The weight gradient of one convolution as 1 or 2 passes (launch + slab reduce each): a decoder layer behind an
// up-sampling runs WG_CLASS + WG_SKIP when the tap-parallel kernel takes it, everything else WG_WHOLE.
struct WgradLayer {
  WgradParams part[2];
  int nparts;
};
// base: geometry of the layer (B, Hv, Wv, C0, C1, H0s, W0s, shift0, Ho, Wo, Cout (padded), KH, KW, stride, pad, M, flops)
int wgrad_layer_plan(WgradLayer& L, const WgradParams& base, int dtype);
size_t wgrad_layer_partial_floats(const WgradLayer& L);  // slab scratch: every pass keeps its own region
int wgrad_layer_launch(const WgradLayer& L, const void* dy, const void* src0, const void* src1, float* partial,
                       float* dw, int CoutReal, int CinReal, int dtype, hipStream_t stream);
// the passes' launches only; their reduce jobs are appended to `tb` for wgrad_reduce_batch_launch on the same stream
int wgrad_layer_launch_deferred(const WgradLayer& L, const void* dy, const void* src0, const void* src1, float* partial,
                                float* dw, int CoutReal, int CinReal, int dtype, WgradReduceBatch& tb,
                                hipStream_t stream, const NetSplit* ns = nullptr);

}  // namespace d3f
