// HBM-bound kernels of the U-Net hot path (pointwise.hip, loss.hip, optim.hip).
#pragma once
#include "common.h"

namespace d3f {

constexpr int PACK_MAX_LAYERS = 64;  // layers per kernel-argument table

// ---- BatchNorm, train mode (K5/K7) ------------------------------------------------------
// stats: per-m-tile partial (sum, sumsq) written by the conv epilogue.
// Produces mean / invstd (saved for backward), folded scale / shift, and updates the
// running statistics (momentum, unbiased variance) like torch.nn.BatchNorm2d.
int bn_finalize_launch(const float* stats, int tiles, int C, int Cpad, long count,
                       const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, float* mean, float* invstd,
                       float* scale, float* shift, hipStream_t stream, const NetSplit* ns = nullptr);
// eval mode: scale/shift from the running statistics
int bn_eval_coeff_launch(const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, int C, float* scale, float* shift,
                         hipStream_t stream);
// the same for every BatchNorm of a network in one launch: scale -> coef[2*C..], shift -> coef[3*C..]
struct BnEvalEntry {
  uint32_t g_off, b_off, rm_off, rv_off;  // floats into params / bnstats
  uint32_t coef_off16;                     // 16-byte units into the workspace
  int32_t C;
};
struct BnEvalTable {
  int n;
  BnEvalEntry e[PACK_MAX_LAYERS];
};
int bn_eval_coeff_all_launch(const float* params, const float* bnstats, void* ws, float eps, const BnEvalTable& t,
                             hipStream_t stream);
// a = act(y*scale + shift + residual),  residual = res (activation) or yr*scale_r + shift_r
int bn_apply_launch(int dtype, const void* y, const float* scale, const float* shift,
                    const void* res, const void* yr, const float* scale_r, const float* shift_r,
                    int relu, void* out, long rows, int C, hipStream_t stream, const NetSplit* ns = nullptr);
// backward: dz = dA * (a > 0 if a given);  partial sums of dz and dz*xhat
int bn_bwd_reduce_launch(int dtype, const void* dA, const void* a, const void* y,
                         const float* mean, const float* invstd, float* partial, int* nblocks,
                         long rows, int C, hipStream_t stream, const float* mask_scale = nullptr,
                         const float* mask_shift = nullptr, const NetSplit* ns = nullptr);
int bn_bwd_reduce_blocks(long rows, int C, int dtype);
int bn_bwd_finalize_launch(const float* partial, int nblocks, int C, long count,
                           const float* gamma, const float* invstd, float* dgamma, float* dbeta,
                           int accumulate, float* coef /*[3][C]*/, hipStream_t stream, const NetSplit* ns = nullptr);
// dy = k1*(dz - k2 - xhat*k3); optionally dz -> dres (+= if dres_acc)
int bn_bwd_apply_launch(int dtype, const void* dA, const void* a, const void* y, const float* mean,
                        const float* invstd, const float* coef, void* dy, void* dres, int dres_acc,
                        long rows, int C, hipStream_t stream, const float* mask_scale = nullptr,
                        const float* mask_shift = nullptr, const NetSplit* ns = nullptr);

// bn_fused.hip: the finalize step folded into the streaming pass (fp32 tensors, C % 32 == 0, few partial rows)
bool bn_fused_finalize_ok(int dtype, int stat_rows, int C);
int bn_finalize_apply_launch(int dtype, const float* stats, int stat_rows, int C, int Cpad, long count, const float* gamma,
                             const float* beta, float eps, float momentum, float* running_mean,
                             float* running_var, float* mean, float* invstd, float* scale, float* shift,
                             const void* y, const void* res, const void* yr, const float* scale_r,
                             const float* shift_r, int relu, void* out, long rows, hipStream_t stream,
                             const NetSplit* ns = nullptr, int plan_nets = 1);
int bn_bwd_finalize_apply_launch(int dtype, const float* partial, int nblocks, int C, long count, const float* gamma,
                                 const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                 int accumulate, float* coef, const void* dA, const void* a, const void* y, void* dy,
                                 void* dres, int dres_acc, long rows, hipStream_t stream,
                                 const float* mask_scale = nullptr, const float* mask_shift = nullptr,
                                 const NetSplit* ns = nullptr, int plan_nets = 1);

// ---- pooling / resampling / layout (K6, K8 backward, boundary) ---------------------------
int maxpool3x3s2_fwd_launch(int dtype, const void* in, void* out, uint8_t* idx, int B, int H, int W,
                            int C, hipStream_t stream, const NetSplit* ns = nullptr);
int maxpool3x3s2_bwd_launch(int dtype, const void* dout, const uint8_t* idx, void* din, int accumulate,
                            int B, int H, int W, int C, hipStream_t stream, const NetSplit* ns = nullptr);
// dlow[b,y,x,c] = sum of the 2x2 block of dfull  (backward of nearest x2 up-sampling)
int sum2x2_launch(int dtype, const void* dfull, void* dlow, int B, int Hl, int Wl, int C,
                  hipStream_t stream, const NetSplit* ns = nullptr);
// NCHW fp32 [B][C][H][W] -> NHWC T [B][H][W][Cpad] (pad channels zero) and back
int nchw_to_nhwc_launch(int dtype, const float* in, void* out, int B, int C, int H, int W, int Cpad,
                        hipStream_t stream, const NetSplit* ns = nullptr);
int nhwc_to_nchw_launch(int dtype, const void* in, float* out, int B, int C, int H, int W, int Cpad,
                        hipStream_t stream);

// K16: uint8 BGR frames <-> normalised activations (mean255 = mean*255, std255 = std*255, RGB order)
int u8bgr_to_nhwc_launch(int dtype, const uint8_t* in, void* out, long npix, int Cpad, const float mean255[3],
                         const float std255[3], hipStream_t stream);
int nchw_to_u8bgr_launch(const float* in, uint8_t* out, int B, long HW, const float mean255[3],
                         const float std255[3], hipStream_t stream);

// input pipeline: uint8 RGB [B][H][W][3] -> NCHW fp32, ((float)u8 / 255 - mean[c]) / std[c]
int u8rgb_to_nchw_launch(const uint8_t* in, float* out, int B, long HW, const float mean[3], const float stdv[3],
                         hipStream_t stream);

// K17: affine_grid + grid_sample(bilinear, zeros, align_corners=False) on NCHW fp32, theta [B][2][3]
int affine_warp_launch(const float* in, const float* theta, float* out, int B, int C, int H, int W,
                       hipStream_t stream);

// ---- weights -------------------------------------------------------------------------------
// PyTorch [Cout][CinReal][KH][KW] fp32 -> forward pack [CoutPad][Kpad] (k = tap*Cin + c) and/or
// data-gradient pack [CinPadRows][KpadD] (k = flipped tap*Cout + co); T = dtype.
int pack_weights_launch(int dtype, const float* w, int Cout, int CinReal, int Cin, int KH, int KW,
                        void* wf, int CoutPad, int Kpad, void* wd, int CinRows, int KpadD, int stride,
                        hipStream_t stream);

// conv(cat(upsample2x(x), skip)) with the up-sampling folded into pre-summed weights (pointwise.hip): per-class
// forward matrices wfc [4][CoutPad][4*C0 + 9*C1], the 4x4 stride-2 data-gradient matrix wd4 [C0Rows][16*CoutD]
// (gradient w.r.t. the low-resolution source) and the skip tensor's 3x3 data-gradient matrix wds [C1Rows][9*CoutD]
int pack_up_launch(int dtype, const float* w, int Cout, int C0, int C1, void* wfc, int CoutPad, void* wd4,
                   int C0Rows, void* wds, int C1Rows, hipStream_t stream);

// every layer of a network in one launch (engine): table passed by value as a kernel argument
constexpr int PACK_NT = 32;        // filters per tile
constexpr int PACK_LDS_ROW = 288;  // floats per filter in a tile: CT channels x taps  (32 x 9)
struct PackEntry {
  uint32_t w_off;               // floats into the flat parameter buffer
  uint32_t wf_off16, wd_off16;  // 16-byte units into the workspace
  uint32_t block0;              // first block of this layer; blocks are (filter tile, channel tile)
  uint16_t Cout, CinReal, Cin, taps, CoutPad, Kpad, CinRows, CoutD, KpadD, has_d, CT, ctiles;
  uint16_t conv_stride;         // 2: data-gradient taps stored parity class by class (dgrad_tap_slot_to_flipped)
  uint16_t pad0;
  uint16_t taps_shr, ct_log2;   // index arithmetic without divisions: q / taps = umulhi(q, taps_mul) >> taps_shr
  uint32_t taps_mul;            // (fast_div_setup, common.h; 0 = one tap), CT = 1 << ct_log2
};
struct PackTable {
  int n;
  PackEntry e[PACK_MAX_LAYERS];
};
int pack_all_launch(int dtype, const float* params, void* ws, const PackTable& t, int blocks,
                    hipStream_t stream);

// ---- noise blend (K12), loss (K13), Adam (K14), EMA (K15) --------------------------------
int noise_blend_launch(const float* x, const float* noise, const float* y_uniform, float lam,
                       float* out, float* r_out, int B, long per_image, hipStream_t stream);
int noise_blend_fixed_launch(const float* x, const float* noise, const float* r, float* out, int B, long per_image,
                             hipStream_t stream);
size_t l1_per_image_workspace_bytes(int B);
int l1_per_image_launch(const float* pred, const float* target, float* out, void* workspace, int B, long per_image,
                        hipStream_t stream);
size_t loss_workspace_floats(int B, int H, int W);
int mse_ssim_loss_launch(const float* pred, const float* target, float in_min, float in_max,
                         float* loss_out /*[3]: loss, mse, ssim*/, float* grad_pred, float* workspace,
                         int B, int H, int W, hipStream_t stream);
int adam_step_launch(float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                     float beta2, float eps, int step, float grad_scale, hipStream_t stream);
void adam_coefficients(float lr, float beta1, float beta2, float eps, int step, float grad_scale, float coef[8]);
int adam_step_dev_launch(float* p, const float* g, float* m, float* v, long n, const float* coef_dev,
                         hipStream_t stream);
int ema_lerp_launch(float* ema, const float* online, long n, float weight, hipStream_t stream);

}  // namespace d3f
