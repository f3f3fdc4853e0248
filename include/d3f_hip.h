/* libd3f_hip.so -- C ABI of the MI355X-native U-Net hot path of d3f
 * (ChainBreak/denoising_diffusion_deep_fake).
 *
 * The reference is pure Python on PyTorch: its "FFI" for this path is the torch op
 * dispatch under `segmentation_models_pytorch.Unet(...)`, `MseStructuralSimilarityLoss`,
 * `torch.optim.Adam` and `ema_pytorch.EMA`.  Each entry point below names the reference
 * call site (file:line under /root/reference) whose device work it replaces.  The Python
 * binding a maintainer adds is a ctypes stub: see INTEGRATION.md and
 * denoising_diffusion_deep_fake_amd/_lib.py.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; d3f_last_error() gives the
 *     thread-local message.
 *   - all pointers are DEVICE pointers borrowed for the duration of the call (memory is
 *     owned by the caller, i.e. PyTorch's allocator); nothing is allocated or freed on the
 *     device by this library.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *     Calls only enqueue work; they never synchronise.
 *   - dtype: 0 = f32, 1 = bf16 (activations / packed weights; master params are f32).
 *   - network boundary tensors are NCHW f32 (what the reference's callers hold);
 *     internal activations are NHWC with channels padded to a 16-byte multiple.
 */
#ifndef D3F_HIP_H
#define D3F_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D3F_F32 0
#define D3F_BF16 1
/* fp32 tensors and fp32 accumulation; the contraction kernels split every fp32 operand exactly into three bf16
 * terms and form each product from six bf16 MFMAs (dropped terms <= 2^-24 |a*b|): fp32-grade results at 6/16 of
 * the fp32 MFMA time.  Everything that is not a contraction is identical to D3F_F32. */
#define D3F_F32X3 2

int d3f_version(void);
const char* d3f_last_error(void);
/* First 16 hex digits of the sha256 over the sources this library was built from (the .hip and .h files of csrc/ and this header; the
 * Makefile bakes it in).  The Python binding refuses a library whose digest differs from the sources next to it
 * (_lib.lib(); D3F_LIB=<path> loads a variant build on purpose and skips the check); bench.py prints it in `config`. */
const char* d3f_source_digest(void);

/* Measurement hooks: when enabled, every launch of the three contraction kernels
 * (class 0 = conv forward, 1 = conv data-gradient, 2 = conv weight-gradient) is bracketed by a pair
 * of HIP events on the launch stream.  collect() waits for them and returns, per class, the summed
 * kernel time (ms), the number of launches and their algorithmic FLOPs, then resets the buffer.
 * max_launches <= 0 disables.  collect() returns 1 if the buffer overflowed.
 * d3f_profile_classes(mask) restricts the bracketing to the classes whose bit is set (default 7 = all):
 * an event pair costs a few microseconds of stream time, so a timed run brackets only what it reports. */
int d3f_profile_enable(int max_launches);
int d3f_profile_classes(int mask);
int d3f_profile_collect(double ms[3], int64_t launches[3], double flops[3]);

/* ---------------------------------------------------------------------------------------
 * Whole-network handle: Unet(encoder_name, encoder_weights=None, in_channels, classes,
 * activation=None) for a fixed (B, H, W, dtype).
 * Replaces: model construction  d3f/train_denoiser/lit_module.py:41-53,
 *                               d3f/train_deep_fake/lit_module.py:49-60
 *           forward             d3f/train_denoiser/lit_module.py:117 (self.model(image_noisy)),
 *                               d3f/train_deep_fake/lit_module.py:173,189,195,266
 *           backward            Lightning's loss.backward() on that module
 * ------------------------------------------------------------------------------------- */
typedef struct d3f_unet* d3f_unet_t;

int d3f_unet_create(const char* encoder_name, int in_channels, int classes, int B, int H, int W,
                    int dtype, d3f_unet_t* out);
int d3f_unet_destroy(d3f_unet_t h);

/* ---------------------------------------------------------------------------------------
 * Two networks, one set of launches.
 * Replaces: the two optimizer steps of one train_deep_fake batch in `mode: "denoise"` --
 *           training_step(batch, batch_idx, optimizer_idx = 0 | 1) -> training_denoise_step_for_one_model("a", batch_a,
 *           model_a) / ("b", batch_b, model_b), d3f/train_deep_fake/lit_module.py:142-181 -- whose forward and backward
 *           passes are INDEPENDENT (model_a never sees domain b) and identical in shape, 8 images each: half of what
 *           fills an MI355X.  A pair handle runs both networks' layer k as ONE kernel launch (gridDim.z carries the
 *           network; a workgroup of network 1 adds byte offsets to its pointers), so every launch has the occupancy of a
 *           16-image batch.  Results are bit for bit those of a single handle created with plan_nets = 2 and run net by
 *           net; swap mode (coupled through the EMA teachers, :183-206) stays sequential.
 * d3f_unet_create_nets: nets = 1 | 2 networks per launch, B images PER network; plan_nets (>= nets): the tile / split-K /
 *           slab / patch-kernel choices count the workgroups of plan_nets networks (nets = 1, plan_nets = 2: the pair's
 *           kernels on one network -- the reference run of the bit-identity test).  d3f_unet_create == (1, 1).
 * Workspace of a pair: ONE buffer of d3f_unet_workspace_bytes(h) bytes = two copies of the single-network layout,
 *           d3f_unet_net_workspace_stride(h) bytes apart (d3f_unet_export on network 1: pass workspace + stride).
 * The pair entry points take the two networks' buffers as arrays of two pointers (any two allocations); train-mode
 *           forward only, per-GPU BatchNorm statistics only (d3f_unet_set_bn_sync is refused).  Packing, forward and
 *           backward of a pair handle go through these three calls; everything per network (noise blend, loss, Adam,
 *           EMA) stays the single-network entry points, called once per network.
 * ------------------------------------------------------------------------------------- */
int d3f_unet_create_nets(const char* encoder_name, int in_channels, int classes, int B, int H, int W, int dtype,
                         int nets, int plan_nets, d3f_unet_t* out);
int d3f_unet_nets(d3f_unet_t h);
size_t d3f_unet_net_workspace_stride(d3f_unet_t h);
int d3f_unet_pair_pack_weights(d3f_unet_t h, const float* const params[2], void* workspace, void* stream);
int d3f_unet_pair_forward(d3f_unet_t h, const float* const params[2], float* const bnstats[2], const float* const x[2],
                          float* const out[2], void* workspace, void* stream);
/* join != 0: every gradient of the segments is final on `stream` on return (d3f_unet_backward); join == 0: the
 * d3f_unet_backward_nojoin contract (final on the engine's side stream; d3f_unet_backward_join later) */
int d3f_unet_pair_backward(d3f_unet_t h, const float* const params[2], const float* const grad_out[2],
                           float* const grads[2], void* workspace, int seg_begin, int seg_end, int join, void* stream);

/* parameter table, in torch named_parameters() order; offsets are in floats into ONE flat
 * f32 buffer that holds every parameter (gradients use the same layout). */
int d3f_unet_num_params(d3f_unet_t h);
int d3f_unet_param_info(d3f_unet_t h, int i, char* name, int name_cap, int32_t shape[4], int* ndim,
                        int64_t* offset);
int64_t d3f_unet_param_floats(d3f_unet_t h);
/* BatchNorm running statistics: flat f32 buffer, per layer running_mean[C] then running_var[C] */
int d3f_unet_num_bn(d3f_unet_t h);
int d3f_unet_bn_info(d3f_unet_t h, int i, char* prefix, int prefix_cap, int* C, int64_t* rm_offset,
                     int64_t* rv_offset);
int64_t d3f_unet_bnstat_floats(d3f_unet_t h);
size_t d3f_unet_workspace_bytes(d3f_unet_t h);
/* algorithmic conv FLOPs (2*MAC, unpadded channels) of one forward / one backward call */
double d3f_unet_forward_flops(d3f_unet_t h);
double d3f_unet_backward_flops(d3f_unet_t h);

/* re-pack the f32 master weights into the kernels' layouts (call after every parameter update) */
int d3f_unet_pack_weights(d3f_unet_t h, const float* params, void* workspace, void* stream);
/* x, out: NCHW f32 [B][in_channels|classes][H][W].  training != 0: BatchNorm uses batch statistics,
 * updates bnstats (momentum 0.1) and keeps what backward needs in the workspace. */
int d3f_unet_forward(d3f_unet_t h, const float* params, float* bnstats, const float* x, float* out,
                     void* workspace, int training, void* stream);
/* The eval-mode forward (training == 0 above: BatchNorm running statistics folded into the conv epilogues) replayed
 * from a hipGraph that is captured on first use per set of pointers (params, bnstats, x, out, workspace) -- the
 * "hipGraph-captured denoise step" of BASELINE.json configs[4]; the reference's frame loop is
 * d3f/script_tools/put_video_through_fake_model.py:111-119 -> d3f/train_deep_fake/lit_module.py:259-270.  Results are
 * bit-identical to d3f_unet_forward(training = 0); parameter VALUES are read at replay time (only pointers are baked
 * in), so the graph survives optimiser / EMA updates followed by d3f_unet_pack_weights. */
int d3f_unet_forward_graph(d3f_unet_t h, const float* params, float* bnstats, const float* x, float* out,
                           void* workspace, void* stream);
/* Inference entry behind LitModule.predict_fake_for_single_frame (d3f/train_deep_fake/lit_module.py:259-300):
 * uint8 BGR frames [B][H][W][3] in device memory -> eval-mode forward (BatchNorm folded into the conv
 * epilogues) -> uint8 BGR frames, with cv2_to_tensor_normalised (:272-283: BGR->RGB, (x - mean*255) / (std*255))
 * and tensor_cv2_to_denormalised (:285-300: x*std*255 + mean*255, .int() truncation, clamp 0..255, RGB->BGR)
 * fused into the first and last kernel.  mean / std: 3 host floats (RGB order).  use_graph != 0: the launch
 * sequence is captured into a hipGraph on first use (per set of pointers) and replayed afterwards. */
int d3f_unet_predict_u8(d3f_unet_t h, const float* params, float* bnstats, const uint8_t* bgr_in, uint8_t* bgr_out,
                        const float mean[3], const float std[3], void* workspace, int use_graph, void* stream);
/* gradients of every parameter (written, not accumulated) for the preceding training forward.
 * The backward pass is cut into d3f_unet_num_segments() buckets so a data-parallel caller can
 * all-reduce bucket k while bucket k+1 computes: run segments [seg_begin, seg_end) in order 0..n;
 * after segment k, grads[begin_k, end_k) (floats) are final. */
int d3f_unet_num_segments(d3f_unet_t h);
/* Which kernel family the plan chose for every launch of a training step (a regression guard: falling back to the
 * implicit GEMM is silent otherwise).  fwd / dgrad count launches by ConvParams::patch (0 = conv_igemm_kernel, 1 / 3-7 =
 * conv_patch_kernel forms, 2 / 8 = conv_stem[_bf16]_kernel, 9-12 = conv_pres_kernel for 64 / 128 / 256 / 512 channels;
 * slot 15 of fwd = conv_winograd_kernel), wgrad by WgradParams::patch (0 = tap-parallel conv_wgrad_kernel, 1-6 = the
 * fp32-MFMA patch kernels incl. the stem's, 7 = conv_wgrad_patch_bf16_kernel).  Replaces nothing in the reference: the
 * conv2d dispatches of smp.Unet under d3f/train_denoiser/lit_module.py:117 choose their kernels inside ATen. */
int d3f_unet_plan_counts(d3f_unet_t h, int32_t fwd[16], int32_t dgrad[16], int32_t wgrad[16]);
int d3f_unet_segment_range(d3f_unet_t h, int segment, int64_t* begin, int64_t* end);
int d3f_unet_backward(d3f_unet_t h, const float* params, const float* grad_out, float* grads,
                      void* workspace, int seg_begin, int seg_end, void* stream);
/* Data-parallel form (BASELINE.json: "RCCL all-reduce of gradients over xGMI overlapped with the backward pass";
 * the reference itself is single-device, d3f/train_deep_fake/start_training.py:43-48).  d3f_unet_backward makes the
 * caller's stream -- the critical path of the backward pass -- wait for the engine's weight-gradient stream before it
 * returns; called once per bucket that stalls the chain four times.  d3f_unet_backward_nojoin enqueues the same work
 * and makes NOTHING wait on the caller's stream: the gradients of the segments are final on the engine's side stream
 * (d3f_unet_side_stream; NULL in D3F_SERIAL_BACKWARD mode = the caller's stream) once everything enqueued there so far
 * has run, so a collective ordered behind THAT stream overlaps the remaining segments.  d3f_unet_backward_join makes
 * `stream` wait for the side stream (call once, before the optimiser reads the gradients). */
int d3f_unet_backward_nojoin(d3f_unet_t h, const float* params, const float* grad_out, float* grads,
                             void* workspace, int seg_begin, int seg_end, void* stream);
int d3f_unet_side_stream(d3f_unet_t h, void** stream_out);
int d3f_unet_backward_join(d3f_unet_t h, void* stream);
/* One whole optimiser step of the reference's noisy -> clean objective as ONE call -- what Lightning's automatic
 * optimisation runs around d3f/train_denoiser/lit_module.py:107-126 (and training_denoise_step_for_one_model,
 * d3f/train_deep_fake/lit_module.py:162-181): pack weights -> blend_random_amount_of_noise_with_each_sample (the noise
 * and the uniform draws come from the caller: RNG stays outside) -> forward -> MseStructuralSimilarityLoss -> backward ->
 * Adam.  use_graph != 0: the ~330 kernel launches over the engine's streams are captured into a hipGraph on first use
 * (per set of pointers) and replayed with one launch; results are bit-identical to the separate calls.  Every buffer is
 * the caller's and must stay where it is between calls (pointers are baked into the graph); `adam_coef` is DEVICE memory
 * holding the 8 floats d3f_adam_coefficients() computes on the host for this step (lr, betas, bias corrections), copied
 * there stream-ordered before the call. */
typedef struct d3f_step_buffers {
  float* params; float* bnstats; float* grads; float* exp_avg; float* exp_avg_sq;  /* flat buffers of the network */
  const float* image;      /* [B][3][H][W] clean batch = the loss target */
  const float* noise;      /* randn, same shape */
  const float* y_uniform;  /* rand(B) */
  float* noisy;            /* scratch [B][3][H][W] */
  float* pred;             /* network output [B][3][H][W] */
  float* grad_pred;        /* d loss / d pred */
  float* loss_out;         /* {loss, mse, ssim} */
  void* loss_workspace;    /* d3f_mse_ssim_loss_workspace_bytes(B, H, W) */
  const float* adam_coef;  /* device, 8 floats */
} d3f_step_buffers;
int d3f_adam_coefficients(float lr, float beta1, float beta2, float eps, int step, float grad_scale, float coef[8]);
int d3f_unet_train_step(d3f_unet_t h, const d3f_step_buffers* buffers, float lambda, float input_min, float input_max,
                        void* workspace, int use_graph, void* stream);
/* Optional synchronised BatchNorm (what pytorch_lightning's Trainer(sync_batchnorm=True) would add to the
 * reference's Trainer(gpus=1) at d3f/train_deep_fake/start_training.py:43-48): with a callback installed, every
 * train-mode BatchNorm takes its batch statistics -- and the two sums of its backward pass -- over ALL ranks' batches.
 * fn must sum-all-reduce `count` floats at `data` (a region of the workspace handed to forward / backward) in place,
 * ordered on `stream`, and return 0; it is called from inside d3f_unet_forward / d3f_unet_backward, once per BatchNorm
 * layer each.  fn == NULL: per-GPU statistics (the default).  The fused finalize kernels are bypassed in this mode. */
typedef int (*d3f_allreduce_fn)(void* ctx, float* data, int64_t count, void* stream);
int d3f_unet_set_bn_sync(d3f_unet_t h, d3f_allreduce_fn fn, void* ctx, int world_size);
/* debugging / tests: copy an internal activation ("<conv name>:y" raw conv output, ":a" post
 * BN+ReLU, ":da" its gradient) to NCHW f32 */
int d3f_unet_export(d3f_unet_t h, const char* name, const void* workspace, float* out_nchw, void* stream);
/* extent of that tensor: dims = {channels (as stored: padded to the vector width), height, width} */
int d3f_unet_export_shape(d3f_unet_t h, const char* name, int32_t dims[3]);

/* ---------------------------------------------------------------------------------------
 * Single operators (the kernels the network is made of; used by the parity tests)
 * ------------------------------------------------------------------------------------- */
typedef struct d3f_conv_desc {
  int32_t B, H, W;    /* extent of the conv input (after the optional x2 up-sampling of src0) */
  int32_t C0, C1;     /* channels of src0 and of the concatenated src1 (0: none); padded to 4 (f32) / 8 (bf16) */
  int32_t upsample0;  /* 1: src0 is [B][H/2][W/2][C0], read through nearest x2 up-sampling; 2: the same, and
                       * d3f_conv_backward_data may hand back dx0 at that LOW resolution (see there) */
  int32_t Cout, KH, KW, stride, pad;
  int32_t CinReal;    /* unpadded input channels of the f32 master weight [Cout][CinReal][KH][KW] */
} d3f_conv_desc;

/* torch.nn.Conv2d weight -> packed forward (which=0) / data-gradient (which=1) operand */
size_t d3f_conv_packed_bytes(int dtype, const d3f_conv_desc* d, int which);
int d3f_conv_pack_weights(int dtype, const d3f_conv_desc* d, const float* w, void* w_fwd, void* w_dgrad,
                          void* stream);
/* y = conv(cat(up(src0), src1)) NHWC; stats != NULL: per-channel (sum, sumsq) partials for
 * d3f_bn_finalize, d3f_conv_stats_floats() floats.  Replaces F.interpolate + torch.cat + conv2d
 * of smp's DecoderBlock / torchvision BasicBlock under lit_module.py:117. */
/* `workspace` (optional, d3f_conv_workspace_bytes(.., which) bytes; which 0 = forward, 1 = data
 * gradient) lets layers whose M x Cout yields too few workgroups split their K loop (split-K slabs
 * + fixed-order reduce); the statistics partial layout then follows the reduce kernel, so pass the
 * same with_workspace flag to d3f_conv_stats_floats. */
size_t d3f_conv_workspace_bytes(int dtype, const d3f_conv_desc* d, int which);
size_t d3f_conv_stats_floats(int dtype, const d3f_conv_desc* d, int with_workspace, int* tiles);
int d3f_conv_forward(int dtype, const d3f_conv_desc* d, const void* src0, const void* src1,
                     const void* w_fwd, void* y, float* stats, void* workspace, void* stream);
/* The Winograd F(2x2, 3x3) form of the same convolution for fp32 layers with a 3x3 / stride 1 / pad 1 kernel, one
 * source, H and W multiples of 16, C0 a multiple of 16 and Cout a multiple of 64 (any other shape: D3F_EINVAL) -- the
 * conv2d of torchvision's BasicBlock under smp.Unet(resnet34), d3f/train_denoiser/lit_module.py:46-52, :117.  The
 * whole-network plan takes it where d3f_conv_winograd_applies() == 1 (at least 256 workgroups of 16x16 pixels x 64
 * filters); this entry runs it on any shape it fits.  u: filters transformed by d3f_conv_winograd_pack from the torch
 * weight [Cout][C0][3][3] (d3f_conv_winograd_filter_bytes bytes).  scale == NULL: y = conv(src0), stats (optional) =
 * one (sum, sumsq) row per workgroup, d3f_conv_winograd_stats_floats() floats for d3f_bn_finalize.  scale != NULL:
 * the eval epilogue y = relu?(conv(src0) * scale[c] + shift[c] + residual?) of the BatchNorm-folded forward. */
int d3f_conv_winograd_applies(int dtype, const d3f_conv_desc* d);
size_t d3f_conv_winograd_filter_bytes(const d3f_conv_desc* d);
size_t d3f_conv_winograd_stats_floats(const d3f_conv_desc* d, int* tiles);
int d3f_conv_winograd_pack(const d3f_conv_desc* d, const float* w, void* u, void* stream);
int d3f_conv_winograd_forward(const d3f_conv_desc* d, const void* src0, const void* u, void* y, float* stats,
                              const float* scale, const float* shift, const void* residual, int relu, void* stream);
/* dx over the conv input: channels [0,C0) -> dx0, [C0,C0+C1) -> dx1; acc*: add to the destination instead of
 * overwriting.  With upsample0, dx0 is the gradient of the LOW-resolution source [B][H/2][W/2][C0] when the layer
 * runs with the up-sampling folded into pre-summed weights (d3f_conv_upsample_folded() == 1: 3x3, stride 1, pad 1,
 * whole k-tiles per tap -- every decoder layer of the network), else the full-resolution [B][H][W][C0] gradient of
 * the up-sampled operand, to be reduced with d3f_upsample2x_backward.  Opt-in third form: a caller that sets
 * upsample0 = 2 in the descriptor AND finds d3f_conv_upsample_summed() == 1 (bf16 storage, 16 -> 32 channels, one source:
 * decoder.blocks.4.conv1 of smp's UnetDecoder, the F.interpolate + conv2d pair under d3f/train_denoiser/lit_module.py:117)
 * gets the 2x2 blocks summed in the launch's epilogue: dx0 is the LOW-resolution gradient as in the folded case.  With
 * upsample0 = 1 that layer keeps the full-resolution contract. */
int d3f_conv_upsample_folded(int dtype, const d3f_conv_desc* d);
int d3f_conv_upsample_summed(int dtype, const d3f_conv_desc* d);
int d3f_conv_backward_data(int dtype, const d3f_conv_desc* d, const void* dy, const void* w_dgrad,
                           void* dx0, void* dx1, int acc0, int acc1, void* workspace, void* stream);
/* dw in torch layout [Cout][CinReal][KH][KW] f32 */
size_t d3f_conv_backward_weight_workspace_bytes(int dtype, const d3f_conv_desc* d);
int d3f_conv_backward_weight(int dtype, const d3f_conv_desc* d, const void* dy, const void* src0,
                             const void* src1, void* workspace, float* dw, void* stream);

/* BatchNorm2d, train mode (eps 1e-5, momentum 0.1): finalize conv statistics, then
 * a = relu?(y*scale + shift + residual); coef = mean[C] invstd[C] scale[C] shift[C] */
int d3f_bn_finalize(const float* stats, int tiles, int C, int64_t count, const float* gamma,
                    const float* beta, float* running_mean, float* running_var, float* coef, void* stream);
int d3f_bn_apply(int dtype, const void* y, const float* coef, int C, int64_t rows, const void* residual,
                 int relu, void* out, void* stream);
/* backward of a = relu?(bn(y) + residual): dy, dgamma, dbeta, and (dres != NULL) dz for the residual */
size_t d3f_bn_backward_workspace_bytes(int dtype, int C, int64_t rows);
int d3f_bn_backward(int dtype, const void* dA, const void* a_or_null, const void* y, const float* coef,
                    const float* gamma, int C, int64_t rows, void* dy, void* dres, float* dgamma,
                    float* dbeta, void* workspace, void* stream);

int d3f_maxpool3x3s2_forward(int dtype, const void* in, void* out, uint8_t* idx, int B, int H, int W, int C, void* stream);
int d3f_maxpool3x3s2_backward(int dtype, const void* dout, const uint8_t* idx, void* din, int accumulate,
                              int B, int H, int W, int C, void* stream);
int d3f_upsample2x_backward(int dtype, const void* dfull, void* dlow, int B, int Hlow, int Wlow, int C, void* stream);
int d3f_nchw_to_nhwc(int dtype, const float* in, void* out, int B, int C, int H, int W, int Cpad, void* stream);
int d3f_nhwc_to_nchw(int dtype, const void* in, float* out, int B, int C, int H, int W, int Cpad, void* stream);
/* Host half of the input pipeline moved to the device (d3f/train_deep_fake/lit_module.py:100-110: A.Normalize(mean, std,
 * max_pixel_value=255) + ToTensorV2 on the HWC uint8 RGB image d3f/dataset/image_dataset.py:37-41 hands the transform):
 * in_hwc [B][H][W][3] uint8 RGB -> out_nchw [B][3][H][W] f32 = ((float)u8 / 255 - mean[c]) / std[c], the transform's
 * own order of fp32 operations (bit-identical); the batch crosses worker IPC and PCIe as bytes. */
int d3f_u8rgb_normalise(const uint8_t* in_hwc, float* out_nchw, int B, int H, int W, const float mean[3],
                        const float std[3], void* stream);

/* GPU-side augmentation of the training step (d3f/train_denoiser/lit_module.py:55-65 RandomAffine, applied at :113):
 * out[b] = grid_sample(in[b], affine_grid(theta[b]), bilinear, zeros padding, align_corners=False), NCHW f32,
 * theta [B][2][3] row-major (normalised output -> input coordinates).  in and out must not alias. */
int d3f_affine_warp(const float* in, const float* theta, float* out, int B, int C, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------
 * Training-step arithmetic around the network
 * ------------------------------------------------------------------------------------- */
/* blend_random_amount_of_noise_with_each_sample + sample_random_number_from_exponential_distribution
 * (d3f/train_denoiser/lit_module.py:128-153 == d3f/train_deep_fake/lit_module.py:208-233) given the
 * caller's draws noise ~ N(0,1) [B][per_image] and y ~ U[0,1) [B]:
 *   r = (1/lam) * log(1 / (y*(1-c) + c)), c = exp(-lam);  out = sqrt(1-r)*x + sqrt(r)*noise */
int d3f_noise_blend(const float* x, const float* noise, const float* y_uniform, float lam, float* out,
                    float* r_out_or_null, int B, int64_t per_image, void* stream);
/* blend_fixed_amount_of_noise_with_each_sample (d3f/balance_training_images/lit_module.py:109-121) given the
 * caller's noise and the per-image ratios r [B] (device): out = sqrt(1-r)*x + sqrt(r)*noise */
int d3f_noise_blend_fixed(const float* x, const float* noise, const float* r, float* out, int B, int64_t per_image,
                          void* stream);
/* compute_difficulty_loss (d3f/balance_training_images/lit_module.py:139-142): out[b] = mean |prediction - target|
 * over image b; deterministic two-pass sum (f64 partials in the workspace) */
size_t d3f_l1_per_image_workspace_bytes(int B);
int d3f_l1_per_image(const float* prediction, const float* target, float* out, void* workspace, int B,
                     int64_t per_image, void* stream);
/* MseStructuralSimilarityLoss(input_min, input_max)(prediction, target)
 * (d3f/loss_functions/structural_similarity_loss.py:14-26; piqa.SSIM defaults) on NCHW f32
 * [B][3][H][W]: loss_out = {loss, mse, ssim}; grad_pred = d loss / d prediction. */
size_t d3f_mse_ssim_loss_workspace_bytes(int B, int H, int W);
int d3f_mse_ssim_loss(const float* pred, const float* target, float input_min, float input_max,
                      float* loss_out, float* grad_pred, void* workspace, int B, int H, int W, void* stream);
/* torch.optim.Adam step over a flat buffer (lit_module.py:95 / train_deep_fake/lit_module.py:116-120);
 * step counts from 1; grad_scale multiplies the gradient first (1/world_size for data parallel) */
int d3f_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, int step, float grad_scale, void* stream);
/* ema.lerp_(online, weight)  (ema_pytorch update used at d3f/train_deep_fake/lit_module.py:185) */
int d3f_ema_lerp(float* ema, const float* online, int64_t n, float weight, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* D3F_HIP_H */
