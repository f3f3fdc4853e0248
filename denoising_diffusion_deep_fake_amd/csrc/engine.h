// Whole-network engine for Unet(resnet18|resnet34): a static plan of the layer graph
// (units = conv [+ BatchNorm + ReLU + residual]), the workspace layout and the forward /
// backward launch sequences.  No device memory is allocated here: the caller (PyTorch)
// owns params, grads, BN statistics and one workspace buffer; the engine only enqueues
// kernels on the caller's stream.
#pragma once
#include <string>
#include <vector>

#include "common.h"
#include "pointwise.h"

namespace d3f {

struct TensorD {
  size_t off = 0;  // bytes into the workspace
  int H = 0, W = 0, C = 0;
  long elems(int B) const { return (long)B * H * W * C; }
};

struct ParamInfo {
  std::string name;
  int shape[4];
  int ndim;
  long offset;  // floats into the flat parameter (and gradient) buffer
  long numel;
};

struct BnInfo {
  std::string prefix;  // module path, e.g. "encoder.layer1.0.bn1"
  int C;
  long rm_off, rv_off;  // floats into the flat BN-statistics buffer
};

struct Unit {
  std::string conv_name, bn_name;
  int Cout = 0, CinReal = 0, C0 = 0, C1 = 0, KH = 0, KW = 0, stride = 1, pad = 0, up0 = 0;
  int Hv = 0, Wv = 0, Ho = 0, Wo = 0;
  int in0 = -1, in1 = -1, y = -1, a = -1;
  bool bn = true, bias = false, relu = true, apply = true, need_dgrad = true;
  int res_tensor = -1, res_unit = -1;
  long w_off = -1, bias_off = -1, g_off = -1, b_off = -1, rm_off = -1, rv_off = -1;
  size_t wf_off = 0, wd_off = 0;  // packed weights (bytes into workspace)
  int CoutPad = 0, Kpad = 0, CinRows = 0, KpadD = 0, CoutD = 0;
  size_t coef_off = 0;  // bytes: mean[C] invstd[C] scale[C] shift[C] k[3C]
  int segment = 0;      // backward bucket this unit belongs to
  size_t dy_off = 0;    // this unit's own dY buffer (kept until its weight-gradient group has run)
  // decoder conv(cat(upsample2x(in0), in1)) with the up-sampling folded into pre-summed weights (pointwise.hip,
  // pack_up_kernel): per-class forward matrices, the 4x4 stride-2 data gradient w.r.t. in0 (written at in0's own
  // resolution: no full-resolution scratch, no 2x2 sum) and the skip tensor's 3x3 data gradient as its own launch
  bool upfold = false;
  size_t wfc_off = 0, wd4_off = 0, wds_off = 0;
  // train-mode forward as Winograd F(2x2, 3x3) (conv_winograd.hip): transformed filters, statistics rows
  bool wino = false;
  size_t wu_off = 0;
  int wino_rows = 0;
  int C0Rows = 0, C1Rows = 0;
  ConvParams dgrad_lo{};
  ConvParams fwd{}, dgrad{};
  // weight gradient as WG_CLASS + WG_SKIP passes (conv_wgrad.hip): the decoder layers behind an up-sampling whose
  // gradient runs on the tap-parallel kernel -- 4/9 of the MACs on the up-sampled channels
  bool wclass = false;
  WgradLayer wl{};
  size_t wslab_off = 0;  // this unit's own slab region
  int Cin() const { return C0 + C1; }
};

enum BwdKind { BW_HEAD, BW_UNIT, BW_SUM2X2, BW_POOL };
struct BwdOp {
  BwdKind kind;
  int unit = -1;
  // BW_UNIT
  int dA = -1;          // gradient tensor wrt the unit's output activation (grad id)
  // BatchNorm-backward reduce fusion: this op's data gradient is the ONLY writer of the next unit op's dA and
  // that unit recomputes its ReLU mask from y -> its reduction rides in this dgrad's epilogue
  int fuse_for_unit = -1;   // producer side: unit whose (dbeta, dgamma) partial sums this dgrad also emits
  int fused_rows = 0;       // consumer side: > 0 = partial rows already written by the producer (skip the reduce)
  bool mask = true;     // apply the ReLU mask of unit.a
  int dres = -1;        // grad tensor receiving dz (identity / downsample path) or -1
  bool dres_acc = false;
  int dst0 = -1, dst1 = -1;  // dgrad destinations (grad ids); dst0 may be the full-res scratch
  bool acc0 = false, acc1 = false;
  bool dst0_is_full_scratch = false;
  // BW_SUM2X2: src = full-res scratch (C channels at 2H x 2W) -> dst0
  int C = 0, Hl = 0, Wl = 0;
  // BW_POOL: dA (pooled grad) -> dst0 (acc0)
  int segment = 0;
};

// Two networks in one engine (common.h, NetSplit): the caller-owned buffers of the second network as byte offsets from
// the first one's (parameters, running statistics, gradients, NCHW input, NCHW prediction, NCHW output gradient).  The
// workspace of a pair is ONE buffer of UnetEngine::workspace_bytes: two copies of the single-network layout,
// net_ws_stride bytes apart.
struct NetIO {
  long params = 0, bnstats = 0, grads = 0, x = 0, out = 0, dout = 0;
};

class UnetEngine {
 public:
  ~UnetEngine();
  // nets: 1, or 2 = train_deep_fake's denoise-mode pair (model_a / model_b, d3f/train_deep_fake/lit_module.py:142-181) as
  // ONE set of launches; plan_nets (>= nets): the tile / split / patch-kernel choices count the workgroups of plan_nets
  // networks -- a single engine built with plan_nets = 2 runs exactly the kernels of the pair, net by net
  int build(const char* encoder, int in_channels, int classes, int B, int H, int W, int dtype, int nets = 1,
            int plan_nets = 1);

  int pack_weights(const float* params, void* ws, hipStream_t s, const NetIO* io = nullptr) const;
  int forward(const float* params, float* bnstats, const float* x, float* out, void* ws, int training,
              hipStream_t s, const NetIO* io = nullptr) const;
  // join != 0: every gradient of the segments is final on `s` when the call returns.  join == 0 (data-parallel
  // callers): nothing makes `s` wait -- the segments' gradients are final on side_stream() once everything this call
  // enqueued there has run (the side stream also waits for the caller's stream at the end of each segment, so an
  // event recorded on it covers the BatchNorm / bias gradients written on the caller's stream too); backward_join()
  // later makes a stream wait for all of it.
  int backward(const float* params, const float* dout, float* grads, void* ws, int seg_begin,
               int seg_end, hipStream_t s, int join = 1, const NetIO* io = nullptr) const;
  int backward_join(hipStream_t s) const;
  hipStream_t side_stream() const;  // creates the engine's streams on first use; nullptr in D3F_SERIAL_BACKWARD mode
  // eval-mode forward on uint8 BGR frames with the K16 pre/post kernels fused in (in_channels = classes = 3);
  // use_graph: the launch sequence is captured once per set of pointers into a hipGraph and replayed
  int predict_u8(const float* params, float* bnstats, const uint8_t* bgr_in, uint8_t* bgr_out,
                 const float mean[3], const float stdv[3], void* ws, int use_graph, hipStream_t s) const;
  // eval-mode forward (f32 NCHW in / out) replayed from a hipGraph captured once per set of pointers -- the
  // "hipGraph-captured denoise step" of BASELINE.json configs[4]; bit-identical to forward(training = 0)
  int forward_graph(const float* params, float* bnstats, const float* x, float* out, void* ws, hipStream_t s) const;
  // One whole training step of the noisy -> clean objective (d3f/train_denoiser/lit_module.py:107-126 + Lightning's
  // backward / optimizer step) as ONE call: pack -> noise blend -> forward -> (MSE + 1 - SSIM) / 2 -> backward -> Adam.
  // use_graph: the ~330 launches over the engine's streams are captured once per set of pointers and replayed with one
  // hipGraphLaunch on the caller's stream (the small configurations are bound by the host's launch loop, not the GPU).
  struct StepArgs {
    float* params; float* bnstats; float* grads; float* exp_avg; float* exp_avg_sq;
    const float* image; const float* noise; const float* y_uniform;
    float* noisy; float* pred; float* gpred; float* loss_out; float* loss_ws;
    const float* adam_coef;  // device: adam_coefficients()
    float lam, lo, hi;
  };
  int train_step(const StepArgs& a, void* ws, int use_graph, hipStream_t s) const;
  // Optional synchronised BatchNorm statistics across data-parallel ranks (SURVEY.md 8e "optional SyncBN": makes
  // N x bs numerically one process with batch N*bs).  fn sum-all-reduces `count` floats of the workspace in place,
  // ordered on `stream`; it is called once per BatchNorm layer in the forward pass (the per-tile (sum, sum of squares)
  // partials) and once in the backward pass (the (sum dz, sum dz*xhat) partials); the finalize kernels then divide by
  // rows * world.  dgamma / dbeta stay LOCAL sums (the gradient all-reduce adds them up like every other gradient).
  typedef int (*AllReduceFn)(void* ctx, float* data, int64_t count, void* stream);
  bool bn_sync_installed() const { return bn_sync_fn_ != nullptr; }
  void set_bn_sync(AllReduceFn fn, void* ctx, int world) { bn_sync_fn_ = fn; bn_sync_ctx_ = ctx; bn_sync_world_ = world > 0 ? world : 1; }
  int export_tensor(const char* name, const void* ws, float* out_nchw, hipStream_t s) const;
  int export_shape(const char* name, int32_t dims[3]) const;

  std::vector<ParamInfo> params;
  std::vector<BnInfo> bns;
  long param_floats = 0, bnstat_floats = 0;
  size_t workspace_bytes = 0;
  int num_segments = 4;
  long seg_grad_begin[8] = {0}, seg_grad_end[8] = {0};
  int B = 0, H = 0, W = 0, dtype = 0, in_channels = 3, classes = 3;
  int nets = 1, plan_nets = 1;     // B = images PER network
  size_t net_ws_stride = 0;        // bytes between the two networks' workspace copies (0 for a single network)
  int cdtype = 0;  // contraction dtype handed to the conv kernels (dtype = storage dtype; differs for D3F_F32X3)
  double fwd_flops = 0, bwd_flops = 0;  // algorithmic conv FLOPs (2*MAC) per call
  std::vector<Unit> units;

 private:
  int esize() const { return dtype == D3F_F32 ? 4 : 2; }
  int wsize() const { return cdtype == D3F_F32X3 ? 6 : esize(); }  // bytes per packed weight (x3: three bf16 planes)
  int ve() const { return dtype == D3F_F32 ? 4 : 8; }
  int bke() const { return dtype == D3F_F32 ? 32 : 64; }
  const TensorD* find_export(const char* name) const;
  int new_tensor(int H_, int W_, int C_);
  int new_grad(int tensor_id);
  size_t alloc(size_t bytes);
  int add_unit(const std::string& conv_name, const std::string& bn_name, int in0, int in1, int up0,
               int Cout, int k, int stride, int pad, bool bn, bool bias, bool relu, bool apply,
               int segment);
  int plan_unit(Unit& u);

  std::vector<TensorD> tensors;   // activations
  std::vector<TensorD> gtensors;  // activation gradients
  std::vector<int> grad_of;       // tensor id -> grad id or -1
  std::vector<bool> grad_init;    // plan-time: has a writer been emitted yet
  std::vector<BwdOp> bwd_ops;
  std::vector<int> fwd_order_;    // unit ids in execution order, -1 = max-pool
  // backward concurrency: weight gradients run on a side stream next to the data-gradient / BN chain.  Every unit
  // keeps its own dY (585 MB at bs 16, 256x256: nothing next to 288 GB), so the side stream never holds the main
  // chain back.
  mutable hipStream_t side_ = nullptr;
  mutable std::vector<hipEvent_t> ev_dy_;  // one per weight-gradient launch of a backward pass
  mutable hipEvent_t ev_join_ = nullptr;
  mutable hipEvent_t ev_seg_ = nullptr;   // "caller's stream has left the segment" (join == 0 calls)
  mutable bool side_dirty_ = false;       // work was put on the side stream that no stream has joined yet
  // weight packing off the critical path: the layouts of encoder.conv1 / layer1 / layer2 (5 % of the parameters) are
  // packed on the caller's stream, the rest on the side stream while those layers already run; the forward pass
  // waits for it in front of the first later layer
  int ensure_streams() const;
  mutable hipEvent_t ev_pack_in_ = nullptr, ev_pack_done_ = nullptr, ev_pack_mid_ = nullptr;
  mutable bool pack_pending_ = false, pack_mid_pending_ = false;
  // Weight packing in three parts: encoder.conv1 on the caller's stream (the forward needs it at once), layer1-2 and
  // then everything else on the side stream, each behind its own event.
  int first_mid_unit_ = -1;   // first unit (index into `units`) of the second part (encoder.layer1)
  int first_late_unit_ = -1;  // first unit of the third part (encoder.layer3)
  // predict_u8 graph: private capture/launch stream + the pointers and constants the captured graph bakes in
  int forward_body(const float* params, float* bnstats, float* out, char* ws, int training, hipStream_t s,
                   const NetSplit* ns = nullptr) const;
  // what a launch of this engine hands to the kernels: null for a single network
  bool make_split(const NetIO* io, long in_delta, NetSplit* ns) const;
  static void net_conv(ConvParams& p, const NetSplit* ns) {
    if (ns != nullptr) {
      p.nets = ns->nets;
      p.net_ws = p.net_out0 = p.net_scale = ns->ws;
    }
  }
  int predict_u8_launches(const float* params, float* bnstats, const uint8_t* bgr_in, uint8_t* bgr_out,
                          const float mean255[3], const float std255[3], char* ws, hipStream_t s) const;
  // captured graphs: one slot per entry point; a slot is re-captured when the pointers / constants it baked in change
  struct GraphSlot {
    hipGraphExec_t exec = nullptr;
    const void* key[5] = {};
    float cst[6] = {};
  };
  template <typename F>
  int graph_replay(GraphSlot& slot, const void* const key[5], const float cst[6], hipStream_t s, F&& launches) const;
  int wait_for_packed_weights(hipStream_t s) const;
  mutable hipStream_t gstream_ = nullptr;
  mutable GraphSlot g_predict_, g_eval_;
  AllReduceFn bn_sync_fn_ = nullptr;
  void* bn_sync_ctx_ = nullptr;
  int bn_sync_world_ = 1;
  int train_step_launches(const StepArgs& a, void* ws, hipStream_t s) const;
  mutable hipGraphExec_t g_step_ = nullptr;
  mutable StepArgs g_step_key_{};
  mutable const void* g_step_ws_ = nullptr;
  mutable hipEvent_t ev_gin_ = nullptr, ev_gout_ = nullptr;
  size_t head_nchw_off = 0;
  size_t ws_top = 0;
  int t_x = -1, t_pool = -1, head = -1, conv1 = -1;
  size_t pool_idx_off = 0, stats_off = 0, bnpart_off = 0, dz_off = 0, dfull_off = 0,
         bsum_off = 0, splitk_off = 0;
  size_t stats_bytes = 0, bnpart_bytes = 0, dy_bytes = 0, dz_bytes = 0, dfull_bytes = 0, splitk_bytes = 0;
};

int channel_sum_nchw_launch(const float* x, int B, int C, long HW, float* partial, float* out,
                            hipStream_t stream, const NetSplit* ns = nullptr);

}  // namespace d3f
