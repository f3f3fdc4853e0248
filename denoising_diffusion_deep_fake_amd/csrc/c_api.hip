// extern "C" surface of libd3f_hip.so: see include/d3f_hip.h for the contract.
#include "../../include/d3f_hip.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "engine.h"

namespace d3f {
static thread_local char g_err[1024] = "";
int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace d3f

// ---- HIP-event profiling of the contraction kernels ------------------------------------------
namespace d3f {
struct ProfRec {
  hipEvent_t a, b;
  int cls;
  double flops;
};
static std::vector<ProfRec> g_prof;  // pre-created event pairs
static size_t g_prof_used = 0;
static bool g_prof_on = false;
static int g_prof_mask = 7;  // bit per class
bool prof_enabled(int cls) { return g_prof_on && ((g_prof_mask >> cls) & 1) && g_prof_used < g_prof.size(); }
void prof_begin(int cls, double flops, hipStream_t s) {
  ProfRec& r = g_prof[g_prof_used];
  r.cls = cls;
  r.flops = flops;
  (void)hipEventRecord(r.a, s);
}
void prof_end(hipStream_t s) {
  (void)hipEventRecord(g_prof[g_prof_used].b, s);
  ++g_prof_used;
}
}  // namespace d3f

using namespace d3f;

struct d3f_unet {
  UnetEngine e;
};

// storage dtype of a compute dtype (D3F_F32X3 keeps fp32 tensors)
static inline int sdt(int dtype) { return dtype == D3F_F32X3 ? D3F_F32 : dtype; }

static int desc_check(int dtype, const d3f_conv_desc* d) {
  D3F_CHECK(d != nullptr, "conv: null descriptor");
  D3F_CHECK(dtype == D3F_F32 || dtype == D3F_BF16 || dtype == D3F_F32X3, "conv: dtype %d", dtype);
  const int ve = sdt(dtype) == D3F_F32 ? 4 : 8;
  D3F_CHECK(d->B >= 0 && d->H > 0 && d->W > 0 && d->C0 > 0 && d->C1 >= 0 && d->Cout > 0, "conv: extent");
  D3F_CHECK(d->C0 % ve == 0 && d->C1 % ve == 0, "conv: channels must be multiples of %d", ve);
  D3F_CHECK(d->KH == d->KW && d->KH >= 1 && d->KH <= 7, "conv: kernel %dx%d", d->KH, d->KW);
  D3F_CHECK(d->stride == 1 || d->stride == 2, "conv: stride %d", d->stride);
  D3F_CHECK(!d->upsample0 || (d->H % 2 == 0 && d->W % 2 == 0), "conv: up-sampled extent must be even");
  D3F_CHECK(d->CinReal > 0 && d->CinReal <= d->C0 + d->C1, "conv: CinReal");
  return 0;
}

struct Geo {
  int Cin, Ho, Wo, CoutPad, Kpad, CoutD, KpadD, CinRows;
};
static Geo geo(int dtype, const d3f_conv_desc* d) {
  const int ve = sdt(dtype) == D3F_F32 ? 4 : 8, bke = sdt(dtype) == D3F_F32 ? 32 : 64;
  Geo g;
  g.Cin = d->C0 + d->C1;
  g.Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
  g.Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  g.CoutPad = (int)round_up(d->Cout, 16);
  g.Kpad = (int)round_up((long)d->KH * d->KW * g.Cin, bke);
  g.CoutD = (int)round_up(d->Cout, ve);
  g.KpadD = (int)round_up((long)d->KH * d->KW * g.CoutD, bke);
  g.CinRows = (int)round_up(g.Cin, 16);
  return g;
}

static bool desc_upfold(int dtype, const d3f_conv_desc* d) {
  return upfold_applies(sdt(dtype), d->upsample0, d->KH, d->stride, d->pad, d->C0, d->C1);
}

static int fwd_params(int dtype, const d3f_conv_desc* d, ConvParams& p, bool allow_splitk = false) {
  if (int rc = desc_check(dtype, d)) return rc;
  const Geo g = geo(dtype, d);
  std::memset(&p, 0, sizeof(p));
  p.B = d->B; p.Hv = d->H; p.Wv = d->W; p.C0 = d->C0; p.C1 = d->C1; p.cin_real = d->CinReal;
  p.shift0 = d->upsample0 ? 1 : 0;
  p.H0s = d->H >> p.shift0; p.W0s = d->W >> p.shift0;
  p.Ho = g.Ho; p.Wo = g.Wo; p.Cout = d->Cout; p.CoutPad = g.CoutPad; p.Kpad = g.Kpad;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
  p.M = d->B * g.Ho * g.Wo;
  p.mode = CONV_RAW_STATS;
  if (desc_upfold(dtype, d)) {  // up-sampling folded into pre-summed weights: rows = one output-parity class
    p.par = 3;
    p.shift0 = 0;
    p.Ho = p.H0s; p.Wo = p.W0s;
    p.M = d->B * p.H0s * p.W0s;
    p.Kpad = 4 * d->C0 + 9 * d->C1;
  }
  return conv_igemm_plan(p, dtype, allow_splitk);
}

// data gradient of an up-sample-folded layer: (lo) 4x4 stride-2 convolution over dY -> gradient of the LOW-resolution
// source [B][H/2][W/2][C0]; (skip) ordinary 3x3 data gradient -> gradient of the skip tensor [B][H][W][C1]
static int upfold_dgrad_params(int dtype, const d3f_conv_desc* d, ConvParams& lo, ConvParams& sk, bool allow_splitk) {
  const Geo g = geo(dtype, d);
  std::memset(&lo, 0, sizeof(lo));
  std::memset(&sk, 0, sizeof(sk));
  lo.B = d->B; lo.C0 = g.CoutD; lo.C1 = 0;
  lo.Hv = lo.H0s = d->H; lo.Wv = lo.W0s = d->W;
  lo.Ho = d->H / 2; lo.Wo = d->W / 2;
  lo.Cout = d->C0; lo.CoutPad = (int)round_up(d->C0, 16); lo.Kpad = 16 * g.CoutD;
  lo.KH = lo.KW = 4; lo.stride = 2; lo.pad = 1;
  lo.M = d->B * lo.Ho * lo.Wo;
  lo.mode = CONV_DGRAD;
  lo.out_c0 = d->C0;
  if (int rc = conv_igemm_plan(lo, dtype, allow_splitk)) return rc;
  if (d->C1 > 0) {
    sk.B = d->B; sk.C0 = g.CoutD; sk.C1 = 0;
    sk.Hv = sk.H0s = sk.Ho = d->H; sk.Wv = sk.W0s = sk.Wo = d->W;
    sk.Cout = d->C1; sk.CoutPad = (int)round_up(d->C1, 16); sk.Kpad = g.KpadD;
    sk.KH = sk.KW = 3; sk.stride = 1; sk.pad = 1;
    sk.M = d->B * d->H * d->W;
    sk.mode = CONV_DGRAD;
    sk.out_c0 = d->C1;
    if (int rc = conv_igemm_plan(sk, dtype, allow_splitk)) return rc;
  }
  return 0;
}

// want_sum2: the caller takes dx0 of an up-sampled single source at the source's LOW resolution where a launch can sum the
// 2x2 blocks itself (d3f_conv_desc::upsample0 == 2, or the availability query d3f_conv_upsample_summed)
static int dgrad_params(int dtype, const d3f_conv_desc* d, ConvParams& p, bool allow_splitk, bool want_sum2) {
  if (int rc = desc_check(dtype, d)) return rc;
  const Geo g = geo(dtype, d);
  const int s2 = d->stride == 2;
  D3F_CHECK(s2 ? (g.Ho * 2 == d->H && g.Wo * 2 == d->W) : (g.Ho == d->H && g.Wo == d->W),
            "conv_backward_data: needs a 'same' (stride 1) or exactly halving (stride 2) conv");
  std::memset(&p, 0, sizeof(p));
  p.mode = CONV_DGRAD;
  p.out_c0 = d->C1 > 0 ? d->C0 : g.Cin;
  p.Cout = g.Cin; p.CoutPad = g.CinRows; p.Kpad = g.KpadD; p.C0 = g.CoutD; p.C1 = 0; p.B = d->B;
  if (parity_dgrad_applies(dtype, d->stride, d->KH, d->pad, g.CoutD, d->C1)) {
    // stride 2: four plain sub-convolutions over dY, one per output-parity class (conv_igemm.hip)
    p.par = d->KH == 3 ? 1 : 2;
    p.Hv = p.Ho = p.H0s = g.Ho; p.Wv = p.Wo = p.W0s = g.Wo;
    p.KH = p.KW = d->KH == 3 ? 2 : 1; p.stride = 1; p.pad = 0;
    p.M = d->B * g.Ho * g.Wo;
    return conv_igemm_plan(p, dtype, false);
  }
  p.Hv = d->H; p.Wv = d->W;
  p.H0s = g.Ho; p.W0s = g.Wo; p.shift0 = s2; p.zi = s2;
  p.Ho = d->H; p.Wo = d->W;
  p.KH = d->KH; p.KW = d->KW; p.stride = 1; p.pad = d->KH - 1 - d->pad;
  p.M = d->B * d->H * d->W;
  // an up-sampled single source without folded weights: ask for the 2x2-summed gradient at the source's own resolution
  // (kept by the plan only where a patch kernel serves it: d3f_conv_upsample_summed) -- only for a caller that said so
  p.sum2 = (want_sum2 && d->upsample0 && d->C1 == 0) ? 1 : 0;
  return conv_igemm_plan(p, dtype, allow_splitk);
}

extern "C" {

int d3f_version(void) { return 100; }

int d3f_profile_enable(int max_launches) {
  if (max_launches <= 0) {
    g_prof_on = false;
    return 0;
  }
  while ((int)g_prof.size() < max_launches) {
    ProfRec r;
    D3F_HIP(hipEventCreate(&r.a));
    D3F_HIP(hipEventCreate(&r.b));
    r.cls = 0;
    r.flops = 0;
    g_prof.push_back(r);
  }
  g_prof_used = 0;
  g_prof_on = true;
  return 0;
}
int d3f_profile_classes(int mask) {
  g_prof_mask = mask & 7;
  return 0;
}

int d3f_profile_collect(double ms[3], int64_t launches[3], double flops[3]) {
  D3F_CHECK(ms && launches && flops, "profile_collect: null argument");
  for (int k = 0; k < 3; ++k) { ms[k] = 0; launches[k] = 0; flops[k] = 0; }
  for (size_t i = 0; i < g_prof_used; ++i) {
    ProfRec& r = g_prof[i];
    D3F_HIP(hipEventSynchronize(r.b));
    float t = 0.f;
    D3F_HIP(hipEventElapsedTime(&t, r.a, r.b));
    ms[r.cls] += (double)t;
    launches[r.cls] += 1;
    flops[r.cls] += r.flops;
  }
  const int dropped = (g_prof_on && g_prof_used >= g_prof.size()) ? 1 : 0;
  g_prof_used = 0;
  return dropped ? set_error(1, "profile buffer filled up: some launches were not timed") : 0;
}
const char* d3f_last_error(void) { return g_err; }

// ---- whole network ------------------------------------------------------------------------
int d3f_unet_create(const char* encoder_name, int in_channels, int classes, int B, int H, int W,
                    int dtype, d3f_unet_t* out) {
  D3F_CHECK(out != nullptr && encoder_name != nullptr, "unet_create: null argument");
  d3f_unet* h = new (std::nothrow) d3f_unet();
  D3F_CHECK(h != nullptr, "unet_create: out of host memory");
  const int rc = h->e.build(encoder_name, in_channels, classes, B, H, W, dtype);
  if (rc != 0) {
    delete h;
    return rc;
  }
  *out = h;
  return 0;
}
int d3f_unet_create_nets(const char* encoder_name, int in_channels, int classes, int B, int H, int W, int dtype,
                         int nets, int plan_nets, d3f_unet_t* out) {
  D3F_CHECK(out != nullptr && encoder_name != nullptr, "unet_create_nets: null argument");
  d3f_unet* h = new (std::nothrow) d3f_unet();
  D3F_CHECK(h != nullptr, "unet_create_nets: out of host memory");
  const int rc = h->e.build(encoder_name, in_channels, classes, B, H, W, dtype, nets, plan_nets);
  if (rc != 0) {
    delete h;
    return rc;
  }
  *out = h;
  return 0;
}
int d3f_unet_destroy(d3f_unet_t h) {
  delete h;
  return 0;
}
int d3f_unet_nets(d3f_unet_t h) { return h ? h->e.nets : -1; }
size_t d3f_unet_net_workspace_stride(d3f_unet_t h) { return h ? h->e.net_ws_stride : 0; }

// the second network's buffers as byte offsets from the first one's (engine.h, NetIO)
static inline long byte_delta(const void* b, const void* a) {
  return (long)(reinterpret_cast<const char*>(b) - reinterpret_cast<const char*>(a));
}
int d3f_unet_pair_pack_weights(d3f_unet_t h, const float* const params[2], void* workspace, void* stream) {
  D3F_CHECK(h && params && params[0] && params[1] && workspace, "pair_pack_weights: null argument");
  D3F_CHECK(h->e.nets == 2, "pair_pack_weights: the handle is not a pair (d3f_unet_create_nets(..., nets = 2, ...))");
  NetIO io;
  io.params = byte_delta(params[1], params[0]);
  return h->e.pack_weights(params[0], workspace, (hipStream_t)stream, &io);
}
int d3f_unet_pair_forward(d3f_unet_t h, const float* const params[2], float* const bnstats[2], const float* const x[2],
                          float* const out[2], void* workspace, void* stream) {
  D3F_CHECK(h && params && bnstats && x && out && workspace, "pair_forward: null argument");
  D3F_CHECK(h->e.nets == 2, "pair_forward: the handle is not a pair (d3f_unet_create_nets(..., nets = 2, ...))");
  for (int n = 0; n < 2; ++n)
    D3F_CHECK(params[n] && bnstats[n] && x[n] && out[n], "pair_forward: null buffer of network %d", n);
  D3F_CHECK(params[0] != params[1] && bnstats[0] != bnstats[1] && out[0] != out[1],
            "pair_forward: the two networks must own distinct parameters, statistics and outputs");
  NetIO io;
  io.params = byte_delta(params[1], params[0]);
  io.bnstats = byte_delta(bnstats[1], bnstats[0]);
  io.x = byte_delta(x[1], x[0]);
  io.out = byte_delta(out[1], out[0]);
  return h->e.forward(params[0], bnstats[0], x[0], out[0], workspace, 1, (hipStream_t)stream, &io);
}
int d3f_unet_pair_backward(d3f_unet_t h, const float* const params[2], const float* const grad_out[2],
                           float* const grads[2], void* workspace, int seg_begin, int seg_end, int join, void* stream) {
  D3F_CHECK(h && params && grad_out && grads && workspace, "pair_backward: null argument");
  D3F_CHECK(h->e.nets == 2, "pair_backward: the handle is not a pair (d3f_unet_create_nets(..., nets = 2, ...))");
  for (int n = 0; n < 2; ++n)
    D3F_CHECK(params[n] && grad_out[n] && grads[n], "pair_backward: null buffer of network %d", n);
  D3F_CHECK(grads[0] != grads[1], "pair_backward: the two networks must own distinct gradient buffers");
  D3F_CHECK(seg_begin >= 0 && seg_end <= h->e.num_segments && seg_begin <= seg_end,
            "pair_backward: segments [%d,%d)", seg_begin, seg_end);
  NetIO io;
  io.params = byte_delta(params[1], params[0]);
  io.grads = byte_delta(grads[1], grads[0]);
  io.dout = byte_delta(grad_out[1], grad_out[0]);
  return h->e.backward(params[0], grad_out[0], grads[0], workspace, seg_begin, seg_end, (hipStream_t)stream, join ? 1 : 0,
                       &io);
}
int d3f_unet_num_params(d3f_unet_t h) { return h ? (int)h->e.params.size() : -1; }
int d3f_unet_param_info(d3f_unet_t h, int i, char* name, int name_cap, int32_t shape[4], int* ndim,
                        int64_t* offset) {
  D3F_CHECK(h && i >= 0 && i < (int)h->e.params.size(), "param_info: index %d", i);
  const ParamInfo& p = h->e.params[i];
  if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s", p.name.c_str());
  for (int k = 0; k < 4; ++k) shape[k] = p.shape[k];
  *ndim = p.ndim;
  *offset = p.offset;
  return 0;
}
int64_t d3f_unet_param_floats(d3f_unet_t h) { return h ? h->e.param_floats : -1; }
int d3f_unet_num_bn(d3f_unet_t h) { return h ? (int)h->e.bns.size() : -1; }
int d3f_unet_bn_info(d3f_unet_t h, int i, char* prefix, int prefix_cap, int* C, int64_t* rm_offset,
                     int64_t* rv_offset) {
  D3F_CHECK(h && i >= 0 && i < (int)h->e.bns.size(), "bn_info: index %d", i);
  const BnInfo& b = h->e.bns[i];
  if (prefix && prefix_cap > 0) snprintf(prefix, (size_t)prefix_cap, "%s", b.prefix.c_str());
  *C = b.C;
  *rm_offset = b.rm_off;
  *rv_offset = b.rv_off;
  return 0;
}
int64_t d3f_unet_bnstat_floats(d3f_unet_t h) { return h ? h->e.bnstat_floats : -1; }
size_t d3f_unet_workspace_bytes(d3f_unet_t h) { return h ? h->e.workspace_bytes : 0; }
double d3f_unet_forward_flops(d3f_unet_t h) { return h ? h->e.fwd_flops : 0.0; }
double d3f_unet_backward_flops(d3f_unet_t h) { return h ? h->e.bwd_flops : 0.0; }

int d3f_unet_pack_weights(d3f_unet_t h, const float* params, void* workspace, void* stream) {
  D3F_CHECK(h && params && workspace, "pack_weights: null argument");
  D3F_CHECK(h->e.nets == 1, "pack_weights: a pair handle takes d3f_unet_pair_pack_weights");
  return h->e.pack_weights(params, workspace, (hipStream_t)stream);
}
int d3f_unet_forward(d3f_unet_t h, const float* params, float* bnstats, const float* x, float* out,
                     void* workspace, int training, void* stream) {
  D3F_CHECK(h && params && bnstats && x && out && workspace, "unet_forward: null argument");
  D3F_CHECK(h->e.nets == 1, "unet_forward: a pair handle takes d3f_unet_pair_forward");
  return h->e.forward(params, bnstats, x, out, workspace, training, (hipStream_t)stream);
}
int d3f_unet_forward_graph(d3f_unet_t h, const float* params, float* bnstats, const float* x, float* out,
                           void* workspace, void* stream) {
  D3F_CHECK(h && params && bnstats && x && out && workspace, "unet_forward_graph: null argument");
  return h->e.forward_graph(params, bnstats, x, out, workspace, (hipStream_t)stream);
}
int d3f_unet_predict_u8(d3f_unet_t h, const float* params, float* bnstats, const uint8_t* bgr_in, uint8_t* bgr_out,
                        const float mean[3], const float std[3], void* workspace, int use_graph, void* stream) {
  D3F_CHECK(h && params && bnstats && bgr_in && bgr_out && mean && std && workspace, "predict_u8: null argument");
  return h->e.predict_u8(params, bnstats, bgr_in, bgr_out, mean, std, workspace, use_graph, (hipStream_t)stream);
}

int d3f_unet_num_segments(d3f_unet_t h) { return h ? h->e.num_segments : -1; }
int d3f_unet_plan_counts(d3f_unet_t h, int32_t fwd[16], int32_t dgrad[16], int32_t wgrad[16]) {
  D3F_CHECK(h && fwd && dgrad && wgrad, "unet_plan_counts: null argument");
  for (int i = 0; i < 16; ++i) fwd[i] = dgrad[i] = wgrad[i] = 0;
  auto slot = [](int id) { return id < 0 ? 0 : id > 14 ? 14 : id; };
  for (const Unit& u : h->e.units) {
    ++fwd[u.wino ? 15 : slot(u.fwd.patch)];
    if (u.need_dgrad) {
      if (u.upfold) {
        ++dgrad[slot(u.dgrad_lo.patch)];
        if (u.C1 > 0) ++dgrad[slot(u.dgrad.patch)];
      } else {
        ++dgrad[slot(u.dgrad.patch)];
      }
    }
    for (int i = 0; i < u.wl.nparts; ++i) ++wgrad[slot(u.wl.part[i].patch)];
  }
  return 0;
}
int d3f_unet_segment_range(d3f_unet_t h, int segment, int64_t* begin, int64_t* end) {
  D3F_CHECK(h && segment >= 0 && segment < h->e.num_segments, "segment_range: segment %d", segment);
  *begin = h->e.seg_grad_begin[segment];
  *end = h->e.seg_grad_end[segment];
  return 0;
}
int d3f_unet_backward(d3f_unet_t h, const float* params, const float* grad_out, float* grads,
                      void* workspace, int seg_begin, int seg_end, void* stream) {
  D3F_CHECK(h && params && grad_out && grads && workspace, "unet_backward: null argument");
  D3F_CHECK(h->e.nets == 1, "unet_backward: a pair handle takes d3f_unet_pair_backward");
  D3F_CHECK(seg_begin >= 0 && seg_end <= h->e.num_segments && seg_begin <= seg_end,
            "unet_backward: segments [%d,%d)", seg_begin, seg_end);
  return h->e.backward(params, grad_out, grads, workspace, seg_begin, seg_end, (hipStream_t)stream);
}
int d3f_unet_backward_nojoin(d3f_unet_t h, const float* params, const float* grad_out, float* grads,
                             void* workspace, int seg_begin, int seg_end, void* stream) {
  D3F_CHECK(h && params && grad_out && grads && workspace, "unet_backward_nojoin: null argument");
  D3F_CHECK(h->e.nets == 1, "unet_backward_nojoin: a pair handle takes d3f_unet_pair_backward(join = 0)");
  D3F_CHECK(seg_begin >= 0 && seg_end <= h->e.num_segments && seg_begin <= seg_end,
            "unet_backward_nojoin: segments [%d,%d)", seg_begin, seg_end);
  return h->e.backward(params, grad_out, grads, workspace, seg_begin, seg_end, (hipStream_t)stream, 0);
}
int d3f_unet_side_stream(d3f_unet_t h, void** stream_out) {
  D3F_CHECK(h && stream_out, "unet_side_stream: null argument");
  *stream_out = (void*)h->e.side_stream();
  return 0;
}
int d3f_unet_backward_join(d3f_unet_t h, void* stream) {
  D3F_CHECK(h, "unet_backward_join: null handle");
  return h->e.backward_join((hipStream_t)stream);
}
int d3f_adam_coefficients(float lr, float beta1, float beta2, float eps, int step, float grad_scale, float coef[8]) {
  D3F_CHECK(coef && step >= 1, "adam_coefficients: step counts from 1");
  adam_coefficients(lr, beta1, beta2, eps, step, grad_scale, coef);
  return 0;
}
int d3f_unet_train_step(d3f_unet_t h, const d3f_step_buffers* b, float lam, float input_min, float input_max,
                        void* workspace, int use_graph, void* stream) {
  D3F_CHECK(h && b && workspace, "unet_train_step: null argument");
  D3F_CHECK(b->params && b->bnstats && b->grads && b->exp_avg && b->exp_avg_sq && b->image && b->noise && b->y_uniform &&
                b->noisy && b->pred && b->grad_pred && b->loss_out && b->loss_workspace && b->adam_coef,
            "unet_train_step: null buffer");
  D3F_CHECK(!h->e.bn_sync_installed(),
            "unet_train_step: synchronised BatchNorm statistics call back into the host inside the pass; the captured "
            "step is the single-GPU form (use forward / backward)");
  UnetEngine::StepArgs a;
  std::memset(&a, 0, sizeof(a));  // (the struct is compared bytewise as the graph's key)
  a.params = b->params; a.bnstats = b->bnstats; a.grads = b->grads; a.exp_avg = b->exp_avg; a.exp_avg_sq = b->exp_avg_sq;
  a.image = b->image; a.noise = b->noise; a.y_uniform = b->y_uniform;
  a.noisy = b->noisy; a.pred = b->pred; a.gpred = b->grad_pred; a.loss_out = b->loss_out;
  a.loss_ws = reinterpret_cast<float*>(b->loss_workspace);
  a.adam_coef = b->adam_coef;
  a.lam = lam; a.lo = input_min; a.hi = input_max;
  return h->e.train_step(a, workspace, use_graph, (hipStream_t)stream);
}
int d3f_unet_set_bn_sync(d3f_unet_t h, d3f_allreduce_fn fn, void* ctx, int world_size) {
  D3F_CHECK(h && (fn == nullptr || world_size >= 1), "unet_set_bn_sync: arguments");
  h->e.set_bn_sync(fn, ctx, world_size);
  return 0;
}
int d3f_unet_export(d3f_unet_t h, const char* name, const void* workspace, float* out_nchw, void* stream) {
  D3F_CHECK(h && name && workspace && out_nchw, "unet_export: null argument");
  return h->e.export_tensor(name, workspace, out_nchw, (hipStream_t)stream);
}
int d3f_unet_export_shape(d3f_unet_t h, const char* name, int32_t dims[3]) {
  D3F_CHECK(h && name && dims, "unet_export_shape: null argument");
  return h->e.export_shape(name, dims);
}

// ---- single operators -----------------------------------------------------------------------
size_t d3f_conv_packed_bytes(int dtype, const d3f_conv_desc* d, int which) {
  if (desc_check(dtype, d) != 0) return 0;
  const Geo g = geo(dtype, d);
  const size_t es = dtype == D3F_F32X3 ? 6 : (dtype == D3F_F32 ? 4 : 2);  // x3: three bf16 planes
  if (desc_upfold(dtype, d)) {
    // forward: four per-class matrices; data gradient: [wd4 (low-resolution source) | wds (skip tensor)]
    const size_t c0r = (size_t)round_up(d->C0, 16), c1r = (size_t)round_up(d->C1, 16);
    return which == 0 ? (size_t)4 * g.CoutPad * (4 * d->C0 + 9 * d->C1) * es
                      : (c0r * 16 * g.CoutD + c1r * g.KpadD) * es;
  }
  return which == 0 ? (size_t)g.CoutPad * g.Kpad * es : (size_t)g.CinRows * g.KpadD * es;
}
int d3f_conv_pack_weights(int dtype, const d3f_conv_desc* d, const float* w, void* w_fwd, void* w_dgrad,
                          void* stream) {
  if (int rc = desc_check(dtype, d)) return rc;
  const Geo g = geo(dtype, d);
  if (desc_upfold(dtype, d)) {
    const size_t es = dtype == D3F_F32X3 ? 6 : (dtype == D3F_F32 ? 4 : 2);
    const int c0r = (int)round_up(d->C0, 16), c1r = (int)round_up(d->C1, 16);
    D3F_CHECK(w_fwd != nullptr && d->CinReal == d->C0 + d->C1, "conv_pack_weights: up-sample folded layer");
    char* wds = w_dgrad ? reinterpret_cast<char*>(w_dgrad) + (size_t)c0r * 16 * g.CoutD * es : nullptr;
    return pack_up_launch(dtype, w, d->Cout, d->C0, d->C1, w_fwd, g.CoutPad, w_dgrad, c0r,
                          d->C1 > 0 ? wds : nullptr, c1r, (hipStream_t)stream);
  }
  return pack_weights_launch(dtype, w, d->Cout, d->CinReal, g.Cin, d->KH, d->KW, w_fwd, g.CoutPad, g.Kpad,
                             w_dgrad, g.CinRows, g.KpadD,
                             parity_dgrad_applies(dtype, d->stride, d->KH, d->pad, g.CoutD, d->C1) ? 2 : 1,
                             (hipStream_t)stream);
}
int d3f_conv_upsample_folded(int dtype, const d3f_conv_desc* d) {
  return (d != nullptr && desc_check(dtype, d) == 0 && desc_upfold(dtype, d)) ? 1 : 0;
}
int d3f_conv_upsample_summed(int dtype, const d3f_conv_desc* d) {
  ConvParams p;
  if (d == nullptr || !d->upsample0 || desc_check(dtype, d) != 0 || desc_upfold(dtype, d)) return 0;
  if (dgrad_params(dtype, d, p, false, true) != 0) return 0;
  return p.sum2 ? 1 : 0;
}
size_t d3f_conv_workspace_bytes(int dtype, const d3f_conv_desc* d, int which) {
  ConvParams p;
  if (which != 0 && desc_check(dtype, d) == 0 && desc_upfold(dtype, d)) {
    ConvParams lo, sk;
    if (upfold_dgrad_params(dtype, d, lo, sk, true) != 0) return 0;
    return std::max(conv_splitk_floats(lo), d->C1 > 0 ? conv_splitk_floats(sk) : (size_t)0) * sizeof(float);
  }
  const int rc = which == 0 ? fwd_params(dtype, d, p, true) : dgrad_params(dtype, d, p, true, d != nullptr && d->upsample0 == 2);
  return rc != 0 ? 0 : conv_splitk_floats(p) * sizeof(float);
}
size_t d3f_conv_stats_floats(int dtype, const d3f_conv_desc* d, int with_workspace, int* tiles) {
  ConvParams p;
  if (fwd_params(dtype, d, p, with_workspace != 0) != 0) return 0;
  if (tiles) *tiles = p.stat_rows;
  return (size_t)p.stat_rows * p.CoutPad * 2;
}
int d3f_conv_forward(int dtype, const d3f_conv_desc* d, const void* src0, const void* src1,
                     const void* w_fwd, void* y, float* stats, void* workspace, void* stream) {
  ConvParams p;
  if (int rc = fwd_params(dtype, d, p, workspace != nullptr)) return rc;
  if (d && d->B == 0) return 0;  // empty batch: nothing to compute (an empty tensor has a null data pointer)
  D3F_CHECK(src0 && w_fwd && y && (d->C1 == 0 || src1), "conv_forward: null argument");
  p.src0 = src0; p.src1 = src1; p.w = w_fwd; p.out0 = y; p.stats = stats; p.mode = CONV_RAW_STATS;
  p.partial = p.splitk > 1 ? reinterpret_cast<float*>(workspace) : nullptr;
  return conv_igemm_launch(p, dtype, (hipStream_t)stream);
}
// ---- Winograd F(2x2, 3x3) form of a stride-1 3x3 fp32 layer on its own (conv_winograd.hip) ----
static int wino_params(const d3f_conv_desc* d, ConvParams& p) {
  if (int rc = fwd_params(D3F_F32, d, p, false)) return rc;
  D3F_CHECK(conv_winograd_fits(p, D3F_F32),
            "conv_winograd: needs 3x3 / stride 1 / pad 1, one source, H and W multiples of 16, channels a multiple of 16, "
            "filters a multiple of 64 (got %dx%d k%d s%d p%d C0=%d C1=%d up=%d Cout=%d)",
            d->H, d->W, d->KH, d->stride, d->pad, d->C0, d->C1, d->upsample0, d->Cout);
  return 0;
}
int d3f_conv_winograd_applies(int dtype, const d3f_conv_desc* d) {
  ConvParams p;
  if (d == nullptr || desc_check(dtype, d) != 0 || fwd_params(dtype, d, p, false) != 0) return 0;
  return conv_winograd_applies(p, dtype) ? 1 : 0;
}
size_t d3f_conv_winograd_filter_bytes(const d3f_conv_desc* d) {
  ConvParams p;
  return (d == nullptr || desc_check(D3F_F32, d) != 0 || fwd_params(D3F_F32, d, p, false) != 0 || !conv_winograd_fits(p, D3F_F32))
             ? 0 : conv_winograd_filter_floats(p) * sizeof(float);
}
size_t d3f_conv_winograd_stats_floats(const d3f_conv_desc* d, int* tiles) {
  ConvParams p;
  if (d == nullptr || desc_check(D3F_F32, d) != 0 || fwd_params(D3F_F32, d, p, false) != 0 || !conv_winograd_fits(p, D3F_F32))
    return 0;
  const int rows = conv_winograd_stat_rows(p);
  if (tiles) *tiles = rows;
  return (size_t)rows * p.CoutPad * 2;
}
int d3f_conv_winograd_pack(const d3f_conv_desc* d, const float* w, void* u, void* stream) {
  ConvParams p;
  if (int rc = wino_params(d, p)) return rc;
  D3F_CHECK(w && u && d->CinReal == d->C0, "conv_winograd_pack: null argument or padded input channels");
  return conv_winograd_pack_launch(w, reinterpret_cast<float*>(u), d->Cout, d->C0, (hipStream_t)stream);
}
int d3f_conv_winograd_forward(const d3f_conv_desc* d, const void* src0, const void* u, void* y, float* stats,
                              const float* scale, const float* shift, const void* residual, int relu, void* stream) {
  ConvParams p;
  if (int rc = wino_params(d, p)) return rc;
  if (d->B == 0) return 0;
  D3F_CHECK(src0 && u && y && ((scale == nullptr) == (shift == nullptr)), "conv_winograd_forward: null argument");
  D3F_CHECK(scale != nullptr || (residual == nullptr && relu == 0),
            "conv_winograd_forward: residual / ReLU belong to the eval epilogue (scale and shift)");
  p.src0 = src0; p.w = u; p.out0 = y;
  if (scale != nullptr) {
    p.mode = CONV_EVAL_FUSED; p.scale = scale; p.shift = shift; p.res = residual; p.relu = relu ? 1 : 0;
  } else {
    p.mode = CONV_RAW_STATS; p.stats = stats; p.stat_rows = conv_winograd_stat_rows(p);
  }
  return conv_winograd_launch(p, (hipStream_t)stream);
}
int d3f_conv_backward_data(int dtype, const d3f_conv_desc* d, const void* dy, const void* w_dgrad,
                           void* dx0, void* dx1, int acc0, int acc1, void* workspace, void* stream) {
  ConvParams p;
  if (desc_check(dtype, d) == 0 && desc_upfold(dtype, d)) {
    // dx0 = gradient of the LOW-resolution source [B][H/2][W/2][C0], dx1 = gradient of the skip tensor
    ConvParams lo, sk;
    if (int rc = upfold_dgrad_params(dtype, d, lo, sk, workspace != nullptr)) return rc;
    if (d->B == 0) return 0;
    D3F_CHECK(dy && w_dgrad && dx0 && (d->C1 == 0 || dx1), "conv_backward_data: null argument");
    const Geo g = geo(dtype, d);
    const size_t es = dtype == D3F_F32X3 ? 6 : (dtype == D3F_F32 ? 4 : 2);
    lo.src0 = dy; lo.w = w_dgrad; lo.out0 = dx0; lo.acc0 = acc0;
    lo.partial = lo.splitk > 1 ? reinterpret_cast<float*>(workspace) : nullptr;
    if (int rc = conv_igemm_launch(lo, dtype, (hipStream_t)stream)) return rc;
    if (d->C1 > 0) {
      sk.src0 = dy;
      sk.w = reinterpret_cast<const char*>(w_dgrad) + (size_t)round_up(d->C0, 16) * 16 * g.CoutD * es;
      sk.out0 = dx1; sk.acc0 = acc1;
      sk.partial = sk.splitk > 1 ? reinterpret_cast<float*>(workspace) : nullptr;
      if (int rc = conv_igemm_launch(sk, dtype, (hipStream_t)stream)) return rc;
    }
    return 0;
  }
  if (int rc = dgrad_params(dtype, d, p, workspace != nullptr, d != nullptr && d->upsample0 == 2)) return rc;
  if (d && d->B == 0) return 0;
  D3F_CHECK(dy && w_dgrad && dx0 && (d->C1 == 0 || dx1), "conv_backward_data: null argument");
  p.src0 = dy; p.w = w_dgrad; p.out0 = dx0; p.out1 = dx1; p.acc0 = acc0; p.acc1 = acc1;
  p.partial = p.splitk > 1 ? reinterpret_cast<float*>(workspace) : nullptr;
  if (p.par == 2 && !acc0) {  // 1x1 stride 2: only even pixels receive a gradient; the others are zero
    const size_t es = sdt(dtype) == D3F_F32 ? 4 : 2;
    D3F_HIP(hipMemsetAsync(dx0, 0, (size_t)4 * p.M * p.Cout * es, (hipStream_t)stream));
  }
  return conv_igemm_launch(p, dtype, (hipStream_t)stream);
}
static int wgrad_params(int dtype, const d3f_conv_desc* d, WgradParams& w) {
  if (int rc = desc_check(dtype, d)) return rc;
  const Geo g = geo(dtype, d);
  std::memset(&w, 0, sizeof(w));
  w.B = d->B; w.Hv = d->H; w.Wv = d->W; w.C0 = d->C0; w.C1 = d->C1;
  w.shift0 = d->upsample0 ? 1 : 0;
  w.H0s = d->H >> w.shift0; w.W0s = d->W >> w.shift0;
  w.Ho = g.Ho; w.Wo = g.Wo; w.Cout = g.CoutD;
  w.KH = d->KH; w.KW = d->KW; w.stride = d->stride; w.pad = d->pad;
  w.M = d->B * g.Ho * g.Wo;
  w.cin_real = d->CinReal;
  return wgrad_plan(w, sdt(dtype));
}
size_t d3f_conv_backward_weight_workspace_bytes(int dtype, const d3f_conv_desc* d) {
  WgradParams w;
  if (wgrad_params(dtype, d, w) != 0) return 0;
  WgradLayer L;
  if (wgrad_layer_plan(L, w, sdt(dtype)) != 0) return 0;
  return wgrad_layer_partial_floats(L) * sizeof(float);
}
int d3f_conv_backward_weight(int dtype, const d3f_conv_desc* d, const void* dy, const void* src0,
                             const void* src1, void* workspace, float* dw, void* stream) {
  WgradParams w;
  if (int rc = wgrad_params(dtype, d, w)) return rc;
  if (d && d->B == 0) {  // empty batch: the gradient is zero
    D3F_CHECK(dw, "conv_backward_weight: null argument");
    D3F_HIP(hipMemsetAsync(dw, 0, (size_t)d->Cout * d->CinReal * d->KH * d->KW * sizeof(float), (hipStream_t)stream));
    return 0;
  }
  D3F_CHECK(dy && src0 && workspace && dw && (d->C1 == 0 || src1), "conv_backward_weight: null argument");
  // the passes the engine runs for this layer (class form behind an up-sampling where it applies), then its slab reduce
  WgradLayer L;
  if (int rc = wgrad_layer_plan(L, w, sdt(dtype))) return rc;
  return wgrad_layer_launch(L, dy, src0, src1, reinterpret_cast<float*>(workspace), dw, d->Cout, d->CinReal, dtype,
                            (hipStream_t)stream);
}

int d3f_bn_finalize(const float* stats, int tiles, int C, int64_t count, const float* gamma,
                    const float* beta, float* running_mean, float* running_var, float* coef, void* stream) {
  D3F_CHECK(stats && gamma && beta && coef && C > 0 && tiles > 0 && count > 0, "bn_finalize: argument");
  return bn_finalize_launch(stats, tiles, C, (int)round_up(C, 16), (long)count, gamma, beta, 1e-5f, 0.1f,
                            running_mean, running_var, coef, coef + C, coef + 2 * C, coef + 3 * C,
                            (hipStream_t)stream);
}
int d3f_bn_apply(int dtype, const void* y, const float* coef, int C, int64_t rows, const void* residual,
                 int relu, void* out, void* stream) {
  D3F_CHECK(y && coef && out, "bn_apply: null argument");
  return bn_apply_launch(sdt(dtype), y, coef + 2 * C, coef + 3 * C, residual, nullptr, nullptr, nullptr, relu,
                         out, (long)rows, C, (hipStream_t)stream);
}
size_t d3f_bn_backward_workspace_bytes(int dtype, int C, int64_t rows) {
  return ((size_t)bn_bwd_reduce_blocks((long)rows, C, sdt(dtype)) * C * 2 + 3 * (size_t)C) * sizeof(float) + 256;
}
int d3f_bn_backward(int dtype, const void* dA, const void* a_or_null, const void* y, const float* coef,
                    const float* gamma, int C, int64_t rows, void* dy, void* dres, float* dgamma,
                    float* dbeta, void* workspace, void* stream) {
  D3F_CHECK(dA && y && coef && gamma && dy && dgamma && dbeta && workspace, "bn_backward: null argument");
  hipStream_t s = (hipStream_t)stream;
  float* part = reinterpret_cast<float*>(workspace);
  const int blocks = bn_bwd_reduce_blocks((long)rows, C, sdt(dtype));
  float* k = part + (size_t)round_up((long)blocks * C * 2, 4);
  int nb = 0;
  if (int rc = bn_bwd_reduce_launch(sdt(dtype), dA, a_or_null, y, coef, coef + C, part, &nb, (long)rows, C, s))
    return rc;
  if (int rc = bn_bwd_finalize_launch(part, nb, C, (long)rows, gamma, coef + C, dgamma, dbeta, 0, k, s))
    return rc;
  return bn_bwd_apply_launch(sdt(dtype), dA, a_or_null, y, coef, coef + C, k, dy, dres, 0, (long)rows, C, s);
}

int d3f_maxpool3x3s2_forward(int dtype, const void* in, void* out, uint8_t* idx, int B, int H, int W,
                             int C, void* stream) {
  D3F_CHECK(in && out && idx, "maxpool: null argument");
  return maxpool3x3s2_fwd_launch(sdt(dtype), in, out, idx, B, H, W, C, (hipStream_t)stream);
}
int d3f_maxpool3x3s2_backward(int dtype, const void* dout, const uint8_t* idx, void* din, int accumulate,
                              int B, int H, int W, int C, void* stream) {
  D3F_CHECK(dout && idx && din, "maxpool: null argument");
  return maxpool3x3s2_bwd_launch(sdt(dtype), dout, idx, din, accumulate, B, H, W, C, (hipStream_t)stream);
}
int d3f_upsample2x_backward(int dtype, const void* dfull, void* dlow, int B, int Hlow, int Wlow, int C,
                            void* stream) {
  D3F_CHECK(dfull && dlow, "upsample2x_backward: null argument");
  return sum2x2_launch(sdt(dtype), dfull, dlow, B, Hlow, Wlow, C, (hipStream_t)stream);
}
int d3f_u8rgb_normalise(const uint8_t* in_hwc, float* out_nchw, int B, int H, int W, const float mean[3],
                        const float std[3], void* stream) {
  if (B == 0) return 0;
  D3F_CHECK(in_hwc && out_nchw && mean && std, "u8rgb_normalise: null argument");
  D3F_CHECK(B > 0 && H > 0 && W > 0, "u8rgb_normalise: bad shape");
  D3F_CHECK(std[0] != 0.f && std[1] != 0.f && std[2] != 0.f, "u8rgb_normalise: zero std");
  return u8rgb_to_nchw_launch(in_hwc, out_nchw, B, (long)H * W, mean, std, (hipStream_t)stream);
}

int d3f_affine_warp(const float* in, const float* theta, float* out, int B, int C, int H, int W, void* stream) {
  if (B == 0) return 0;
  D3F_CHECK(in && theta && out && in != out, "affine_warp: null or aliased argument");
  D3F_CHECK(B >= 0 && C > 0 && H > 0 && W > 0, "affine_warp: bad shape");
  return affine_warp_launch(in, theta, out, B, C, H, W, (hipStream_t)stream);
}

int d3f_nchw_to_nhwc(int dtype, const float* in, void* out, int B, int C, int H, int W, int Cpad, void* stream) {
  D3F_CHECK(in && out && Cpad >= C, "nchw_to_nhwc: argument");
  return nchw_to_nhwc_launch(sdt(dtype), in, out, B, C, H, W, Cpad, (hipStream_t)stream);
}
int d3f_nhwc_to_nchw(int dtype, const void* in, float* out, int B, int C, int H, int W, int Cpad, void* stream) {
  D3F_CHECK(in && out && Cpad >= C, "nhwc_to_nchw: argument");
  return nhwc_to_nchw_launch(sdt(dtype), in, out, B, C, H, W, Cpad, (hipStream_t)stream);
}

// ---- training-step arithmetic ---------------------------------------------------------------
int d3f_noise_blend(const float* x, const float* noise, const float* y_uniform, float lam, float* out,
                    float* r_out_or_null, int B, int64_t per_image, void* stream) {
  if (B == 0 || per_image == 0) return 0;
  D3F_CHECK(x && noise && y_uniform && out, "noise_blend: null argument");
  return noise_blend_launch(x, noise, y_uniform, lam, out, r_out_or_null, B, (long)per_image,
                            (hipStream_t)stream);
}
int d3f_noise_blend_fixed(const float* x, const float* noise, const float* r, float* out, int B, int64_t per_image,
                          void* stream) {
  if (B == 0 || per_image == 0) return 0;
  D3F_CHECK(x && noise && r && out, "noise_blend_fixed: null argument");
  return noise_blend_fixed_launch(x, noise, r, out, B, (long)per_image, (hipStream_t)stream);
}
size_t d3f_l1_per_image_workspace_bytes(int B) { return l1_per_image_workspace_bytes(B); }
int d3f_l1_per_image(const float* prediction, const float* target, float* out, void* workspace, int B,
                     int64_t per_image, void* stream) {
  if (B == 0) return 0;
  D3F_CHECK(prediction && target && out && workspace, "l1_per_image: null argument");
  return l1_per_image_launch(prediction, target, out, workspace, B, (long)per_image, (hipStream_t)stream);
}
size_t d3f_mse_ssim_loss_workspace_bytes(int B, int H, int W) {
  return loss_workspace_floats(B, H, W) * sizeof(float);
}
int d3f_mse_ssim_loss(const float* pred, const float* target, float input_min, float input_max,
                      float* loss_out, float* grad_pred, void* workspace, int B, int H, int W, void* stream) {
  D3F_CHECK(pred && target && loss_out && grad_pred && workspace, "mse_ssim_loss: null argument");
  return mse_ssim_loss_launch(pred, target, input_min, input_max, loss_out, grad_pred,
                              reinterpret_cast<float*>(workspace), B, H, W, (hipStream_t)stream);
}
int d3f_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, int step, float grad_scale, void* stream) {
  D3F_CHECK(params && grads && exp_avg && exp_avg_sq, "adam_step: null argument");
  return adam_step_launch(params, grads, exp_avg, exp_avg_sq, (long)n, lr, beta1, beta2, eps, step,
                          grad_scale, (hipStream_t)stream);
}
int d3f_ema_lerp(float* ema, const float* online, int64_t n, float weight, void* stream) {
  D3F_CHECK(ema && online, "ema_lerp: null argument");
  return ema_lerp_launch(ema, online, (long)n, weight, (hipStream_t)stream);
}

}  // extern "C"
