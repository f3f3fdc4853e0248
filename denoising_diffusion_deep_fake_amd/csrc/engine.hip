// Whole-network plan and launch sequences for Unet(resnet18|resnet34) -- the network the
// reference instantiates at d3f/train_denoiser/lit_module.py:46-52 and
// d3f/train_deep_fake/lit_module.py:53-59 (structure: SURVEY.md Appendix A.1).
//
// Parameter order (flat buffer) = torch named_parameters() order of that module tree, so the
// Python side can expose ordinary nn.Parameters that are views into one flat buffer and keep
// smp-compatible state_dict keys.
#include "engine.h"
#include <vector>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace d3f {

// ------------------------------------------------------------------------------------------
// tiny reduction used for the head's bias gradient: out[c] = sum_{b,hw} x[b][c][hw]
// ------------------------------------------------------------------------------------------
// grid = (CS_PP segments of a plane, C, B): a block sums one contiguous stretch of one (image, channel) plane with
// 16-byte loads; partial[c][b * CS_PP + segment], summed in that order by the final kernel (fixed order: reproducible)
constexpr int CS_PP = 16;
__global__ __launch_bounds__(256) void channel_sum_partial_kernel(const float* __restrict__ x, int C, long HW,
                                                                  float* __restrict__ partial, int nb, long net_in,
                                                                  long net_ws) {
  __shared__ float red[256];
  // two networks in one launch (common.h, NetSplit): blockIdx.z = net * nb + image
  const int seg = blockIdx.x, c = blockIdx.y, b = (int)blockIdx.z % nb;
  if ((int)blockIdx.z >= nb) {
    net_shift(x, net_in); net_shift(partial, net_ws);
  }
  const float* __restrict__ plane = x + ((long)b * C + c) * HW;
  const long per = (HW + CS_PP - 1) / CS_PP, lo = (long)seg * per, hi = lo + per < HW ? lo + per : HW;
  float s = 0.f;
  if ((HW & 3) == 0 && (per & 3) == 0) {
    for (long i = lo + 4 * threadIdx.x; i < hi; i += 4 * 256) {
      const float4 v = *reinterpret_cast<const float4*>(plane + i);
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (long i = lo + threadIdx.x; i < hi; i += 256) s += plane[i];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[((long)c * nb + b) * CS_PP + seg] = red[0];
}
// one workgroup per channel: thread t sums partial[c][t], [t + 256], ... in f64, the 256 sums are added pairwise in a fixed
// tree (reproducible).  (One THREAD per channel walked its 256 partials as dependent loads: 54 us for three channels.)
__global__ __launch_bounds__(256) void channel_sum_final_kernel(const float* __restrict__ partial, int C, int n,
                                                                float* __restrict__ out, long net_ws, long net_grad) {
  if (blockIdx.z != 0) {  // two networks in one launch: the head's bias gradient lands in net 1's flat gradient
    net_shift(partial, net_ws); net_shift(out, net_grad);
  }
  __shared__ double red[256];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[(long)c * n + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = (float)red[0];
}
size_t channel_sum_partial_floats(int B, int C) { return (size_t)B * C * CS_PP; }
int channel_sum_nchw_launch(const float* x, int B, int C, long HW, float* partial, float* out,
                            hipStream_t stream, const NetSplit* ns) {
  const NetSplit nv = net_split_or_single(ns);
  D3F_CHECK(B * nv.nets <= 65535 && C <= 65535, "channel sum: grid (%d, %d)", C, B);
  hipLaunchKernelGGL(channel_sum_partial_kernel, dim3(CS_PP, C, B * nv.nets), dim3(256), 0, stream, x, C, HW, partial, B,
                     nv.in, nv.ws);
  D3F_HIP(hipGetLastError());
  hipLaunchKernelGGL(channel_sum_final_kernel, dim3(C, 1, nv.nets), dim3(256), 0, stream, partial, C, B * CS_PP, out,
                     nv.ws, nv.grad);
  D3F_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
size_t UnetEngine::alloc(size_t bytes) {
  const size_t off = ws_top;
  ws_top += (size_t)round_up((long)bytes, 256);
  return off;
}

int UnetEngine::new_tensor(int H_, int W_, int C_) {
  TensorD t;
  t.H = H_; t.W = W_; t.C = C_;
  t.off = alloc((size_t)t.elems(B) * esize());
  tensors.push_back(t);
  grad_of.push_back(-1);
  return (int)tensors.size() - 1;
}

int UnetEngine::new_grad(int tid) {
  if (grad_of[tid] >= 0) return grad_of[tid];
  TensorD g = tensors[tid];
  g.off = alloc((size_t)g.elems(B) * esize());
  gtensors.push_back(g);
  grad_init.push_back(false);
  grad_of[tid] = (int)gtensors.size() - 1;
  return grad_of[tid];
}

static void add_param(std::vector<ParamInfo>& ps, long& top, const std::string& name, int ndim,
                      int s0, int s1, int s2, int s3, long* off_out) {
  ParamInfo p;
  p.name = name;
  p.ndim = ndim;
  p.shape[0] = s0; p.shape[1] = s1; p.shape[2] = s2; p.shape[3] = s3;
  p.numel = (long)s0 * (ndim > 1 ? s1 : 1) * (ndim > 2 ? s2 : 1) * (ndim > 3 ? s3 : 1);
  p.offset = top;
  top += p.numel;
  *off_out = p.offset;
  ps.push_back(p);
}

int UnetEngine::add_unit(const std::string& conv_name, const std::string& bn_name, int in0, int in1,
                         int up0, int Cout, int k, int stride, int pad, bool bn, bool bias, bool relu,
                         bool apply, int segment) {
  Unit u;
  u.conv_name = conv_name;
  u.bn_name = bn_name;
  u.in0 = in0; u.in1 = in1; u.up0 = up0;
  u.Cout = Cout; u.KH = u.KW = k; u.stride = stride; u.pad = pad;
  u.bn = bn; u.bias = bias; u.relu = relu; u.apply = apply; u.segment = segment;
  const TensorD& t0 = tensors[in0];
  u.C0 = t0.C;
  u.Hv = t0.H << up0;
  u.Wv = t0.W << up0;
  u.C1 = in1 >= 0 ? tensors[in1].C : 0;
  D3F_CHECK(in1 < 0 || (tensors[in1].H == u.Hv && tensors[in1].W == u.Wv),
            "unit %s: skip extent mismatch", conv_name.c_str());
  u.CinReal = (in0 == t_x) ? in_channels : u.Cin();
  u.Ho = (u.Hv + 2 * pad - k) / stride + 1;
  u.Wo = (u.Wv + 2 * pad - k) / stride + 1;
  u.need_dgrad = (in0 != t_x);
  D3F_CHECK(stride == 1 || (u.Hv % 2 == 0 && u.Wv % 2 == 0 && u.Ho * 2 == u.Hv && u.Wo * 2 == u.Wv),
            "unit %s: stride-2 conv needs an even extent", conv_name.c_str());

  add_param(params, param_floats, conv_name + ".weight", 4, Cout, u.CinReal, k, k, &u.w_off);
  if (bias) add_param(params, param_floats, conv_name + ".bias", 1, Cout, 1, 1, 1, &u.bias_off);
  if (bn) {
    add_param(params, param_floats, bn_name + ".weight", 1, Cout, 1, 1, 1, &u.g_off);
    add_param(params, param_floats, bn_name + ".bias", 1, Cout, 1, 1, 1, &u.b_off);
    BnInfo bi;
    bi.prefix = bn_name;
    bi.C = Cout;
    bi.rm_off = bnstat_floats;
    bi.rv_off = bnstat_floats + Cout;
    bnstat_floats += 2L * Cout;
    u.rm_off = bi.rm_off;
    u.rv_off = bi.rv_off;
    bns.push_back(bi);
    u.y = new_tensor(u.Ho, u.Wo, Cout);
    u.a = apply ? new_tensor(u.Ho, u.Wo, Cout) : -1;
    u.coef_off = alloc((size_t)7 * Cout * sizeof(float));
  }
  u.CoutPad = (int)round_up(Cout, 16);
  u.Kpad = (int)round_up((long)k * k * u.Cin(), bke());
  u.wf_off = alloc((size_t)u.CoutPad * u.Kpad * wsize());
  u.CoutD = (int)round_up(Cout, ve());
  if (u.need_dgrad) {
    u.KpadD = (int)round_up((long)k * k * u.CoutD, bke());
    u.CinRows = (int)round_up(u.Cin(), 16);
    u.wd_off = alloc((size_t)u.CinRows * u.KpadD * wsize());
  }
  u.upfold = upfold_applies(dtype, up0, k, stride, pad, u.C0, u.C1);
  if (u.upfold) {
    u.C0Rows = (int)round_up(u.C0, 16);
    u.C1Rows = (int)round_up(u.C1, 16);
    u.wfc_off = alloc((size_t)4 * u.CoutPad * (4 * u.C0 + 9 * u.C1) * wsize());
    u.wd4_off = alloc((size_t)u.C0Rows * 16 * u.CoutD * wsize());
    u.wds_off = alloc((size_t)std::max(1, u.C1Rows) * round_up(9L * u.CoutD, bke()) * wsize());
  }
  units.push_back(u);
  return (int)units.size() - 1;
}

int UnetEngine::plan_unit(Unit& u) {
  ConvParams& f = u.fwd;
  std::memset(&f, 0, sizeof(f));
  f.B = B; f.Hv = u.Hv; f.Wv = u.Wv; f.C0 = u.C0; f.C1 = u.C1; f.cin_real = u.CinReal;
  f.H0s = u.Hv >> u.up0; f.W0s = u.Wv >> u.up0; f.shift0 = u.up0; f.zi = 0;
  f.Ho = u.Ho; f.Wo = u.Wo; f.Cout = u.Cout; f.CoutPad = u.CoutPad; f.Kpad = u.Kpad;
  f.KH = u.KH; f.KW = u.KW; f.stride = u.stride; f.pad = u.pad;
  f.M = B * u.Ho * u.Wo;
  f.mode = u.bn ? CONV_RAW_STATS : CONV_HEAD_NCHW;
  f.plan_nets = plan_nets;
  const long rows_full = (long)B * u.Ho * u.Wo;
  if (u.upfold) {  // rows = one output-parity class; src0 described at its own (low) resolution
    f.par = 3;
    f.shift0 = 0;
    f.Ho = f.H0s; f.Wo = f.W0s;
    f.M = B * f.H0s * f.W0s;
    f.Kpad = 4 * u.C0 + 9 * u.C1;
  }
  if (int rc = conv_igemm_plan(f, cdtype, true)) return rc;
  if (conv_splitk_floats(f) * sizeof(float) > splitk_bytes) splitk_bytes = conv_splitk_floats(f) * sizeof(float);
  const double macs = (double)rows_full * u.Cout * u.KH * u.KW * u.CinReal;  // algorithmic
  f.flops = 2.0 * macs;
  fwd_flops += 2.0 * macs;
  bwd_flops += 2.0 * macs;  // weight gradient
  u.wino = u.bn && !u.upfold && cdtype == D3F_F32 && conv_winograd_applies(f, cdtype);
  if (u.wino) {
    u.wino_rows = conv_winograd_stat_rows(f);
    u.wu_off = alloc(conv_winograd_filter_floats(f) * sizeof(float));
  }
  if (u.bn) {
    const size_t sb = (size_t)std::max(f.stat_rows, u.wino_rows) * u.CoutPad * 2 * sizeof(float);
    if (sb > stats_bytes) stats_bytes = sb;
    const size_t pb = (size_t)bn_bwd_reduce_blocks(rows_full, u.Cout, dtype) * u.Cout * 2 * sizeof(float);
    if (pb > bnpart_bytes) bnpart_bytes = pb;
  }
  const size_t dyb = (size_t)rows_full * u.CoutD * esize();
  if (dyb > dy_bytes) dy_bytes = dyb;
  if (!u.apply && u.bn && dyb > dz_bytes) dz_bytes = dyb;

  WgradParams g;
  std::memset(&g, 0, sizeof(g));
  g.B = B; g.Hv = u.Hv; g.Wv = u.Wv; g.C0 = u.C0; g.C1 = u.C1;
  g.H0s = u.Hv >> u.up0; g.W0s = u.Wv >> u.up0; g.shift0 = u.up0;
  g.Ho = u.Ho; g.Wo = u.Wo; g.Cout = u.CoutD;
  g.KH = u.KH; g.KW = u.KW; g.stride = u.stride; g.pad = u.pad; g.M = (int)rows_full;
  g.cin_real = u.CinReal;
  g.flops = 2.0 * macs;
  g.plan_nets = plan_nets;
  if (int rc = wgrad_layer_plan(u.wl, g, dtype)) return rc;
  u.wclass = u.wl.part[0].part != WG_WHOLE;
  u.wslab_off = alloc(wgrad_layer_partial_floats(u.wl) * sizeof(float));
  u.dy_off = alloc(dyb);

  if (u.need_dgrad && u.upfold) {
    // (1) gradient w.r.t. the low-resolution source: 4x4 stride-2 pad-1 convolution over dY with pre-summed weights
    ConvParams& l = u.dgrad_lo;
    std::memset(&l, 0, sizeof(l));
    l.B = B; l.C0 = u.CoutD; l.C1 = 0;
    l.Hv = l.H0s = u.Hv; l.Wv = l.W0s = u.Wv;
    l.Ho = u.Hv / 2; l.Wo = u.Wv / 2;
    l.Cout = u.C0; l.CoutPad = u.C0Rows; l.Kpad = 16 * u.CoutD;
    l.KH = l.KW = 4; l.stride = 2; l.pad = 1;
    l.M = B * l.Ho * l.Wo;
    l.mode = CONV_DGRAD;
    l.out_c0 = u.C0;
    l.plan_nets = plan_nets;
    if (int rc = conv_igemm_plan(l, cdtype, true)) return rc;
    if (conv_splitk_floats(l) * sizeof(float) > splitk_bytes) splitk_bytes = conv_splitk_floats(l) * sizeof(float);
    l.flops = 2.0 * macs * u.C0 / u.Cin();
    // (2) gradient w.r.t. the skip tensor: an ordinary 3x3 data gradient with C1 outputs
    ConvParams& d = u.dgrad;
    std::memset(&d, 0, sizeof(d));
    if (u.C1 > 0) {
      d.B = B; d.C0 = u.CoutD; d.C1 = 0;
      d.Hv = d.H0s = d.Ho = u.Hv; d.Wv = d.W0s = d.Wo = u.Wv;
      d.Cout = u.C1; d.CoutPad = u.C1Rows; d.Kpad = u.KpadD;
      d.KH = d.KW = 3; d.stride = 1; d.pad = 1;
      d.M = (int)rows_full;
      d.mode = CONV_DGRAD;
      d.out_c0 = u.C1;
      d.plan_nets = plan_nets;
      if (int rc = conv_igemm_plan(d, cdtype, true)) return rc;
      if (conv_splitk_floats(d) * sizeof(float) > splitk_bytes) splitk_bytes = conv_splitk_floats(d) * sizeof(float);
      d.flops = 2.0 * macs * u.C1 / u.Cin();
    }
    bwd_flops += 2.0 * macs;
  } else if (u.need_dgrad) {
    ConvParams& d = u.dgrad;
    std::memset(&d, 0, sizeof(d));
    const int s2 = u.stride == 2 ? 1 : 0;
    D3F_CHECK(s2 || (u.Ho == u.Hv && u.Wo == u.Wv), "unit %s: dgrad expects a 'same' conv",
              u.conv_name.c_str());
    d.B = B; d.C0 = u.CoutD; d.C1 = 0;
    d.Cout = u.Cin(); d.CoutPad = u.CinRows; d.Kpad = u.KpadD;
    d.mode = CONV_DGRAD;
    d.plan_nets = plan_nets;
    d.out_c0 = u.C1 > 0 ? u.C0 : u.Cin();
    if (parity_dgrad_applies(dtype, u.stride, u.KH, u.pad, u.CoutD, u.C1)) {
      // stride 2: four plain sub-convolutions over dY, one per output-parity class (conv_igemm.hip)
      d.par = u.KH == 3 ? 1 : 2;
      d.Hv = d.Ho = d.H0s = u.Ho; d.Wv = d.Wo = d.W0s = u.Wo;
      d.KH = d.KW = u.KH == 3 ? 2 : 1; d.stride = 1; d.pad = 0;
      d.M = B * u.Ho * u.Wo;
    } else {
      d.Hv = u.Hv; d.Wv = u.Wv;  // extent of the (zero-inserted) dY == extent of dX
      d.H0s = u.Ho; d.W0s = u.Wo; d.shift0 = s2; d.zi = s2;
      d.Ho = u.Hv; d.Wo = u.Wv;
      d.KH = u.KH; d.KW = u.KW; d.stride = 1; d.pad = u.KH - 1 - u.pad;
      d.M = B * u.Hv * u.Wv;
      // a source read through the up-sampling without folded weights (bf16: decoder block 4 conv1): ask for the gradient
      // at the source's own resolution, 2x2 blocks summed in the epilogue; the plan keeps the request only when a patch
      // kernel takes the launch (conv_patch.hip), otherwise full-resolution scratch + sum2x2 as before
      d.sum2 = (u.up0 && u.C1 == 0) ? 1 : 0;
    }
    if (int rc = conv_igemm_plan(d, cdtype, true)) return rc;
    if (conv_splitk_floats(d) * sizeof(float) > splitk_bytes) splitk_bytes = conv_splitk_floats(d) * sizeof(float);
    d.flops = 2.0 * macs;
    bwd_flops += 2.0 * macs;
    if (u.up0 && !d.sum2) {
      const size_t fb = (size_t)d.M * u.C0 * esize();
      if (fb > dfull_bytes) dfull_bytes = fb;
    }
  }
  return 0;
}

int UnetEngine::build(const char* encoder, int in_channels_, int classes_, int B_, int H_, int W_,
                      int dtype_, int nets_, int plan_nets_) {
  D3F_CHECK(dtype_ == D3F_F32 || dtype_ == D3F_BF16 || dtype_ == D3F_F32X3, "unet: dtype %d", dtype_);
  D3F_CHECK((nets_ == 1 || nets_ == 2) && (plan_nets_ == 1 || plan_nets_ == 2) && plan_nets_ >= nets_,
            "unet: nets=%d, plan_nets=%d (one network, or the pair of train_deep_fake's denoise mode)", nets_, plan_nets_);
  nets = nets_;
  plan_nets = plan_nets_;
  D3F_CHECK(B_ >= 1 && H_ >= 32 && W_ >= 32, "unet: bad shape B=%d H=%d W=%d", B_, H_, W_);
  D3F_CHECK(H_ % 32 == 0 && W_ % 32 == 0,
            "Wrong input shape height=%d, width=%d. Expected image height and width divisible by 32.",
            H_, W_);
  D3F_CHECK(in_channels_ >= 1 && in_channels_ <= 8 && classes_ >= 1 && classes_ <= 16,
            "unet: in_channels=%d classes=%d unsupported", in_channels_, classes_);
  int nblocks[4];
  if (std::strcmp(encoder, "resnet34") == 0) { nblocks[0] = 3; nblocks[1] = 4; nblocks[2] = 6; nblocks[3] = 3; }
  else if (std::strcmp(encoder, "resnet18") == 0) { nblocks[0] = nblocks[1] = nblocks[2] = nblocks[3] = 2; }
  else return set_error(-1, "Wrong encoder name `%s`, supported encoders: ['resnet18', 'resnet34']", encoder);
  B = B_; H = H_; W = W_; cdtype = dtype_; dtype = dtype_ == D3F_F32X3 ? D3F_F32 : dtype_; in_channels = in_channels_; classes = classes_;
  D3F_CHECK((long)B * H * W * 64 < (1L << 31), "unet: activation too large for 32-bit pixel indices");

  // ---- graph --------------------------------------------------------------------------
  t_x = new_tensor(H, W, (int)round_up(in_channels, ve()));
  conv1 = add_unit("encoder.conv1", "encoder.bn1", t_x, -1, 0, 64, 7, 2, 3, true, false, true, true, 3);
  if (conv1 < 0) return conv1;
  const int f1 = units[conv1].a;
  t_pool = new_tensor(H / 4, W / 4, 64);
  pool_idx_off = alloc((size_t)tensors[t_pool].elems(B));
  int cur = t_pool, inpl = 64;
  int feat[6] = {-1, f1, -1, -1, -1, -1};
  struct Blk { int u1, u2, ud; };
  std::vector<Blk> enc_blocks;
  for (int li = 1; li <= 4; ++li) {
    const int planes = 64 << (li - 1);
    const int seg = li == 4 ? 1 : (li == 3 ? 2 : 3);
    for (int bi = 0; bi < nblocks[li - 1]; ++bi) {
      const int stride = (bi == 0 && li > 1) ? 2 : 1;
      const std::string pre = "encoder.layer" + std::to_string(li) + "." + std::to_string(bi);
      Blk blk;
      blk.u1 = add_unit(pre + ".conv1", pre + ".bn1", cur, -1, 0, planes, 3, stride, 1, true, false, true, true, seg);
      if (blk.u1 < 0) return blk.u1;
      blk.u2 = add_unit(pre + ".conv2", pre + ".bn2", units[blk.u1].a, -1, 0, planes, 3, 1, 1, true, false, true, true, seg);
      if (blk.u2 < 0) return blk.u2;
      blk.ud = -1;
      if (stride != 1 || inpl != planes) {
        blk.ud = add_unit(pre + ".downsample.0", pre + ".downsample.1", cur, -1, 0, planes, 1, stride, 0, true, false, false, false, seg);
        if (blk.ud < 0) return blk.ud;
        units[blk.u2].res_unit = blk.ud;
      } else {
        units[blk.u2].res_tensor = cur;
      }
      enc_blocks.push_back(blk);
      cur = units[blk.u2].a;
      inpl = planes;
    }
    feat[li + 1] = cur;
  }
  const int dec_out[5] = {256, 128, 64, 32, 16};
  const int skips[5] = {feat[4], feat[3], feat[2], feat[1], -1};
  int x = feat[5];
  struct DBlk { int u1, u2; };
  std::vector<DBlk> dec_blocks;
  for (int i = 0; i < 5; ++i) {
    const std::string pre = "decoder.blocks." + std::to_string(i);
    DBlk d;
    d.u1 = add_unit(pre + ".conv1.0", pre + ".conv1.1", x, skips[i], 1, dec_out[i], 3, 1, 1, true, false, true, true, 0);
    if (d.u1 < 0) return d.u1;
    d.u2 = add_unit(pre + ".conv2.0", pre + ".conv2.1", units[d.u1].a, -1, 0, dec_out[i], 3, 1, 1, true, false, true, true, 0);
    if (d.u2 < 0) return d.u2;
    dec_blocks.push_back(d);
    x = units[d.u2].a;
  }
  head = add_unit("segmentation_head.0", "", x, -1, 0, classes, 3, 1, 1, false, true, false, false, 0);
  if (head < 0) return head;

  for (auto& u : units)
    if (int rc = plan_unit(u)) return rc;
  for (int ui = 0; ui < (int)units.size() && first_late_unit_ < 0; ++ui)
    if (units[ui].conv_name.rfind("encoder.layer3.", 0) == 0) first_late_unit_ = ui;
  for (int ui = 0; ui < (int)units.size() && first_mid_unit_ < 0; ++ui)
    if (units[ui].conv_name.rfind("encoder.layer1.", 0) == 0) first_mid_unit_ = ui;
  if (first_late_unit_ <= 0 || first_mid_unit_ <= 0 || first_mid_unit_ >= first_late_unit_) first_mid_unit_ = -1;

  // ---- backward schedule (static: first writer writes, later writers accumulate) --------
  auto grad_dst = [&](int tid, bool* acc) {
    const int g = new_grad(tid);
    *acc = grad_init[g];
    grad_init[g] = true;
    return g;
  };
  auto emit_unit = [&](int ui, int dA, bool mask, int dres, bool dres_acc) {
    const Unit& u = units[ui];
    BwdOp op;
    op.kind = BW_UNIT; op.unit = ui; op.dA = dA; op.mask = mask; op.dres = dres; op.dres_acc = dres_acc;
    op.segment = u.segment;
    if (u.need_dgrad) {
      if (u.upfold) {  // two launches: low-resolution source (dst0) and skip tensor (dst1), both written directly
        op.dst0 = grad_dst(u.in0, &op.acc0);
        if (u.in1 >= 0) op.dst1 = grad_dst(u.in1, &op.acc1);
        D3F_CHECK(u.segment == 0, "plan: unit %s: a folded up-sampling layer outside the decoder bucket", u.conv_name.c_str());
      } else if (u.up0 && !u.dgrad.sum2) {
        op.dst0_is_full_scratch = true;
        if (u.in1 >= 0) op.dst1 = grad_dst(u.in1, &op.acc1);
      } else {
        op.dst0 = grad_dst(u.in0, &op.acc0);
      }
    }
    bwd_ops.push_back(op);
    if (u.need_dgrad && u.up0 && !u.upfold && !u.dgrad.sum2) {
      BwdOp s;
      s.kind = BW_SUM2X2; s.unit = ui; s.C = u.C0; s.Hl = u.Hv / 2; s.Wl = u.Wv / 2; s.segment = u.segment;
      bool acc;
      s.dst0 = grad_dst(u.in0, &acc);
      D3F_CHECK(!acc, "plan: up-sampled tensor has two consumers");
      bwd_ops.push_back(s);
    }
    return 0;
  };
  {
    BwdOp h;
    h.kind = BW_HEAD; h.unit = head; h.segment = 0;
    h.dst0 = grad_dst(units[head].in0, &h.acc0);
    bwd_ops.push_back(h);
  }
  for (int i = 4; i >= 0; --i) {
    const DBlk& d = dec_blocks[i];
    if (int rc = emit_unit(d.u2, grad_of[units[d.u2].a], true, -1, false)) return rc;
    if (int rc = emit_unit(d.u1, grad_of[units[d.u1].a], true, -1, false)) return rc;
  }
  for (int bi = (int)enc_blocks.size() - 1; bi >= 0; --bi) {
    const Blk& b = enc_blocks[bi];
    const Unit& u2 = units[b.u2];
    D3F_CHECK(grad_of[u2.a] >= 0, "plan: block output without gradient");
    int dres;
    bool dres_acc = false;
    if (b.ud >= 0) dres = -2;  // dz scratch feeds the downsample branch
    else dres = grad_dst(u2.res_tensor, &dres_acc);
    if (int rc = emit_unit(b.u2, grad_of[u2.a], true, dres, dres_acc)) return rc;
    if (int rc = emit_unit(b.u1, grad_of[units[b.u1].a], true, -1, false)) return rc;
    if (b.ud >= 0)
      if (int rc = emit_unit(b.ud, -2, false, -1, false)) return rc;
  }
  {
    BwdOp p;
    p.kind = BW_POOL; p.segment = 3;
    p.dA = grad_of[t_pool];
    D3F_CHECK(p.dA >= 0, "plan: pool output without gradient");
    p.dst0 = grad_dst(f1, &p.acc0);
    bwd_ops.push_back(p);
  }
  if (int rc = emit_unit(conv1, grad_of[f1], true, -1, false)) return rc;

  // ---- BatchNorm-backward reduce fusion (f32 storage): the data gradient that is the LAST writer of an activation
  // gradient also emits the (dbeta, dgamma) partial sums of the unit that consumes it -- when that unit is the very
  // next op (bnpart is one stream-ordered scratch).  The producer may accumulate (it sums first, then reduces the
  // final values); a consumer with a residual add takes its ReLU mask from its activation instead of from y.
  static const bool no_fused_reduce = prof_knob("D3F_NO_FUSED_BN_REDUCE") != nullptr;  // debugging knob: separate reduce launches
  if (!no_fused_reduce) {
    auto writes = [&](const BwdOp& o, int gid) {
      if (o.kind == BW_UNIT || o.kind == BW_HEAD) {
        if (units[o.unit].need_dgrad && !o.dst0_is_full_scratch && o.dst0 == gid) return true;
        if (o.dst1 == gid) return true;
        if (o.kind == BW_UNIT && o.dres == gid) return true;
      }
      return (o.kind == BW_SUM2X2 || o.kind == BW_POOL) && o.dst0 == gid;
    };
    for (size_t j = 0; j + 1 < bwd_ops.size(); ++j) {
      BwdOp& pj = bwd_ops[j];
      if (!(pj.kind == BW_UNIT || pj.kind == BW_HEAD)) continue;
      const Unit& up = units[pj.unit];
      const ConvParams& pd = up.upfold ? up.dgrad_lo : up.dgrad;  // the launch that writes dst0
      if (!up.need_dgrad || pd.par || pj.dst0_is_full_scratch || (pj.dst1 >= 0 && !up.upfold) || pj.dst0 < 0) continue;
      BwdOp& ck = bwd_ops[j + 1];  // the consumer must be the very next op (bnpart is a stream-ordered scratch)
      if (ck.kind != BW_UNIT || ck.dA != pj.dst0) continue;
      bool last = true;  // no later writer, and (a dres write of op j itself happens BEFORE its data gradient)
      for (size_t k = j + 1; k < bwd_ops.size(); ++k) last = last && !writes(bwd_ops[k], pj.dst0);
      const Unit& uc = units[ck.unit];
      if (!last || !(uc.bn && ck.mask)) continue;
      D3F_CHECK(pd.Cout == uc.Cout && pd.M == (pd.sum2 ? 4 : 1) * B * uc.Ho * uc.Wo && pd.out_c0 == pd.Cout,
                "plan: fused BatchNorm reduce shape mismatch (%s -> %s)", up.conv_name.c_str(), uc.conv_name.c_str());
      pj.fuse_for_unit = ck.unit;
      ck.fused_rows = pd.splitk > 1 ? pd.stat_rows : pd.tiles_m;
      bnpart_bytes = std::max(bnpart_bytes, (size_t)ck.fused_rows * uc.Cout * 2 * sizeof(float));
    }
  }

  // ---- scratch ------------------------------------------------------------------------
  stats_off = alloc(stats_bytes);
  bnpart_off = alloc(bnpart_bytes);
  dz_off = alloc(dz_bytes);
  dfull_off = alloc(dfull_bytes);
  bsum_off = alloc(channel_sum_partial_floats(B, classes) * sizeof(float));
  splitk_off = alloc(splitk_bytes);
  head_nchw_off = alloc((size_t)B * classes * H * W * sizeof(float));  // predict_u8: head output before K16 post
  // a pair keeps two copies of this layout in one buffer, a whole number of 4 KB pages apart
  net_ws_stride = nets > 1 ? (size_t)round_up((long)ws_top, 4096) : 0;
  workspace_bytes = nets > 1 ? net_ws_stride * nets : ws_top;

  // ---- gradient buckets (contiguous slices of the flat gradient, ready in this order) ----
  long layer3_start = -1, layer4_start = -1, dec_start = -1;
  for (const auto& p : params) {
    if (layer3_start < 0 && p.name.rfind("encoder.layer3.", 0) == 0) layer3_start = p.offset;
    if (layer4_start < 0 && p.name.rfind("encoder.layer4.", 0) == 0) layer4_start = p.offset;
    if (dec_start < 0 && p.name.rfind("decoder.", 0) == 0) dec_start = p.offset;
  }
  num_segments = 4;
  seg_grad_begin[0] = dec_start;     seg_grad_end[0] = param_floats;
  seg_grad_begin[1] = layer4_start;  seg_grad_end[1] = dec_start;
  seg_grad_begin[2] = layer3_start;  seg_grad_end[2] = layer4_start;
  seg_grad_begin[3] = 0;             seg_grad_end[3] = layer3_start;
  fwd_order_.clear();
  fwd_order_.push_back(conv1);
  fwd_order_.push_back(-1);  // max-pool
  for (const auto& b : enc_blocks) {
    fwd_order_.push_back(b.u1);
    if (b.ud >= 0) fwd_order_.push_back(b.ud);
    fwd_order_.push_back(b.u2);
  }
  for (const auto& d : dec_blocks) {
    fwd_order_.push_back(d.u1);
    fwd_order_.push_back(d.u2);
  }
  fwd_order_.push_back(head);
  return 0;
}

// ------------------------------------------------------------------------------------------
// ONE side stream per device for every engine of the process.  A module holds several engines -- train_deep_fake's swap
// mode runs four networks (two students, two EMA teachers: d3f/train_deep_fake/lit_module.py:36-40), each with a plan per
// shape -- and they run one after the other on the caller's stream; a side stream per ENGINE gave that step five streams,
// and with more than three streams in use the ROCm 7 runtime (four hardware queues by default) makes every launch of the
// step slow: swap mode 36.2 ms per combined batch with per-engine streams, 14.1 ms with two streams in all, 13.9 ms with
// GPU_MAX_HW_QUEUES=2 (profiles/README.md, round 6).  Sharing only ever ADDS ordering between engines, never removes any.
static int shared_side_stream(hipStream_t* out) {
  constexpr int MAXDEV = 64;
  static hipStream_t streams[MAXDEV] = {};
  int dev = 0;
  D3F_HIP(hipGetDevice(&dev));
  D3F_CHECK(dev >= 0 && dev < MAXDEV, "side stream: device index %d", dev);
  if (streams[dev] == nullptr) {
    // lowest priority: the weight gradients fill the machine behind the dependent chain on the caller's stream
    // (BatchNorm backward -> data gradient), whose workgroups should get freed CUs first (+0.5-1 % measured).
    // A CU mask on this stream (hipExtStreamCreateWithCUMask, every 2nd..8th CU
    // left to the caller's stream) was tried and halves the throughput on this platform.
    int least = 0, greatest = 0;
    D3F_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    D3F_HIP(hipStreamCreateWithPriority(&streams[dev], hipStreamNonBlocking, least));
  }
  *out = streams[dev];  // lives as long as the process (engines come and go; nothing is ever left running on it unjoined)
  return 0;
}

int UnetEngine::ensure_streams() const {
  if (side_ != nullptr) return 0;
  if (int rc = shared_side_stream(&side_)) return rc;
  D3F_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
  D3F_HIP(hipEventCreateWithFlags(&ev_seg_, hipEventDisableTiming));
  D3F_HIP(hipEventCreateWithFlags(&ev_pack_in_, hipEventDisableTiming));
  D3F_HIP(hipEventCreateWithFlags(&ev_pack_done_, hipEventDisableTiming));
  D3F_HIP(hipEventCreateWithFlags(&ev_pack_mid_, hipEventDisableTiming));
  return 0;
}

bool UnetEngine::make_split(const NetIO* io, long in_delta, NetSplit* ns) const {
  if (nets <= 1) return false;
  ns->nets = nets;
  ns->ws = (long)net_ws_stride;
  ns->par = io->params;
  ns->grad = io->grads;
  ns->bn = io->bnstats;
  ns->in = in_delta;
  ns->out = io->out;
  return true;
}

int UnetEngine::pack_weights(const float* params0_, void* ws0_, hipStream_t s, const NetIO* io) const {
  D3F_CHECK((int)units.size() <= PACK_MAX_LAYERS, "pack_weights: %d layers exceed the table", (int)units.size());
  D3F_CHECK(nets == 1 || io != nullptr, "pack_weights: a pair engine needs the second network's offsets");
  const int ve = dtype == D3F_F32 ? 4 : 8;
  static const bool sync_pack = getenv("D3F_NO_ASYNC_PACK") != nullptr;  // debugging knob: everything on the caller's stream
  const bool async = !sync_pack && first_late_unit_ > 0;
  const int mid = async ? first_mid_unit_ : -1;
  // (a pair: every layout launch once per network -- these are HBM-bound, chip-filling launches off the critical path)
  auto params_of = [&](int n) { return reinterpret_cast<const float*>(reinterpret_cast<const char*>(params0_) + (n ? io->params : 0)); };
  auto ws_of = [&](int n) { return reinterpret_cast<char*>(ws0_) + (size_t)n * net_ws_stride; };
  // parts: 0 = the first layers (caller's stream: encoder.conv1, or everything before layer3 without a middle part),
  // 1 = layer1-2 (side stream, own event; empty without a middle part), 2 = the rest (side stream)
  auto part_of = [&](int ui) { return ui >= first_late_unit_ ? 2 : (mid > 0 && ui >= mid) ? 1 : 0; };
  // (the data-gradient layouts in a separate, later pass were measured slower: profiles/README.md round 3)
  for (int part = 0; part < 3; ++part) {
    if (part == 1 && mid <= 0) continue;
    PackTable t;
    t.n = 0;
    uint32_t blocks = 0;
    for (int ui = 0; ui < (int)units.size(); ++ui) {
      const Unit& u = units[ui];
      if (first_late_unit_ > 0 && part_of(ui) != part) continue;
      if (first_late_unit_ <= 0 && part != 0) continue;
      if (u.upfold) continue;  // packed by pack_up_launch below (none of the plain layouts is read for these layers)
      PackEntry& e = t.e[t.n++];
      const int CoutD = (int)round_up(u.Cout, ve);
      const long nf = (long)u.CoutPad * u.Kpad, nd = u.need_dgrad ? (long)u.CinRows * u.KpadD : 0;
      D3F_CHECK(u.Kpad >= u.KH * u.KW * u.Cin() && (!u.need_dgrad || u.KpadD >= u.KH * u.KW * CoutD),
                "pack_weights: padded K too small");
      D3F_CHECK(u.wf_off % 16 == 0 && u.wd_off % 16 == 0 && (u.wf_off >> 4) < 0xffffffffull &&
                    (u.wd_off >> 4) < 0xffffffffull && nf + nd < 0x7fffffffl && u.Kpad < 65536 && u.KpadD < 65536,
                "pack_weights: layer out of table range");
      e.w_off = (uint32_t)u.w_off;
      e.wf_off16 = (uint32_t)(u.wf_off >> 4);
      e.wd_off16 = (uint32_t)(u.wd_off >> 4);
      e.block0 = blocks;
      e.Cout = (uint16_t)u.Cout; e.CinReal = (uint16_t)u.CinReal; e.Cin = (uint16_t)u.Cin();
      e.taps = (uint16_t)(u.KH * u.KW); e.CoutPad = (uint16_t)u.CoutPad; e.Kpad = (uint16_t)u.Kpad;
      e.CinRows = (uint16_t)u.CinRows; e.CoutD = (uint16_t)CoutD; e.KpadD = (uint16_t)u.KpadD;
      e.has_d = u.need_dgrad ? 1 : 0;
      e.conv_stride = (u.need_dgrad && u.dgrad.par) ? 2 : 1;
      const int taps = u.KH * u.KW;
      D3F_CHECK(taps <= PACK_LDS_ROW, "pack_weights: %d taps exceed the tile", taps);
      int CT = 32;
      while (CT * taps > PACK_LDS_ROW) CT >>= 1;
      const int crows = std::max(u.Cin(), u.need_dgrad ? u.CinRows : 0);
      const int nrows = std::max(u.CoutPad, u.need_dgrad ? CoutD : 0);
      e.CT = (uint16_t)CT;
      {
        unsigned mul, shr;
        fast_div_setup((unsigned)taps, &mul, &shr);
        e.taps_mul = mul;
        e.taps_shr = (uint16_t)shr;
        e.ct_log2 = (uint16_t)__builtin_ctz((unsigned)CT);
      }
      e.ctiles = (uint16_t)((crows + CT - 1) / CT);
      blocks += (uint32_t)e.ctiles * (uint32_t)((nrows + PACK_NT - 1) / PACK_NT);
    }
    hipStream_t ps = s;
    if (part >= 1 && async) {
      if (int rc = ensure_streams()) return rc;
      if (part == 1 || mid <= 0) {
        D3F_HIP(hipEventRecord(ev_pack_in_, s));  // the parameter update (and every reader of the old layouts) is done
        D3F_HIP(hipStreamWaitEvent(side_, ev_pack_in_, 0));
      }
      ps = side_;
    }
    for (int n = 0; n < nets; ++n) {
      const float* params_ = params_of(n);
      char* ws = ws_of(n);
      if (t.n > 0)
        if (int rc = pack_all_launch(cdtype, params_, ws, t, (int)blocks, ps)) return rc;
      // transformed filters of the Winograd layers of this part, behind its plain layouts
      for (int ui = 0; ui < (int)units.size(); ++ui) {
        const Unit& u = units[ui];
        if (!u.wino) continue;
        if (first_late_unit_ > 0 ? part_of(ui) != part : part != 0) continue;
        if (int rc = conv_winograd_pack_launch(params_ + u.w_off, reinterpret_cast<float*>(ws + u.wu_off), u.Cout, u.Cin(), ps))
          return rc;
      }
    }
    if (part == 1 && async) {
      D3F_HIP(hipEventRecord(ev_pack_mid_, side_));
      pack_mid_pending_ = true;
    }
    if (part == 2 || (first_late_unit_ <= 0 && part == 0)) {
      for (int n = 0; n < nets; ++n) {
        const float* params_ = params_of(n);
        char* ws = ws_of(n);
        for (const Unit& u : units)  // the folded decoder layers: all late
          if (u.upfold)
            if (int rc = pack_up_launch(cdtype, params_ + u.w_off, u.Cout, u.C0, u.C1, ws + u.wfc_off, u.CoutPad,
                                        u.need_dgrad ? ws + u.wd4_off : nullptr, u.C0Rows,
                                        (u.need_dgrad && u.C1 > 0) ? ws + u.wds_off : nullptr, u.C1Rows, ps))
              return rc;
      }
    }
    if (part == 2 && async) {
      D3F_HIP(hipEventRecord(ev_pack_done_, side_));
      pack_pending_ = true;
    }
  }
  return 0;
}

static inline float* coef_ptr(char* ws, const Unit& u, int which) {
  return reinterpret_cast<float*>(ws + u.coef_off) + (long)which * u.Cout;
}

int UnetEngine::forward(const float* params_, float* bnstats, const float* x, float* out, void* ws_,
                        int training, hipStream_t s, const NetIO* io) const {
  char* ws = reinterpret_cast<char*>(ws_);
  D3F_CHECK(nets == 1 || (io != nullptr && training), "unet: the pair engine runs train-mode passes and needs the second "
            "network's offsets");
  NetSplit split{};
  const NetSplit* ns = make_split(io, io ? io->x : 0, &split) ? &split : nullptr;
  if (int rc = nchw_to_nhwc_launch(dtype, x, ws + tensors[t_x].off, B, in_channels, H, W, tensors[t_x].C, s, ns))
    return rc;
  return forward_body(params_, bnstats, out, ws, training, s, ns);
}

// everything after the input layout conversion; `out` = NCHW fp32 destination of the head
int UnetEngine::forward_body(const float* params_, float* bnstats, float* out, char* ws, int training,
                             hipStream_t s, const NetSplit* ns) const {
  auto T = [&](int tid) { return ws + tensors[tid].off; };
  D3F_CHECK(ns == nullptr || (training && bn_sync_fn_ == nullptr), "unet: the pair engine runs train-mode passes with "
            "per-GPU BatchNorm statistics");
  if (!training) {  // folded BatchNorm coefficients of all layers: one launch
    BnEvalTable t;
    t.n = 0;
    for (const Unit& u : units)
      if (u.bn) {
        D3F_CHECK(t.n < PACK_MAX_LAYERS && u.coef_off % 16 == 0, "eval coefficients: table overflow");
        BnEvalEntry& e = t.e[t.n++];
        e.g_off = (uint32_t)u.g_off; e.b_off = (uint32_t)u.b_off;
        e.rm_off = (uint32_t)u.rm_off; e.rv_off = (uint32_t)u.rv_off;
        e.coef_off16 = (uint32_t)(u.coef_off >> 4);
        e.C = u.Cout;
      }
    if (int rc = bn_eval_coeff_all_launch(params_, bnstats, ws, 1e-5f, t, s)) return rc;
  }
  for (int ui : fwd_order_) {
    if (ui < 0) {
      const TensorD& f1 = tensors[units[conv1].a];
      if (int rc = maxpool3x3s2_fwd_launch(dtype, T(units[conv1].a), T(t_pool),
                                           reinterpret_cast<uint8_t*>(ws + pool_idx_off), B, f1.H, f1.W,
                                           f1.C, s, ns))
        return rc;
      continue;
    }
    const Unit& u = units[ui];
    if (pack_mid_pending_ && ui >= first_mid_unit_) {  // layer1-2: second part
      D3F_HIP(hipStreamWaitEvent(s, ev_pack_mid_, 0));
      pack_mid_pending_ = false;
    }
    if (pack_pending_ && ui >= first_late_unit_) {  // packed weights of the later layers come from the side stream
      D3F_HIP(hipStreamWaitEvent(s, ev_pack_done_, 0));
      pack_pending_ = false;
    }
    ConvParams p = u.fwd;
    p.src0 = T(u.in0);
    p.src1 = u.in1 >= 0 ? T(u.in1) : nullptr;
    p.w = ws + (u.upfold ? u.wfc_off : u.wf_off);
    net_conv(p, ns);
    if (!u.bn) {  // segmentation head
      p.mode = CONV_HEAD_NCHW;
      p.out0 = out;
      p.scale = params_ + u.bias_off;
      if (ns != nullptr) {  // the prediction and the bias live outside the workspace
        p.net_out0 = ns->out;
        p.net_scale = ns->par;
      }
      if (int rc = conv_igemm_launch(p, cdtype, s)) return rc;
      continue;
    }
    const Unit* ds = u.res_unit >= 0 ? &units[u.res_unit] : nullptr;
    if (training) {
      p.mode = CONV_RAW_STATS;
      p.out0 = T(u.y);
      p.stats = reinterpret_cast<float*>(ws + stats_off);
      p.partial = p.splitk > 1 ? reinterpret_cast<float*>(ws + splitk_off) : nullptr;
      if (u.wino) {  // Winograd F(2x2, 3x3): its own filter layout and one statistics row per workgroup
        p.w = ws + u.wu_off;
        p.stat_rows = u.wino_rows;
        if (int rc = conv_winograd_launch(p, s)) return rc;
      } else if (int rc = conv_igemm_launch(p, cdtype, s)) {
        return rc;
      }
      const long rows = (long)B * u.Ho * u.Wo;  // (p.M counts one output-parity class for a folded layer)
      const bool sync = bn_sync_fn_ != nullptr;
      if (sync) {  // statistics over every rank's batch: the partial rows are summed across ranks in place
        if (int rc = bn_sync_fn_(bn_sync_ctx_, p.stats, (int64_t)p.stat_rows * u.CoutPad * 2, (void*)s))
          return set_error(rc, "BatchNorm statistics all-reduce failed in the forward pass (%s)", u.bn_name.c_str());
      }
      const long count = rows * (sync ? bn_sync_world_ : 1);
#ifdef D3F_PROFILING
      // timing-only ablation (wrong activations; profiling builds only): D3F_ABLATE_FWD_BN=1 replaces the streaming
      // BatchNorm + ReLU pass of every layer WITHOUT a residual add by the coefficient-only finalize launch -- the ceiling
      // of "apply the two coefficients in the consumer's patch staging instead"  before any consumer cost
      static const bool abl_fbn = getenv("D3F_ABLATE_FWD_BN") != nullptr;
      if (abl_fbn && !sync && u.apply && u.res_tensor < 0 && ds == nullptr) {
        if (int rc = bn_finalize_launch(p.stats, p.stat_rows, u.Cout, u.CoutPad, count, params_ + u.g_off, params_ + u.b_off,
                                        1e-5f, 0.1f, bnstats + u.rm_off, bnstats + u.rv_off, coef_ptr(ws, u, 0),
                                        coef_ptr(ws, u, 1), coef_ptr(ws, u, 2), coef_ptr(ws, u, 3), s, ns))
          return rc;
        continue;
      }
#endif
      if (!sync && u.apply && bn_fused_finalize_ok(dtype, p.stat_rows, u.Cout)) {
        // finalize folded into the streaming pass (bn_fused.hip): one launch instead of two
        if (int rc = bn_finalize_apply_launch(dtype, p.stats, p.stat_rows, u.Cout, u.CoutPad, rows, params_ + u.g_off,
                                              params_ + u.b_off, 1e-5f, 0.1f, bnstats + u.rm_off, bnstats + u.rv_off,
                                              coef_ptr(ws, u, 0), coef_ptr(ws, u, 1), coef_ptr(ws, u, 2),
                                              coef_ptr(ws, u, 3), T(u.y),
                                              u.res_tensor >= 0 ? T(u.res_tensor) : nullptr, ds ? T(ds->y) : nullptr,
                                              ds ? coef_ptr(ws, *ds, 2) : nullptr, ds ? coef_ptr(ws, *ds, 3) : nullptr,
                                              u.relu ? 1 : 0, T(u.a), rows, s, ns, plan_nets))
          return rc;
        continue;
      }
      if (int rc = bn_finalize_launch(p.stats, p.stat_rows, u.Cout, u.CoutPad, count,
                                      params_ + u.g_off, params_ + u.b_off, 1e-5f, 0.1f,
                                      bnstats + u.rm_off, bnstats + u.rv_off, coef_ptr(ws, u, 0),
                                      coef_ptr(ws, u, 1), coef_ptr(ws, u, 2), coef_ptr(ws, u, 3), s, ns))
        return rc;
      if (u.apply) {
        if (int rc = bn_apply_launch(dtype, T(u.y), coef_ptr(ws, u, 2), coef_ptr(ws, u, 3),
                                     u.res_tensor >= 0 ? T(u.res_tensor) : nullptr,
                                     ds ? T(ds->y) : nullptr, ds ? coef_ptr(ws, *ds, 2) : nullptr,
                                     ds ? coef_ptr(ws, *ds, 3) : nullptr, u.relu ? 1 : 0, T(u.a),
                                     rows, u.Cout, s, ns))
          return rc;
      }
    } else {
      p.mode = CONV_EVAL_FUSED;
      p.scale = coef_ptr(ws, u, 2);
      p.shift = coef_ptr(ws, u, 3);
      p.relu = u.relu ? 1 : 0;
      p.partial = p.splitk > 1 ? reinterpret_cast<float*>(ws + splitk_off) : nullptr;  // small batches: split K
      if (u.apply) {
        p.out0 = T(u.a);
        p.res = u.res_tensor >= 0 ? T(u.res_tensor) : (ds ? T(ds->y) : nullptr);
      } else {
        p.out0 = T(u.y);  // downsample branch: bn(conv(x)) lands in its y slot
        p.res = nullptr;
      }
      if (u.wino) {
        p.w = ws + u.wu_off;
        if (int rc = conv_winograd_launch(p, s)) return rc;
      } else if (int rc = conv_igemm_launch(p, cdtype, s)) {
        return rc;
      }
    }
  }
  return 0;
}

int UnetEngine::predict_u8_launches(const float* params_, float* bnstats, const uint8_t* bgr_in, uint8_t* bgr_out,
                                    const float mean255[3], const float std255[3], char* ws, hipStream_t s) const {
  if (int rc = u8bgr_to_nhwc_launch(dtype, bgr_in, ws + tensors[t_x].off, (long)B * H * W, tensors[t_x].C, mean255,
                                    std255, s))
    return rc;
  float* head_out = reinterpret_cast<float*>(ws + head_nchw_off);
  if (int rc = forward_body(params_, bnstats, head_out, ws, 0, s)) return rc;
  return nchw_to_u8bgr_launch(head_out, bgr_out, B, (long)H * W, mean255, std255, s);
}

int UnetEngine::predict_u8(const float* params_, float* bnstats, const uint8_t* bgr_in, uint8_t* bgr_out,
                           const float mean[3], const float stdv[3], void* ws_, int use_graph,
                           hipStream_t s) const {
  D3F_CHECK(in_channels == 3 && classes == 3, "predict_u8: 3-channel frames only (in %d, out %d)", in_channels, classes);
  char* ws = reinterpret_cast<char*>(ws_);
  float m255[3], s255[3];
  for (int c = 0; c < 3; ++c) {  // fp32 products, as torch.tensor(mean) * 255 gives
    m255[c] = mean[c] * 255.0f;
    s255[c] = stdv[c] * 255.0f;
  }
  if (!use_graph) return predict_u8_launches(params_, bnstats, bgr_in, bgr_out, m255, s255, ws, s);
  // hipGraph path: at B = 1 the ~100 launches of an eval forward are launch-bound; capture them once
  const void* key[5] = {params_, bnstats, bgr_in, bgr_out, ws_};
  const float cst[6] = {m255[0], m255[1], m255[2], s255[0], s255[1], s255[2]};
  return graph_replay(g_predict_, key, cst, s, [&](hipStream_t gs) {
    return predict_u8_launches(params_, bnstats, bgr_in, bgr_out, m255, s255, ws, gs);
  });
}

int UnetEngine::forward_graph(const float* params_, float* bnstats, const float* x, float* out, void* ws_,
                              hipStream_t s) const {
  const void* key[5] = {params_, bnstats, x, out, ws_};
  const float cst[6] = {0, 0, 0, 0, 0, 0};
  return graph_replay(g_eval_, key, cst, s, [&](hipStream_t gs) { return forward(params_, bnstats, x, out, ws_, 0, gs); });
}

int UnetEngine::train_step_launches(const StepArgs& a, void* ws_, hipStream_t s) const {
  const long per_image = (long)in_channels * H * W;
  if (int rc = pack_weights(a.params, ws_, s)) return rc;
  if (int rc = noise_blend_launch(a.image, a.noise, a.y_uniform, a.lam, a.noisy, nullptr, B, per_image, s)) return rc;
  if (int rc = forward(a.params, a.bnstats, a.noisy, a.pred, ws_, 1, s)) return rc;
  if (int rc = mse_ssim_loss_launch(a.pred, a.image, a.lo, a.hi, a.loss_out, a.gpred, a.loss_ws, B, H, W, s)) return rc;
  if (int rc = backward(a.params, a.gpred, a.grads, ws_, 0, num_segments, s, 1)) return rc;
  return adam_step_dev_launch(a.params, a.grads, a.exp_avg, a.exp_avg_sq, param_floats, a.adam_coef, s);
}

int UnetEngine::train_step(const StepArgs& a, void* ws_, int use_graph, hipStream_t s) const {
  D3F_CHECK(in_channels == 3 && classes == 3, "train_step: the (MSE + 1 - SSIM) / 2 objective is defined on 3-channel images");
  if (!use_graph) return train_step_launches(a, ws_, s);
  if (int rc = wait_for_packed_weights(s)) return rc;  // (a pack enqueued eagerly before this call)
  if (int rc = ensure_streams()) return rc;
  if (gstream_ == nullptr) {
    D3F_HIP(hipStreamCreateWithFlags(&gstream_, hipStreamNonBlocking));
    D3F_HIP(hipEventCreateWithFlags(&ev_gin_, hipEventDisableTiming));
    D3F_HIP(hipEventCreateWithFlags(&ev_gout_, hipEventDisableTiming));
  }
  const bool same = g_step_ != nullptr && g_step_ws_ == ws_ && memcmp(&a, &g_step_key_, sizeof(StepArgs)) == 0;
  if (!same) {
    if (g_step_) {
      (void)hipGraphExecDestroy(g_step_);
      g_step_ = nullptr;
    }
    // The capture spans three streams: the capture stream (the dependent chain), the weight-gradient stream and the
    // late weight packing join it through the events the eager path uses (record on a capturing stream + wait on
    // another pulls that stream into the capture) and are joined back before the capture ends (backward joins the side
    // stream; the pack events are waited for by the forward pass).
    hipGraph_t graph = nullptr;
    D3F_HIP(hipStreamBeginCapture(gstream_, hipStreamCaptureModeThreadLocal));
    const int rc = train_step_launches(a, ws_, gstream_);
    const hipError_t e = hipStreamEndCapture(gstream_, &graph);
    pack_pending_ = pack_mid_pending_ = false;  // (consumed inside the capture)
    side_dirty_ = false;
    if (rc != 0) {
      if (graph) (void)hipGraphDestroy(graph);
      return rc;
    }
    D3F_HIP(e);
    const hipError_t ei = hipGraphInstantiate(&g_step_, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    D3F_HIP(ei);
    g_step_key_ = a;
    g_step_ws_ = ws_;
  }
  D3F_HIP(hipGraphLaunch(g_step_, s));  // stream-ordered on the caller's stream: no extra events
  return 0;
}

// the packed weights of the later layers may still be in flight on the side stream: order `s` behind them (outside
// any capture: the replay stream is ordered after `s`)
int UnetEngine::wait_for_packed_weights(hipStream_t s) const {
  if (pack_mid_pending_) {
    D3F_HIP(hipStreamWaitEvent(s, ev_pack_mid_, 0));
    pack_mid_pending_ = false;
  }
  if (pack_pending_) {
    D3F_HIP(hipStreamWaitEvent(s, ev_pack_done_, 0));
    pack_pending_ = false;
  }
  return 0;
}

template <typename F>
int UnetEngine::graph_replay(GraphSlot& slot, const void* const key[5], const float cst[6], hipStream_t s,
                             F&& launches) const {
  if (int rc = wait_for_packed_weights(s)) return rc;
  if (gstream_ == nullptr) {
    D3F_HIP(hipStreamCreateWithFlags(&gstream_, hipStreamNonBlocking));
    D3F_HIP(hipEventCreateWithFlags(&ev_gin_, hipEventDisableTiming));
    D3F_HIP(hipEventCreateWithFlags(&ev_gout_, hipEventDisableTiming));
  }
  const bool same = slot.exec != nullptr && memcmp(key, slot.key, sizeof(slot.key)) == 0 &&
                    memcmp(cst, slot.cst, sizeof(slot.cst)) == 0;
  if (!same) {
    if (slot.exec) {
      (void)hipGraphExecDestroy(slot.exec);
      slot.exec = nullptr;
    }
    hipGraph_t graph = nullptr;
    D3F_HIP(hipStreamBeginCapture(gstream_, hipStreamCaptureModeThreadLocal));
    const int rc = launches(gstream_);
    const hipError_t e = hipStreamEndCapture(gstream_, &graph);
    if (rc != 0) {
      if (graph) (void)hipGraphDestroy(graph);
      return rc;
    }
    D3F_HIP(e);
    const hipError_t ei = hipGraphInstantiate(&slot.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    D3F_HIP(ei);
    memcpy(slot.key, key, sizeof(slot.key));
    memcpy(slot.cst, cst, sizeof(slot.cst));
  }
  // order the replay after the caller's stream (input, packed weights) and the caller after the replay
  D3F_HIP(hipEventRecord(ev_gin_, s));
  D3F_HIP(hipStreamWaitEvent(gstream_, ev_gin_, 0));
  D3F_HIP(hipGraphLaunch(slot.exec, gstream_));
  D3F_HIP(hipEventRecord(ev_gout_, gstream_));
  D3F_HIP(hipStreamWaitEvent(s, ev_gout_, 0));
  return 0;
}

UnetEngine::~UnetEngine() {
  if (g_predict_.exec) (void)hipGraphExecDestroy(g_predict_.exec);
  if (g_eval_.exec) (void)hipGraphExecDestroy(g_eval_.exec);
  if (g_step_) (void)hipGraphExecDestroy(g_step_);
  if (ev_gin_) (void)hipEventDestroy(ev_gin_);
  if (ev_gout_) (void)hipEventDestroy(ev_gout_);
  if (gstream_) (void)hipStreamDestroy(gstream_);
  for (hipEvent_t e : ev_dy_)
    if (e) (void)hipEventDestroy(e);
  if (ev_join_) (void)hipEventDestroy(ev_join_);
  if (ev_seg_) (void)hipEventDestroy(ev_seg_);
  if (ev_pack_in_) (void)hipEventDestroy(ev_pack_in_);
  if (ev_pack_done_) (void)hipEventDestroy(ev_pack_done_);
  if (ev_pack_mid_) (void)hipEventDestroy(ev_pack_mid_);
  // (side_ is the process-wide stream of this device: not ours to destroy)
}

// Backward of segments [seg_begin, seg_end).  Per unit: BN backward (writes the unit's dY) -> {weight gradient,
// data gradient}.  The two gradients are independent, and the following unit's BN-backward kernels are
// HBM-bound while the weight gradient is MFMA-bound, so the weight gradients (+ their slab reduces) run on
// a side stream: main records "dY ready", side waits for it.  One weight-gradient launch (+ slab reduce) per layer
// (grouped launches of identical layers fill every wave slot in front of the chain's next kernels: measured slower,
// profiles/README.md round 2); every unit owns its dY, so nothing on the main chain ever waits for the side stream,
// which is joined before returning: after the call every gradient of the segments is final on the caller's stream.
static bool serial_backward() {
  static const bool serial = getenv("D3F_SERIAL_BACKWARD") != nullptr;  // debugging knob: no side stream
  return serial;
}

hipStream_t UnetEngine::side_stream() const {
  if (serial_backward() || ensure_streams() != 0) return nullptr;
  return side_;
}

int UnetEngine::backward_join(hipStream_t s) const {
  if (side_ != nullptr && side_dirty_) {
    D3F_HIP(hipEventRecord(ev_join_, side_));
    D3F_HIP(hipStreamWaitEvent(s, ev_join_, 0));
    side_dirty_ = false;
  }
  return 0;
}

int UnetEngine::backward(const float* params_, const float* dout, float* grads, void* ws_,
                         int seg_begin, int seg_end, hipStream_t s, int join, const NetIO* io) const {
  char* ws = reinterpret_cast<char*>(ws_);
  D3F_CHECK(nets == 1 || (io != nullptr && bn_sync_fn_ == nullptr), "unet: the pair engine needs the second network's "
            "offsets and runs with per-GPU BatchNorm statistics");
  NetSplit split{};
  const NetSplit* ns = make_split(io, io ? io->dout : 0, &split) ? &split : nullptr;
  auto T = [&](int tid) { return ws + tensors[tid].off; };
  auto G = [&](int gid) { return gid == -2 ? ws + dz_off : ws + gtensors[gid].off; };
  const bool serial = serial_backward();
#ifdef D3F_PROFILING
  // timing-only ablation (wrong gradients; profiling builds only): D3F_ABLATE_BACKWARD contains w (skip weight
  // gradients), d (data gradients), b (BatchNorm backward kernels) -- what each class costs on the critical path
  static const char* abl = getenv("D3F_ABLATE_BACKWARD");
  const bool skip_w = abl && strchr(abl, 'w'), skip_d = abl && strchr(abl, 'd'), skip_b = abl && strchr(abl, 'b');
  const bool skip_r = abl && strchr(abl, 'r');  // r: the slab reduces of the weight gradients (gradients stay unwritten)
#else
  constexpr bool skip_w = false, skip_d = false, skip_b = false, skip_r = false;
#endif
  if (!serial)
    if (int rc = ensure_streams()) return rc;
  hipStream_t ws_stream = serial ? s : side_;
  size_t next_event = 0;
  bool side_used = false;
  // Weight-gradient launches waiting for their "dY ready" event: one event for every `defer` launches (an event
  // record costs the chain a few us of dispatch latency, a deferred launch starts later).  Measured (r02_ap/aq, 1 / 2 / 3 /
  // 4 launches per event): 3 gives the shortest data-gradient launches (class 3.80 -> 3.64 ms per step) at an equal or
  // slightly shorter step; 2 and 4 are 0.5-1 % slower.  Pending launches never cross a gradient-bucket (segment) boundary.
  constexpr int defer = 3;
  std::vector<int> pending;
  int pending_segment = -1;
  bool head_bias_pending = false;
  // Slab reduces: one launch right behind every layer (every unit owns its slab region).  One launch per gradient
  // bucket (47 -> 4-6 launches, same sums) was measured equal or slightly SLOWER per step (r03): the big launches at the
  // bucket ends delay the buckets' "gradients final" point and the stream's tail more than the 40 launches cost.
  auto flush_pending = [&]() -> int {
    if (pending.empty()) return 0;
    if (!serial) {
      if (next_event == ev_dy_.size()) {
        hipEvent_t e = nullptr;
        D3F_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ev_dy_.push_back(e);
      }
      D3F_HIP(hipEventRecord(ev_dy_[next_event], s));
      D3F_HIP(hipStreamWaitEvent(side_, ev_dy_[next_event], 0));
      ++next_event;
      side_used = true;
    }
    if (head_bias_pending) {  // the head's bias gradient: two small launches that nothing on the chain waits for
      const Unit& uh = units[head];
      if (int rc = channel_sum_nchw_launch(dout, B, uh.Cout, (long)uh.Ho * uh.Wo, reinterpret_cast<float*>(ws + bsum_off),
                                           grads + uh.bias_off, ws_stream, ns))
        return rc;
      head_bias_pending = false;
    }
    for (int ui : pending) {
      const Unit& u = units[ui];
      WgradReduceBatch red;
      if (int rc = wgrad_layer_launch_deferred(u.wl, ws + u.dy_off, T(u.in0), u.in1 >= 0 ? T(u.in1) : nullptr,
                                               reinterpret_cast<float*>(ws + u.wslab_off), grads + u.w_off, u.Cout,
                                               u.CinReal, cdtype, red, ws_stream, ns))
        return rc;
      if (!skip_r)
        if (int rc = wgrad_reduce_batch_launch(red, ws_stream)) return rc;
    }
    pending.clear();
    return 0;
  };
  for (const BwdOp& op : bwd_ops) {
    if (op.segment < seg_begin || op.segment >= seg_end) continue;
    if (op.segment != pending_segment) {
      if (int rc = flush_pending()) return rc;  // the bucket's gradients are final behind these launches
      pending_segment = op.segment;
    }
    if (op.kind == BW_SUM2X2) {
      if (int rc = sum2x2_launch(dtype, ws + dfull_off, G(op.dst0), B, op.Hl, op.Wl, op.C, s, ns)) return rc;
      continue;
    }
    if (op.kind == BW_POOL) {
      const TensorD& f1 = tensors[units[conv1].a];
      if (int rc = maxpool3x3s2_bwd_launch(dtype, G(op.dA), reinterpret_cast<uint8_t*>(ws + pool_idx_off),
                                           G(op.dst0), op.acc0 ? 1 : 0, B, f1.H, f1.W, f1.C, s, ns))
        return rc;
      continue;
    }
    const Unit& u = units[op.unit];
    const long rows = (long)B * u.Ho * u.Wo;
    char* dy = ws + u.dy_off;
    if (op.kind == BW_HEAD) {
      if (int rc = nchw_to_nhwc_launch(dtype, dout, dy, B, u.Cout, u.Ho, u.Wo, u.CoutD, s, ns)) return rc;
      if (skip_w) {  // (profiling ablation without weight-gradient launches: nothing would flush it)
        if (int rc = channel_sum_nchw_launch(dout, B, u.Cout, (long)u.Ho * u.Wo,
                                             reinterpret_cast<float*>(ws + bsum_off), grads + u.bias_off, s, ns))
          return rc;
      } else {
        head_bias_pending = true;  // with the head's weight gradient, on the weight-gradient stream (flush_pending)
      }
    } else {
      int nb = 0;
      float* mean = coef_ptr(ws, u, 0);
      float* invstd = coef_ptr(ws, u, 1);
      float* k = coef_ptr(ws, u, 4);
      float* bnpart = reinterpret_cast<float*>(ws + bnpart_off);
      // ReLU mask: layers without a residual recompute it from y (identical to a > 0: the forward apply used the
      // same fp32 y*scale + shift on the same stored y, and rounding a positive fp32 value to bf16 never gives
      // zero); residual layers read the saved activation
      const bool from_y = op.mask && u.res_tensor < 0 && u.res_unit < 0;
      const void* amask = (op.mask && !from_y) ? T(u.a) : nullptr;
      const float* msc = from_y ? coef_ptr(ws, u, 2) : nullptr;
      const float* msf = from_y ? coef_ptr(ws, u, 3) : nullptr;
      if (skip_b) {
        nb = 1;
      } else if (op.fused_rows > 0) {
        nb = op.fused_rows;  // the producing data gradient already left the partial sums in bnpart
      } else if (int rc = bn_bwd_reduce_launch(dtype, G(op.dA), amask, T(u.y), mean, invstd, bnpart, &nb, rows,
                                               u.Cout, s, msc, msf, ns)) {
        return rc;
      }
      const bool sync = bn_sync_fn_ != nullptr && !skip_b;
      if (sync) {
        // synchronised statistics: dgamma / dbeta from the LOCAL sums (they are summed over ranks with the other
        // gradients), the coefficients of dy from the sums over every rank's batch
        if (int rc = bn_bwd_finalize_launch(bnpart, nb, u.Cout, rows, params_ + u.g_off, invstd, grads + u.g_off,
                                            grads + u.b_off, 0, k, s))
          return rc;
        if (int rc = bn_sync_fn_(bn_sync_ctx_, bnpart, (int64_t)nb * u.Cout * 2, (void*)s))
          return set_error(rc, "BatchNorm statistics all-reduce failed in the backward pass (%s)", u.bn_name.c_str());
        if (int rc = bn_bwd_finalize_launch(bnpart, nb, u.Cout, rows * bn_sync_world_, params_ + u.g_off, invstd, nullptr,
                                            nullptr, 0, k, s))
          return rc;
        if (int rc = bn_bwd_apply_launch(dtype, G(op.dA), amask, T(u.y), mean, invstd, k, dy,
                                         op.dres == -1 ? nullptr : G(op.dres), op.dres_acc ? 1 : 0, rows,
                                         u.Cout, s, msc, msf))
          return rc;
#ifdef D3F_PROFILING
      } else if (!skip_b && getenv("D3F_ABLATE_BWD_BN") != nullptr && op.dres == -1 && from_y) {
        // timing-only ablation (wrong gradients): the backward streaming pass of the layers without a residual replaced by
        // the coefficient-only finalize launch -- the ceiling of VERDICT r5 item 4b (dy made in the data gradient's staging)
        if (int rc = bn_bwd_finalize_launch(bnpart, nb, u.Cout, rows, params_ + u.g_off, invstd, grads + u.g_off,
                                            grads + u.b_off, 0, k, s, ns))
          return rc;
#endif
      } else if (!skip_b && bn_fused_finalize_ok(dtype, nb, u.Cout)) {
        // finalize folded into the streaming pass (bn_fused.hip)
        if (int rc = bn_bwd_finalize_apply_launch(dtype, bnpart, nb, u.Cout, rows, params_ + u.g_off, mean, invstd,
                                                  grads + u.g_off, grads + u.b_off, 0, k, G(op.dA), amask, T(u.y), dy,
                                                  op.dres == -1 ? nullptr : G(op.dres), op.dres_acc ? 1 : 0, rows, s,
                                                  msc, msf, ns, plan_nets))
          return rc;
      } else if (!skip_b) {
        if (int rc = bn_bwd_finalize_launch(bnpart, nb, u.Cout, rows, params_ + u.g_off, invstd,
                                            grads + u.g_off, grads + u.b_off, 0, k, s, ns))
          return rc;
        if (int rc = bn_bwd_apply_launch(dtype, G(op.dA), amask, T(u.y), mean, invstd, k, dy,
                                         op.dres == -1 ? nullptr : G(op.dres), op.dres_acc ? 1 : 0, rows,
                                         u.Cout, s, msc, msf, ns))
          return rc;
      }
    }
    // weight gradient (side stream), once the unit's dY exists
    if (!skip_w) {
      pending.push_back(op.unit);
      if ((int)pending.size() >= defer)
        if (int rc = flush_pending()) return rc;
    }
    // data gradient (main stream)
    if (u.need_dgrad && !skip_d && u.upfold) {
      // up-sampling folded into the weights: the gradient of the low-resolution source is a 4x4 stride-2 convolution
      // over dY written at its own resolution (it feeds the next BatchNorm backward: first), the skip tensor's
      // gradient an ordinary 3x3 data gradient
      ConvParams l = u.dgrad_lo;
      l.src0 = dy;
      l.w = ws + u.wd4_off;
      l.out0 = G(op.dst0);
      l.acc0 = op.acc0 ? 1 : 0;
      l.partial = l.splitk > 1 ? reinterpret_cast<float*>(ws + splitk_off) : nullptr;
      net_conv(l, ns);
      if (op.fuse_for_unit >= 0) {
        const Unit& uc = units[op.fuse_for_unit];
        l.bn_y = T(uc.y);
        l.bn_coef = coef_ptr(ws, uc, 0);
        l.bn_partial = reinterpret_cast<float*>(ws + bnpart_off);
        l.bn_a = (uc.res_tensor >= 0 || uc.res_unit >= 0) ? T(uc.a) : nullptr;
      }
      if (int rc = conv_igemm_launch(l, cdtype, s)) return rc;
      if (op.dst1 >= 0) {
        // the skip tensor's gradient (not read before the encoder stages; a third stream for these launches was measured
        // 1 % slower -- MFMA-bound work next to the chain's own kernels -- profiles/README.md round 2)
        ConvParams d = u.dgrad;
        d.src0 = dy;
        d.w = ws + u.wds_off;
        d.out0 = G(op.dst1);
        d.acc0 = op.acc1 ? 1 : 0;
        d.partial = d.splitk > 1 ? reinterpret_cast<float*>(ws + splitk_off) : nullptr;
        net_conv(d, ns);
        if (int rc = conv_igemm_launch(d, cdtype, s)) return rc;
      }
    } else if (u.need_dgrad && !skip_d) {
      ConvParams d = u.dgrad;
      d.src0 = dy;
      d.w = ws + u.wd_off;
      if (op.dst0_is_full_scratch) {
        d.out0 = ws + dfull_off;
        d.acc0 = 0;
      } else {
        d.out0 = G(op.dst0);
        d.acc0 = op.acc0 ? 1 : 0;
      }
      d.out1 = op.dst1 >= 0 ? G(op.dst1) : nullptr;
      d.acc1 = op.acc1 ? 1 : 0;
      d.partial = d.splitk > 1 ? reinterpret_cast<float*>(ws + splitk_off) : nullptr;
      net_conv(d, ns);
      if (d.par == 2 && !d.acc0)  // 1x1 stride 2: only even pixels receive a gradient; the others are zero
        for (int n = 0; n < nets; ++n)
          D3F_HIP(hipMemsetAsync(reinterpret_cast<char*>(d.out0) + (size_t)n * net_ws_stride, 0,
                                 (size_t)4 * d.M * d.Cout * esize(), s));
      if (op.fuse_for_unit >= 0) {
        const Unit& uc = units[op.fuse_for_unit];
        d.bn_y = T(uc.y);
        d.bn_coef = coef_ptr(ws, uc, 0);
        d.bn_partial = reinterpret_cast<float*>(ws + bnpart_off);
        d.bn_a = (uc.res_tensor >= 0 || uc.res_unit >= 0) ? T(uc.a) : nullptr;
      }
      if (int rc = conv_igemm_launch(d, cdtype, s)) return rc;
    }
  }
  if (int rc = flush_pending()) return rc;
  if (!serial && side_used) side_dirty_ = true;
  if (!serial && !join) {
    // data-parallel caller: the caller's stream (the critical path) is NOT held back.  The side stream waits for the
    // caller's stream instead (BatchNorm / bias gradients of these segments are written there), so "everything on the
    // side stream so far" == "every gradient of these segments is final": the collective waits on that, nothing else.
    D3F_HIP(hipEventRecord(ev_seg_, s));
    D3F_HIP(hipStreamWaitEvent(side_, ev_seg_, 0));
    side_dirty_ = true;
  } else if (!serial) {  // join: every gradient of these segments is final on the caller's stream
    if (int rc = backward_join(s)) return rc;
  }
  return 0;
}

// "<conv name>:y" raw conv output, ":a" post BN(+residual)+ReLU activation, ":da" gradient w.r.t. that activation
const TensorD* UnetEngine::find_export(const char* name) const {
  const std::string n(name);
  const size_t colon = n.rfind(':');
  if (colon == std::string::npos) return nullptr;
  const std::string un = n.substr(0, colon), kind = n.substr(colon + 1);
  if (kind != "y" && kind != "a" && kind != "da") return nullptr;
  for (const Unit& u : units) {
    if (u.conv_name != un) continue;
    const int tid = kind == "y" ? u.y : u.a;
    if (tid < 0) return nullptr;
    if (kind == "da") return grad_of[tid] >= 0 ? &gtensors[grad_of[tid]] : nullptr;
    return &tensors[tid];
  }
  return nullptr;
}

int UnetEngine::export_tensor(const char* name, const void* ws_, float* out_nchw, hipStream_t s) const {
  const TensorD* t = find_export(name);
  D3F_CHECK(t != nullptr, "export: no tensor named '%s' (expected '<conv name>:y|a|da')", name);
  return nhwc_to_nchw_launch(dtype, reinterpret_cast<const char*>(ws_) + t->off, out_nchw, B, t->C, t->H, t->W, t->C, s);
}

int UnetEngine::export_shape(const char* name, int32_t dims[3]) const {
  const TensorD* t = find_export(name);
  D3F_CHECK(t != nullptr, "export: no tensor named '%s' (expected '<conv name>:y|a|da')", name);
  dims[0] = t->C; dims[1] = t->H; dims[2] = t->W;
  return 0;
}

}  // namespace d3f
