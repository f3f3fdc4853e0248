"""GPU: the optional bf16 activation mode (compute_dtype="bf16": bf16 activations / packed weights,
f32 accumulation, f32 master weights, f32 statistics and weight gradients).

Stated tolerance.  Single operators, inputs pre-rounded to bf16: products are exact in f32, so the only
difference to an f32 evaluation is the final rounding of the output to bf16 (2^-9 relative per element):
rel-L2 <= 4e-3 for forward / data gradient (measured 1.7e-3), <= 1e-5 for the f32 weight gradient
(measured 1e-7).  Whole network: this BatchNorm-heavy net amplifies rounding ~100x (f32: 7e-6 forward
error from 6e-8 rounding); with 2^-8 activation rounding the measured distance to the f32 CPU oracle is
6.7e-2 / 7.5e-2 rel-L2 forward (B=4 64x64 / B=8 128x128), loss within 3e-4, flat-gradient cosine 0.80 /
0.88.  Gates = those measurements with 1.5x head-room: forward rel-L2 <= 0.11, |loss - oracle| <= 5e-4, gradient
cosine >= 0.72 (a dropped layer or a wrong operand plane is far outside).  The TIGHT bf16 gate is the layer-by-layer,
teacher-forced one at the bottom of this file (every conv output within 4.4e-3, every activation within 5.5e-3 of the
float64 oracle fed the HIP run's own inputs, at 4x64x64 and at the 16x256x256 headline shape): the end-to-end distance
above measures bf16's drift through 47 BatchNorm-normalised layers, not the kernels.  bf16 is a throughput mode, the f32
path is the parity-graded one (tests/test_gpu_unet.py, tests/test_gpu_parity_layers.py); full-size bf16 properties:
tests/test_gpu_fullsize.py."""
import pytest
import torch
import torch.nn.functional as F

from util import rel_l2, to_nchw, to_nhwc

pytestmark = pytest.mark.gpu

CASES = [
    (2, 16, 16, 64, 0, 64, 3, 1, 1, False),
    (2, 16, 16, 64, 64, 32, 3, 1, 1, True),
    (2, 16, 16, 32, 0, 16, 3, 1, 1, True),
    (2, 16, 16, 16, 0, 16, 3, 1, 1, False),
    (1, 8, 8, 256, 0, 256, 3, 1, 1, False),
    (2, 16, 16, 64, 0, 128, 3, 2, 1, False),
    (2, 16, 16, 64, 0, 128, 1, 2, 0, False),
    (2, 8, 64, 16, 0, 16, 3, 1, 1, False),        # conv_patch_kernel<bf16, 16, 16> (4 x 64 tiles, 16x16x16 bf16 MFMA)
    (1, 12, 128, 16, 0, 8, 3, 1, 1, False),       # ... two tiles per row, fewer filters than the tile
    (2, 8, 64, 32, 0, 32, 3, 1, 1, False),        # conv_patch_kernel<bf16, 32, 32> (64-byte pixels, 16x16x32 bf16 MFMA): forward + data gradient
    (1, 12, 128, 32, 0, 16, 3, 1, 1, False),      # ... 16 of the 32 filter columns real
    (3, 4, 64, 32, 0, 24, 3, 1, 1, False),        # ... a ragged filter count, one tile row per image
    (12, 64, 64, 64, 0, 64, 3, 1, 1, False),      # conv_pres_kernel<64, 64, 2, 2, 1, 2>: patch-resident layer1 form (384 workgroups), forward + data gradient
    (8, 64, 64, 64, 0, 64, 3, 1, 1, False),       # ... at 8 images (train_deep_fake's per-net batch): 256 workgroups, the lower limit of the plan's rule
    (12, 32, 32, 128, 0, 128, 3, 1, 1, False),    # conv_pres_kernel<128, 32, 4, 1, 1, 1>: layer2 form, 4 filter groups per tile
    (12, 16, 16, 256, 0, 256, 3, 1, 1, False),    # conv_pres_kernel<256, 16, 2, 1, 2, 1>: layer3 form, two k-groups summed through LDS
    (16, 8, 8, 512, 0, 512, 3, 1, 1, False),      # conv_pres_kernel<512, 8, 1, 1, 4, 2>: layer4 form, a whole 8 x 8 image per workgroup, four k-groups
    (8, 8, 8, 512, 0, 512, 3, 1, 1, False),       # ... at 8 images: 128 workgroups, the lower limit of the plan's rule for this form
    (2, 8, 64, 64, 0, 32, 3, 1, 1, False),        # data gradient 32 -> 64 channels: two 32-filter workgroups per tile (grid.y) of the bf16 32-channel patch kernel
    (1, 12, 128, 48, 0, 32, 3, 1, 1, False),      # ... 32 -> 48: the second filter half is half empty
    (2, 8, 64, 16, 0, 3, 3, 1, 1, False),         # head: data gradient through conv_patch_kernel<bf16, 16, 16, false, 8> (dY 3 -> 8 channels, staged as 16)
    (1, 12, 128, 16, 0, 3, 3, 1, 1, False),       # ... two tiles per row
    (2, 8, 64, 32, 0, 16, 3, 1, 1, True),         # conv_patch_kernel<bf16, 32, 16, UP>: forward through the up-sampling from a low-resolution patch
    (1, 12, 128, 32, 0, 8, 3, 1, 1, True),        # ... two tiles per row, 8 of 16 filter columns real
    (1, 32, 32, 128, 64, 64, 3, 1, 1, True),      # class-form weight gradient (WG_CLASS + WG_SKIP), bf16 MFMA
    (3, 12, 20, 128, 64, 128, 3, 1, 1, True),     # ... ragged class grid
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_conv_bf16(case):
    from denoising_diffusion_deep_fake_amd import ops
    B, H, W, C0, C1, Co, k, s, pd, up = case
    g = torch.Generator().manual_seed(0)
    h0, w0 = (H // 2, W // 2) if up else (H, W)
    x0 = torch.randn(B, C0, h0, w0, generator=g).bfloat16().float()
    x1 = torch.randn(B, C1, H, W, generator=g).bfloat16().float() if C1 else None
    w = (torch.randn(Co, C0 + C1, k, k, generator=g) / ((C0 + C1) * k * k) ** 0.5).bfloat16().float()
    xin = F.interpolate(x0, scale_factor=2, mode="nearest") if up else x0
    if C1:
        xin = torch.cat([xin, x1], 1)
    xin = xin.requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(xin, wr, None, s, pd)
    dy = torch.randn(y_ref.shape, generator=g).bfloat16().float()
    y_ref.backward(dy)
    d = ops.make_desc(B, H, W, C0, C1, Co, k, s, pd, up)
    s0 = to_nhwc(x0).bfloat16().cuda()
    s1 = to_nhwc(x1).bfloat16().cuda() if C1 else None
    wf, wd = ops.pack_weights(d, w.cuda(), dtype=ops.BF16)
    if Co % 8 == 0:  # (the 3-channel head's forward runs through the NCHW epilogue of the whole-network path)
        y, stats, tiles = ops.conv_forward(d, s0, s1, wf, dtype=ops.BF16, splitk=True)
        assert rel_l2(to_nchw(y.float().cpu()), y_ref) < 4e-3
        st = stats.view(tiles, (Co + 15) // 16 * 16, 2).double().sum(0).cpu()
        # statistics come from the f32 accumulators, not from the rounded outputs.  Layers with the up-sampling folded
        # into the weights multiply by bf16(w1 + w2 [+ w3 + w4]) -- one more bf16 rounding of the (pre-rounded) test
        # weights than the reference's bf16(w1) x + bf16(w2) x; with fp32 master weights both forms round once
        folded = bool(up) and bool(ops.conv_upsample_folded(d, ops.BF16))
        assert rel_l2(st[:Co, 1], (y_ref.detach().double() ** 2).sum((0, 2, 3))) < (2e-3 if folded else 1e-5)
    dyh = to_nhwc(dy, (Co + 7) // 8 * 8).bfloat16().cuda()
    dx0, dx1 = ops.conv_backward_data(d, dyh, wd, dtype=ops.BF16, splitk=True)
    want0 = xin.grad[:, :C0]
    if up:  # an up-sampled source gets its gradient at its own (low) resolution
        x0r = x0.clone().requires_grad_(True)
        F.interpolate(x0r, scale_factor=2, mode="nearest").backward(want0)
        want0 = x0r.grad
    assert rel_l2(to_nchw(dx0.float().cpu()), want0) < 4e-3
    if C1:
        assert rel_l2(to_nchw(dx1.float().cpu()), xin.grad[:, C0:]) < 4e-3
    if up and not C1 and not ops.conv_upsample_folded(d, ops.BF16):
        # the same gradient on the full-resolution contract (descriptor upsample0 = 1: what a C caller that never asked
        # for the 2x2-summed epilogue gets) -- d3f_conv_backward_data + d3f_upsample2x_backward
        dx0_full, _ = ops.conv_backward_data(d, dyh, wd, dtype=ops.BF16, splitk=True, summed=False)
        assert dx0_full.shape == dx0.shape and rel_l2(to_nchw(dx0_full.float().cpu()), want0) < 4e-3
    dw = ops.conv_backward_weight(d, dyh, s0, s1, dtype=ops.BF16)
    assert rel_l2(dw.cpu(), wr.grad) < 1e-5


@pytest.mark.parametrize("shape", [(2, 64, 96), (3, 36, 52)], ids=str)
def test_stem_weight_gradient_bf16_half_vector(shape):
    """encoder.conv1 (7x7 stride 2) in bf16 storage: 3 image channels padded to one 16-byte vector of 8.  With the real
    channel count known (<= 4) the patch kernel stages only the first four (variant 6); the gradient must equal the
    eight-channel kernel's (variant 5) bit for bit and torch's to fp32 accuracy (the operands are exact in bf16)."""
    from denoising_diffusion_deep_fake_amd import ops
    B, H, W = shape
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 3, H, W, generator=g).bfloat16().float()
    dy = torch.randn(B, 64, H // 2, W // 2, generator=g).bfloat16().float()
    want = torch.nn.grad.conv2d_weight(x, (64, 3, 7, 7), dy, stride=2, padding=3)
    xh = to_nhwc(x, 8).bfloat16().cuda()
    dyh = to_nhwc(dy).bfloat16().cuda()
    d6 = ops.make_desc(B, H, W, 8, 0, 64, 7, 2, 3, False, cin_real=3)
    d5 = ops.make_desc(B, H, W, 8, 0, 64, 7, 2, 3, False, cin_real=8)
    g6 = ops.conv_backward_weight(d6, dyh, xh, None, dtype=ops.BF16)
    g5 = ops.conv_backward_weight(d5, dyh, xh, None, dtype=ops.BF16)
    assert tuple(g6.shape) == (64, 3, 7, 7) and tuple(g5.shape) == (64, 8, 7, 7)
    assert torch.equal(g6, g5[:, :3]) and float(g5[:, 3:].abs().max()) == 0.0
    assert rel_l2(g6.cpu(), want) < 1e-5


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 48, 128)], ids=str)
def test_stem_forward_bf16_patch_kernel(shape):
    """encoder.conv1 (7x7 stride 2, 3 -> 64) in bf16 storage through conv_stem_bf16_kernel: the image's 3 channels sit in
    one 16-byte vector of 8, the kernel stages the first four (the packed weights of channels 3 .. 7 are zero) and
    contracts four taps per v_mfma_f32_32x32x16_bf16.  Output and statistics against torch on the bf16-rounded operands;
    finite garbage in the pad channels of the input must not change a bit."""
    from denoising_diffusion_deep_fake_amd import ops
    B, H, W = shape
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, 3, H, W, generator=g).bfloat16().float()
    w = (torch.randn(64, 3, 7, 7, generator=g) / (3 * 49) ** 0.5).bfloat16().float()
    y_ref = F.conv2d(x, w, None, 2, 3)
    d = ops.make_desc(B, H, W, 8, 0, 64, 7, 2, 3, False, cin_real=3)
    wf, _ = ops.pack_weights(d, w.cuda(), dtype=ops.BF16)
    xh = to_nhwc(x, 8).bfloat16().cuda()
    y, stats, tiles = ops.conv_forward(d, xh, None, wf, dtype=ops.BF16)
    assert tiles == B * (H // 2 // 8) * (W // 2 // 32)          # one statistics row per 8 x 32 tile: the patch kernel ran
    assert rel_l2(to_nchw(y.float().cpu()), y_ref) < 4e-3
    st = stats.view(tiles, 64, 2).double().sum(0).cpu()
    assert rel_l2(st[:, 1], (y_ref.double() ** 2).sum((0, 2, 3))) < 1e-5
    assert (st[:, 0] - y_ref.double().sum((0, 2, 3))).abs().max() < 1e-3 * (1 + st[:, 1].sqrt().max())
    dirty = xh.clone()
    dirty[..., 3:] = torch.randn(dirty[..., 3:].shape, generator=g).bfloat16().cuda()
    y2, stats2, _ = ops.conv_forward(d, dirty, None, wf, dtype=ops.BF16)
    assert torch.equal(y2, y) and torch.equal(stats2, stats)


def test_unet_bf16_training_step():
    import oracle
    from denoising_diffusion_deep_fake_amd import Unet, ops
    torch.manual_seed(1)
    ref = oracle.Unet("resnet34", None, 3, 3, None).train()
    net = Unet("resnet34", None, 3, 3, None, compute_dtype="bf16")
    net.load_state_dict(ref.state_dict())
    net = net.cuda().train()
    B, S = 4, 64
    x = oracle.synthetic_face_crops(B, S, seed=11)
    g = torch.Generator().manual_seed(5)
    noise = torch.randn(x.shape, generator=g)
    r = torch.rand(B, generator=g) * 0.5 + 0.05
    noisy = oracle.step_oracle.blend_with_given_noise(x, noise, r)
    crit = oracle.MseStructuralSimilarityLoss(-1.0, 1.0)
    pr = ref(noisy)
    lr = crit(pr, x)
    lr.backward()
    pred = net(noisy.cuda())
    lossv, gp = ops.mse_ssim_loss(pred.detach(), x.cuda())
    pred.backward(gp)
    assert pred.dtype == torch.float32            # boundary tensors stay f32
    # gates = what was measured (6.7e-2 / 7.5e-2 forward, |dloss| 3e-4, cosine 0.80 / 0.88) with 1.5x head-room
    assert rel_l2(pred, pr) < 0.11
    assert abs(lossv[0].item() - lr.item()) < 5e-4
    g32 = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    cos = F.cosine_similarity(net.flat_grads.cpu().double(), g32.double(), dim=0).item()
    assert cos > 0.72, cos
    assert torch.isfinite(net.flat_grads).all()
    net.eval()
    ref.eval()
    with torch.no_grad():
        assert rel_l2(net(noisy.cuda()), ref(noisy)) < 0.05


class _TeacherForcedReLU(torch.nn.Module):
    """records the oracle's own ReLU output, hands the NEXT layer the HIP run's activation instead (queue in unit order)"""

    def __init__(self, queue, recorded):
        super().__init__()
        self.queue, self.recorded = queue, recorded

    def forward(self, x):
        self.recorded.append(F.relu(x))
        return self.queue.pop(0).to(x.dtype)


@pytest.mark.timeout(900)
# (8, 256, 256): train_deep_fake's per-net batch on its OWN plan (swap mode, the sequential loop): since round 6 the
# patch-resident convolution takes layer1-3 from 256 workgroups and layer4 from 128 (conv_pres_applies)
@pytest.mark.parametrize("shape", [(4, 64, 64), (16, 256, 256), (16, 128, 128), (2, 448, 448), (8, 256, 256)],
                         ids=["4x64x64", "headline_16x256x256", "config1_16x128x128", "authors_2x448x448", "paired_8x256x256"])
def test_bf16_every_layer_teacher_forced(shape):
    """The bf16 mode layer by layer, WITHOUT the drift that makes the end-to-end distance to an fp32 run ~7e-2: the float64
    oracle is fed the HIP run's own (bf16-valued) activation in front of every layer, so each comparison sees one layer's
    arithmetic only -- bf16 rounding of the fp32 master weights (and of the pre-summed weights of the folded decoder
    layers), fp32 accumulation, one rounding of the output.  Measured on MI355X (4x64x64 and 16x256x256 alike): conv
    outputs <= 2.9e-3 rel-L2, activations <= 3.7e-3; gates = 1.5x that (4.4e-3 / 5.5e-3).  A wrong tap, a dropped channel tile, a BatchNorm
    coefficient from the wrong partial row or a stale packed weight is O(1e-1 .. 1)."""
    import oracle
    from oracle.pinned import conv_outputs, swap_relus, unit_names
    from denoising_diffusion_deep_fake_amd import Unet
    B, H, W = shape
    torch.manual_seed(3)
    ref = oracle.Unet("resnet34", None, 3, 3, None).train()
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
    net = Unet("resnet34", None, 3, 3, None, compute_dtype="bf16")
    net.load_state_dict(ref.state_dict())
    net = net.cuda().train()
    x = oracle.synthetic_face_crops(B, (H, W), seed=21)
    with torch.no_grad():
        net(x.cuda())
    names = unit_names()
    ds_names = [f"encoder.layer{li}.0.downsample.0" for li in (2, 3, 4)]
    hip_y = {n: net.export_activation(n + ":y").cpu() for n in names + ds_names}
    hip_a = {n: net.export_activation(n + ":a").cpu() for n in names}
    del net
    torch.cuda.empty_cache()
    ref64 = ref.double()
    chan = {n: dict(ref64.named_modules())[n].out_channels for n in names + ds_names}
    queue = [hip_a[n][:, :chan[n]] for n in names]
    recorded = []
    swap_relus(ref64, lambda: _TeacherForcedReLU(queue, recorded))
    store, hooks = conv_outputs(ref64)
    with torch.no_grad():
        ref64(x.double())
    for h in hooks:
        h.remove()
    assert not queue and len(recorded) == len(names)
    worst_y = worst_a = 0.0
    for n in names + ds_names:
        e = rel_l2(hip_y[n][:, :chan[n]], store[n])
        assert e < 4.4e-3, ("y", n, e)
        worst_y = max(worst_y, e)
    for n, a64 in zip(names, recorded):
        e = rel_l2(hip_a[n][:, :chan[n]], a64)
        assert e < 5.5e-3, ("a", n, e)
        worst_a = max(worst_a, e)
    print(f"bf16 teacher-forced {shape}: worst conv output {worst_y:.2e}, worst activation {worst_a:.2e}")


@pytest.mark.timeout(900)
def test_bf16_eval_every_layer_teacher_forced_at_the_headline_batch():
    """EVAL mode in bf16 storage at 16 x 256 x 256, layer by layer: the eval forward runs the round-5 bf16 kernels with their
    FUSED epilogues -- conv_pres_kernel (folded BatchNorm + residual + ReLU on the BasicBlock conv2 layers), the bf16 patch
    kernels and conv_stem_bf16_kernel -- which no train-mode gate touches.  The float64 oracle (eval mode, the same running
    statistics) is fed the HIP run's own activation in front of every layer, so every comparison sees one layer's
    conv + folded BatchNorm (+ residual) + ReLU: one bf16 rounding of the output (train mode rounds y first: 3.7e-3; here
    2^-9 once).  Measured on MI355X: worst activation 2.63e-3 (decoder.blocks.4.conv2.0), head output 1.60e-3; gates = x 1.5
    (4.0e-3 / 2.4e-3).  A missing residual, ReLU or a wrong coefficient row is O(1).
    Reference path: d3f/train_deep_fake/lit_module.py:259-270 (eval forward of predict_fake)."""
    import oracle
    from oracle.pinned import swap_relus, unit_names
    from denoising_diffusion_deep_fake_amd import Unet
    B, H, W = 16, 256, 256
    torch.manual_seed(13)
    ref = oracle.Unet("resnet34", None, 3, 3, None)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
    ref.eval()
    net = Unet("resnet34", None, 3, 3, None, compute_dtype="bf16")
    net.load_state_dict(ref.state_dict())
    net = net.cuda().eval()
    x = oracle.synthetic_face_crops(B, (H, W), seed=23)
    with torch.no_grad():
        out_hip = net(x.cuda()).cpu()
    names = unit_names()
    chan = {n: dict(ref.named_modules())[n].out_channels for n in names}
    hip_a = {n: net.export_activation(n + ":a").cpu()[:, :chan[n]] for n in names}
    del net
    torch.cuda.empty_cache()
    ref64 = ref.double()
    queue = [hip_a[n] for n in names]
    recorded = []
    swap_relus(ref64, lambda: _TeacherForcedReLU(queue, recorded))
    with torch.no_grad():
        out64 = ref64(x.double())
    assert not queue and len(recorded) == len(names)
    worst = ("", 0.0)
    for n, a64 in zip(names, recorded):
        e = rel_l2(hip_a[n], a64)
        worst = max(worst, (n, e), key=lambda t: t[1])
        assert e < 4.0e-3, ("a", n, e)
    # the head sees the HIP run's last activation too: fp32 output of a bf16 conv on bf16 operands
    e_out = rel_l2(out_hip, out64)
    print(f"bf16 eval teacher-forced {(B, H, W)}: worst activation {worst[1]:.2e} ({worst[0]}), head output {e_out:.2e}")
    assert e_out < 2.4e-3, e_out


# gates of the teacher-forced BACKWARD test = measured on MI355X x 1.5 (gpurun_out/r05_a, worst tensor per class at
# 4x64x64 / 16x128x128 / 16x256x256): activation gradients 3.79 / 3.84 / 3.86e-3 (decoder.blocks.4.conv1.0); conv weight
# gradients 4.56 / 5.27 / 8.92e-3 (encoder.layer1.1.conv1.weight: a sum over 65 536 pixels of products whose dy factor
# carries 2^-9 of bf16 rounding -- the sum itself cancels to ~1/8 of its terms' root-sum-square at this batch, so the
# RELATIVE error grows with the pixel count); BatchNorm affine / bias gradients 5.09 / 4.22 / 4.47e-3
BF16_BWD_TOL_DA = 5.8e-3
BF16_BWD_TOL_W = 1.35e-2
BF16_BWD_TOL_BN = 7.7e-3
BF16_BWD_TOL_W_PAIR_DATA = 3.1e-2   # (8, 256, 256) on the paired case's data: measured 2.03e-2 x 1.5, see the test


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("shape", [(4, 64, 64), (16, 128, 128), (16, 256, 256), (8, 256, 256, "pair"), (2, 448, 448)],
                         ids=["4x64x64", "config1_16x128x128", "headline_16x256x256", "paired_8x256x256_network_1_of_a_pair",
                              "authors_2x448x448"])
def test_bf16_every_layer_backward_teacher_forced(shape):
    """The bf16 engine plan's BACKWARD pass tensor by tensor (it differs from the fp32 plan: at 16 x 256 x 256 the
    patch-resident conv_pres_kernel for 35 data gradients, the bf16 patch kernels incl. the 2x2-summed data gradient of
    decoder block 4 conv1 with its fused BatchNorm partial sums, conv_wgrad_patch_bf16_kernel for 22 weight gradients, the
    two-chunk bf16 tap-parallel loop; at the smaller shapes 64x32 k-split tiles and bf16 split-K; half-vector stem weight
    gradient, bf16 `bn_fused` backward, no Winograd -- `d3f_unet_plan_counts`, tests/test_cpu_lib.py).  The float64
    oracle is teacher-forced in BOTH directions (oracle/pinned.py:teacher_forced_backward): the forward sees the HIP run's
    bf16 activation in front of every layer with ReLU / max-pool decisions pinned to it, and the gradient arriving at
    every unit's activation is replaced by the HIP run's exported ":da" after the oracle's own value was recorded.  Every
    ":da" and each of the 143 parameter gradients therefore compares ONE layer's data gradient + BatchNorm backward +
    weight gradient on identical operands: what is left is bf16 rounding of the stored dy / y / packed weights
    (2^-9 per element).  A missed accumulate, a wrong bucket edge, tap or slab is O(1e-1 .. 1).
    Replaces autograd through /root/reference/d3f/train_denoiser/lit_module.py:117-119 in the bf16 mode (BASELINE configs[2]).
    (8, 256, 256, "pair"): BASELINE configs[3]'s per-net batch the way the trainer runs it since round 6 -- TWO bf16 networks
    stepped as one set of launches (UnetPair, d3f/train_deep_fake/lit_module.py:142-181); the tensors checked are those of
    network 1, the one whose workgroups add the second network's offsets to every pointer.  (2, 448, 448): the authors'
    resolution, ragged against every tile."""
    import gc

    import oracle
    from oracle.pinned import teacher_forced_backward, unit_names
    from denoising_diffusion_deep_fake_amd import Unet, UnetPair, ops
    B, H, W = shape[:3]
    paired = len(shape) > 3

    def oracle_net(seed):
        torch.manual_seed(seed)
        ref = oracle.Unet("resnet34", None, 3, 3, None).train()
        with torch.no_grad():
            for m in ref.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.uniform_(0.5, 1.5)
                    m.bias.normal_(0, 0.1)
            ref.segmentation_head[0].bias.normal_(0, 0.1)
        return ref

    def hip_net(ref):
        net = Unet("resnet34", None, 3, 3, None, compute_dtype="bf16")
        net.load_state_dict(ref.state_dict())
        return net.cuda().train()
    ref = oracle_net(3)
    net = hip_net(ref)
    x = oracle.synthetic_face_crops(B, (H, W), seed=21)
    tgt = oracle.synthetic_face_crops(B, (H, W), seed=22).cuda()
    names = unit_names()
    if paired:  # network 0: another network on another batch; `ref` / `net` / `x` are network 1
        other = hip_net(oracle_net(4))
        pair = UnetPair(other, net)
        p0, pred = pair(oracle.synthetic_face_crops(B, (H, W), seed=31).cuda(), x.cuda())
        _, gout = ops.mse_ssim_loss(pred.detach(), tgt)
        torch.autograd.backward([p0, pred], [ops.mse_ssim_loss(p0.detach(), tgt)[1], gout])
        export = lambda n: pair.export_activation(1, n)  # noqa: E731
    else:
        pred = net(x.cuda())
        _, gout = ops.mse_ssim_loss(pred.detach(), tgt)
        pred.backward(gout)
        export = net.export_activation
    chan = {n: dict(ref.named_modules())[n].out_channels for n in names}
    hip_a = {n: export(n + ":a").cpu()[:, :chan[n]] for n in names}
    hip_da = {n: export(n + ":da").cpu()[:, :chan[n]] for n in names}
    hip_grads = {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}
    gout = gout.cpu()
    del net, pred, export
    if paired:
        del pair, other, p0
    torch.cuda.empty_cache()
    gc.collect()
    dact, grads = teacher_forced_backward(ref, x, gout, hip_a, hip_da, names)
    assert set(dact) == set(names) and set(grads) == set(hip_grads)
    worst = {"da": ("", 0.0), "w": ("", 0.0), "bn": ("", 0.0)}
    bad = []
    # the paired case's data: ONE tensor (encoder.layer1.0.conv1.weight) measures 2.03e-2 -- on network 1 of the pair, on the
    # same network stepped alone on the pair's plan (2.03e-2, bit-identical) and on its own 8-image plan (1.80e-2); every other
    # conv weight gradient of that run is <= 4.4e-3, and another network / batch measures 4.8e-3 for the same tensor
    # (profiles/r06_bf16_wgrad_conditioning_probe.txt: a property of bf16 storage on this batch, not of a kernel or plan;
    # neither bf16-rounding the oracle's weights nor the sum's root-sum-square conditioning, 1.5, accounts for it)
    tol_w = BF16_BWD_TOL_W_PAIR_DATA if paired else BF16_BWD_TOL_W
    for n in names:
        e = rel_l2(hip_da[n], dact[n])
        worst["da"] = max(worst["da"], (n, e), key=lambda t: t[1])
        if not e < BF16_BWD_TOL_DA:
            bad.append(("da", n, e))
    for k, g in grads.items():
        e = rel_l2(hip_grads[k], g)
        kind = "w" if g.dim() == 4 else "bn"
        worst[kind] = max(worst[kind], (k, e), key=lambda t: t[1])
        if not e < (tol_w if kind == "w" else BF16_BWD_TOL_BN):
            bad.append((kind, k, e))
    print(f"bf16 teacher-forced backward {shape}: worst activation gradient {worst['da'][1]:.2e} ({worst['da'][0]}), "
          f"worst conv weight gradient {worst['w'][1]:.2e} ({worst['w'][0]}), worst BatchNorm / bias gradient "
          f"{worst['bn'][1]:.2e} ({worst['bn'][0]})")
    assert not bad, bad
