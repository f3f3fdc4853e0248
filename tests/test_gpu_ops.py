"""GPU parity of the single HIP operators (through the C ABI) against plain fp32 PyTorch on CPU.

Tolerances (fp32 path): the MFMA f32 instructions are exact-f32 fmaf chains, so differences come
only from summation order: rel-L2 <= 1e-5 for contractions, <= 1e-6 for pointwise kernels.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import max_rel, rel_l2, to_nchw, to_nhwc

pytestmark = pytest.mark.gpu

TOL_CONV = 1e-5


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from denoising_diffusion_deep_fake_amd import ops as o
    return o


def _conv_case(ops, B, H, W, C0, C1, Cout, k, stride, pad, up=False, cin_real=None, seed=0, splitk=False,
               dtype=0):
    g = torch.Generator().manual_seed(seed)
    cin = C0 + C1
    cr = cin_real or cin
    h0, w0 = (H // 2, W // 2) if up else (H, W)
    x0 = torch.randn(B, C0 if C1 or not cin_real else cr, h0, w0, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    w = torch.randn(Cout, cr, k, k, generator=g) / (cr * k * k) ** 0.5
    # ---- reference -------------------------------------------------------------------------
    xin = F.interpolate(x0, scale_factor=2, mode="nearest") if up else x0
    if x1 is not None:
        xin = torch.cat([xin, x1], 1)
    xin = xin.requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(xin, wr, None, stride, pad)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    # ---- HIP -------------------------------------------------------------------------------
    d = ops.make_desc(B, H, W, C0, C1, Cout, k, stride, pad, up, cr)
    s0 = to_nhwc(x0, C0).cuda()
    s1 = to_nhwc(x1).cuda() if x1 is not None else None
    wf, wd = ops.pack_weights(d, w.cuda(), dtype)
    if Cout % 4 == 0:  # the 3-channel head runs through the NCHW epilogue of the whole-network path
        y, stats, tiles = ops.conv_forward(d, s0, s1, wf, dtype, splitk=splitk)
        torch.cuda.synchronize()
        y_h = to_nchw(y.cpu())
        assert rel_l2(y_h, y_ref) < TOL_CONV, ("fwd", rel_l2(y_h, y_ref))
        # statistics partials: per-channel sum / sumsq of the conv output
        cpad = (Cout + 15) // 16 * 16
        st = stats.view(tiles, cpad, 2).double().sum(0).cpu()
        ref_s1 = y_ref.detach().double().sum((0, 2, 3))
        ref_s2 = (y_ref.detach().double() ** 2).sum((0, 2, 3))
        assert max_rel(st[:Cout, 1], ref_s2) < 1e-5
        assert (st[:Cout, 0] - ref_s1).abs().max() < 1e-3 * (1 + ref_s2.sqrt().max())
    # weight gradient
    co_pad = (Cout + 3) // 4 * 4
    dy_h = to_nhwc(dy, co_pad).cuda()
    dw = ops.conv_backward_weight(d, dy_h, s0, s1, dtype)
    assert rel_l2(dw.cpu(), wr.grad) < TOL_CONV, ("wgrad", rel_l2(dw.cpu(), wr.grad))
    # data gradient (the padded first conv never needs one)
    if cin_real is None:
        dx0, dx1 = ops.conv_backward_data(d, dy_h, wd, dtype, splitk=splitk)
        dxr = xin.grad
        if up:
            # dx0 = gradient of the LOW-resolution source (4x4 stride-2 kernel with pre-summed weights, or the
            # full-resolution gradient reduced 2x2): autograd reference of the up-sampling
            x0r = x0.clone().requires_grad_(True)
            F.interpolate(x0r, scale_factor=2, mode="nearest").backward(dxr[:, :C0])
            assert rel_l2(to_nchw(dx0.cpu()), x0r.grad) < TOL_CONV, ("dgrad0 (low)", rel_l2(to_nchw(dx0.cpu()), x0r.grad))
            # the 2x2 sum kernel on its own (backward of nearest up-sampling)
            full = to_nhwc(dxr[:, :C0].contiguous(), C0).cuda()
            assert rel_l2(to_nchw(ops.upsample2x_backward(full).cpu()), x0r.grad) < 1e-6
        else:
            assert rel_l2(to_nchw(dx0.cpu()), dxr[:, :C0]) < TOL_CONV, ("dgrad0", rel_l2(to_nchw(dx0.cpu()), dxr[:, :C0]))
        if C1:
            assert rel_l2(to_nchw(dx1.cpu()), dxr[:, C0:]) < TOL_CONV
        # accumulate flag
        base = torch.randn(dx0.shape, generator=g).cuda()
        acc, _ = ops.conv_backward_data(d, dy_h, wd, dtype, dx0=base.clone(), dx1=dx1, acc0=True, splitk=splitk)
        assert rel_l2(acc.cpu(), (base + dx0).cpu()) < 1e-6


CASES = [
    # B, H, W, C0, C1, Cout, k, stride, pad, up
    (2, 16, 16, 64, 0, 64, 3, 1, 1, False),      # layer1 block conv (64x64 tile)
    (2, 16, 16, 64, 0, 128, 3, 2, 1, False),     # stride-2 block entry
    (2, 16, 16, 64, 0, 128, 1, 2, 0, False),     # 1x1 stride-2 downsample
    (3, 10, 10, 32, 0, 64, 3, 1, 1, False),      # ragged M (300 rows)
    (1, 8, 8, 256, 0, 256, 3, 1, 1, False),      # deep layer, long K
    (2, 16, 16, 64, 64, 32, 3, 1, 1, True),      # decoder: upsample + concat -> 32 (256x32 tile)
    (2, 16, 16, 32, 0, 16, 3, 1, 1, True),       # decoder block 4 conv1 (256x16 tile, 16x16 MFMA)
    (2, 16, 16, 16, 0, 16, 3, 1, 1, False),      # small-channel K path
    (1, 32, 32, 128, 64, 64, 3, 1, 1, True),     # decoder block 2
    (2, 24, 40, 16, 0, 3, 3, 1, 1, False),       # head shape (3 outputs), ragged 8x16 patch tiles
    (1, 24, 40, 64, 64, 32, 3, 1, 1, True),      # decoder 3 conv1 with ragged patch tiles
    (2, 8, 64, 16, 0, 16, 3, 1, 1, False),       # full-resolution 16-channel layer: conv_patch_kernel (4x64 tiles)
    (1, 12, 128, 16, 0, 8, 3, 1, 1, False),      # conv_patch_kernel, two tiles per row, fewer filters than the tile
    (2, 8, 64, 16, 0, 3, 3, 1, 1, False),        # head: data gradient through conv_patch_kernel<4, 16> (dY 3 -> 4 channels)
    (1, 12, 128, 16, 0, 3, 3, 1, 1, False),      # ... two tiles per row
    # weight gradient in output-parity-class form (WG_CLASS + WG_SKIP passes, conv_wgrad.hip): ragged class grid
    # (3 x 6 x 10 = 180 low-resolution pixels: partial last chunk, a chunk = 3 rows + 2 columns of the grid) ...
    (3, 12, 20, 128, 64, 128, 3, 1, 1, True),
    (2, 16, 16, 64, 0, 64, 3, 1, 1, True),       # ... and a layer with no skip tensor (WG_CLASS alone)
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_conv_fwd_dgrad_wgrad(ops, case):
    _conv_case(ops, *case)


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_conv_splitk_paths(ops, case):
    # same shapes with a workspace: the planner splits the K loop wherever M x Cout is small
    _conv_case(ops, *case, splitk=True)


@pytest.mark.parametrize("splitk", [False, True])
@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_conv_f32x3(ops, case, splitk):
    """D3F_F32X3: fp32 tensors, every product formed from six bf16 MFMAs over an exact 3-way split of both
    operands.  Same tolerance as the fp32-MFMA path (the dropped terms are below fp32 rounding)."""
    _conv_case(ops, *case, splitk=splitk, dtype=ops.F32X3)


def test_conv_f32x3_first_layer_and_large_tiles(ops):
    _conv_case(ops, 2, 32, 32, 4, 0, 64, 7, 2, 3, False, cin_real=3, dtype=ops.F32X3)
    _conv_case(ops, 4, 128, 128, 32, 0, 128, 3, 1, 1, False, dtype=ops.F32X3)
    _conv_case(ops, 4, 128, 128, 32, 0, 64, 3, 1, 1, False, seed=1, dtype=ops.F32X3)


def test_conv_first_layer_7x7(ops):
    # encoder.conv1: 3 real input channels padded to 4, 7x7 stride 2
    _conv_case(ops, 2, 32, 32, 4, 0, 64, 7, 2, 3, False, cin_real=3)
    # wide enough for conv_stem_kernel (8 x 32 output tiles): image borders on all four sides of one / several tiles
    _conv_case(ops, 2, 32, 64, 4, 0, 64, 7, 2, 3, False, cin_real=3, seed=2)
    _conv_case(ops, 1, 48, 128, 4, 0, 64, 7, 2, 3, False, cin_real=3, seed=3)


def test_conv_large_tiles(ops):
    # enough rows for the 128x128 and 128x64 tile configurations
    _conv_case(ops, 4, 128, 128, 32, 0, 128, 3, 1, 1, False)
    _conv_case(ops, 4, 128, 128, 32, 0, 64, 3, 1, 1, False, seed=1)


# Winograd F(2x2, 3x3) forward on its own (d3f_conv_winograd_*; the whole-network plan takes it from 256 workgroups of
# 16x16 pixels x 64 filters up): B, H, W, Cin, Cout, applies
WINOGRAD_CASES = [
    (1, 16, 16, 16, 64, False),      # one workgroup, one 16-channel chunk: every image border inside one patch
    (2, 32, 48, 48, 128, False),     # odd chunk count (3), two filter blocks, non-square
    (4, 128, 128, 64, 64, True),     # exactly 256 workgroups: the first shape the plan takes
    (4, 128, 112, 64, 64, False),    # 224 workgroups, just below: the plan keeps the implicit GEMM
    (64, 32, 32, 128, 128, True),    # layer2 at 256x256, bs 64 (BASELINE configs[4]): 512 workgroups, 8 chunks
]


@pytest.mark.parametrize("case", WINOGRAD_CASES, ids=[str(c) for c in WINOGRAD_CASES])
def test_conv_winograd_forward(ops, case):
    """conv2d of a BasicBlock (d3f/train_denoiser/lit_module.py:117) through conv_winograd_kernel alone: raw output +
    BatchNorm statistics rows (train mode) and the folded-BatchNorm epilogue with / without residual and ReLU (eval
    mode), held to the same 1e-5 as every other contraction; the plan's selection rule at its threshold."""
    B, H, W, Cin, Cout, applies = case
    g = torch.Generator().manual_seed(Cin + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    y_ref = F.conv2d(x, w, None, 1, 1)
    d = ops.make_desc(B, H, W, Cin, 0, Cout, 3, 1, 1)
    assert ops.conv_winograd_applies(d) == applies
    xs = to_nhwc(x).cuda()
    u = ops.conv_winograd_pack(d, w.cuda())
    y, stats, tiles = ops.conv_winograd_forward(d, xs, u)
    torch.cuda.synchronize()
    assert tiles == B * (H // 16) * (W // 16)
    assert rel_l2(to_nchw(y.cpu()), y_ref) < TOL_CONV, ("winograd fwd", rel_l2(to_nchw(y.cpu()), y_ref))
    st = stats.view(tiles, -1, 2).double().sum(0).cpu()
    ref_s1, ref_s2 = y_ref.double().sum((0, 2, 3)), (y_ref.double() ** 2).sum((0, 2, 3))
    assert max_rel(st[:Cout, 1], ref_s2) < 1e-5
    assert (st[:Cout, 0] - ref_s1).abs().max() < 1e-3 * (1 + ref_s2.sqrt().max())
    # the implicit GEMM (what d3f_conv_forward runs, and what the plan keeps below the threshold) on the same operands
    wf, _ = ops.pack_weights(d, w.cuda(), dgrad=False)
    y_ig, _, _ = ops.conv_forward(d, xs, None, wf)
    assert rel_l2(to_nchw(y_ig.cpu()), y_ref) < TOL_CONV
    assert rel_l2(y.cpu(), y_ig.cpu()) < TOL_CONV
    # eval epilogue: relu?(conv * scale + shift + residual?)
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.3
    res = torch.randn(B, Cout, H, W, generator=g)
    for with_res, relu in ((False, False), (False, True), (True, True), (True, False)):
        ref = y_ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        if with_res:
            ref = ref + res
        if relu:
            ref = ref.clamp_min(0)
        out, st2, _ = ops.conv_winograd_forward(d, xs, u, scale=scale.cuda(), shift=shift.cuda(),
                                                residual=to_nhwc(res).cuda() if with_res else None, relu=relu)
        assert st2 is None
        assert rel_l2(to_nchw(out.cpu()), ref) < TOL_CONV, ("winograd eval", with_res, relu)
        if relu:
            assert (out >= 0).all()


def test_conv_winograd_rejects_shapes_it_cannot_run(ops):
    from denoising_diffusion_deep_fake_amd._lib import D3FError
    for bad in (ops.make_desc(1, 16, 24, 16, 0, 64, 3, 1, 1),     # width not a multiple of 16
                ops.make_desc(1, 16, 16, 16, 0, 32, 3, 1, 1),     # filters not a multiple of 64
                ops.make_desc(1, 16, 16, 8, 0, 64, 3, 1, 1),      # channels not a multiple of 16
                ops.make_desc(1, 16, 16, 16, 0, 64, 3, 2, 1),     # stride 2
                ops.make_desc(1, 16, 16, 16, 16, 64, 3, 1, 1)):   # concatenated source
        assert not ops.conv_winograd_applies(bad)
        with pytest.raises(D3FError):
            ops.conv_winograd_pack(bad, torch.zeros(bad.Cout, bad.CinReal, 3, 3, device="cuda"))
    assert not ops.conv_winograd_applies(ops.make_desc(4, 128, 128, 64, 0, 64, 3, 1, 1), ops.BF16)
    # residual / ReLU without the eval epilogue's scale and shift
    d = ops.make_desc(1, 16, 16, 16, 0, 64, 3, 1, 1)
    u = ops.conv_winograd_pack(d, torch.zeros(64, 16, 3, 3, device="cuda"))
    with pytest.raises(D3FError):
        ops.conv_winograd_forward(d, torch.zeros(1, 16, 16, 16, device="cuda"), u, relu=True)


def test_batchnorm_train_fwd_bwd(ops):
    g = torch.Generator().manual_seed(3)
    B, C, H, W = 4, 64, 12, 12
    x = torch.randn(B, 32, H, W, generator=g)
    w = torch.randn(C, 32, 3, 3, generator=g) * 0.1
    res = torch.randn(B, C, H, W, generator=g)
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.1
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    bn.train()
    wr = w.clone()
    y_ref = F.conv2d(x, wr, None, 1, 1).requires_grad_(True)
    resr = res.clone().requires_grad_(True)
    a_ref = F.relu(bn(y_ref) + resr)
    dA = torch.randn(a_ref.shape, generator=g)
    a_ref.backward(dA)

    d = ops.make_desc(B, H, W, 32, 0, C, 3, 1, 1)
    wf, _ = ops.pack_weights(d, w.cuda(), dgrad=False)
    y, stats, tiles = ops.conv_forward(d, to_nhwc(x).cuda(), None, wf)
    rm = torch.zeros(C).cuda()
    rv = torch.ones(C).cuda()
    coef = ops.bn_finalize(stats, tiles, C, B * H * W, gamma.cuda(), beta.cuda(), rm, rv)
    a = ops.bn_apply(y, coef, residual=to_nhwc(res).cuda(), relu=True)
    assert rel_l2(to_nchw(a.cpu()), a_ref) < 1e-5
    assert rel_l2(rm.cpu(), bn.running_mean) < 1e-5
    assert rel_l2(rv.cpu(), bn.running_var) < 1e-5
    dy, dres, dgamma, dbeta = ops.bn_backward(to_nhwc(dA).cuda(), a, y, coef, gamma.cuda(), want_dres=True)
    assert rel_l2(to_nchw(dy.cpu()), y_ref.grad) < 2e-5
    assert rel_l2(to_nchw(dres.cpu()), resr.grad) < 1e-6
    assert rel_l2(dgamma.cpu(), bn.weight.grad) < 2e-5
    assert rel_l2(dbeta.cpu(), bn.bias.grad) < 2e-5


def test_maxpool_fwd_bwd(ops):
    g = torch.Generator().manual_seed(4)
    x = F.relu(torch.randn(2, 8, 12, 16, generator=g))  # post-ReLU: many exact ties at 0
    x[0, 0, 2:5, 2:5] = 1.25                               # exact positive ties too
    xr = x.clone().requires_grad_(True)
    y_ref = F.max_pool2d(xr, 3, 2, 1)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    out, idx = ops.maxpool_forward(to_nhwc(x).cuda())
    assert torch.equal(to_nchw(out.cpu()), y_ref.detach())
    din = ops.maxpool_backward(to_nhwc(dy).cuda(), idx, 12, 16)
    # ATen's tie order (first maximum in scan order) is reproduced; a mis-routed tie would be an
    # O(1) error.  Overlapping windows add in a different order -> float tolerance, not bitwise.
    torch.testing.assert_close(to_nchw(din.cpu()), xr.grad, rtol=1e-6, atol=1e-6)
    base = torch.ones_like(din)
    acc = ops.maxpool_backward(to_nhwc(dy).cuda(), idx, 12, 16, din=base.clone())
    torch.testing.assert_close(acc.cpu(), (base + din).cpu())


@pytest.mark.parametrize("shape,bf16", [((3, 24, 20, 36), False), ((2, 64, 32, 64), False), ((2, 24, 16, 40), True),
                                        ((1, 64, 8, 96), True)], ids=str)
def test_maxpool_backward_shapes(ops, shape, bf16):
    """channel counts that are no power of two (the column decode divides by C / 4 or C / 8 through a multiply-high),
    rows wider than one workgroup, both storage types, with and without accumulation into an existing gradient"""
    B, C, H, W = shape
    dt = ops.BF16 if bf16 else ops.F32
    tdt = torch.bfloat16 if bf16 else torch.float32
    g = torch.Generator().manual_seed(11)
    x = F.relu(torch.randn(B, C, H, W, generator=g))
    dy = torch.randn(B, C, H // 2, W // 2, generator=g)
    if bf16:
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    y_ref = F.max_pool2d(xr, 3, 2, 1)
    y_ref.backward(dy)
    out, idx = ops.maxpool_forward(to_nhwc(x).to(tdt).cuda(), dtype=dt)
    assert torch.equal(to_nchw(out.float().cpu()), y_ref.detach())
    din = ops.maxpool_backward(to_nhwc(dy).to(tdt).cuda(), idx, H, W, dtype=dt)
    tol = dict(rtol=1e-2, atol=1e-2) if bf16 else dict(rtol=1e-6, atol=1e-6)  # bf16: the sum of up to 4 windows is rounded
    torch.testing.assert_close(to_nchw(din.float().cpu()), xr.grad, **tol)
    base = torch.full_like(din, 0.5)
    acc = ops.maxpool_backward(to_nhwc(dy).to(tdt).cuda(), idx, H, W, dtype=dt, din=base.clone())
    torch.testing.assert_close(acc.float().cpu(), (din.float() + 0.5).cpu(), **tol)


def test_noise_blend_matches_golden(ops, golden_dir):
    g = np.load(golden_dir / "blend.npz")
    x = torch.from_numpy(g["x"])
    # sampler: r from the reference's uniform draws (logf differs from libm by <= 2 ulp)
    y = torch.from_numpy(g["sampler_y"]).reshape(-1)
    xs = torch.zeros(16, 4)
    _, r = ops.noise_blend(xs.cuda(), xs.cuda(), y.cuda(), 5.0, return_r=True)
    np.testing.assert_allclose(r.cpu().numpy(), g["sampler_r_lam5"].reshape(-1), rtol=3e-6, atol=1e-9)
    # blend given the reference's noise: recover y from r is not possible, so check via the oracle
    # formula on device draws instead, and the golden outputs through explicit (noise, r)
    import oracle
    for lam in (3, 5, 8):
        torch.manual_seed(0)
        yy = torch.rand(4)
        noise = torch.from_numpy(g[f"denoiser_lam{lam}_seed0_noise"])
        out, r = ops.noise_blend(x.cuda(), noise.cuda(), yy.cuda(), float(lam), return_r=True)
        c = 1 / np.exp(lam)
        r_ref = 1 / lam * torch.log(1 / (yy * (1 - c) + c))
        np.testing.assert_allclose(r.cpu().numpy(), r_ref.numpy(), rtol=3e-6)
        ref = oracle.step_oracle.blend_with_given_noise(x, noise, r.cpu())
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-7)


def test_mse_ssim_loss_fwd_bwd(ops, golden_dir):
    import oracle
    g = torch.Generator().manual_seed(7)
    for (B, H, W) in [(2, 32, 32), (1, 64, 96), (3, 40, 33)]:
        t = oracle.synthetic_face_crops(B, 64, seed=5)[:, :, :H, :W].contiguous()
        p = (t + 0.3 * torch.randn(t.shape, generator=g)).clone()
        p[0, 0, :4, :4] = 1.7   # outside [-1,1]: exercises the clip and its zero gradient
        p[0, 1, 5, 5] = -1.0    # exactly on the clip boundary (gradient passes, like torch)
        pr = p.clone().requires_grad_(True)
        p64 = p.double().requires_grad_(True)
        crit = oracle.MseStructuralSimilarityLoss(-1.0, 1.0)
        loss_ref = crit(pr, t)
        loss_ref.backward()
        loss64 = crit(p64, t.double())
        loss64.backward()
        out, grad = ops.mse_ssim_loss(p.cuda(), t.cuda())
        out = out.cpu()
        assert abs(out[0].item() - loss64.item()) < 2e-6
        assert abs(out[1].item() - F.mse_loss(p, t).item()) < 1e-6
        # sigma = E[x^2] - mu^2 cancels in fp32: the gradient's fp32 noise floor is ~1e-5.  Stated
        # tolerance: rel-L2 5e-5 vs the float64 evaluation, and within 4x the CPU-fp32 oracle's own error
        e_hip, e_cpu = rel_l2(grad.cpu(), p64.grad), rel_l2(pr.grad, p64.grad)
        assert e_hip < 5e-5 and e_hip < max(4 * e_cpu, 5e-6), (e_hip, e_cpu)
    # first-party combination formula against the reference golden (ssim term = ours)
    gl = np.load(golden_dir / "loss_first_party.npz")
    p, t = torch.from_numpy(gl["pred"]), torch.from_numpy(gl["target"])
    out, _ = ops.mse_ssim_loss(p.cuda(), t.cuda())
    out = out.cpu()
    # golden: loss computed by the reference with ssim := 0.25  ->  (mse + 0.75)/2
    mse_from_golden = 2 * float(gl["loss_with_ssim_0.25"]) - 0.75
    assert abs(out[1].item() - mse_from_golden) < 1e-6
    assert abs(out[0].item() - (out[1].item() + 1 - out[2].item()) / 2) < 1e-7


def test_adam_and_ema(ops):
    g = torch.Generator().manual_seed(9)
    n = 1003  # not a multiple of 4: exercises the tail
    p = torch.randn(n, generator=g)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=0.02, betas=(0.5, 0.999))
    ph = p.clone().cuda()
    m = torch.zeros(n).cuda()
    v = torch.zeros(n).cuda()
    for step in range(1, 6):
        grad = torch.randn(n, generator=g)
        pr.grad = grad.clone()
        opt.step()
        ops.adam_step(ph, grad.cuda(), m, v, 0.02, 0.5, 0.999, 1e-8, step)
        assert max_rel(ph.cpu(), pr.detach()) < 2e-6
    a = torch.randn(n, generator=g)
    b = torch.randn(n, generator=g)
    for w in (0.3, 1 - 0.9999, 0.75):
        e = a.clone().cuda()
        ops.ema_lerp(e, b.cuda(), w)
        torch.testing.assert_close(e.cpu(), torch.lerp(a, b, w), rtol=1e-6, atol=1e-7)


def test_affine_warp_matches_torch_grid_sample(ops):
    """K17 against F.grid_sample(F.affine_grid(theta), bilinear, zeros, align_corners=False) on CPU.  Tolerance:
    the sampling coordinates are computed in a different order of fp32 operations (1-2 ulp of a coordinate up to
    W ~ 1e-4 pixel), so interpolation weights differ by ~1e-4 * |neighbour difference|: rel-L2 <= 2e-5 on smooth
    face-like crops, max abs 2e-4 on white noise in [-1, 1]."""
    import math
    import oracle
    g = torch.Generator().manual_seed(3)
    B, H, W = 5, 64, 96
    ang = (torch.rand(B, generator=g) * 2 - 1) * math.radians(15)
    sc = torch.rand(B, generator=g) * 0.4 + 0.8
    tx, ty = (torch.rand(B, generator=g) * 2 - 1) * 0.4, (torch.rand(B, generator=g) * 2 - 1) * 0.4
    cos, sin = torch.cos(ang) / sc, torch.sin(ang) / sc
    theta = torch.stack([torch.stack([cos, -sin, tx], 1), torch.stack([sin, cos, ty], 1)], 1)
    theta[0] = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])      # identity
    theta[1] = torch.tensor([[0.5, 0.0, 1.2], [0.0, 0.5, -1.1]])     # mostly outside the image: zero border
    for x in (oracle.synthetic_face_crops(B, 64, seed=2)[:, :, :, :64].repeat(1, 1, 1, 2)[:, :, :H, :W].contiguous(),
              torch.rand(B, 3, H, W, generator=g) * 2 - 1):
        want = F.grid_sample(x, F.affine_grid(theta, list(x.shape), align_corners=False), mode="bilinear",
                             padding_mode="zeros", align_corners=False)
        got = ops.affine_warp(x.cuda(), theta.cuda()).cpu()
        assert rel_l2(got, want) < 2e-5 or (got - want).abs().max() < 2e-4, (rel_l2(got, want), (got - want).abs().max())
        assert (got - want).abs().max() < 2e-4
        assert (got[0] - x[0]).abs().max() < 1e-5   # identity
    with pytest.raises(ValueError):
        ops.affine_warp(x.cuda(), theta[:2].cuda())


def test_layout_roundtrip(ops):
    x = torch.randn(2, 3, 8, 10)
    h = ops.nchw_to_nhwc(x.cuda(), 4)
    assert h.shape == (2, 8, 10, 4) and torch.equal(h[..., 3].cpu(), torch.zeros(2, 8, 10))
    assert torch.equal(ops.nhwc_to_nchw(h, 3).cpu(), x)


def test_balance_kernels_match_reference_golden(ops, golden_dir):
    """fixed-ratio blend and per-image L1 (balance_training_images) against vectors produced by the reference."""
    g = np.load(golden_dir / "balance.npz")
    x = torch.from_numpy(g["x"]).cuda()
    for ratio in (0.7, 0.25):
        out = ops.noise_blend_fixed(x, torch.from_numpy(g[f"blend_ratio{ratio}_noise"]).cuda(), ratio)
        assert torch.equal(out.cpu(), torch.from_numpy(g[f"blend_ratio{ratio}_out"]))   # bit-exact
    dl = ops.l1_per_image(torch.from_numpy(g["pred"]).cuda(), x)
    assert max_rel(dl.cpu(), torch.from_numpy(g["difficulty_loss"])) < 1e-6
    # a size where every partial block has work, and a per-image ratio tensor
    big_p, big_t = torch.randn(3, 3, 64, 96), torch.randn(3, 3, 64, 96)
    ref = (big_p - big_t).abs().double().mean(dim=(1, 2, 3)).float()
    assert max_rel(ops.l1_per_image(big_p.cuda(), big_t.cuda()).cpu(), ref) < 1e-6
    r = torch.tensor([0.1, 0.5, 0.9])
    want = torch.sqrt(1 - r.view(-1, 1, 1, 1)) * big_p + torch.sqrt(r.view(-1, 1, 1, 1)) * big_t
    # 1 ulp: torch's CPU sqrt is not correctly rounded on every host (sqrt(0.9f) comes out one ulp high on the GPU
    # box's CPU; the kernel's __fsqrt_rn value is the correctly rounded one) -- the reference-generated fixtures
    # above are matched bit for bit
    assert max_rel(ops.noise_blend_fixed(big_p.cuda(), big_t.cuda(), r).cpu(), want) < 2.5e-7


def test_empty_batches_are_no_ops(ops):
    """zero-sized inputs: every entry point returns without launching (no fault, no error)."""
    z = torch.zeros(0, 3, 32, 32, device="cuda")
    assert ops.noise_blend(z, z, torch.zeros(0, device="cuda"), 5.0).shape == z.shape
    assert ops.noise_blend_fixed(z, z, 0.7).shape == z.shape
    assert ops.l1_per_image(z, z).shape == (0,)
    assert ops.affine_warp(z, torch.zeros(0, 2, 3, device="cuda")).shape == z.shape
    d = ops.make_desc(0, 16, 16, 32, 0, 32, 3, 1, 1, False)
    w = torch.randn(32, 32, 3, 3, device="cuda")
    wf, wd = ops.pack_weights(d, w)
    y, stats, tiles = ops.conv_forward(d, torch.zeros(0, 16, 16, 32, device="cuda"), None, wf)
    assert y.shape == (0, 16, 16, 32)
    dx0, _ = ops.conv_backward_data(d, torch.zeros(0, 16, 16, 32, device="cuda"), wd)
    assert dx0.shape == (0, 16, 16, 32)
    dw = ops.conv_backward_weight(d, torch.zeros(0, 16, 16, 32, device="cuda"), torch.zeros(0, 16, 16, 32, device="cuda"), None)
    torch.cuda.synchronize()
    assert dw.shape == w.shape and float(dw.abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(3, 32, 48), (1, 7, 5)], ids=str)
def test_u8rgb_normalise_is_the_host_transform_bit_for_bit(shape):
    """d3f_u8rgb_normalise (the `uint8_batches: true` input pipeline): HWC uint8 RGB -> normalised NCHW fp32 must equal
    the host transform NormalizeToTensor = A.Normalize(mean, std, max_pixel_value=255) + ToTensorV2
    (/root/reference/d3f/train_deep_fake/lit_module.py:100-110) BIT FOR BIT, every byte value in every channel."""
    import numpy as np
    from denoising_diffusion_deep_fake_amd import ops
    from denoising_diffusion_deep_fake_amd.dataset.image_dataset import NormalizeToTensor
    B, H, W = shape
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(B, H, W, 3), dtype=np.uint8)
    img.reshape(-1, 3)[:256] = np.arange(256, dtype=np.uint8)[:, None][:min(256, B * H * W)]   # every byte value, every channel
    for mean, std in (([0.5, 0.5, 0.5], [0.5, 0.5, 0.5]), ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225]),
                      ([128 / 255.0] * 3, [128 / 255.0] * 3)):
        t = NormalizeToTensor(mean, std)
        want = torch.stack([t(image=img[b])["image"] for b in range(B)])
        got = ops.u8rgb_normalise(torch.from_numpy(img).cuda(), mean, std)
        assert got.shape == (B, 3, H, W) and got.dtype == torch.float32
        assert torch.equal(got.cpu(), want), (mean, std, (got.cpu() - want).abs().max())
    with pytest.raises(ValueError):
        ops.u8rgb_normalise(torch.zeros(2, 3, 8, 8, dtype=torch.uint8, device="cuda"), [0.5] * 3, [0.5] * 3)
