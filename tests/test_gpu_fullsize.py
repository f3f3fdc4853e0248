"""GPU, BASELINE.json full sizes (256x256, bs 16, f32): size-independent properties of every layer shape and of the
whole step.  (The oracle comparison AT these sizes -- every layer's output and activation against the float64 oracle,
every gradient against the mask-pinned float64 oracle, 256x256 at bs 16 and bs 8, and the B=64 eval forward -- is
tests/test_gpu_parity_layers.py; the properties here are cheap enough to run per layer shape and per mode.)
  * adjoint identities of each contraction trio:  <conv(x; w), dy> = <x, dgrad(dy; w)> = <w, wgrad(dy; x)>
    (ties the forward, data-gradient and weight-gradient kernels of a layer shape to each other);
  * linearity of the forward kernel;
  * bitwise run-to-run reproducibility of the whole training step (fixed-order reductions, no float atomics);
  * the loss goes down over real optimiser steps on a fixed batch;
  * the other BASELINE configurations at their full sizes: bf16 at 256x256 bs 16 (adjoint identities at bf16 tolerance,
    learning), the paired-domain step at 256x256 bs 8 x 2 nets (bitwise reproducible, each net only touches its own
    gradient), the eval-mode fused epilogues at B=64 256x256 against the train-path apply on the same statistics."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# B, H, W, C0, C1, Cout, k, stride, pad, up  -- layer shapes of the 256x256 bs-16 network
SHAPES = [
    (16, 64, 64, 64, 0, 64, 3, 1, 1, False),       # layer1
    (16, 64, 64, 64, 0, 128, 3, 2, 1, False),      # layer2.0 (stride 2)
    (16, 64, 64, 64, 0, 128, 1, 2, 0, False),      # layer2.0 downsample
    (16, 16, 16, 256, 0, 256, 3, 1, 1, False),     # layer3 (split-K)
    (16, 8, 8, 512, 0, 512, 3, 1, 1, False),       # layer4 (split-K)
    (16, 32, 32, 256, 128, 128, 3, 1, 1, True),    # decoder 1 conv1: upsample + concat
    (16, 128, 128, 64, 64, 32, 3, 1, 1, True),     # decoder 3 conv1 (patch wgrad, 4 ci slices)
    (16, 256, 256, 32, 0, 16, 3, 1, 1, True),      # decoder 4 conv1
    (16, 256, 256, 16, 0, 16, 3, 1, 1, False),     # decoder 4 conv2
]


def _dot(a, b):
    return (a.double() * b.double()).sum().item()


@pytest.mark.parametrize("mode", ["f32", "f32x3"])
@pytest.mark.parametrize("shape", SHAPES, ids=[str(s) for s in SHAPES])
def test_adjoint_identities(shape, mode):
    from denoising_diffusion_deep_fake_amd import ops
    DT = ops.F32 if mode == "f32" else ops.F32X3
    B, H, W, C0, C1, Co, k, s, pd, up = shape
    g = torch.Generator(device="cuda").manual_seed(0)
    h0, w0 = (H // 2, W // 2) if up else (H, W)
    x0 = torch.randn(B, h0, w0, C0, device="cuda", generator=g)
    x1 = torch.randn(B, H, W, C1, device="cuda", generator=g) if C1 else None
    w = torch.randn(Co, C0 + C1, k, k, device="cuda", generator=g) / ((C0 + C1) * k * k) ** 0.5
    d = ops.make_desc(B, H, W, C0, C1, Co, k, s, pd, up)
    wf, wd = ops.pack_weights(d, w, DT)
    y, _, _ = ops.conv_forward(d, x0, x1, wf, DT, splitk=True)
    dy = torch.randn(y.shape, device="cuda", generator=g)
    dx0, dx1 = ops.conv_backward_data(d, dy, wd, DT, splitk=True)
    dw = ops.conv_backward_weight(d, dy, x0, x1, DT)
    lhs = _dot(y, dy)
    # <x, dX>: the up-sampled source enters through its nearest x2 expansion = sum over the 2x2 block of dX0
    # (an up-sampled source gets its gradient at its own low resolution: ops.conv_backward_data)
    via_x = _dot(x0, dx0) + (_dot(x1, dx1) if C1 else 0.0)
    via_w = _dot(w, dw)
    scale = max(abs(lhs), (y.double().norm() * dy.double().norm()).item() * 1e-3)
    assert abs(lhs - via_x) < 2e-5 * scale, (lhs, via_x)
    assert abs(lhs - via_w) < 2e-5 * scale, (lhs, via_w)
    # linearity of the forward kernel
    x0b = torch.randn(x0.shape, device="cuda", generator=g)
    yb, _, _ = ops.conv_forward(d, x0b, x1, wf, DT, splitk=True)
    ysum, _, _ = ops.conv_forward(d, x0 + x0b, x1, wf, DT, splitk=True)
    if C1:
        y_skip_only, _, _ = ops.conv_forward(d, torch.zeros_like(x0), x1, wf, DT, splitk=True)
        ref = y + yb - y_skip_only
    else:
        ref = y + yb
    assert ((ysum - ref).norm() / ref.norm()).item() < 5e-6


def _train_steps(n, seed=0):
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    from denoising_diffusion_deep_fake_amd.train_denoiser.lit_module import LitModule
    torch.manual_seed(seed)
    lit = LitModule(batch_size=16, learning_rate=0.003, max_epochs=1, cosine_scheduler_max_epoch=100, num_workers=0,
                    encoder_name="resnet34", noise_exponential_sampling_lambda=5, mean=[128] * 3, std=[128] * 3,
                    synthetic=True, image_size=256, augment=False).cuda().train()
    (opt,), _ = lit.configure_optimizers()
    x = synthetic_face_crops(16, 256, seed=1234, device="cuda")
    losses = []
    for i in range(n):
        torch.manual_seed(1000 + i)
        opt.zero_grad(set_to_none=True)
        loss = lit.training_step({"image": x, "index": None}, i)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return losses, lit.model.flat_params.clone(), lit.model.flat_grads.clone()


def test_training_step_is_bitwise_reproducible_and_learns():
    l1, p1, g1 = _train_steps(6)
    l2, p2, g2 = _train_steps(6)
    assert l1 == l2, (l1, l2)
    assert torch.equal(g1, g2) and torch.equal(p1, p2)
    assert all(torch.isfinite(torch.tensor(l1)))
    assert min(l1[3:]) < l1[0], l1  # the noisy->clean objective improves on a fixed batch


BF16_SHAPES = [SHAPES[0], SHAPES[1], SHAPES[3], SHAPES[5], SHAPES[7]]


@pytest.mark.parametrize("shape", BF16_SHAPES, ids=[str(s) for s in BF16_SHAPES])
def test_adjoint_identities_bf16_full_size(shape):
    """BASELINE config 2/3 dtype at the full layer shapes: operands pre-rounded to bf16, so the products are exact
    and the only differences are the bf16 roundings of y and dX (2^-9 per element, random sign): the three inner
    products agree to a few 1e-4 of |y||dy|, the f32 weight gradient to 1e-5."""
    from denoising_diffusion_deep_fake_amd import ops
    B, H, W, C0, C1, Co, k, s, pd, up = shape
    g = torch.Generator(device="cuda").manual_seed(0)
    h0, w0 = (H // 2, W // 2) if up else (H, W)
    x0 = torch.randn(B, h0, w0, C0, device="cuda", generator=g).bfloat16()
    x1 = torch.randn(B, H, W, C1, device="cuda", generator=g).bfloat16() if C1 else None
    w = (torch.randn(Co, C0 + C1, k, k, device="cuda", generator=g) / ((C0 + C1) * k * k) ** 0.5).bfloat16().float()
    d = ops.make_desc(B, H, W, C0, C1, Co, k, s, pd, up)
    wf, wd = ops.pack_weights(d, w, ops.BF16)
    y, _, _ = ops.conv_forward(d, x0, x1, wf, ops.BF16, splitk=True)
    cop = (Co + 7) // 8 * 8
    dy = torch.zeros(y.shape[:3] + (cop,), device="cuda", dtype=torch.bfloat16)
    dy[..., :Co] = torch.randn(y.shape[:3] + (Co,), device="cuda", generator=g).bfloat16()
    dx0, dx1 = ops.conv_backward_data(d, dy, wd, ops.BF16, splitk=True)
    dw = ops.conv_backward_weight(d, dy, x0, x1, ops.BF16)
    lhs = _dot(y[..., :Co], dy[..., :Co])
    via_x = _dot(x0, dx0) + (_dot(x1, dx1) if C1 else 0.0)
    via_w = _dot(w, dw)
    scale = (y.double().norm() * dy.double().norm()).item()
    assert abs(lhs - via_x) < 1e-3 * scale, (lhs, via_x, scale)
    assert abs(lhs - via_w) < 1e-3 * scale, (lhs, via_w, scale)   # y is rounded to bf16, dw is not
    assert torch.isfinite(dw).all()


def test_bf16_training_step_full_size_learns_and_is_reproducible():
    def run():
        from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
        from denoising_diffusion_deep_fake_amd.train_denoiser.lit_module import LitModule
        torch.manual_seed(0)
        lit = LitModule(batch_size=16, learning_rate=0.003, max_epochs=1, cosine_scheduler_max_epoch=100, num_workers=0,
                        encoder_name="resnet34", noise_exponential_sampling_lambda=5, mean=[128] * 3, std=[128] * 3,
                        synthetic=True, image_size=256, augment=False, precision="bf16").cuda().train()
        (opt,), _ = lit.configure_optimizers()
        x = synthetic_face_crops(16, 256, seed=1234, device="cuda")
        losses = []
        for i in range(6):
            torch.manual_seed(1000 + i)
            opt.zero_grad(set_to_none=True)
            loss = lit.training_step({"image": x, "index": None}, i)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        return losses, lit.model.flat_grads.clone()
    l1, g1 = run()
    l2, g2 = run()
    assert l1 == l2 and torch.equal(g1, g2)
    assert all(torch.isfinite(torch.tensor(l1))) and min(l1[3:]) < l1[0], l1


def test_paired_domain_step_full_size():
    """BASELINE config 3 at its size: two nets, 256x256, bs 8 per domain, denoise mode: run to run bitwise equal, both
    flat gradients finite, and an optimiser step of net a leaves net b (weights, gradient, Adam state) untouched."""
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    from denoising_diffusion_deep_fake_amd.train_deep_fake.lit_module import LitModule

    def run():
        torch.manual_seed(3)
        lit = LitModule(mode="denoise", batch_size=8, learning_rate=0.01, adam_b1=0.5, adam_b2=0.999, max_epochs=1,
                        cosine_scheduler_max_epoch=50, num_workers=0, encoder_name="resnet34",
                        noise_exponential_sampling_lambda=3, mean_a=[0.5] * 3, std_a=[0.5] * 3, mean_b=[0.5] * 3,
                        std_b=[0.5] * 3, synthetic=True, image_size=256, augment=False).cuda().train()
        opts, _ = lit.configure_optimizers()
        batch = {k: {"image": synthetic_face_crops(8, 256, seed=7 + i, device="cuda"), "index": None}
                 for i, k in enumerate("ab")}
        lit.model_b.prepare()
        wb0 = lit.model_b.flat_params.clone()
        out = []
        for it in range(2):
            for oi, opt in enumerate(opts):
                torch.manual_seed(50 + 2 * it + oi)
                opt.zero_grad(set_to_none=True)
                loss = lit.training_step(batch, it, oi)
                loss.backward()
                if it == 0 and oi == 0:   # before net b has ever been stepped
                    assert lit.model_b.flat_grads is None and torch.equal(lit.model_b.flat_params, wb0)
                opt.step()
                out.append(loss.item())
        return out, lit.model_a.flat_grads.clone(), lit.model_b.flat_grads.clone(), lit.model_a.flat_params.clone()
    l1, ga1, gb1, pa1 = run()
    l2, ga2, gb2, pa2 = run()
    assert l1 == l2 and torch.equal(ga1, ga2) and torch.equal(gb1, gb2) and torch.equal(pa1, pa2)
    assert torch.isfinite(ga1).all() and torch.isfinite(gb1).all() and not torch.equal(ga1, gb1)


def test_paired_domain_fused_step_full_size_equals_the_sequential_loop():
    """BASELINE configs[3] as the trainer runs it (round 6): the denoise-mode batch of 2 x 8 images at 256x256 through
    trainer.optimizer_steps' FUSED route (both nets' forward / backward as one set of launches, UnetPair) -- two batches,
    both Adam steps each -- against Lightning's one-after-the-other loop on the same kernel choices (`pair_plan`): losses,
    both flat gradients and both updated parameter buffers bit-identical, and the fused route run to run reproducible
    (reference: d3f/train_deep_fake/lit_module.py:142-181)."""
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    from denoising_diffusion_deep_fake_amd.train_deep_fake.lit_module import LitModule
    from denoising_diffusion_deep_fake_amd.trainer import optimizer_steps

    def run(**kw):
        torch.manual_seed(3)
        lit = LitModule(mode="denoise", batch_size=8, learning_rate=0.01, adam_b1=0.5, adam_b2=0.999, max_epochs=1,
                        cosine_scheduler_max_epoch=50, num_workers=0, encoder_name="resnet34",
                        noise_exponential_sampling_lambda=3, mean_a=[0.5] * 3, std_a=[0.5] * 3, mean_b=[0.5] * 3,
                        std_b=[0.5] * 3, synthetic=True, image_size=256, augment=False, **kw).cuda().train()
        opts, _ = lit.configure_optimizers()
        opt_params = [[p for g in o.param_groups for p in g["params"]] for o in opts]
        batch = {k: {"image": synthetic_face_crops(8, 256, seed=7 + i, device="cuda"), "index": None}
                 for i, k in enumerate("ab")}
        torch.manual_seed(50)
        losses = []
        for it in range(2):
            optimizer_steps(lit, opts, opt_params, batch, it, True, None)
            losses += [float(lit._logged["loss_denoise/train_a"]), float(lit._logged["loss_denoise/train_b"])]
        return (losses, lit.model_a.flat_grads.clone(), lit.model_b.flat_grads.clone(), lit.model_a.flat_params.clone(),
                lit.model_b.flat_params.clone(), lit._pair is not None)
    f1, f2, s = run(), run(), run(pair_fused=False, pair_plan=True)
    assert f1[5] and f2[5] and not s[5]
    for k in range(5):
        assert (f1[k] == f2[k]) if k == 0 else torch.equal(f1[k], f2[k]), ("fused run to run", k)
        assert (f1[k] == s[k]) if k == 0 else torch.equal(f1[k], s[k]), ("fused vs sequential", k)
    assert all(l == l for l in f1[0]) and not torch.equal(f1[1], f1[2])


@pytest.mark.timeout(900)
def test_swap_step_full_size_against_the_oracle():
    """The reference's 200-epoch phase at BASELINE configs[3]'s size (`mode: "swap"`, 8 x 256 x 256 per domain, both
    optimizer indices; d3f/train_deep_fake/lit_module.py:183-206 and swap_config.yml): per index the EMA teacher of the
    OTHER net is updated and renders the fake under no_grad with train-mode BatchNorm, the explicit noise draws blend it,
    the student denoises it -- `loss_swap`, `swap_difference` and the student's flat gradient against oracle.swap_step in
    float64 (index 0 also against the CPU-fp32 oracle as the yardstick), and index 1 sees net a AFTER its Adam step through
    its teacher."""
    import numpy as np

    import oracle
    from denoising_diffusion_deep_fake_amd.train_deep_fake.lit_module import LitModule
    from util import rel_l2
    torch.manual_seed(5)
    lam = 8
    lit = LitModule(mode="swap", batch_size=8, learning_rate=0.01, adam_b1=0.5, adam_b2=0.999, max_epochs=1,
                    cosine_scheduler_max_epoch=200, num_workers=0, encoder_name="resnet34",
                    noise_exponential_sampling_lambda=lam, mean_a=[0.5] * 3, std_a=[0.5] * 3, mean_b=[0.5] * 3,
                    std_b=[0.5] * 3, synthetic=True, image_size=256, ema_beta=0.9999, ema_update_every=1,
                    augment=False).cuda().train()
    opts, _ = lit.configure_optimizers()
    xs = {"a": oracle.synthetic_face_crops(8, 256, seed=41), "b": oracle.synthetic_face_crops(8, 256, seed=42)}
    batch = {k: {"image": v.cuda(), "index": None} for k, v in xs.items()}
    crit = oracle.MseStructuralSimilarityLoss(-1.0, 1.0)

    def replica(model, dtype):
        ref = oracle.Unet("resnet34", None, 3, 3, None).train()
        ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
        return ref.to(dtype)

    def draws(seed, shape):
        torch.manual_seed(seed)
        noise = torch.randn(shape, device="cuda")
        y = torch.rand(size=(shape[0], 1, 1, 1), device="cuda")
        c = 1 / np.exp(lam)
        return noise.cpu(), 1 / lam * torch.log(1 / (y.reshape(-1).cpu() * (1 - c) + c))

    for oi, (name, student, other) in enumerate((("a", lit.model_a, lit.model_b), ("b", lit.model_b, lit.model_a))):
        out = {}
        for dtype in ((torch.float32, torch.float64) if oi == 0 else (torch.float64,)):
            st = replica(student, dtype)
            teacher = oracle.EMA(replica(other, dtype), beta=0.9999, update_every=1)   # first update(): a copy of `other`
            noise, r = draws(99 + oi, xs[name].shape)
            loss, diff, _, _ = oracle.swap_step(xs[name].to(dtype), st, teacher, crit, noise.to(dtype), r.to(dtype))
            loss.backward()
            out[dtype] = (loss.item(), diff.item(), torch.cat([p.grad.reshape(-1) for p in st.parameters()]))
        torch.manual_seed(99 + oi)
        opts[oi].zero_grad(set_to_none=True)
        loss = lit.training_step(batch, 0, oi)
        loss.backward()
        l64, d64, g64 = out[torch.float64]
        l32, d32, g32 = out.get(torch.float32, (None, None, None))
        assert abs(loss.item() - l64) < (max(4 * abs(l32 - l64), 5e-6) if l32 is not None else 2e-5), (oi, loss.item(), l32, l64)
        sd = float(lit._logged[f"swap_difference/{name}"])
        assert abs(sd - d64) < 1e-5 * d64, (oi, sd, d64)
        e_hip = rel_l2(student.flat_grads, g64)
        e_cpu = rel_l2(g32, g64) if g32 is not None else None
        # (unpinned ReLU / max-pool masks: the floor of two fp32 evaluations; the 2e-4 mask-pinned gate on every tensor at
        # this size is tests/test_gpu_parity_layers.py)
        assert e_hip < 5e-2 and (e_cpu is None or e_hip < max(10 * e_cpu, 2e-5)), (oi, e_hip, e_cpu)
        opts[oi].step()
    assert lit.ema_model_a._host_step == 1 and lit.ema_model_b._host_step == 1


def test_eval_fused_epilogue_matches_train_path_apply_at_b64():
    """BASELINE config 4's shape (B=64, 256x256, eval mode): the eval kernels fold BatchNorm (+ residual + ReLU) into
    the conv epilogue and pick other tiles / split-K factors than the small tests.  With the running statistics set to
    the batch statistics of this very input, the eval forward must reproduce the train forward layer by layer
    (`:a` through d3f_unet_export) and at the output.  The statistics go through the engine's own update
    (running = 0.9 * 0 + 0.1 * batch, unbiased variance), are scaled back by 10 and by (n-1)/n: three extra fp32
    roundings of scale/shift, amplified like any rounding by this BatchNorm-heavy net -> 2e-5 rel-L2, not bit equality."""
    from denoising_diffusion_deep_fake_amd import Unet
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    torch.manual_seed(0)
    net = Unet("resnet34", None, 3, 3, None).cuda().train()
    x = synthetic_face_crops(64, 256, seed=5, device="cuda")
    names = ["encoder.conv1", "encoder.layer1.0.conv2", "encoder.layer2.0.conv2", "encoder.layer3.5.conv2",
             "encoder.layer4.2.conv2", "decoder.blocks.0.conv1.0", "decoder.blocks.2.conv2.0", "decoder.blocks.4.conv2.0"]
    bns = [(name, m) for name, m in net.named_modules() if isinstance(m, torch.nn.BatchNorm2d)]
    assert len(bns) == 46
    with torch.no_grad():
        for _, m in bns:
            m.running_mean.zero_()
            m.running_var.zero_()
        out_train = net(x)
        train_a = {n: net.export_activation(n + ":a") for n in names}
        for name, m in bns:
            conv = (name.replace(".bn1", ".conv1").replace(".bn2", ".conv2").replace("downsample.1", "downsample.0")
                    .replace("conv1.1", "conv1.0").replace("conv2.1", "conv2.0"))
            b, _, h, w = net.export_activation_shape(conv + ":y")
            n_el = b * h * w
            m.running_mean.mul_(10.0)
            m.running_var.mul_(10.0 * (n_el - 1) / n_el)   # unbiased -> the biased variance the train path divides by
        net.eval()
        out_eval = net(x)
        for n in names:
            e = net.export_activation(n + ":a")
            err = ((e - train_a[n]).norm() / train_a[n].norm()).item()
            assert err < 2e-5, (n, err)
        assert ((out_eval - out_train).norm() / out_train.norm()).item() < 2e-5


@pytest.mark.timeout(600)
def test_sample50_hipgraph_replay_equals_eager_loop_bitwise():
    """BASELINE.json configs[4] as written: 50 eval-mode forwards of B=64 at 256x256, output fed back, with the denoise
    step replayed from ONE captured hipGraph (Unet.forward_graph, d3f_unet_forward_graph) -- every one of the 50
    replays must equal the eager forward of the same input bit for bit, and an optimiser-style parameter update between
    two loops must be picked up by the already-captured graph (only pointers are baked in).  Reference loop:
    d3f/script_tools/put_video_through_fake_model.py:111-119 -> d3f/train_deep_fake/lit_module.py:259-270."""
    from denoising_diffusion_deep_fake_amd import Unet
    from denoising_diffusion_deep_fake_amd._lib import D3FError
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    torch.manual_seed(5)
    net = Unet("resnet34", None, 3, 3, None).cuda()
    with pytest.raises(D3FError):
        net.train().forward_graph(torch.zeros(1, 3, 32, 32, device="cuda"))
    net.eval()
    x0 = synthetic_face_crops(64, 256, seed=3, device="cuda")
    xbuf, ybuf = torch.empty_like(x0), torch.empty_like(x0)
    with torch.no_grad():
        for loop in range(2):
            xbuf.copy_(x0)
            y = x0
            for it in range(50):
                y = net(y).clamp_(-1.0, 1.0)                      # eager: ~50 launches from Python per forward
                out = net.forward_graph(xbuf, out=ybuf)           # replay of the graph captured at it == 0, loop == 0
                assert out.data_ptr() == ybuf.data_ptr()
                torch.clamp(ybuf, -1.0, 1.0, out=xbuf)
                assert torch.equal(y, xbuf), (loop, it)
            assert torch.isfinite(y).all() and y.abs().max() > 0
            net.flat_params.mul_(1.01)    # what an optimiser / EMA update does: values change, pointers stay
            net.mark_params_changed()
