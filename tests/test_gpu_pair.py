"""GPU: the two networks of train_deep_fake's denoise mode stepped as ONE set of kernel launches (UnetPair,
d3f_unet_pair_forward / d3f_unet_pair_backward).

Reference: d3f/train_deep_fake/lit_module.py:142-181 -- optimizer 0 trains model_a on batch a, optimizer 1 trains model_b on
batch b, nothing shared.  Lightning's order of the two steps is unobservable, so the fused step must compute the VALUES of
the one-after-the-other loop.  What is asserted here:
  * bit for bit: a pair pass == each network stepped alone on the pair's kernel choices (`Unet.set_plan_nets(2)`) -- outputs,
    every parameter gradient, BatchNorm running statistics, at shapes that exercise the k-split tiles, split-K slabs +
    reduce, the Winograd / patch / stem kernels and the class-form weight gradients, in fp32, f32x3 and bf16 storage;
  * the trainer's fused route == the sequential loop on the same kernel choices after several batches, bit for bit (state
    dict incl. Adam-updated parameters), and within fp32 rounding of the sequential loop on its own (8-image) plan;
  * the layer-by-layer float64 gates on BOTH networks of a pair: tests/test_gpu_parity_layers.py.
"""
import copy

import pytest
import torch

from util import rel_l2

pytestmark = pytest.mark.gpu


def _two_nets(dtype, seeds=(3, 4)):
    from denoising_diffusion_deep_fake_amd import Unet
    nets = []
    for seed in seeds:
        torch.manual_seed(seed)
        net = Unet("resnet34", None, 3, 3, None, compute_dtype=dtype)
        with torch.no_grad():
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.uniform_(0.5, 1.5)
                    m.bias.normal_(0, 0.1)
            net.segmentation_head[0].bias.normal_(0, 0.1)
        nets.append(net.cuda().train())
    return nets


def _pass(forward, xs, tgts):
    """forward -> loss gradient -> backward; returns (predictions, output gradients)"""
    from denoising_diffusion_deep_fake_amd import ops
    preds = forward(*xs)
    gouts = [ops.mse_ssim_loss(p.detach(), t)[1] for p, t in zip(preds, tgts)]
    torch.autograd.backward(list(preds), gouts)
    return [p.detach().clone() for p in preds], gouts


@pytest.mark.parametrize("dtype,shape", [("f32", (2, 64, 64)), ("f32", (3, 64, 96)), ("f32", (8, 128, 128)),
                                         ("f32x3", (2, 64, 64)), ("bf16", (2, 64, 64)), ("bf16", (8, 128, 128))],
                         ids=lambda v: v if isinstance(v, str) else "x".join(map(str, v)))
def test_pair_pass_is_bitwise_each_network_alone_on_the_pair_plan(dtype, shape):
    from denoising_diffusion_deep_fake_amd import UnetPair
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    B, H, W = shape
    nets = _two_nets(dtype)
    twins = [copy.deepcopy(n).cuda().train().set_plan_nets(2) for n in nets]
    xs = [synthetic_face_crops(B, (H, W), seed=30 + i, device="cuda") for i in range(2)]
    tgts = [synthetic_face_crops(B, (H, W), seed=40 + i, device="cuda") for i in range(2)]
    pair = UnetPair(*nets)
    for step in range(2):  # twice: the second pass runs on statistics / packed weights the first one left behind
        for n in nets + twins:
            for p in n.parameters():
                p.grad = None
        preds, _ = _pass(pair, xs, tgts)
        for i in range(2):
            (alone,), _ = _pass(lambda x, i=i: (twins[i](x),), [xs[i]], [tgts[i]])
            assert torch.equal(preds[i], alone), (step, i, "prediction", rel_l2(preds[i], alone))
            assert torch.equal(nets[i].flat_grads, twins[i].flat_grads), (
                step, i, "gradient", rel_l2(nets[i].flat_grads, twins[i].flat_grads))
            assert torch.equal(nets[i].flat_bn_stats, twins[i].flat_bn_stats), (step, i, "running statistics")
            assert int(nets[i].encoder.bn1.num_batches_tracked) == step + 1
    # the two halves really are different networks on different data
    assert not torch.equal(preds[0], preds[1]) and not torch.equal(nets[0].flat_grads, nets[1].flat_grads)
    # ... and a network of the pair still runs alone (its own single plan) right after a pair pass
    with torch.no_grad():
        assert torch.isfinite(nets[0](xs[0])).all()


@pytest.mark.parametrize("dtype,shape", [("f32", (2, 64, 96)), ("bf16", (2, 64, 64)), ("f32", (8, 128, 128))],
                         ids=lambda v: v if isinstance(v, str) else "x".join(map(str, v)))
def test_pair_on_a_poisoned_workspace(monkeypatch, dtype, shape):
    """Both copies of the pair's workspace filled with NaN bit patterns before the first call (D3F_POISON_WORKSPACE): everything
    network 1 reads from ITS copy -- packed-weight padding, statistics rows, coefficient arrays, split-K and gradient slabs,
    the zeroed parts of stride-2 gradients -- must have been written by a launch that carried network 1; a kernel that
    initialised only network 0's copy shows up as NaN, one that read network 0's copy as a mismatch with the twin."""
    from denoising_diffusion_deep_fake_amd import UnetPair
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    monkeypatch.setenv("D3F_POISON_WORKSPACE", "1")
    B, H, W = shape
    nets = _two_nets(dtype)
    twins = [copy.deepcopy(n).cuda().train().set_plan_nets(2) for n in nets]
    xs = [synthetic_face_crops(B, (H, W), seed=60 + i, device="cuda") for i in range(2)]
    tgts = [synthetic_face_crops(B, (H, W), seed=70 + i, device="cuda") for i in range(2)]
    pair = UnetPair(*nets)
    for _ in range(2):  # second pass: re-uses the arenas after a backward
        for n in nets + twins:
            for p in n.parameters():
                p.grad = None
        preds, _ = _pass(pair, xs, tgts)
        for i in range(2):
            (alone,), _ = _pass(lambda x, i=i: (twins[i](x),), [xs[i]], [tgts[i]])
            assert torch.isfinite(preds[i]).all() and torch.isfinite(nets[i].flat_grads).all(), i
            assert torch.equal(preds[i], alone) and torch.equal(nets[i].flat_grads, twins[i].flat_grads), i
    assert pair.last_engine.workspace.numel() == 2 * pair.last_engine.net_stride


def test_pair_sees_parameter_updates_even_if_each_network_ran_alone_in_between():
    """The pair's engines hold their own packed copy of both networks' weights.  Sequence of a training loop with a preview
    callback: pair pass -> both Adam steps (raw-pointer updates of the flat buffers) -> EACH network run alone (its own plan
    re-packs and is up to date) -> pair pass.  The second pair pass must compute on the UPDATED weights: bit for bit the
    values of twins that went through the same updates alone (a cleared 'dirty' flag would leave the pair one update behind)."""
    from denoising_diffusion_deep_fake_amd import UnetPair
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    from denoising_diffusion_deep_fake_amd.optim import FusedAdam
    nets = _two_nets("f32")
    twins = [copy.deepcopy(n).cuda().train().set_plan_nets(2) for n in nets]
    opts = [FusedAdam(n.parameters(), lr=0.01, betas=(0.5, 0.999), module=n) for n in nets + twins]
    xs = [synthetic_face_crops(2, 64, seed=30 + i, device="cuda") for i in range(2)]
    tgts = [synthetic_face_crops(2, 64, seed=40 + i, device="cuda") for i in range(2)]
    pair = UnetPair(*nets)
    first, _ = _pass(pair, xs, tgts)
    for i in range(2):
        _pass(lambda x, i=i: (twins[i](x),), [xs[i]], [tgts[i]])
    for o in opts:
        o.step()
    with torch.no_grad():  # each network alone, on its own plan (a preview / validation forward between two batches)
        alone_now = [nets[i](xs[i]) for i in range(2)]
    for n in nets + twins:
        for p in n.parameters():
            p.grad = None
    second, _ = _pass(pair, xs, tgts)
    for i in range(2):
        (want,), _ = _pass(lambda x, i=i: (twins[i](x),), [xs[i]], [tgts[i]])
        assert torch.equal(second[i], want), (i, rel_l2(second[i], want))
        assert not torch.equal(second[i], first[i])                      # the update moved the prediction
        assert rel_l2(second[i], alone_now[i]) < 1e-1                     # ... to where the network itself is now
        assert torch.equal(nets[i].flat_grads, twins[i].flat_grads)


def test_pair_against_the_networks_own_single_plans_within_rounding():
    """the 8-image plan of a network alone picks other tiles (k-split 32x32 instead of 64x64 / 128x64, no Winograd): same
    mathematics, another summation order -- the pair must agree with it to fp32 rounding (forward) and to the mask-flip
    floor of two fp32 evaluations (gradients; the binding gradient gate is the mask-pinned one in test_gpu_parity_layers)"""
    from denoising_diffusion_deep_fake_amd import UnetPair
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    B, S = 4, 128
    nets = _two_nets("f32")
    twins = [copy.deepcopy(n).cuda().train() for n in nets]
    xs = [synthetic_face_crops(B, S, seed=30 + i, device="cuda") for i in range(2)]
    tgts = [synthetic_face_crops(B, S, seed=40 + i, device="cuda") for i in range(2)]
    preds, _ = _pass(UnetPair(*nets), xs, tgts)
    for i in range(2):
        (alone,), _ = _pass(lambda x, i=i: (twins[i](x),), [xs[i]], [tgts[i]])
        assert rel_l2(preds[i], alone) < 2e-5, rel_l2(preds[i], alone)
        assert rel_l2(nets[i].flat_grads, twins[i].flat_grads) < 3e-2
        assert rel_l2(nets[i].flat_bn_stats, twins[i].flat_bn_stats) < 1e-5


def test_pair_protocol_errors_and_accumulation():
    from denoising_diffusion_deep_fake_amd import D3FError, Unet, UnetPair
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    a, b = _two_nets("f32")
    with pytest.raises(TypeError):
        UnetPair(a, a)
    with pytest.raises(ValueError):
        UnetPair(a, Unet("resnet18", None, 3, 3, None).cuda())
    pair = UnetPair(a, b)
    x = synthetic_face_crops(2, 64, seed=1, device="cuda")
    with pytest.raises(RuntimeError):
        pair(x, x[:1])
    with pytest.raises(D3FError):
        pair(x.cpu(), x.cpu())
    b.eval()
    with pytest.raises(D3FError, match="train-mode"):
        pair(x, x)
    b.train()
    pa, pb = pair(x, x)
    pa.sum().backward()              # a loss on one network only: the other one's gradient is exactly zero
    assert float(b.flat_grads.abs().max()) == 0.0 and float(a.flat_grads.abs().max()) > 0.0
    for p in list(a.parameters()) + list(b.parameters()):
        p.grad = None
    pa, pb = pair(x, x)
    (pa.sum() + pb.sum()).backward()
    g1 = a.segmentation_head[0].bias.grad.clone()
    pa, pb = pair(x, x)              # .grad not cleared: accumulation, as any nn.Module
    (pa.sum() + pb.sum()).backward()
    torch.testing.assert_close(a.segmentation_head[0].bias.grad, 2 * g1, rtol=1e-4, atol=1e-4)
    with torch.no_grad():            # inference-only pair pass: no graph, no lease
        qa, qb = pair(x, x)
    assert qa.shape == (2, 3, 64, 64) and not qa.requires_grad
    # the C ABI refuses the single-network entry points on a pair handle and the other way round
    import ctypes as C
    from denoising_diffusion_deep_fake_amd import _lib
    L = _lib.lib()
    h = C.c_void_p()
    assert L.d3f_unet_create_nets(b"resnet34", 3, 3, 2, 64, 64, _lib.F32, 2, 1, C.byref(h)) != 0   # plan_nets < nets
    assert L.d3f_unet_create_nets(b"resnet34", 3, 3, 2, 64, 64, _lib.F32, 3, 3, C.byref(h)) != 0
    _lib.check(L.d3f_unet_create_nets(b"resnet34", 3, 3, 2, 64, 64, _lib.F32, 2, 2, C.byref(h)))
    try:
        assert L.d3f_unet_nets(h) == 2 and L.d3f_unet_net_workspace_stride(h) * 2 == L.d3f_unet_workspace_bytes(h)
        ws = torch.empty(16, dtype=torch.uint8, device="cuda")
        assert L.d3f_unet_pack_weights(h, _lib.ptr(a.flat_params), _lib.ptr(ws), None) != 0
        assert b"pair" in L.d3f_last_error()
    finally:
        L.d3f_unet_destroy(h)


HP_FAKE = dict(mode="denoise", batch_size=2, learning_rate=0.01, adam_b1=0.5, adam_b2=0.999, max_epochs=1,
               cosine_scheduler_max_epoch=50, num_workers=0, encoder_name="resnet34",
               noise_exponential_sampling_lambda=3, mean_a=[0.5] * 3, std_a=[0.5] * 3, mean_b=[0.5] * 3,
               std_b=[0.5] * 3, synthetic=True, image_size=64, synthetic_length=8, ema_beta=0.9999,
               ema_update_every=1, augment=True)


def test_trainer_fused_denoise_step_equals_the_sequential_loop(tmp_path):
    """Trainer.fit over train_deep_fake's LitModule, denoise mode, augmentation on: the fused route (default) against
    Lightning's loop -- toggle, zero_grad, training_step(batch, i, optimizer_idx), backward, step, per optimizer -- run on
    the same kernel choices (`pair_plan: true`): same seeds -> the whole state dict (both nets after 4 batches x 2 Adam
    steps, BatchNorm statistics, counters) bit-identical; against the loop on each net's own plan: fp32 rounding."""
    from denoising_diffusion_deep_fake_amd.train_deep_fake.lit_module import LitModule
    from denoising_diffusion_deep_fake_amd.trainer import Trainer

    def run(**kw):
        torch.manual_seed(11)
        lit = LitModule(**dict(HP_FAKE, default_root_dir=str(tmp_path), **kw))
        torch.manual_seed(12)
        tr = Trainer(max_epochs=1, default_root_dir=tmp_path, enable_checkpointing=False).fit(lit)
        torch.cuda.synchronize()
        return tr.global_step, {k: v.detach().clone() for k, v in lit.state_dict().items()}, lit

    steps_f, sd_f, lit_f = run()
    steps_s, sd_s, lit_s = run(pair_fused=False, pair_plan=True)
    steps_o, sd_o, lit_o = run(pair_fused=False)
    assert lit_f._pair is not None and lit_s._pair is None and lit_o._pair is None     # the routes really differed
    assert steps_f == steps_s == steps_o == 8                                          # 4 batches x 2 optimizers
    assert sd_f.keys() == sd_s.keys()
    for k in sd_f:
        assert torch.equal(sd_f[k], sd_s[k]), k
    logged = {k: float(v) for k, v in lit_f._logged.items() if k.startswith("loss_denoise")}
    assert set(logged) == {"loss_denoise/train_a", "loss_denoise/train_b"}
    assert logged == {k: float(v) for k, v in lit_s._logged.items() if k.startswith("loss_denoise")}
    # (Adam's m / sqrt(v) turns rounding-level differences of near-zero gradients into lr-sized steps, so after 8 updates the
    # two plans' PARAMETERS are not comparable element by element; the pass-level comparison is the test above.  The runs
    # must still describe the same training: the last batch's losses agree to a few per cent.)
    for k, v in logged.items():
        assert abs(v - float(lit_o._logged[k])) < 0.05 * abs(v), (k, v, float(lit_o._logged[k]))
    # swap mode and `pair_fused: false` never take the fused route
    swap = LitModule(**dict(HP_FAKE, mode="swap"))
    assert not swap.pair_fused_active() and not lit_s.pair_fused_active() and lit_f.pair_fused_active()


def test_fused_step_with_a_ragged_batch_pair_falls_back(tmp_path):
    """CombinedLoader(max_size_cycle) over datasets of different length: the last batch of the shorter one is ragged, the
    two halves differ in shape, and the trainer must take the sequential loop for that batch (and the fused route for the
    others) without a word"""
    from denoising_diffusion_deep_fake_amd.train_deep_fake.lit_module import LitModule
    from denoising_diffusion_deep_fake_amd.trainer import optimizer_steps
    torch.manual_seed(5)
    lit = LitModule(**dict(HP_FAKE, augment=False)).cuda().train()
    opts, _ = lit.configure_optimizers()
    opt_params = [[p for g in o.param_groups for p in g["params"]] for o in opts]
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    full = {k: {"image": synthetic_face_crops(2, 64, seed=7 + i, device="cuda"), "index": None} for i, k in enumerate("ab")}
    ragged = {"a": full["a"], "b": {"image": full["b"]["image"][:1].contiguous(), "index": None}}
    assert lit.pair_fused_active(full, opts) and not lit.pair_fused_active(ragged, opts)
    optimizer_steps(lit, opts, opt_params, full, 0, True, None)
    assert lit._pair is not None and lit._pair.last_engine is not None
    before = lit._pair.last_engine.serial
    optimizer_steps(lit, opts, opt_params, ragged, 1, True, None)
    assert lit._pair.last_engine.serial == before                     # the pair did not run
    optimizer_steps(lit, opts, opt_params, full, 2, True, None)
    assert lit._pair.last_engine.serial == before + 1
    assert all(torch.isfinite(p).all() for p in lit.parameters())
