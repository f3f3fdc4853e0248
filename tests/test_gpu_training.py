"""GPU: the training-step orchestration rows of SURVEY.md section 8 (a4, a5, a9, a10) -- the two
LitModules, EMA, predict_fake, the trainer loop with checkpoints and the CLI -- against the CPU oracle."""
import copy

import numpy as np
import pytest
import torch
import yaml

from util import max_rel, rel_l2

pytestmark = pytest.mark.gpu

HP_DENOISER = dict(batch_size=4, learning_rate=0.02, max_epochs=2, cosine_scheduler_max_epoch=2, num_workers=0,
                   encoder_name="resnet34", noise_exponential_sampling_lambda=5, mean=[128, 128, 128],
                   std=[128, 128, 128], synthetic=True, image_size=64, augment=False, synthetic_length=8)
HP_FAKE = dict(mode="denoise", batch_size=2, learning_rate=0.01, adam_b1=0.5, adam_b2=0.999, max_epochs=1,
               cosine_scheduler_max_epoch=50, num_workers=0, encoder_name="resnet34",
               noise_exponential_sampling_lambda=3, mean_a=[0.5] * 3, std_a=[0.5] * 3, mean_b=[0.5] * 3,
               std_b=[0.5] * 3, synthetic=True, image_size=64, synthetic_length=4, ema_beta=0.9999,
               ema_update_every=1, augment=False)


def _device_draws(seed, shape):
    """replay the device RNG stream of blend_random_amount_of_noise_with_each_sample"""
    torch.manual_seed(seed)
    noise = torch.randn(shape, device="cuda")
    y = torch.rand(size=(shape[0], 1, 1, 1), device="cuda")
    return noise.cpu(), y.reshape(-1).cpu()


def _oracle_r(y, lam):
    c = 1 / np.exp(lam)
    return 1 / lam * torch.log(1 / (y * (1 - c) + c))


def test_train_denoiser_step_matches_oracle():
    import oracle
    from denoising_diffusion_deep_fake_amd.train_denoiser.lit_module import LitModule
    torch.manual_seed(0)
    lit = LitModule(**HP_DENOISER).cuda().train()
    ref = oracle.Unet("resnet34", None, 3, 3, None).train()
    ref.load_state_dict({k: v.cpu() for k, v in lit.model.state_dict().items()})
    ref64 = copy.deepcopy(ref).double()
    crit = oracle.MseStructuralSimilarityLoss(-1.0, 1.0)
    x = oracle.synthetic_face_crops(4, 64, seed=21)
    noise, y = _device_draws(123, x.shape)
    torch.manual_seed(123)
    loss = lit.training_step({"image": x.cuda(), "index": None}, 0)
    noisy = oracle.step_oracle.blend_with_given_noise(x, noise, _oracle_r(y, 5))
    l32 = crit(ref(noisy), x)
    l64 = crit(ref64(noisy.double()), x.double())
    assert abs(loss.item() - l64.item()) < max(4 * abs(l32.item() - l64.item()), 5e-6)
    (opt,), (sched,) = lit.configure_optimizers()
    before = lit.model.flat_params.clone()
    loss.backward()
    opt.step()
    delta = (lit.model.flat_params - before).abs()
    assert 0.019 < delta.max().item() <= 0.0201  # Adam's first step moves every element by ~lr
    assert float(lit._logged["loss"]) == pytest.approx(loss.item())


@pytest.mark.parametrize("mode", ["denoise", "swap"])
def test_train_deep_fake_steps(mode):
    import oracle
    from denoising_diffusion_deep_fake_amd.train_deep_fake.lit_module import LitModule
    torch.manual_seed(1)
    lit = LitModule(**dict(HP_FAKE, mode=mode)).cuda().train()
    assert (lit.ema_model_a is not None) == (mode == "swap")
    keys = list(lit.state_dict().keys())
    assert "model_a.encoder.conv1.weight" in keys and "model_b.segmentation_head.0.bias" in keys
    if mode == "swap":
        assert "ema_model_a.ema_model.encoder.conv1.weight" in keys and "ema_model_b.step" in keys
        assert not any(k.startswith("ema_model_a.online_model") for k in keys)
    opts, scheds = lit.configure_optimizers()
    xa = oracle.synthetic_face_crops(2, 64, seed=31).cuda()
    xb = oracle.synthetic_face_crops(2, 64, seed=32).cuda()
    batch = {"a": {"image": xa, "index": None}, "b": {"image": xb, "index": None}}
    # oracle replica of net a for the first optimiser step
    ref = oracle.Unet("resnet34", None, 3, 3, None).train()
    ref.load_state_dict({k: v.cpu() for k, v in lit.model_a.state_dict().items()})
    crit = oracle.MseStructuralSimilarityLoss(-1.0, 1.0)
    losses = []
    for oi, opt in enumerate(opts):
        opt.zero_grad(set_to_none=True)
        if oi == 0 and mode == "denoise":
            noise, y = _device_draws(77, xa.shape)
            torch.manual_seed(77)
        loss = lit.training_step(batch, 0, oi)
        if oi == 0 and mode == "denoise":
            noisy = oracle.step_oracle.blend_with_given_noise(xa.cpu(), noise, _oracle_r(y, 3))
            l_ref = crit(ref(noisy), xa.cpu())
            assert abs(loss.item() - l_ref.item()) < 2e-5
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses))
    if mode == "swap":
        # first update() copies online -> EMA (step <= update_after_step), and the teacher saw train-mode BN
        assert lit.ema_model_b._host_step == 1 and lit.ema_model_a._host_step == 1
        assert "swap_difference/a" in lit._logged and "loss_swap/train_b" in lit._logged
    else:
        assert "loss_denoise/train_a" in lit._logged


def test_swap_step_values_match_oracle():
    """`training_swap_step_for_one_model` (d3f/train_deep_fake/lit_module.py:183-206) value for value: EMA teacher of
    the other domain (train-mode BatchNorm, no_grad) renders the fake, explicit noise draws blend it, the student
    denoises it; `loss_swap`, `swap_difference` and the student's gradient against oracle.swap_step in float64."""
    import oracle
    from denoising_diffusion_deep_fake_amd.train_deep_fake.lit_module import LitModule
    torch.manual_seed(5)
    lit = LitModule(**dict(HP_FAKE, mode="swap", augment=False)).cuda().train()
    opts, _ = lit.configure_optimizers()
    xa = oracle.synthetic_face_crops(2, 64, seed=41)
    xb = oracle.synthetic_face_crops(2, 64, seed=42)
    batch = {"a": {"image": xa.cuda(), "index": None}, "b": {"image": xb.cuda(), "index": None}}

    def replica(model, dtype):
        ref = oracle.Unet("resnet34", None, 3, 3, None).train()
        ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
        return ref.to(dtype)
    crit = oracle.MseStructuralSimilarityLoss(-1.0, 1.0)
    out = {}
    for dtype in (torch.float32, torch.float64):
        student = replica(lit.model_a, dtype)
        teacher = oracle.EMA(replica(lit.model_b, dtype), beta=0.9999, update_every=1)
        noise, y = _device_draws(99, xa.shape)
        loss, diff, fake, pred = oracle.swap_step(xa.to(dtype), student, teacher, crit, noise.to(dtype),
                                                  _oracle_r(y, 3).to(dtype))
        loss.backward()
        out[dtype] = (loss.item(), diff.item(), torch.cat([p.grad.reshape(-1) for p in student.parameters()]))
    torch.manual_seed(99)
    opts[0].zero_grad(set_to_none=True)
    loss = lit.training_step(batch, 0, 0)      # optimizer 0: net a learns from the EMA copy of net b
    loss.backward()
    l32, d32, g32 = out[torch.float32]
    l64, d64, g64 = out[torch.float64]
    assert abs(loss.item() - l64) < max(4 * abs(l32 - l64), 5e-6), (loss.item(), l32, l64)
    sd = float(lit._logged["swap_difference/a"])
    assert abs(sd - d64) < max(4 * abs(d32 - d64), 1e-6 * d64), (sd, d32, d64)
    assert float(lit._logged["loss_swap/train_a"]) == pytest.approx(loss.item())
    e_hip, e_cpu = rel_l2(lit.model_a.flat_grads, g64), rel_l2(g32, g64)
    # (unpinned masks: the noise floor of tests/test_gpu_unet.py; the mask-pinned 2e-4 gate is test_gpu_parity_layers.py)
    assert e_hip < 5e-2 and e_hip < max(10 * e_cpu, 2e-5), (e_hip, e_cpu)
    assert lit.ema_model_b._host_step == 1 and lit.model_b.flat_grads is None  # only the student was trained


def test_ema_matches_oracle():
    import oracle
    from denoising_diffusion_deep_fake_amd import Unet
    from denoising_diffusion_deep_fake_amd.optim import EMA
    torch.manual_seed(2)
    net = Unet("resnet34", None, 3, 3, None).cuda().prepare()
    ref = oracle.Unet("resnet34", None, 3, 3, None)
    ref.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    ema = EMA(net, beta=0.9999, update_every=1, update_after_step=2)
    ema_ref = oracle.EMA(ref, beta=0.9999, update_every=1, update_after_step=2)
    g = torch.Generator().manual_seed(3)
    for step in range(6):
        d = torch.randn(net.flat_params.numel(), generator=g) * 0.01
        with torch.no_grad():
            net.flat_params.add_(d.cuda())
            off = 0
            for p in ref.parameters():
                p.add_(d[off:off + p.numel()].view_as(p))
                off += p.numel()
            for bn_h, bn_r in zip([m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)],
                                  [m for m in ref.modules() if isinstance(m, torch.nn.BatchNorm2d)]):
                bn_h.running_mean.add_(0.01 * (step + 1))
                bn_r.running_mean.add_(0.01 * (step + 1))
        ema.update()
        ema_ref.update()
        assert abs(ema.get_current_decay() - ema_ref.get_current_decay()) < 1e-12
        sd, sd_ref = ema.ema_model.state_dict(), ema_ref.ema_model.state_dict()
        for k in ("encoder.conv1.weight", "decoder.blocks.2.conv1.0.weight", "encoder.layer2.0.bn1.running_mean",
                  "segmentation_head.0.bias"):
            assert max_rel(sd[k], sd_ref[k]) < 2e-6, (step, k)
    assert ema.initted.item() and ema.step.item() == 6


def test_predict_fake_matches_oracle():
    import oracle
    from denoising_diffusion_deep_fake_amd.train_deep_fake.lit_module import LitModule
    torch.manual_seed(4)
    lit = LitModule(**HP_FAKE).cuda().eval()
    ref = oracle.Unet("resnet34", None, 3, 3, None).eval()
    ref.load_state_dict({k: v.cpu() for k, v in lit.model_a.state_dict().items()})
    rng = np.random.default_rng(0)
    frame_bgr = rng.integers(0, 256, size=(64, 96, 3), dtype=np.uint8)
    out = lit.predict_fake(frame_bgr, "a")
    assert out.shape == frame_bgr.shape and out.dtype == np.uint8
    mean, std = torch.tensor([0.5] * 3), torch.tensor([0.5] * 3)
    rgb = torch.from_numpy(np.ascontiguousarray(frame_bgr[:, :, ::-1])).float().permute(2, 0, 1)
    x = ((rgb - mean.reshape(3, 1, 1) * 255) / (std.reshape(3, 1, 1) * 255)).unsqueeze(0)
    with torch.no_grad():
        y = ref(x)
    want = oracle.tensor_to_uint8_denormalised(y, mean, std).numpy()[:, :, ::-1]
    diff = np.abs(out.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.01  # .int() truncation flips on float-noise ties only


def test_predict_u8_fused_and_graph_paths_are_bit_exact():
    """K16 + hipGraph: the fused uint8 path, eager and graph-replayed, must give exactly the
    bytes of the unfused reference sequence (cv2_to_tensor_normalised -> eval forward -> tensor_cv2_to_denormalised)
    on the same device arithmetic; frames, weights and batch size change between replays."""
    from denoising_diffusion_deep_fake_amd.train_deep_fake.lit_module import LitModule
    torch.manual_seed(6)
    lit = LitModule(**HP_FAKE).cuda().eval()
    net = lit.model_a
    with torch.no_grad():  # non-trivial running statistics
        for name, buf in net.named_buffers():
            if name.endswith("running_mean"):
                buf.normal_(0, 0.1)
            elif name.endswith("running_var"):
                buf.uniform_(0.5, 1.5)
    net.mark_params_changed()
    mean, std = [0.4, 0.5, 0.6], [0.5, 0.45, 0.55]
    mt, st = torch.tensor(mean, device="cuda"), torch.tensor(std, device="cuda")
    rng = np.random.default_rng(1)

    def unfused(frame):
        with torch.no_grad():
            return lit.tensor_cv2_to_denormalised(net(lit.cv2_to_tensor_normalised(frame, mt, st)), mt, st)

    buf_in = torch.empty((64, 96, 3), dtype=torch.uint8, device="cuda")
    buf_out = torch.empty((1, 64, 96, 3), dtype=torch.uint8, device="cuda")
    for it in range(4):
        frame = rng.integers(0, 256, size=(64, 96, 3), dtype=np.uint8)
        want = unfused(frame)
        buf_in.copy_(torch.from_numpy(frame))
        eager = net.predict_u8(buf_in, mean, std, graph=False).cpu().numpy()
        replay = net.predict_u8(buf_in, mean, std, graph=True, out=buf_out).cpu().numpy()
        assert np.array_equal(eager, want), (it, np.abs(eager.astype(int) - want.astype(int)).max())
        assert np.array_equal(replay, want), it
        if it == 1:  # a parameter update between replays: the graph must see the re-packed weights
            with torch.no_grad():
                net.segmentation_head[0].bias.add_(0.05)
                next(iter(net.parameters())).mul_(1.01)
    # the LitModule entry (host frame in, host frame out) and a batch of frames
    frame = rng.integers(0, 256, size=(64, 96, 3), dtype=np.uint8)
    lit.hparams.mean_b, lit.hparams.std_b = mean, std
    assert np.array_equal(lit.predict_fake(frame, "a"), unfused(frame))
    batch = torch.from_numpy(rng.integers(0, 256, size=(3, 32, 64, 3), dtype=np.uint8)).cuda()
    got = net.predict_u8(batch, mean, std).cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], unfused(batch[i].cpu().numpy()))
    with pytest.raises(RuntimeError):
        net.predict_u8(torch.zeros((30, 64, 3), dtype=torch.uint8, device="cuda"), mean, std)


def test_balance_training_images_module(tmp_path):
    """the balance LitModule trains at a fixed noise ratio, scores every image in eval mode with
    the per-image L1 kernel, bins the scores like the reference and writes the class list the reference forgot."""
    import oracle
    from denoising_diffusion_deep_fake_amd.balance_training_images.lit_module import LitModule
    from denoising_diffusion_deep_fake_amd.trainer import Trainer
    torch.manual_seed(8)
    out_list = tmp_path / "classes.txt"
    lit = LitModule(batch_size=4, learning_rate=0.01, max_epochs=1, num_workers=0, encoder_name="resnet34",
                    ratio_of_noise=0.7, number_of_classes=4, mean=[128] * 3, std=[128] * 3, synthetic=True,
                    synthetic_length=12, image_size=64, output_image_list_path=str(out_list))
    trainer = Trainer(max_epochs=1, default_root_dir=str(tmp_path / "logs"), enable_checkpointing=False)
    trainer.fit(lit)
    index, classes = lit.difficulty_index
    assert sorted(index.tolist()) == list(range(12)) and classes.min() == 0 and classes.max() == 3
    lines = out_list.read_text().strip().splitlines()
    assert len(lines) == 12 and all(len(l.split("\t")) == 2 for l in lines)
    # the binning is the reference's arithmetic (golden-pinned in the oracle) on the device-computed losses
    lit.eval()
    batch = next(iter(lit.val_dataloader()))
    batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    torch.manual_seed(1)
    out = lit.validation_step(batch, 0)
    torch.manual_seed(1)
    noisy = oracle.blend_fixed_amount_of_noise(batch["image"].cpu(), 0.7, noise=torch.randn_like(batch["image"]).cpu())
    with torch.no_grad():
        pred = lit.model(noisy.cuda()).cpu()
    want = oracle.difficulty_loss(pred, batch["image"].cpu())
    assert max_rel(out["loss"], want) < 1e-5
    assert torch.equal(lit.compute_difficulty_index_for_each_loss(out["loss"]), oracle.difficulty_index(out["loss"], 4))


def test_trainer_fit_checkpoint_resume(tmp_path):
    from denoising_diffusion_deep_fake_amd.train_denoiser.lit_module import LitModule
    from denoising_diffusion_deep_fake_amd.trainer import Trainer
    hp = dict(HP_DENOISER, default_root_dir=str(tmp_path))
    torch.manual_seed(5)
    lit = LitModule(**hp)
    tr = Trainer(max_epochs=2, log_every_n_steps=1, default_root_dir=tmp_path).fit(lit)
    assert tr.global_step == 4 and tr.current_epoch == 2  # 8 synthetic images / bs 4 = 2 steps per epoch
    ckpts = sorted((tr.log_dir / "checkpoints").glob("*.ckpt"))
    assert ckpts, "default ModelCheckpoint writes one file per epoch"
    ck = torch.load(ckpts[-1], map_location="cpu", weights_only=False)
    assert set(ck) >= {"state_dict", "hyper_parameters", "optimizer_states", "lr_schedulers", "epoch", "global_step"}
    assert "model.encoder.conv1.weight" in ck["state_dict"]
    assert (tr.log_dir / "metrics.csv").read_text().count("loss=") == 4
    # cosine schedule stepped once per epoch: lr(2 of T_max=2) = 0
    assert tr.optimizers[0].param_groups[0]["lr"] == pytest.approx(0.0, abs=1e-9)
    again = LitModule.load_from_checkpoint(ckpts[-1]).cuda().eval()
    lit.eval()
    x = torch.randn(1, 3, 64, 64).cuda()
    with torch.no_grad():
        torch.testing.assert_close(again(x), lit(x))
    tr2 = Trainer(max_epochs=3, default_root_dir=tmp_path).fit(LitModule.load_from_checkpoint(ckpts[-1]), ckpt_path=ckpts[-1])
    assert tr2.current_epoch == 3 and tr2.global_step == 6 and tr2.optimizers[0]._step == 6


def test_cli_smoke(tmp_path):
    from click.testing import CliRunner
    from denoising_diffusion_deep_fake_amd.main import cli
    cfg = tmp_path / "denoise.yml"
    cfg.write_text(yaml.safe_dump(dict(HP_DENOISER, default_root_dir=str(tmp_path / "logs"))))
    r = CliRunner().invoke(cli, ["denoise", "--config", str(cfg), "--max_steps", "2"])
    assert r.exit_code == 0, r.output + str(r.exception)
    cfg2 = tmp_path / "fake.yml"
    cfg2.write_text(yaml.safe_dump(dict(HP_FAKE, default_root_dir=str(tmp_path / "logs2"))))
    r = CliRunner().invoke(cli, ["train", "new", "--config_path", str(cfg2), "--max_steps", "2"])
    assert r.exit_code == 0, r.output + str(r.exception)
    last = next((tmp_path / "logs2").glob("version_*/checkpoints/last.ckpt"))
    swap = tmp_path / "swap.yml"
    swap.write_text(yaml.safe_dump(dict(mode="swap", max_epochs=1, ema_beta=0.9999, ema_update_every=1,
                                        noise_exponential_sampling_lambda=8)))
    r = CliRunner().invoke(cli, ["train", "modify", "--config_path", str(swap), "--checkpoint_path", str(last),
                                 "--max_steps", "2"])
    assert r.exit_code == 0, r.output + str(r.exception)
    # d3f balance on real image files: PIL reader -> training -> scoring -> class list
    from PIL import Image
    (tmp_path / "imgs").mkdir()
    rng = np.random.default_rng(0)
    names = []
    for i in range(6):
        Image.fromarray(rng.integers(0, 256, size=(64, 64, 3), dtype=np.uint8)).save(tmp_path / "imgs" / f"{i}.png")
        names.append(f"imgs/{i}.png")
    (tmp_path / "images.txt").write_text("\n".join(names) + "\n")
    bal = tmp_path / "balance.yml"
    bal.write_text(yaml.safe_dump(dict(batch_size=3, learning_rate=0.01, max_epochs=1, num_workers=0,
                                       encoder_name="resnet34", ratio_of_noise=0.7, number_of_classes=3,
                                       mean=[128, 128, 128], std=[128, 128, 128],
                                       default_root_dir=str(tmp_path / "logs3"))))
    out_list = tmp_path / "classes.txt"
    r = CliRunner().invoke(cli, ["balance", "--config", str(bal), "--input_list", str(tmp_path / "images.txt"),
                                 "--output_list", str(out_list)])
    assert r.exit_code == 0, r.output + str(r.exception)
    rows = [l.split("\t") for l in out_list.read_text().strip().splitlines()]
    assert [r_[0] for r_ in rows] == names and {int(r_[1]) for r_ in rows} <= {0, 1, 2}


def test_shift_scale_rotate_on_device():
    """train_deep_fake's ShiftScaleRotate(p=0.7) on the GPU (d3f/train_deep_fake/lit_module.py:99-111): p=0 is the
    identity, unselected samples pass through bit-exactly, selected ones equal grid_sample(bilinear, zeros) of the
    same matrices, and about 70 % of the samples are selected."""
    import torch.nn.functional as F
    import oracle
    from denoising_diffusion_deep_fake_amd.train_deep_fake.lit_module import LitModule, ShiftScaleRotate
    x = oracle.synthetic_face_crops(6, 64, seed=5)[:, :, :, :48].contiguous()
    xc = x.cuda()
    assert torch.equal(ShiftScaleRotate(p=0.0)(xc), xc)
    aug = ShiftScaleRotate(shift_limit=0.2, scale_limit=0.1, rotate_limit=15, p=0.7)
    draws = {"apply": torch.tensor([True, False, True, True, False, True]).cuda(),
             "angle": torch.tensor([15.0, 3.0, -7.5, 0.0, 9.0, -15.0]).cuda(),
             "scale": torch.tensor([1.1, 1.0, 0.9, 1.0, 1.05, 0.95]).cuda(),
             "dx": torch.tensor([0.2, 0.1, -0.2, 0.0, 0.0, 0.05]).cuda(),
             "dy": torch.tensor([-0.2, 0.0, 0.1, 0.0, 0.1, -0.05]).cuda()}
    got = aug(xc, draws).cpu()
    th = ShiftScaleRotate.theta(*(draws[k].cpu() for k in ("angle", "scale", "dx", "dy")), 64, 48)
    want = F.grid_sample(x, F.affine_grid(th, list(x.shape), align_corners=False), mode="bilinear",
                         padding_mode="zeros", align_corners=False)
    for b in range(6):
        if draws["apply"][b]:
            assert (got[b] - want[b]).abs().max() < 2e-5, b
            assert (got[b] == 0).any() or b == 3   # the zero border shows up (normalised units)
        else:
            assert torch.equal(got[b], x[b])
    torch.manual_seed(0)
    frac = torch.stack([aug.draw(64, "cuda")["apply"].float().mean() for _ in range(20)]).mean().item()
    assert 0.64 < frac < 0.76
    d = aug.draw(4096, "cuda")
    assert d["angle"].abs().max() <= 15 and (d["scale"] - 1).abs().max() <= 0.1 + 1e-6 and d["dx"].abs().max() <= 0.2
    # wired into the working training path, off with augment=False
    assert LitModule(**dict(HP_FAKE, augment=True)).augmentation is not None
    assert LitModule(**HP_FAKE).augmentation is None
    lit = LitModule(**dict(HP_FAKE, augment=True)).cuda().train()
    xa, xb = oracle.synthetic_face_crops(2, 64, seed=1).cuda(), oracle.synthetic_face_crops(2, 64, seed=2).cuda()
    loss = lit.training_step({"a": {"image": xa, "index": None}, "b": {"image": xb, "index": None}}, 0, 1)
    assert torch.isfinite(loss)


def test_adam_state_is_torch_adam_state_both_ways():
    """Checkpoint drop-in (d3f/train_deep_fake/start_training.py:19-23 resumes Lightning checkpoints whose
    optimizer_states are torch.optim.Adam.state_dict()): FusedAdam writes that per-parameter form, reads it back,
    and torch.optim.Adam itself can load what FusedAdam saved -- the moments and the step count survive."""
    from denoising_diffusion_deep_fake_amd import Unet
    from denoising_diffusion_deep_fake_amd.optim import FusedAdam
    torch.manual_seed(11)
    net = Unet("resnet34", None, 3, 3, None).cuda().train()
    twin = copy.deepcopy(net).cuda().train()
    twin.load_state_dict(net.state_dict())
    fused = FusedAdam(net.parameters(), lr=0.01, betas=(0.5, 0.999), module=net)
    plain = torch.optim.Adam(twin.parameters(), lr=0.01, betas=(0.5, 0.999))
    x = torch.randn(2, 3, 64, 64, device="cuda")
    g = torch.randn(2, 3, 64, 64, device="cuda")

    def step(model, opt):
        opt.zero_grad(set_to_none=True)
        model(x).backward(g)
        opt.step()
    for _ in range(2):
        step(net, fused)
        step(twin, plain)
    sd_f, sd_p = fused.state_dict(), plain.state_dict()
    assert set(sd_f) == {"state", "param_groups"} and len(sd_f["state"]) == len(sd_p["state"]) == len(list(net.parameters()))
    for i in (0, 5, len(sd_p["state"]) - 1):
        assert float(sd_f["state"][i]["step"]) == float(sd_p["state"][i]["step"]) == 2.0
        assert sd_f["state"][i]["exp_avg"].shape == sd_p["state"][i]["exp_avg"].shape
        # (two HIP nets stepped by the two optimisers: after step 1 their weights differ by rounding, so step 2's
        # gradients differ by the ReLU mask flips of any two fp32 evaluations)
        assert rel_l2(sd_f["state"][i]["exp_avg"], sd_p["state"][i]["exp_avg"]) < 2e-3
        assert rel_l2(sd_f["state"][i]["exp_avg_sq"], sd_p["state"][i]["exp_avg_sq"]) < 2e-3
    # cross-load: the reference's optimizer state into the fused one, the fused one into torch.optim.Adam
    fused2 = FusedAdam(net.parameters(), lr=0.5, betas=(0.9, 0.9), module=net)
    fused2.load_state_dict(sd_p)
    assert fused2._step == 2 and fused2.param_groups[0]["lr"] == 0.01 and tuple(fused2.param_groups[0]["betas"]) == (0.5, 0.999)
    plain2 = torch.optim.Adam(twin.parameters(), lr=0.5)
    plain2.load_state_dict(sd_f)
    net.load_state_dict(twin.state_dict())           # same weights again, then one more step on each side
    step(net, fused2)
    step(twin, plain2)
    # (Adam's third step moves every weight by ~lr; with the moments lost the two sides would differ by ~lr / |w|)
    assert rel_l2(net.flat_params, torch.cat([p.detach().reshape(-1) for p in twin.parameters()])) < 1e-3
    # an optimizer state that does not fit is refused, not zero-filled
    bad = {"state": {0: sd_p["state"][0]}, "param_groups": sd_p["param_groups"]}
    with pytest.raises(ValueError):
        FusedAdam(net.parameters(), lr=0.01, module=net).load_state_dict(bad)


def test_resume_from_a_reference_style_checkpoint(tmp_path):
    """A hand-built Lightning 1.x checkpoint dict as the reference's Trainer writes it -- smp key names under
    `model.`, torch.optim.Adam optimizer_states, CosineAnnealingLR state -- resumes with weights, Adam moments, step
    count and lr intact; a checkpoint with a foreign key is refused (strict), not resumed from random init."""
    import oracle
    from denoising_diffusion_deep_fake_amd.train_denoiser.lit_module import LitModule
    from denoising_diffusion_deep_fake_amd.trainer import Trainer
    torch.manual_seed(3)
    ref = oracle.Unet("resnet34", None, 3, 3, None).train()
    opt = torch.optim.Adam(ref.parameters(), lr=0.02)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=4)
    crit = oracle.MseStructuralSimilarityLoss(-1.0, 1.0)
    x = oracle.synthetic_face_crops(2, 64, seed=9)
    for _ in range(2):   # "epoch 0" of a reference run: two optimiser steps on the CPU
        opt.zero_grad()
        crit(ref(x), x).backward()
        opt.step()
    sched.step()
    hp = dict(HP_DENOISER, max_epochs=2, cosine_scheduler_max_epoch=4)
    ckpt = {"epoch": 0, "global_step": 2, "pytorch-lightning_version": "1.9.5",
            "state_dict": {"model." + k: v for k, v in ref.state_dict().items()},
            "optimizer_states": [opt.state_dict()], "lr_schedulers": [sched.state_dict()],
            "hyper_parameters": hp}
    path = tmp_path / "reference.ckpt"
    torch.save(ckpt, path)
    lit = LitModule.load_from_checkpoint(path)
    tr = Trainer(max_epochs=1, default_root_dir=tmp_path, enable_checkpointing=False).fit(lit, ckpt_path=path)
    # max_epochs=1 and the checkpoint closed epoch 0: nothing is trained, only restored
    assert tr.current_epoch == 1 and tr.global_step == 2
    o = tr.optimizers[0]
    assert o._step == 2 and o.param_groups[0]["lr"] == pytest.approx(opt.param_groups[0]["lr"])
    want_m = torch.cat([opt.state[p]["exp_avg"].reshape(-1) for p in ref.parameters()])
    want_v = torch.cat([opt.state[p]["exp_avg_sq"].reshape(-1) for p in ref.parameters()])
    assert torch.equal(o.exp_avg.cpu(), want_m) and torch.equal(o.exp_avg_sq.cpu(), want_v)
    assert torch.equal(lit.model.flat_params.cpu(), torch.cat([p.detach().reshape(-1) for p in ref.parameters()]))
    # ... and training continues from there: epoch 1 = two more steps
    tr2 = Trainer(max_epochs=2, default_root_dir=tmp_path, enable_checkpointing=False).fit(
        LitModule.load_from_checkpoint(path), ckpt_path=path)
    assert tr2.global_step == 4 and tr2.optimizers[0]._step == 4
    bad_sd = dict(ckpt["state_dict"])
    bad_sd["model.encoder.conv1.weight_x"] = bad_sd.pop("model.encoder.conv1.weight")
    bad = dict(ckpt, state_dict=bad_sd)
    torch.save(bad, tmp_path / "bad.ckpt")
    with pytest.raises(RuntimeError):
        Trainer(max_epochs=1, default_root_dir=tmp_path, enable_checkpointing=False).fit(LitModule(**hp), ckpt_path=tmp_path / "bad.ckpt")


def test_ragged_last_batch_and_mid_epoch_resume(tmp_path):
    """the reference's loaders keep the ragged last batch (no drop_last, d3f/train_denoiser/lit_module.py:82-87):
    10 images at bs 4 are 3 steps per epoch, the last one on 2 images; a checkpoint written mid-epoch resumes inside
    that epoch instead of skipping the rest of it."""
    from denoising_diffusion_deep_fake_amd.train_denoiser.lit_module import LitModule
    from denoising_diffusion_deep_fake_amd.trainer import Trainer
    hp = dict(HP_DENOISER, synthetic_length=10, max_epochs=1, default_root_dir=str(tmp_path))
    torch.manual_seed(2)
    lit = LitModule(**hp)
    assert len(lit.train_dataloader()) == 3
    seen = []

    def recording(module):
        inner = module.training_step

        def step(batch, batch_idx):
            seen.append([int(i) for i in batch["index"]])
            return inner(batch, batch_idx)
        module.training_step = step
        return module
    tr = Trainer(max_epochs=1, default_root_dir=tmp_path, enable_checkpointing=False, max_steps=2).fit(recording(lit))
    # max_steps ended the run INSIDE epoch 0: the epoch is not counted as done, the per-epoch cosine schedule has not
    # stepped, and a checkpoint written now is marked mid-epoch (what ModelCheckpoint(train_time_interval=...) sees)
    assert tr.global_step == 2 and tr.current_epoch == 0 and tr._batches_done == 2
    assert tr.lr_schedulers[0].last_epoch == 0
    tr.save_checkpoint(tmp_path / "mid.ckpt")
    first = [i for b in seen for i in b]
    seen.clear()
    torch.manual_seed(12345)  # a resumed process starts with another global RNG state
    tr2 = Trainer(max_epochs=1, default_root_dir=tmp_path, enable_checkpointing=False).fit(
        recording(LitModule.load_from_checkpoint(tmp_path / "mid.ckpt")), ckpt_path=tmp_path / "mid.ckpt")
    assert tr2.global_step == 3 and tr2.current_epoch == 1   # one remaining (ragged) batch of epoch 0 was trained
    # ... and it is the batch the interrupted epoch had NOT trained yet: the epoch's permutation is replayed from the
    # checkpoint's loader seed, so every image is seen exactly once across the interruption
    rest = [i for b in seen for i in b]
    assert len(first) == 8 and len(rest) == 2 and sorted(first + rest) == list(range(10))
    assert tr2.lr_schedulers[0].last_epoch == 1


@pytest.mark.parametrize("dtype,shape", [("f32", (4, 64, 64)), ("bf16", (4, 64, 64)), ("f32", (16, 128, 128))],
                         ids=["f32_4x64", "bf16_4x64", "f32_16x128"])
def test_graph_train_step_is_bitwise_the_eager_step(dtype, shape):
    """GraphTrainStep (d3f_unet_train_step: pack -> blend -> forward -> loss -> backward -> Adam captured into one
    hipGraph) against the step Lightning's automatic optimisation runs around training_step
    (d3f/train_denoiser/lit_module.py:107-126): same seeds, five steps with a learning-rate change in between --
    parameters, BatchNorm statistics, Adam moments and every loss value must be bit-identical, replayed (graph), eager
    inside the one call (use_graph=False) and through the separate calls."""
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    from denoising_diffusion_deep_fake_amd.train_denoiser.lit_module import LitModule
    B, H, W = shape
    hp = dict(HP_DENOISER, batch_size=B, image_size=H, precision=dtype, augment=False)
    data = [synthetic_face_crops(B, (H, W), seed=30 + i, device="cuda") for i in range(2)]

    def run(mode):
        torch.manual_seed(5)
        lit = LitModule(**dict(hp, graph_step=mode != "separate")).cuda().train()
        (opt,), _ = lit.configure_optimizers()
        lit.attach_optimizers([opt])
        if mode != "separate":
            from denoising_diffusion_deep_fake_amd.graph_step import GraphTrainStep
            lit.__dict__["_graph_step"] = GraphTrainStep(lit.model, opt, hp["noise_exponential_sampling_lambda"], -1.0, 1.0,
                                                         use_graph=mode == "graph")
        torch.manual_seed(77)
        losses = []
        for i in range(5):
            if i == 3:
                opt.param_groups[0]["lr"] *= 0.5   # what the per-epoch cosine schedule does
            if mode == "separate":
                opt.zero_grad(set_to_none=True)
                loss = lit.training_step({"image": data[i % 2], "index": None}, i)
                loss.backward()
                opt.step()
            else:
                loss = lit.training_step({"image": data[i % 2], "index": None}, i)
            losses.append(float(loss.item()))
        m = lit.model
        return (losses, m.flat_params.clone(), m.flat_bn_stats.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(),
                opt._step, m.flat_grads.clone())

    ref = run("separate")
    for mode in ("graph", "one_call_eager"):
        got = run(mode)
        assert got[0] == ref[0], (mode, got[0], ref[0])
        for k in range(1, 5):
            assert torch.equal(got[k], ref[k]), (mode, k)
        assert got[5] == ref[5] == 5 and torch.equal(got[6], ref[6])
    assert ref[0][-1] < ref[0][0]   # and it learns


@pytest.mark.parametrize("dtype,shape", [("f32", (4, 64, 64)), ("bf16", (4, 64, 64)), ("f32", (16, 128, 128))],
                         ids=["f32_4x64", "bf16_4x64", "f32_16x128"])
def test_adam_overlap_tail_is_bitwise_the_plain_step(dtype, shape):
    """FusedAdam(overlap_tail=True): the Adam update of every gradient bucket but the last runs inside backward() on the
    engine's side stream, step() updates the rest -- parameters, moments, BatchNorm statistics, losses and the gradients
    left in .grad must equal the plain step bit for bit (five steps, a learning-rate change in between; replaces
    torch.optim.Adam.step at d3f/train_denoiser/lit_module.py:95 under Lightning's automatic optimisation)."""
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    from denoising_diffusion_deep_fake_amd.train_denoiser.lit_module import LitModule
    B, H, W = shape
    hp = dict(HP_DENOISER, batch_size=B, image_size=H, precision=dtype, augment=False)
    data = [synthetic_face_crops(B, (H, W), seed=30 + i, device="cuda") for i in range(2)]

    def run(tail):
        torch.manual_seed(5)
        lit = LitModule(**dict(hp, optimizer_overlap_tail=tail)).cuda().train()
        (opt,), _ = lit.configure_optimizers()
        assert opt.overlap_tail == tail
        torch.manual_seed(77)
        losses, early = [], 0
        for i in range(5):
            if i == 3:
                opt.param_groups[0]["lr"] *= 0.5
            opt.zero_grad(set_to_none=True)
            loss = lit.training_step({"image": data[i % 2], "index": None}, i)
            loss.backward()
            early += opt._early is not None
            opt.step()
            assert opt._early is None
            losses.append(float(loss.item()))
        m = lit.model
        return (losses, m.flat_params.clone(), m.flat_bn_stats.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(),
                m.flat_grads.clone(), opt._step, early)

    ref, got = run(False), run(True)
    assert ref[7] == 0 and got[7] == 5          # the early part really ran inside every backward
    assert got[0] == ref[0] and got[6] == ref[6] == 5
    for k in range(1, 6):
        assert torch.equal(got[k], ref[k]), k


def test_adam_overlap_tail_contract_violations_are_loud():
    """gradient accumulation or a hyper-parameter change between backward() and step() raise instead of applying a mixed
    update; gradients kept outside the flat buffer keep the whole update in step()"""
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    from denoising_diffusion_deep_fake_amd.train_denoiser.lit_module import LitModule
    hp = dict(HP_DENOISER, optimizer_overlap_tail=True)
    x = synthetic_face_crops(4, 64, seed=3, device="cuda")
    torch.manual_seed(1)
    lit = LitModule(**hp).cuda().train()
    (opt,), _ = lit.configure_optimizers()
    # (a) a second backward before step() (gradient accumulation)
    opt.zero_grad(set_to_none=True)
    lit.training_step({"image": x, "index": None}, 0).backward()
    assert opt._early is not None
    lit.training_step({"image": x, "index": None}, 1).backward()
    with pytest.raises(RuntimeError, match="another backward"):
        opt.step()
    # ... while gradients that do not land in the flat buffer directly keep the whole update in step()
    torch.manual_seed(1)
    lit = LitModule(**hp).cuda().train()
    (opt,), _ = lit.configure_optimizers()
    lit.training_step({"image": x, "index": None}, 0).backward()
    opt.step()
    opt.zero_grad(set_to_none=False)
    lit.training_step({"image": x, "index": None}, 1).backward()
    assert opt._early is None
    opt.step()
    # (b) lr changed between backward and step
    torch.manual_seed(1)
    lit = LitModule(**hp).cuda().train()
    (opt,), _ = lit.configure_optimizers()
    opt.zero_grad(set_to_none=True)
    lit.training_step({"image": x, "index": None}, 0).backward()
    opt.param_groups[0]["lr"] *= 0.5
    with pytest.raises(RuntimeError, match="changed between backward"):
        opt.step()
    # (c) the gradients edited in place between backward and step (gradient clipping / GradScaler.unscale_ would
    # otherwise reach only the last bucket's 6 % of the parameters) -- detected through the flat buffer's version counter
    torch.manual_seed(1)
    lit = LitModule(**hp).cuda().train()
    (opt,), _ = lit.configure_optimizers()
    opt.zero_grad(set_to_none=True)
    lit.training_step({"image": x, "index": None}, 0).backward()
    assert opt._early is not None
    torch.nn.utils.clip_grad_norm_(lit.model.parameters(), 1e-3)
    with pytest.raises(RuntimeError, match="modified in place"):
        opt.step()
    # ... and a later optimiser for the same module with the flag off takes the module's hook away from the first one
    from denoising_diffusion_deep_fake_amd.optim import FusedAdam
    opt2 = FusedAdam(lit.model.parameters(), lr=0.01, module=lit.model, overlap_tail=False)
    opt2.zero_grad(set_to_none=True)
    lit.training_step({"image": x, "index": None}, 1).backward()
    assert opt2._early is None and opt.early_updates == 1
    opt2.step()


def test_graph_step_metrics_rows_are_per_step_values(tmp_path):
    """The Trainer keeps logged device scalars until the next flush: with graph_step on every step's loss must be its own
    tensor (GraphTrainStep returns a stream-ordered copy, not a view of the buffer each replay overwrites) -- the rows
    of metrics.csv equal the eager run's, step for step, and differ from each other."""
    import re
    from denoising_diffusion_deep_fake_amd.train_denoiser.lit_module import LitModule
    from denoising_diffusion_deep_fake_amd.trainer import Trainer

    def rows(graph):
        d = tmp_path / ("graph" if graph else "eager")
        torch.manual_seed(5)
        lit = LitModule(**dict(HP_DENOISER, graph_step=graph, default_root_dir=str(d)))
        torch.manual_seed(6)
        tr = Trainer(max_epochs=2, log_every_n_steps=1, default_root_dir=d, enable_checkpointing=False,
                     flush_every=100).fit(lit)
        return [float(v) for v in re.findall(r"\bloss=([-+0-9.e]+)", (tr.log_dir / "metrics.csv").read_text())]

    eager, graph = rows(False), rows(True)
    assert len(eager) == 4 and len(set(eager)) == 4
    assert graph == eager, (graph, eager)


def test_graph_train_step_refuses_data_parallel_and_eval():
    from denoising_diffusion_deep_fake_amd import Unet
    from denoising_diffusion_deep_fake_amd._lib import D3FError
    from denoising_diffusion_deep_fake_amd.graph_step import GraphTrainStep
    from denoising_diffusion_deep_fake_amd.optim import FusedAdam
    net = Unet("resnet34", None, 3, 3, None).cuda().train()
    opt = FusedAdam(net.parameters(), lr=1e-3, module=net)
    step = GraphTrainStep(net, opt, 5.0)
    x = torch.zeros(2, 3, 32, 32, device="cuda")
    net.set_grad_sync(lambda seg, g: None)
    with pytest.raises(D3FError, match="single-GPU"):
        step(x)
    net.set_grad_sync(None)
    net._rt["bn_sync"] = (True, 1)   # synchronised BatchNorm statistics call back into Python inside the pass
    with pytest.raises(D3FError, match="single-GPU"):
        step(x)
    net._rt["bn_sync"] = None
    net.eval()
    with pytest.raises(D3FError, match="train mode"):
        step(x)
    with pytest.raises(TypeError):
        GraphTrainStep(net, torch.optim.Adam(net.parameters()), 5.0)


def test_denoise_mode_nets_on_two_streams_equal_the_sequential_loop(tmp_path):
    """train_deep_fake denoise mode: the two optimizer steps of a batch are independent (net a / domain a, net b /
    domain b: d3f/train_deep_fake/lit_module.py:142-181), so the trainer overlaps them on two streams
    (LitModule.optimizer_streams, trainer.optimizer_steps).  Same seeds -> the state after several batches must be
    bit-identical to Lightning's one-after-the-other loop; swap mode (coupled through the EMA teachers) stays sequential."""
    from denoising_diffusion_deep_fake_amd.train_deep_fake.lit_module import LitModule
    from denoising_diffusion_deep_fake_amd.trainer import Trainer

    def run(concurrent):
        torch.manual_seed(11)
        # (pair_fused off: the comparison partner is Lightning's one-after-the-other loop on the nets' own plans; the fused
        # two-network route of round 6 has its own tests, tests/test_gpu_pair.py)
        lit = LitModule(**dict(HP_FAKE, synthetic_length=8, concurrent_optimizers=concurrent, augment=True,
                               pair_fused=False, default_root_dir=str(tmp_path)))
        torch.manual_seed(12)
        tr = Trainer(max_epochs=1, default_root_dir=tmp_path, enable_checkpointing=False).fit(lit)
        torch.cuda.synchronize()
        streams = lit.optimizer_streams(torch.device("cuda", 0))
        return tr.global_step, {k: v.detach().clone() for k, v in lit.state_dict().items()}, streams

    steps_c, sd_c, streams_c = run(True)
    steps_s, sd_s, streams_s = run(False)
    assert streams_c is not None and len(streams_c) == 2 and streams_s is None
    assert steps_c == steps_s == 8   # 4 batches x 2 optimizers
    assert sd_c.keys() == sd_s.keys()
    for k in sd_c:
        assert torch.equal(sd_c[k], sd_s[k]), k
    swap = LitModule(**dict(HP_FAKE, mode="swap"))
    assert swap.optimizer_streams(torch.device("cuda", 0)) is None


def test_fit_on_an_image_list_uint8_batches_equal_float_batches(tmp_path):
    """End to end on the reference's dataset format (images.txt + PIL, /root/reference/d3f/dataset/image_dataset.py:19-44):
    Trainer.fit with `uint8_batches: true` (workers hand over HWC uint8 images, Normalize + ToTensor on the GPU through
    d3f_u8rgb_normalise, pinned staging) must train EXACTLY like the host transform NormalizeToTensor -- the normalisation is
    bit-identical, so with the same seeds every logged loss and every parameter is bit-equal; two spawned DataLoader workers
    in the uint8 run exercise the worker -> pinned buffer -> device path the round-5 throughput numbers were measured on."""
    import re
    import numpy as np
    from PIL import Image
    from denoising_diffusion_deep_fake_amd.train_denoiser.lit_module import LitModule
    from denoising_diffusion_deep_fake_amd.trainer import Trainer
    rng = np.random.default_rng(3)
    (tmp_path / "images").mkdir()
    names = []
    for i in range(12):
        Image.fromarray(rng.integers(0, 256, size=(64, 64, 3), dtype=np.uint8)).save(tmp_path / "images" / f"{i}.png")
        names.append(f"images/{i}.png")
    (tmp_path / "images.txt").write_text("\n".join(names) + "\n")

    def run(u8, workers):
        torch.manual_seed(11)
        hp = dict(HP_DENOISER, synthetic=False, input_image_list_path=str(tmp_path / "images.txt"), uint8_batches=u8,
                  num_workers=workers, max_epochs=2, default_root_dir=str(tmp_path / f"run_u8{int(u8)}_{workers}"))
        lit = LitModule(**hp)
        torch.manual_seed(12)
        tr = Trainer(max_epochs=2, log_every_n_steps=1, default_root_dir=tmp_path / f"run_u8{int(u8)}_{workers}",
                     enable_checkpointing=False, flush_every=100).fit(lit)
        losses = [float(v) for v in re.findall(r"\bloss=([-+0-9.e]+)", (tr.log_dir / "metrics.csv").read_text())]
        return losses, lit.model.flat_params.clone()

    l_f32, p_f32 = run(False, 0)
    l_u8, p_u8 = run(True, 0)
    assert len(l_f32) == 6 and l_f32 == l_u8 and torch.equal(p_f32, p_u8)
    l_w, p_w = run(True, 2)   # spawned workers + pinned staging: same batches (the sampler's permutation is seeded), same result
    assert l_w == l_f32 and torch.equal(p_w, p_f32)
