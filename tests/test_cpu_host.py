"""CPU: host-side mirror of the reference's Python surface (no kernels involved)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from denoising_diffusion_deep_fake_amd.dataset import ImageDataset, SyntheticFaceDataset
from denoising_diffusion_deep_fake_amd.dataset.image_dataset import NormalizeToTensor
from denoising_diffusion_deep_fake_amd.lightning import AttributeDict, LightningModule

ROOT = Path(__file__).resolve().parent.parent


def test_image_dataset_contract(tmp_path):
    from PIL import Image
    (tmp_path / "imgs").mkdir()
    rng = np.random.default_rng(0)
    arrs = []
    for i in range(3):
        a = rng.integers(0, 256, size=(32, 32, 3), dtype=np.uint8)
        Image.fromarray(a).save(tmp_path / "imgs" / f"{i}.png")
        arrs.append(a)
    (tmp_path / "images.txt").write_text("imgs/0.png\nimgs/1.png\nimgs/2.png\n")
    ds = ImageDataset(tmp_path / "images.txt", transform=NormalizeToTensor([0.5] * 3, [0.5] * 3))
    assert len(ds) == 3
    item = ds[1]
    assert set(item) == {"image", "index"} and item["index"] == 1
    img = item["image"]
    assert img.shape == (3, 32, 32) and img.dtype == torch.float32
    want = (arrs[1].astype(np.float32) / 255 - 0.5) / 0.5   # RGB order, like cv2 BGR->RGB in the reference
    np.testing.assert_allclose(img.numpy(), want.transpose(2, 0, 1), rtol=0, atol=1e-6)
    raw = ImageDataset(tmp_path / "images.txt")[0]["image"]
    assert raw.dtype == np.uint8 and raw.shape == (32, 32, 3)
    # `uint8_batches: true`: the workers hand over the decoded HWC uint8 image; the default collate stacks [B, H, W, 3]
    from torch.utils.data import DataLoader
    from denoising_diffusion_deep_fake_amd.dataset.image_dataset import ToUint8Tensor
    u8 = ImageDataset(tmp_path / "images.txt", transform=ToUint8Tensor())
    batch = next(iter(DataLoader(u8, batch_size=3)))
    assert batch["image"].dtype == torch.uint8 and tuple(batch["image"].shape) == (3, 32, 32, 3)
    assert np.array_equal(batch["image"].numpy(), np.stack(arrs)) and batch["index"].tolist() == [0, 1, 2]


def test_synthetic_dataset_matches_oracle_generator():
    import oracle
    ds = SyntheticFaceDataset(4, 64, seed=1234)
    x = ds[2]["image"]
    assert x.shape == (3, 64, 64) and x.abs().max() <= 1
    assert torch.equal(x, oracle.synthetic_face_crops(1, 64, seed=1236)[0])


def test_hparams_surface():
    class M(LightningModule):
        def __init__(self, **kwargs):
            super().__init__()
            self.save_hyperparameters()
            self.lin = torch.nn.Linear(2, 2)

    m = M(batch_size=4, encoder_name="resnet34", mean=[0.5, 0.5, 0.5])
    assert m.hparams.batch_size == 4 and m.hparams["encoder_name"] == "resnet34"
    assert isinstance(m.hparams, AttributeDict) and m.global_step == 0 and m.current_epoch == 0
    m.log("loss", torch.tensor(1.5))
    assert float(m._logged["loss"]) == 1.5
    assert m.device.type == "cpu"


def test_checkpoint_roundtrip(tmp_path):
    class M(LightningModule):
        def __init__(self, **kwargs):
            super().__init__()
            self.save_hyperparameters()
            self.lin = torch.nn.Linear(self.hparams.width, 2)

    m = M(width=3, learning_rate=0.1)
    torch.save({"state_dict": m.state_dict(), "hyper_parameters": dict(m.hparams)}, tmp_path / "a.ckpt")
    m2 = M.load_from_checkpoint(tmp_path / "a.ckpt", learning_rate=0.5)
    assert m2.hparams.width == 3 and m2.hparams.learning_rate == 0.5
    assert torch.equal(m2.lin.weight, m.lin.weight)


def test_ema_decay_schedule_matches_oracle():
    import oracle
    from denoising_diffusion_deep_fake_amd.optim import EMA

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))

    ours = EMA.__new__(EMA)
    ours.update_after_step, ours.inv_gamma, ours.power, ours.min_value, ours.beta = 100, 1.0, 2 / 3, 0.0, 0.9999
    ref = oracle.EMA(Fake(), beta=0.9999)
    for step in (0, 50, 101, 102, 150, 1000, 10 ** 7):
        ref.step.fill_(step)
        assert abs(ours.get_current_decay(step) - ref.get_current_decay()) < 1e-12


def test_balance_host_logic_matches_reference_golden(tmp_path):
    """balance_training_images host arithmetic (no kernels): binning against vectors produced by the reference,
    concatenation of validation outputs, and the class list file."""
    from pathlib import Path
    import oracle  # noqa: F401  (same golden file pins the oracle in test_oracle_golden.py)
    from denoising_diffusion_deep_fake_amd.balance_training_images.lit_module import LitModule
    g = np.load(Path(__file__).resolve().parent / "golden" / "balance.npz")
    losses = torch.from_numpy(g["losses"])
    for ncls in (10, 4):
        stub = type("S", (), {"hparams": AttributeDict(number_of_classes=ncls)})()
        idx = LitModule.compute_difficulty_index_for_each_loss(stub, losses)
        assert torch.equal(idx, torch.from_numpy(g[f"difficulty_index_{ncls}"]))
    outs = [{"index": torch.tensor([2, 0]), "loss": torch.tensor([0.5, 0.1])},
            {"index": torch.tensor([1]), "loss": torch.tensor([0.9])}]
    cat = LitModule.concat_validation_output(None, outs)
    assert cat["index"].tolist() == [2, 0, 1] and torch.allclose(cat["loss"], torch.tensor([0.5, 0.1, 0.9]))
    (tmp_path / "images.txt").write_text("a.png\nb.png\nc.png\n")
    stub = type("S", (), {"hparams": AttributeDict(input_image_list_path=str(tmp_path / "images.txt"), synthetic=False),
                          "_data_path": lambda self: self.hparams["input_image_list_path"]})()
    LitModule.write_output_list(stub, str(tmp_path / "out.txt"), cat["index"], torch.tensor([3, 0, 1]))
    assert (tmp_path / "out.txt").read_text() == "a.png\t0\nb.png\t1\nc.png\t3\n"


def test_cli_has_the_reference_commands():
    from denoising_diffusion_deep_fake_amd.main import cli
    assert {"train", "denoise", "balance"} <= set(cli.commands)
    assert {"new", "resume", "modify"} <= set(cli.commands["train"].commands)


def test_bench_self_launch_command(monkeypatch):
    """bench.py --gpus N without a launcher: the parent builds the torch.distributed.run command for a CHILD process and
    never initialises the GPU itself (an exec / HIP call in the parent is what the GPU box forbids)."""
    import importlib.util
    import subprocess
    import sys
    import types
    spec = importlib.util.spec_from_file_location("d3f_bench", str(ROOT / "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                   # the child's exit code is relayed
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert not torch.cuda.is_initialized()


def test_shift_scale_rotate_theta_is_cv2_warp_affine():
    """ShiftScaleRotate.theta must describe cv2.warpAffine(img, M) with M = getRotationMatrix2D(centre, angle, scale)
    + (dx W, dy H) -- what albumentations.ShiftScaleRotate applies (d3f/train_deep_fake/lit_module.py:102-108).
    Checked in pixel space: for every output pixel, M applied to the source position affine_grid names must give
    the output pixel back."""
    import math
    import torch.nn.functional as F
    from denoising_diffusion_deep_fake_amd.train_deep_fake.lit_module import ShiftScaleRotate
    H, W = 24, 40
    angle, scale = torch.tensor([11.0, -15.0, 0.0]), torch.tensor([1.08, 0.9, 1.0])
    dx, dy = torch.tensor([0.2, -0.13, 0.0]), torch.tensor([-0.05, 0.2, 0.0])
    th = ShiftScaleRotate.theta(angle, scale, dx, dy, H, W)
    assert torch.allclose(th[2], torch.tensor([[1.0, 0, 0], [0, 1.0, 0]]), atol=1e-7)   # null draw = identity
    grid = F.affine_grid(th, [3, 1, H, W], align_corners=False)        # [B, H, W, 2] normalised source positions
    sx = ((grid[..., 0] + 1) * W - 1) / 2                                # -> source pixel coordinates
    sy = ((grid[..., 1] + 1) * H - 1) / 2
    cx, cy = (W - 1) / 2, (H - 1) / 2
    for b in range(3):
        a = math.radians(angle[b].item())
        al, be = scale[b].item() * math.cos(a), scale[b].item() * math.sin(a)
        m = torch.tensor([[al, be, (1 - al) * cx - be * cy + dx[b].item() * W],
                          [-be, al, be * cx + (1 - al) * cy + dy[b].item() * H]])
        ox = m[0, 0] * sx[b] + m[0, 1] * sy[b] + m[0, 2]
        oy = m[1, 0] * sx[b] + m[1, 1] * sy[b] + m[1, 2]
        xs = torch.arange(W, dtype=torch.float32).expand(H, W)
        ys = torch.arange(H, dtype=torch.float32).reshape(H, 1).expand(H, W)
        assert (ox - xs).abs().max() < 1e-3 and (oy - ys).abs().max() < 1e-3


def test_combined_loader_reiterates_the_short_loader():
    """mode max_size_cycle: the shorter loader starts a FRESH pass when it runs out (Lightning's CombinedLoader),
    it is not replayed from a cache of its first pass."""
    from denoising_diffusion_deep_fake_amd.trainer import CombinedLoader

    class Counting:
        def __init__(self, n):
            self.n, self.passes = n, 0

        def __len__(self):
            return self.n

        def __iter__(self):
            self.passes += 1
            return iter([(self.passes, i) for i in range(self.n)])
    long, short = Counting(5), Counting(2)
    got = list(CombinedLoader({"a": long, "b": short}))
    assert len(got) == 5 and [g["a"][1] for g in got] == [0, 1, 2, 3, 4]
    assert [g["b"] for g in got] == [(1, 0), (1, 1), (2, 0), (2, 1), (3, 0)]
    assert long.passes == 1 and short.passes == 3


def test_shard_loader_keeps_loader_options_and_refuses_custom_samplers():
    """generator / worker_init_fn / timeout survive the re-build; a sampler that is neither sequential nor
    random (e.g. WeightedRandomSampler over the `balance` classes) is refused instead of silently replaced."""
    import pytest
    from torch.utils.data import DataLoader, WeightedRandomSampler
    from torch.utils.data.distributed import DistributedSampler
    from denoising_diffusion_deep_fake_amd.dataset.image_dataset import SyntheticFaceDataset
    from denoising_diffusion_deep_fake_amd.trainer import shard_loader
    ds = SyntheticFaceDataset(8, 32)
    g = torch.Generator().manual_seed(5)

    def init(worker_id):
        pass
    base = DataLoader(ds, batch_size=2, shuffle=True, generator=g, worker_init_fn=init, timeout=0)
    sharded = shard_loader(base, 2, 1, seed=9)
    assert isinstance(sharded.sampler, DistributedSampler) and sharded.sampler.seed == 9 and sharded.sampler.shuffle
    assert sharded.generator is g and sharded.worker_init_fn is init and sharded.batch_size == 2
    assert shard_loader(base, 1, 0) is base
    weighted = DataLoader(ds, batch_size=2, sampler=WeightedRandomSampler([1.0] * 8, 8))
    with pytest.raises(TypeError, match="WeightedRandomSampler"):
        shard_loader(weighted, 2, 0)


def test_single_gpu_epoch_permutation_is_replayable():
    """mid-epoch resume passes over the first `done` batches, so the epoch's permutation must be a function
    of (loader seed, epoch) -- also on one GPU, where the loader carries a plain RandomSampler."""
    from torch.utils.data import DataLoader
    from denoising_diffusion_deep_fake_amd.dataset.image_dataset import SyntheticFaceDataset
    from denoising_diffusion_deep_fake_amd.trainer import _set_epoch

    def epoch_indices(epoch, seed, global_seed):
        torch.manual_seed(global_seed)  # another process, another global RNG state
        loader = DataLoader(SyntheticFaceDataset(12, 32), batch_size=4, shuffle=True)
        _set_epoch(loader, epoch, seed)
        return [int(i) for b in loader for i in b["index"]]
    assert epoch_indices(3, 77, 1) == epoch_indices(3, 77, 2)
    assert epoch_indices(3, 77, 1) != epoch_indices(4, 77, 1)
    assert epoch_indices(3, 77, 1) != epoch_indices(3, 78, 1)
    own = torch.Generator().manual_seed(1)   # a generator the user supplied is left alone
    loader = DataLoader(SyntheticFaceDataset(12, 32), batch_size=4, shuffle=True, generator=own)
    _set_epoch(loader, 0, 5)
    assert loader.sampler.generator is own


def test_lightning_loop_record_gives_the_mid_epoch_position():
    """a checkpoint the reference wrote with ModelCheckpoint(train_time_interval=...) (d3f/train_deep_fake/
    lit_module.py:127-140) carries pytorch_lightning 1.x's `loops` record, not our d3f key"""
    from denoising_diffusion_deep_fake_amd.trainer import lightning_batches_done
    prog = {"total": {"ready": 130, "completed": 130, "started": 130, "processed": 130},
            "current": {"ready": 30, "completed": 30, "started": 30, "processed": 30}, "is_last_batch": False}
    ck = {"epoch": 1, "loops": {"fit_loop": {"epoch_loop.batch_progress": prog}}}
    assert lightning_batches_done(ck) == 30
    ck["loops"]["fit_loop"]["epoch_loop.batch_progress"] = dict(prog, is_last_batch=True)
    assert lightning_batches_done(ck) is None
    assert lightning_batches_done({"epoch": 1}) is None and lightning_batches_done({"loops": None}) is None


def test_console_script_entry_point_resolves():
    """the reference installs `d3f = d3f.main:cli` (setup.py:7-11); pyproject.toml declares the same script and the
    packages it needs, and the target resolves to the click group with the reference's commands"""
    import importlib
    from pathlib import Path

    import tomli
    root = Path(__file__).resolve().parent.parent
    meta = tomli.loads((root / "pyproject.toml").read_text())
    target = meta["project"]["scripts"]["d3f"]
    assert target == "d3f.main:cli"
    mod, attr = target.split(":")
    cli = getattr(importlib.import_module(mod), attr)
    assert {"train", "denoise", "balance"} <= set(cli.commands)
    found = set(meta["tool"]["setuptools"]["packages"]["find"]["include"])
    assert {"d3f", "denoising_diffusion_deep_fake_amd*"} <= found
    data = meta["tool"]["setuptools"]["package-data"]["denoising_diffusion_deep_fake_amd"]
    assert any(p.endswith("libd3f_hip.so") for p in data) and any(p.endswith(".yml") for p in data)


def test_division_by_multiply_high_is_exact(tmp_path):
    """csrc/common.h: the conv prologue, the max-pool backward and the pack kernels decode indices with
    q = umulhi(n, mul) >> shr instead of an integer division; the host-made constants must make that exact for every
    dividend below 2^31 (tests/aux/fast_div_check.hip restates the device formula with a 64-bit product)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("no hipcc")
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "fast_div_check"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++20",
                    "-I", str(root / "denoising_diffusion_deep_fake_amd" / "csrc"),
                    str(root / "tests" / "aux" / "fast_div_check.hip"), "-o", str(exe)], check=True, timeout=600)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "bad 0" in out.stdout, out.stdout + out.stderr


def test_fused_adam_host_side_contracts():
    """FusedAdam's host logic that needs no GPU: the plain zero_grad (set_to_none) clears every .grad like
    torch.optim.Optimizer.zero_grad, set_to_none=False keeps torch's behaviour; overlap_tail registers / leaves the
    early-update hook of its Unet; the hparam reaches the optimiser from both LitModules (torch.optim.Adam stands at
    d3f/train_denoiser/lit_module.py:95 and d3f/train_deep_fake/lit_module.py:116-120 in the reference)."""
    import torch

    from denoising_diffusion_deep_fake_amd import Unet
    from denoising_diffusion_deep_fake_amd.optim import FusedAdam
    net = Unet("resnet34", None, 3, 3, None)
    opt = FusedAdam(net.parameters(), lr=1e-3, module=net)
    assert not opt.overlap_tail and net._rt.get("early_update") is None
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    opt.zero_grad(set_to_none=False)
    assert all(p.grad is not None for p in net.parameters())
    opt.zero_grad()
    assert all(p.grad is None for p in net.parameters())
    opt2 = FusedAdam(net.parameters(), lr=1e-3, module=net, overlap_tail=True)
    assert opt2.overlap_tail and net._rt["early_update"] == opt2._early_update
    with pytest.raises(NotImplementedError):
        opt2.param_groups[0]["weight_decay"] = 0.1
        opt2._hyper()
    from denoising_diffusion_deep_fake_amd.train_denoiser.lit_module import LitModule
    hp = dict(batch_size=2, learning_rate=0.02, max_epochs=1, cosine_scheduler_max_epoch=1, num_workers=0,
              encoder_name="resnet34", noise_exponential_sampling_lambda=5, mean=[128] * 3, std=[128] * 3, synthetic=True,
              image_size=64, augment=False)
    (o,), _ = LitModule(**hp).configure_optimizers()
    assert not o.overlap_tail
    (o,), _ = LitModule(**dict(hp, optimizer_overlap_tail=True)).configure_optimizers()
    assert o.overlap_tail


def test_unet_pair_host_protocol():
    """UnetPair (the two nets of train_deep_fake's denoise mode as one set of launches, d3f/train_deep_fake/lit_module.py:142-181)
    without a GPU: constructor errors, and that copying / pickling whatever holds a pair never duplicates its plans (C
    handles) -- a copy starts without engines and is bound to the COPIED networks."""
    import copy
    import pickle

    import pytest
    from denoising_diffusion_deep_fake_amd import D3FError, Unet, UnetPair
    a, b = Unet("resnet18", None, 3, 3, None), Unet("resnet18", None, 3, 3, None)
    with pytest.raises(TypeError):
        UnetPair(a, a)
    with pytest.raises(TypeError):
        UnetPair(a, object())
    with pytest.raises(ValueError):
        UnetPair(a, Unet("resnet34", None, 3, 3, None))
    with pytest.raises(ValueError):
        UnetPair(a, Unet("resnet18", None, 3, 3, None, compute_dtype="bf16"))
    pair = UnetPair(a, b)
    holder = copy.deepcopy({"pair": pair, "a": a, "b": b})
    assert holder["pair"].nets == (holder["a"], holder["b"]) and holder["pair"].nets[0] is not a
    assert holder["pair"]._engines == {} and holder["pair"].last_engine is None
    again = pickle.loads(pickle.dumps(pair))
    assert again._engines == {} and len(again.nets) == 2
    import torch
    with pytest.raises(D3FError):
        pair(torch.zeros(1, 3, 32, 32), torch.zeros(1, 3, 32, 32))   # CPU tensors: there is no fallback
    with pytest.raises(ValueError):
        a.set_plan_nets(3)
    assert a.set_plan_nets(2) is a and a.plan_nets == 2 and copy.deepcopy(a).plan_nets == 2
