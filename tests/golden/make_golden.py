"""Generate tests/golden/*.npz by running the REFERENCE's own first-party functions.

Run once in the build container (needs /root/reference; never runs on the GPU box):
    python tests/golden/make_golden.py

The reference's third-party imports (cv2, pytorch_lightning, segmentation_models_pytorch,
piqa, kornia, albumentations, ema_pytorch, torchvision, matplotlib...) are not installed
anywhere offline, so they are replaced by inert mocks *only so the reference modules
import*; every golden value below comes from code that does not touch a mock:
  * LitModule.blend_random_amount_of_noise_with_each_sample and
    LitModule.sample_random_number_from_exponential_distribution
        (d3f/train_denoiser/lit_module.py:128-153; d3f/train_deep_fake/lit_module.py:208-233)
  * MseStructuralSimilarityLoss.normalise_between_zero_and_one and .forward with the
    piqa SSIM term replaced by a constant (pins the (mse + (1 - ssim)) / 2 combination,
    d3f/loss_functions/structural_similarity_loss.py:14-26)
  * LitModule.tensor_cv2_to_denormalised integer semantics
        (d3f/train_deep_fake/lit_module.py:285-296; the trailing cv2 RGB->BGR flip is
        mocked, so the fixture stores the tensor handed to it = HWC uint8 RGB)
  * balance_training_images LitModule: blend_fixed_amount_of_noise_with_each_sample,
    compute_difficulty_loss, compute_difficulty_index_for_each_loss
        (d3f/balance_training_images/lit_module.py:109-121, 139-142, 181-193)
Only data (inputs / outputs) is written; no reference source is copied.
"""
import sys
import types
from pathlib import Path
from types import SimpleNamespace
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
OUT = Path(__file__).resolve().parent


def _install_stubs():
    class _LightningModule(torch.nn.Module):
        pass

    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = _LightningModule
    pl.Trainer = mock.MagicMock()
    cb = types.ModuleType("pytorch_lightning.callbacks")
    cb.LearningRateMonitor = mock.MagicMock()
    cb.ModelCheckpoint = mock.MagicMock()
    pl.callbacks = cb
    sys.modules["pytorch_lightning"] = pl
    sys.modules["pytorch_lightning.callbacks"] = cb
    for name in ["cv2", "segmentation_models_pytorch", "piqa", "kornia",
                 "kornia.augmentation", "albumentations", "albumentations.pytorch",
                 "ema_pytorch", "torchvision", "torchvision.transforms",
                 "torchvision.transforms.functional", "torchvision.io", "torchvision.utils",
                 "matplotlib", "matplotlib.pyplot",
                 "PIL", "PIL.Image", "tqdm"]:
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock(name=name)


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    from d3f.train_denoiser.lit_module import LitModule as DenoiseLit
    from d3f.train_deep_fake.lit_module import LitModule as FakeLit
    from d3f.loss_functions.structural_similarity_loss import MseStructuralSimilarityLoss

    # ---- (1) noise blend, incl. RNG call order -------------------------------------
    blend = {}
    gx = torch.Generator().manual_seed(99)
    x = torch.tanh(torch.randn(4, 3, 8, 8, generator=gx))
    blend["x"] = x.numpy()
    for cls, tag in [(DenoiseLit, "denoiser"), (FakeLit, "deepfake")]:
        for lam in (3, 5, 8):
            for seed in (0, 7):
                self = SimpleNamespace(
                    hparams=SimpleNamespace(noise_exponential_sampling_lambda=lam),
                    device=torch.device("cpu"))
                self.sample_random_number_from_exponential_distribution = (
                    lambda b, l, _s=self, _c=cls:
                    _c.sample_random_number_from_exponential_distribution(_s, b, l))
                torch.manual_seed(seed)
                out = cls.blend_random_amount_of_noise_with_each_sample(self, x)
                # the r the reference drew (same stream position: after randn_like)
                torch.manual_seed(seed)
                noise = torch.randn_like(x)
                r = cls.sample_random_number_from_exponential_distribution(self, 4, lam)
                blend[f"{tag}_lam{lam}_seed{seed}_out"] = out.numpy()
                blend[f"{tag}_lam{lam}_seed{seed}_r"] = r.numpy()
                blend[f"{tag}_lam{lam}_seed{seed}_noise"] = noise.numpy()
    # sampler distribution check: fixed uniform draws
    self = SimpleNamespace(device=torch.device("cpu"))
    torch.manual_seed(123)
    y = torch.rand(size=(16, 1, 1, 1))
    torch.manual_seed(123)
    blend["sampler_y"] = y.numpy()
    blend["sampler_r_lam5"] = DenoiseLit.sample_random_number_from_exponential_distribution(
        self, 16, 5).numpy()
    np.savez_compressed(OUT / "blend.npz", **blend)

    # ---- (2)+(3) loss first-party arithmetic ------------------------------------------
    loss = {}
    crit = MseStructuralSimilarityLoss.__new__(MseStructuralSimilarityLoss)
    torch.nn.Module.__init__(crit)
    crit.input_min_value, crit.input_max_value = -1.0, 1.0
    crit.mse = torch.nn.MSELoss()
    ramp = torch.linspace(-1.75, 1.75, 29).reshape(1, 1, 1, 29)
    loss["ramp"] = ramp.numpy()
    loss["ramp_normalised"] = crit.normalise_between_zero_and_one(ramp).numpy()
    g = torch.Generator().manual_seed(5)
    p = torch.randn(2, 3, 16, 16, generator=g) * 0.8
    t = torch.tanh(torch.randn(2, 3, 16, 16, generator=g))
    loss["pred"], loss["target"] = p.numpy(), t.numpy()
    for const in (1.0, 0.25):
        seen = {}

        def fake_ssim(a, b, _c=const, _seen=seen):
            _seen["a"], _seen["b"] = a.clone(), b.clone()
            return torch.tensor(_c)

        crit.ssim = fake_ssim
        loss[f"loss_with_ssim_{const}"] = crit.forward(p, t).numpy()
        loss["ssim_arg_pred"] = seen["a"].numpy()
        loss["ssim_arg_target"] = seen["b"].numpy()
    np.savez_compressed(OUT / "loss_first_party.npz", **loss)

    # ---- (4) uint8 de-normalisation ----------------------------------------------------
    den = {}
    g = torch.Generator().manual_seed(11)
    tin = torch.randn(1, 3, 6, 5, generator=g) * 1.4  # exceeds [-1,1] -> exercises clamp
    mean = torch.tensor([0.5, 0.4, 0.6])
    std = torch.tensor([0.5, 0.25, 0.3])
    captured = {}
    import d3f.train_deep_fake.lit_module as fl

    def fake_cvt(img, code):
        captured["img"] = np.array(img)
        return img

    fl.cv2.cvtColor = fake_cvt
    fl.cv2.COLOR_RGB2BGR = 4
    self = SimpleNamespace()
    FakeLit.tensor_cv2_to_denormalised(self, tin.clone(), mean, std)
    den["tensor"], den["mean"], den["std"] = tin.numpy(), mean.numpy(), std.numpy()
    den["uint8_rgb_hwc"] = captured["img"]
    np.savez_compressed(OUT / "denormalise.npz", **den)
    # ---- (5) balance_training_images first-party arithmetic -----------------------------
    sys.modules.setdefault("d3f.helpers", mock.MagicMock(name="d3f.helpers"))
    from d3f.balance_training_images.lit_module import LitModule as BalanceLit
    bal = {}
    g = torch.Generator().manual_seed(21)
    xb = torch.tanh(torch.randn(5, 3, 8, 8, generator=g))
    bal["x"] = xb.numpy()
    for ratio in (0.7, 0.25):
        self = SimpleNamespace(hparams=SimpleNamespace(ratio_of_noise=ratio), device=torch.device("cpu"))
        torch.manual_seed(3)
        bal[f"blend_ratio{ratio}_out"] = BalanceLit.blend_fixed_amount_of_noise_with_each_sample(self, xb).numpy()
        torch.manual_seed(3)
        bal[f"blend_ratio{ratio}_noise"] = torch.randn_like(xb).numpy()
    pred = torch.randn(5, 3, 8, 8, generator=g)
    bal["pred"] = pred.numpy()
    dl = BalanceLit.compute_difficulty_loss(SimpleNamespace(), pred, xb)
    bal["difficulty_loss"] = dl.numpy()
    losses = torch.tensor([0.31, 0.05, 0.95, 0.5, 0.050001, 0.949, 0.2, 0.77, 0.95, 0.05])
    bal["losses"] = losses.numpy()
    for ncls in (10, 4):
        self = SimpleNamespace(hparams=SimpleNamespace(number_of_classes=ncls))
        bal[f"difficulty_index_{ncls}"] = BalanceLit.compute_difficulty_index_for_each_loss(self, losses).numpy()
    np.savez_compressed(OUT / "balance.npz", **bal)
    print("wrote", [p.name for p in OUT.glob("*.npz")])


if __name__ == "__main__":
    main()
