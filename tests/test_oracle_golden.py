"""CPU: the oracle against the golden vectors generated from the reference's own
first-party functions (tests/golden/make_golden.py), plus the structural cross-checks
available for the un-vendored U-Net."""
import numpy as np
import pytest
import torch

import oracle
from oracle import step_oracle


def test_blend_matches_reference_golden(golden_dir):
    g = np.load(golden_dir / "blend.npz")
    x = torch.from_numpy(g["x"])
    for tag in ("denoiser", "deepfake"):
        for lam in (3, 5, 8):
            for seed in (0, 7):
                torch.manual_seed(seed)
                out = oracle.blend_random_amount_of_noise_with_each_sample(x, lam)
                np.testing.assert_array_equal(out.numpy(), g[f"{tag}_lam{lam}_seed{seed}_out"])
                # deterministic core given (noise, r)
                out2 = step_oracle.blend_with_given_noise(
                    x, torch.from_numpy(g[f"{tag}_lam{lam}_seed{seed}_noise"]),
                    torch.from_numpy(g[f"{tag}_lam{lam}_seed{seed}_r"]))
                np.testing.assert_array_equal(out2.numpy(), g[f"{tag}_lam{lam}_seed{seed}_out"])


def test_sampler_matches_reference_golden(golden_dir):
    g = np.load(golden_dir / "blend.npz")
    torch.manual_seed(123)
    r = oracle.sample_random_number_from_exponential_distribution(16, 5)
    np.testing.assert_array_equal(r.numpy(), g["sampler_r_lam5"])
    assert (r > 0).all() and (r <= 1).all()


def test_loss_first_party_matches_reference_golden(golden_dir):
    g = np.load(golden_dir / "loss_first_party.npz")
    crit = oracle.MseStructuralSimilarityLoss(-1.0, 1.0)
    np.testing.assert_array_equal(
        crit.normalise_between_zero_and_one(torch.from_numpy(g["ramp"])).numpy(),
        g["ramp_normalised"])
    p, t = torch.from_numpy(g["pred"]), torch.from_numpy(g["target"])
    np.testing.assert_array_equal(crit.normalise_between_zero_and_one(p).numpy(), g["ssim_arg_pred"])
    np.testing.assert_array_equal(crit.normalise_between_zero_and_one(t).numpy(), g["ssim_arg_target"])
    mse = torch.nn.functional.mse_loss(p, t)
    for const in (1.0, 0.25):
        want = g[f"loss_with_ssim_{const}"]
        got = (mse + (1.0 - torch.tensor(const))) / 2.0
        np.testing.assert_array_equal(got.numpy(), want)


def test_denormalise_matches_reference_golden(golden_dir):
    g = np.load(golden_dir / "denormalise.npz")
    out = oracle.tensor_to_uint8_denormalised(
        torch.from_numpy(g["tensor"]), torch.from_numpy(g["mean"]), torch.from_numpy(g["std"]))
    np.testing.assert_array_equal(out.numpy(), g["uint8_rgb_hwc"])


def test_unet_structure():
    m = oracle.Unet("resnet34", None, 3, 3, None)
    assert oracle.count_parameters(m) == 24_436_659
    enc = sum(p.numel() for p in m.encoder.parameters())
    dec = sum(p.numel() for p in m.decoder.parameters())
    head = sum(p.numel() for p in m.segmentation_head.parameters())
    assert (enc, dec, head) == (21_284_672, 3_151_552, 435)
    keys = list(m.state_dict().keys())
    for k in ["encoder.conv1.weight", "encoder.layer2.0.downsample.0.weight",
              "encoder.layer4.2.bn2.running_var", "decoder.blocks.0.conv1.0.weight",
              "decoder.blocks.4.conv2.1.num_batches_tracked", "segmentation_head.0.bias"]:
        assert k in keys
    assert m.decoder.blocks[0].conv1[0].weight.shape == (256, 768, 3, 3)
    assert m.decoder.blocks[3].conv1[0].weight.shape == (32, 128, 3, 3)
    assert m.decoder.blocks[4].conv1[0].weight.shape == (16, 32, 3, 3)
    x = torch.randn(1, 3, 64, 96)
    assert m(x).shape == x.shape
    with pytest.raises(RuntimeError):
        m(torch.randn(1, 3, 48, 64))
    with pytest.raises(KeyError):
        oracle.Unet("resnet18", None, 3, 3, None)


def test_ssim_properties():
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 3, 32, 32, generator=g)
    s = oracle.ssim(x, x)
    torch.testing.assert_close(s, torch.ones(2), atol=1e-6, rtol=0)
    y = torch.rand(2, 3, 32, 32, generator=g)
    s2 = oracle.ssim(x, y)
    assert (s2 < 0.2).all()
    torch.testing.assert_close(s2, oracle.ssim(y, x))
    k = oracle.loss_oracle.gaussian_kernel_1d()
    assert k.shape == (11,) and abs(k.sum().item() - 1) < 1e-6 and k.argmax() == 5


def test_ema_schedule():
    net = torch.nn.Linear(2, 2)
    ema = oracle.EMA(net, beta=0.9999, update_every=1)
    for _ in range(101):
        with torch.no_grad():
            net.weight.add_(1.0)
        ema.update()
        torch.testing.assert_close(ema.ema_model.weight, net.weight)  # copy phase
    with torch.no_grad():
        net.weight.add_(1.0)
    ema.update()  # step 101 -> first lerp call copies once (initted) then decay(epoch=1)
    assert ema.initted.item()
    d = ema.get_current_decay()
    assert abs(d - (1 - 2 ** (-2 / 3))) < 1e-9


def test_balance_first_party_matches_reference_golden(golden_dir):
    g = np.load(golden_dir / "balance.npz")
    x = torch.from_numpy(g["x"])
    for ratio in (0.7, 0.25):
        out = oracle.blend_fixed_amount_of_noise(x, ratio, noise=torch.from_numpy(g[f"blend_ratio{ratio}_noise"]))
        assert torch.equal(out, torch.from_numpy(g[f"blend_ratio{ratio}_out"]))
    dl = oracle.difficulty_loss(torch.from_numpy(g["pred"]), x)
    assert torch.equal(dl, torch.from_numpy(g["difficulty_loss"]))
    for ncls in (10, 4):
        idx = oracle.difficulty_index(torch.from_numpy(g["losses"]), ncls)
        assert torch.equal(idx, torch.from_numpy(g[f"difficulty_index_{ncls}"]))
