"""Oracle: MseStructuralSimilarityLoss restated in plain torch (CPU).

TEST INFRASTRUCTURE (see oracle/__init__.py).
  * First-party part -- normalise_between_zero_and_one and the (mse + (1-ssim))/2
    combination -- follows d3f/loss_functions/structural_similarity_loss.py:14-26 and
    is PINNED by tests/golden/loss_first_party.npz.
  * SSIM itself is `piqa.SSIM()` (structural_similarity_loss.py:2,11,19), an
    un-vendored dependency: restated from its published defaults (window 11,
    sigma 1.5, 3 channels, value_range 1, k1 .01, k2 .03, valid filtering,
    reduction mean) -- PARITY UNPINNED.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def gaussian_kernel_1d(size=11, sigma=1.5, dtype=torch.float32):
    k = torch.arange(size, dtype=dtype) - (size - 1) / 2
    k = torch.exp(-(k ** 2) / (2 * sigma ** 2))
    return k / k.sum()


def _filter(x, g):
    """separable depthwise 'valid' gaussian filtering of [B,C,H,W]."""
    c = x.shape[1]
    kh = g.view(1, 1, -1, 1).repeat(c, 1, 1, 1)
    kw = g.view(1, 1, 1, -1).repeat(c, 1, 1, 1)
    x = F.conv2d(x, kh, groups=c)
    return F.conv2d(x, kw, groups=c)


def ssim(x, y, window_size=11, sigma=1.5, value_range=1.0, k1=0.01, k2=0.03):
    """per-image SSIM (mean over channels and valid pixels) -> [B]."""
    g = gaussian_kernel_1d(window_size, sigma, x.dtype).to(x.device)
    c1 = (k1 * value_range) ** 2
    c2 = (k2 * value_range) ** 2
    mu_x, mu_y = _filter(x, g), _filter(y, g)
    mu_xx, mu_yy, mu_xy = mu_x ** 2, mu_y ** 2, mu_x * mu_y
    s_xx = _filter(x ** 2, g) - mu_xx
    s_yy = _filter(y ** 2, g) - mu_yy
    s_xy = _filter(x * y, g) - mu_xy
    cs = (2 * s_xy + c2) / (s_xx + s_yy + c2)
    ss = (2 * mu_xy + c1) / (mu_xx + mu_yy + c1) * cs
    return ss.mean(dim=(1, 2, 3))


class MseStructuralSimilarityLoss(nn.Module):
    """structural_similarity_loss.py:5-26 with SSIM restated above."""

    def __init__(self, input_min_value, input_max_value):
        super().__init__()
        self.input_min_value = input_min_value
        self.input_max_value = input_max_value

    def forward(self, prediction, target):
        mse_loss = F.mse_loss(prediction, target)
        prediction = self.normalise_between_zero_and_one(prediction)
        target = self.normalise_between_zero_and_one(target)
        ssim_loss = 1.0 - ssim(prediction, target).mean()
        return (mse_loss + ssim_loss) / 2.0

    def normalise_between_zero_and_one(self, x):
        x = (x - self.input_min_value) / (self.input_max_value - self.input_min_value)
        return x.clip(0.0, 1.0)
