"""Oracle: the training-step arithmetic around the U-Net, restated on CPU torch.

TEST INFRASTRUCTURE (see oracle/__init__.py).
PINNED by tests/golden (generated from the reference's own functions):
  * blend_random_amount_of_noise_with_each_sample / sample_random_number_from_
    exponential_distribution -- d3f/train_denoiser/lit_module.py:128-153 (identical
    bodies at d3f/train_deep_fake/lit_module.py:208-233); RNG call order is
    randn_like(batch) first, then rand(B,1,1,1).
  * tensor_to_uint8_denormalised -- d3f/train_deep_fake/lit_module.py:285-300
    (.int() truncation BEFORE clamp).
  * blend_fixed_amount_of_noise / difficulty_loss / difficulty_index --
    d3f/balance_training_images/lit_module.py:109-121, 139-142, 181-193 (tests/golden/balance.npz).
PARITY UNPINNED (un-vendored `ema_pytorch`, restated from upstream defaults,
SURVEY.md Appendix A.3): EMA.
swap_step restates the reference's own orchestration (d3f/train_deep_fake/lit_module.py:183-206) on top of
those parts.
"""
import copy
import math

import torch
import torch.nn.functional as F


def sample_random_number_from_exponential_distribution(batch_size, lam, device="cpu",
                                                       generator=None):
    # lit_module.py:141-153
    y = torch.rand(size=(batch_size, 1, 1, 1), device=device, generator=generator)
    c = 1 / math.exp(lam)
    return 1 / lam * torch.log(1 / (y * (1 - c) + c))


def blend_random_amount_of_noise_with_each_sample(batch, lam, generator=None):
    # lit_module.py:128-139 -- randn first, rand second
    if generator is None:
        noise = torch.randn_like(batch)
    else:
        noise = torch.randn(batch.shape, dtype=batch.dtype, device=batch.device,
                            generator=generator)
    r = sample_random_number_from_exponential_distribution(
        batch.shape[0], lam, batch.device, generator)
    return torch.sqrt(1 - r) * batch + torch.sqrt(r) * noise


def blend_with_given_noise(batch, noise, r):
    """Deterministic core of the blend (what the HIP kernel is checked against when
    the Gaussian noise / ratio are supplied explicitly)."""
    r = r.view(-1, 1, 1, 1)
    return torch.sqrt(1 - r) * batch + torch.sqrt(r) * noise


def blend_fixed_amount_of_noise(batch, ratio_of_noise, noise=None):
    # balance_training_images/lit_module.py:109-121
    if noise is None:
        noise = torch.randn_like(batch)
    r = torch.ones((batch.shape[0], 1, 1, 1), device=batch.device) * ratio_of_noise
    return torch.sqrt(1 - r) * batch + torch.sqrt(r) * noise


def difficulty_loss(predicted, target):
    # balance_training_images/lit_module.py:139-142: per-image mean absolute error
    return torch.abs(predicted - target).mean(dim=(1, 2, 3))


def difficulty_index(loss, number_of_classes):
    # balance_training_images/lit_module.py:181-193: min-max normalise, clamp below 1, bin
    loss_normalised = (loss - loss.min()) / (loss.max() - loss.min())
    loss_normalised = loss_normalised.clamp(0, 0.99999)
    return (loss_normalised * number_of_classes).long()


def tensor_to_uint8_denormalised(tensor, mean, std):
    """train_deep_fake/lit_module.py:285-296 up to (and excluding) the cv2 RGB->BGR
    flip: [1,3,H,W] float -> HWC uint8 RGB."""
    t = tensor.squeeze(0).clone()
    t *= std.reshape(3, 1, 1) * 255
    t += mean.reshape(3, 1, 1) * 255
    t = t.permute(1, 2, 0)
    t = t.int()
    t = t.clamp(0, 255)
    return t.to(torch.uint8)


class EMA(torch.nn.Module):
    """ema_pytorch.EMA(model, beta, update_every, include_online_model=False) with
    upstream defaults update_after_step=100, inv_gamma=1, power=2/3, min_value=0."""

    def __init__(self, model, beta=0.9999, update_every=1, update_after_step=100,
                 inv_gamma=1.0, power=2 / 3, min_value=0.0):
        super().__init__()
        self.online_model = [model]  # not registered (include_online_model=False)
        self.ema_model = copy.deepcopy(model)
        self.ema_model.requires_grad_(False)
        self.beta, self.update_every, self.update_after_step = beta, update_every, update_after_step
        self.inv_gamma, self.power, self.min_value = inv_gamma, power, min_value
        self.register_buffer("initted", torch.tensor(False))
        self.register_buffer("step", torch.tensor(0))

    def get_current_decay(self):
        epoch = max(self.step.item() - self.update_after_step - 1, 0.0)
        if epoch <= 0:
            return 0.0
        value = 1 - (1 + epoch / self.inv_gamma) ** -self.power
        return min(max(value, self.min_value), self.beta)

    @torch.no_grad()
    def copy_params_from_model_to_ema(self):
        src = self.online_model[0]
        for (_, e), (_, o) in zip(self.ema_model.named_parameters(), src.named_parameters()):
            e.copy_(o)
        for (_, e), (_, o) in zip(self.ema_model.named_buffers(), src.named_buffers()):
            e.copy_(o)

    @torch.no_grad()
    def update(self):
        step = self.step.item()
        self.step += 1
        if step % self.update_every != 0:
            return
        if step <= self.update_after_step:
            self.copy_params_from_model_to_ema()
            return
        if not self.initted.item():
            self.copy_params_from_model_to_ema()
            self.initted.fill_(True)
        decay = self.get_current_decay()
        src = self.online_model[0]
        for (_, e), (_, o) in zip(self.ema_model.named_parameters(), src.named_parameters()):
            if e.is_floating_point():
                e.lerp_(o, 1 - decay)
        for (_, e), (_, o) in zip(self.ema_model.named_buffers(), src.named_buffers()):
            if e.is_floating_point():
                e.lerp_(o, 1 - decay)

    def forward(self, *a, **k):
        return self.ema_model(*a, **k)


def synthetic_face_crops(batch, size, seed=1234, device="cpu"):
    """spatially-correlated fp32 NCHW crops in [-1,1]."""
    h, w = (size, size) if isinstance(size, int) else size  # an int, or (height, width)
    g = torch.Generator().manual_seed(seed)
    low = torch.randn(batch, 3, h // 16, w // 16, generator=g) * 0.5
    fine = torch.randn(batch, 3, h, w, generator=g) * 0.05
    x = F.interpolate(low, size=(h, w), mode="bilinear", align_corners=False) + fine
    return torch.tanh(x).to(device)


def training_step(model, criterion, optimizer, image, noise, r):
    """One noisy->clean x0-prediction step (train_denoiser/lit_module.py:107-126 +
    Lightning's zero_grad/backward/step), with the Gaussian noise and the blend
    ratio supplied so that both sides of a parity test see identical inputs."""
    noisy = blend_with_given_noise(image, noise, r)
    optimizer.zero_grad(set_to_none=True)
    pred = model(noisy)
    loss = criterion(pred, image)
    loss.backward()
    optimizer.step()
    return loss.detach(), pred.detach()


def swap_step(real, real_model, fake_model, criterion, noise, r):
    """d3f/train_deep_fake/lit_module.py:183-206 (`training_swap_step_for_one_model`) with the Gaussian noise and
    the blend ratio supplied: the EMA teacher of the OTHER domain is updated, renders a fake from the real batch
    under no_grad (the EMA wrapper is a registered sub-module of a training LightningModule, so its BatchNorm
    layers use batch statistics), the fake is noised, and the student denoises it back to the real image.
    Returns (loss, swap_difference, fake, prediction); loss carries the graph."""
    fake_model.update()                                          # :185
    with torch.no_grad():
        fake = fake_model(real)                                  # :189
        swap_diff = F.mse_loss(real, fake)                       # :191
        noisy_fake = blend_with_given_noise(fake, noise, r)      # :193
    real_prediction = real_model(noisy_fake)                     # :195
    loss = criterion(real_prediction, real)                      # :197
    return loss, swap_diff, fake, real_prediction
