"""GPU, BASELINE.json full sizes (256x256, bs 16, f32), where the CPU oracle is too slow to be the checker:
size-independent properties instead.
  * adjoint identities of each contraction trio:  <conv(x; w), dy> = <x, dgrad(dy; w)> = <w, wgrad(dy; x)>
    (ties the forward, data-gradient and weight-gradient kernels of a layer shape to each other);
  * linearity of the forward kernel;
  * bitwise run-to-run reproducibility of the whole training step (fixed-order reductions, no float atomics);
  * the loss goes down over real optimiser steps on a fixed batch."""
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
    dx0_low = ops.upsample2x_backward(dx0) if up else dx0
    via_x = _dot(x0, dx0_low) + (_dot(x1, dx1) if C1 else 0.0)
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
