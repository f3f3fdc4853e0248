"""GPU: layer-by-layer parity of the whole network through d3f_unet_export -- the binding whole-network gate.

Forward: every unit's raw conv output (":y") and post-BatchNorm/ReLU activation (":a") against the float64 oracle,
each within NOISE x the CPU-fp32 oracle's own distance from float64.

Backward: the network is piecewise linear, so two fp32 evaluations differ by ReLU / max-pool mask flips and their
gradients by O(1e-3..1e-2) (tests/test_gpu_unet.py) -- a gate that loose would pass a missed accumulate.  Here the
float64 oracle runs with the masks PINNED to the HIP run's own decisions (ReLU: a_hip > 0; max-pool: the argmax of the
HIP activation), which makes the graph linear in every remaining operation: every activation gradient (":da") and
every parameter gradient must then agree to 2e-4 (measured on MI355X: worst tensor 9e-5 fp32 MFMA / 1.1e-4 f32x3 -- the
BatchNorm affine gradients of the first layers, sums over up to 49 k positions of products of two fp32-rounded
factors; flat gradient 6e-5) -- a wrong tap, routing, bucket edge or accumulate flag is O(1)."""
import copy

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from util import rel_l2

pytestmark = pytest.mark.gpu
NOISE = 4.0
PINNED_TOL = 2e-4


class _PinnedReLU(nn.Module):
    """ReLU whose decisions come from a queue of external masks (one per call, in execution order); records its
    outputs so that their gradients can be read after backward."""

    def __init__(self, masks, outputs):
        super().__init__()
        self.masks, self.outputs = masks, outputs

    def forward(self, x):
        out = x * self.masks.pop(0).to(x.dtype)
        out.retain_grad()
        self.outputs.append(out)
        return out


class _RecordingReLU(nn.Module):
    def __init__(self, outputs):
        super().__init__()
        self.outputs = outputs

    def forward(self, x):
        out = F.relu(x)
        self.outputs.append(out)
        return out


class _PinnedMaxPool(nn.Module):
    def __init__(self, index):
        super().__init__()
        self.index = index  # [B, C, Ho, Wo] flat positions in H*W, from the HIP activation

    def forward(self, x):
        B, Cc = x.shape[:2]
        return x.flatten(2).gather(2, self.index.flatten(2)).view(B, Cc, *self.index.shape[2:])


def _swap_relus(model, factory):
    for mod in list(model.modules()):
        for name, child in list(mod.named_children()):
            if isinstance(child, nn.ReLU):
                setattr(mod, name, factory())


def _unit_names(net):
    """units with a post-activation tensor, in execution order = the order of ReLU calls in the oracle forward"""
    names = ["encoder.conv1"]
    for li, n in enumerate((3, 4, 6, 3), start=1):
        for bi in range(n):
            names += [f"encoder.layer{li}.{bi}.conv1", f"encoder.layer{li}.{bi}.conv2"]
    for i in range(5):
        names += [f"decoder.blocks.{i}.conv1.0", f"decoder.blocks.{i}.conv2.0"]
    return names


def _conv_outputs(model):
    store, hooks = {}, []
    for name, mod in model.named_modules():
        if isinstance(mod, nn.Conv2d):
            hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: store.__setitem__(name, o)))
    return store, hooks


@pytest.mark.parametrize("dtype", ["f32", "f32x3"])
def test_every_layer_forward_and_mask_pinned_backward(dtype):
    import oracle
    from denoising_diffusion_deep_fake_amd import Unet, ops
    torch.manual_seed(7)
    ref = oracle.Unet("resnet34", None, 3, 3, None).train()
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
        ref.segmentation_head[0].bias.normal_(0, 0.1)
    net = Unet("resnet34", None, 3, 3, None, compute_dtype=dtype)
    net.load_state_dict(ref.state_dict())
    net = net.cuda().train()
    B, S = 3, 64
    x = oracle.synthetic_face_crops(B, S, seed=17)
    names = _unit_names(net)

    # ---- HIP run: forward, loss gradient, backward; export every unit ----
    pred = net(x.cuda())
    _, gout = ops.mse_ssim_loss(pred.detach(), oracle.synthetic_face_crops(B, S, seed=18).cuda())
    pred.backward(gout)
    hip_y = {n: net.export_activation(n + ":y").cpu() for n in names}
    hip_a = {n: net.export_activation(n + ":a").cpu() for n in names}
    hip_da = {n: net.export_activation(n + ":da").cpu() for n in names}
    ds_names = [f"encoder.layer{li}.0.downsample.0" for li in (2, 3, 4)]
    hip_y.update({n: net.export_activation(n + ":y").cpu() for n in ds_names})
    hip_grads = {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}

    # ---- (i) forward, layer by layer, unpinned: hip vs float64, next to cpu-fp32 vs float64 ----
    ref64 = copy.deepcopy(ref).double()
    outs = {}
    for tag, model, inp in (("f32", copy.deepcopy(ref), x), ("f64", ref64, x.double())):
        acts = []
        _swap_relus(model, lambda: _RecordingReLU(acts))
        store, hooks = _conv_outputs(model)
        with torch.no_grad():
            model(inp)
        for h in hooks:
            h.remove()
        outs[tag] = (dict(store), dict(zip(names, acts)))
        assert len(acts) == len(names)
    worst = 0.0
    for n in names + ds_names:
        c = outs["f64"][0][n].shape[1]
        e_hip, e_cpu = rel_l2(hip_y[n][:, :c], outs["f64"][0][n]), rel_l2(outs["f32"][0][n], outs["f64"][0][n])
        assert e_hip < max(NOISE * e_cpu, 2e-6), ("y", n, e_hip, e_cpu)
        worst = max(worst, e_hip / max(e_cpu, 1e-12))
    for n in names:
        c = outs["f64"][1][n].shape[1]
        e_hip, e_cpu = rel_l2(hip_a[n][:, :c], outs["f64"][1][n]), rel_l2(outs["f32"][1][n], outs["f64"][1][n])
        assert e_hip < max(NOISE * e_cpu, 2e-6), ("a", n, e_hip, e_cpu)

    # ---- (ii) backward with the HIP run's masks pinned into the float64 oracle ----
    pinned = copy.deepcopy(ref).double().train()
    masks = [(hip_a[n][:, :outs["f64"][1][n].shape[1]] > 0) for n in names]
    acts = []
    _swap_relus(pinned, lambda: _PinnedReLU(masks, acts))
    stem = hip_a["encoder.conv1"][:, :64]
    _, idx = F.max_pool2d(stem, 3, 2, 1, return_indices=True)
    pinned.encoder.maxpool = _PinnedMaxPool(idx)
    out = pinned(x.double())
    assert not masks, "every mask consumed: ReLU call order == unit order"
    assert rel_l2(pred, out) < 1e-4   # pinning moves the forward only where a sign was within rounding of zero
    out.backward(gout.cpu().double())
    bad = []
    for n, a in zip(names, acts):
        if a.grad is None:
            continue
        e = rel_l2(hip_da[n][:, :a.shape[1]], a.grad)
        if e > PINNED_TOL:
            bad.append(("da", n, e))
    for k, p in pinned.named_parameters():
        e = rel_l2(hip_grads[k], p.grad)
        if e > PINNED_TOL:
            bad.append(("grad", k, e))
    assert not bad, bad
    flat = rel_l2(torch.cat([hip_grads[k].reshape(-1) for k, _ in pinned.named_parameters()]),
                  torch.cat([p.grad.reshape(-1) for _, p in pinned.named_parameters()]))
    assert flat < PINNED_TOL / 2, flat
    print(f"[{dtype}] forward worst hip/cpu-fp32 distance ratio {worst:.2f}; pinned flat gradient rel-L2 {flat:.2e}")
