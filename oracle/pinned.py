"""Oracle helpers for the mask-pinned gradient gate.  TEST INFRASTRUCTURE (see oracle/__init__.py).

The U-Net is piecewise linear, so two fp32 evaluations differ by ReLU / max-pool mask flips and their gradients by
O(1e-3..1e-2).  Running the float64 oracle with the masks PINNED to the decisions another run took (ReLU: `a > 0` of that
run's activations; max-pool: the argmax of that run's stem activation) makes the graph linear in every remaining
operation, so every activation gradient and every parameter gradient must agree to rounding (2e-4 in the tests) -- a
wrong tap, routing, bucket edge or accumulate flag is O(1).  Used by tests/test_gpu_parity_layers.py and by
__graft_entry__.smoke(); restates autograd through d3f/train_denoiser/lit_module.py:117-119.
"""
import copy

import torch.nn as nn
import torch.nn.functional as F


class PinnedReLU(nn.Module):
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


class RecordingReLU(nn.Module):
    def __init__(self, outputs):
        super().__init__()
        self.outputs = outputs

    def forward(self, x):
        out = F.relu(x)
        self.outputs.append(out)
        return out


class PinnedMaxPool(nn.Module):
    def __init__(self, index):
        super().__init__()
        self.index = index  # [B, C, Ho, Wo] flat positions in H*W, from the other run's activation

    def forward(self, x):
        B, Cc = x.shape[:2]
        return x.flatten(2).gather(2, self.index.flatten(2)).view(B, Cc, *self.index.shape[2:])


def swap_relus(model, factory):
    for mod in list(model.modules()):
        for name, child in list(mod.named_children()):
            if isinstance(child, nn.ReLU):
                setattr(mod, name, factory())


def unit_names(blocks=(3, 4, 6, 3)):
    """units with a post-activation tensor, in execution order = the order of ReLU calls in the oracle forward"""
    names = ["encoder.conv1"]
    for li, n in enumerate(blocks, start=1):
        for bi in range(n):
            names += [f"encoder.layer{li}.{bi}.conv1", f"encoder.layer{li}.{bi}.conv2"]
    for i in range(5):
        names += [f"decoder.blocks.{i}.conv1.0", f"decoder.blocks.{i}.conv2.0"]
    return names


def conv_outputs(model):
    store, hooks = {}, []
    for name, mod in model.named_modules():
        if isinstance(mod, nn.Conv2d):
            hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: store.__setitem__(name, o)))
    return store, hooks


def pinned_backward(ref, x, grad_out, activations, names):
    """float64 copy of `ref` (train mode), ReLU / max-pool decisions pinned to `activations` ({unit name: NCHW
    activation of the other run, channel padding already cut}); runs forward + backward(grad_out).
    Returns (output, {unit name: d loss / d activation}, {parameter name: gradient})."""
    pinned = copy.deepcopy(ref).double().train()
    masks = [activations[n] > 0 for n in names]
    acts = []
    swap_relus(pinned, lambda: PinnedReLU(masks, acts))
    _, idx = F.max_pool2d(activations["encoder.conv1"].float(), 3, 2, 1, return_indices=True)
    pinned.encoder.maxpool = PinnedMaxPool(idx)
    out = pinned(x.double())
    assert not masks, "every mask consumed: ReLU call order == unit order"
    out.backward(grad_out.double())
    dact = {n: a.grad for n, a in zip(names, acts) if a.grad is not None}
    grads = {k: p.grad for k, p in pinned.named_parameters()}
    return out.detach(), dact, grads


class TeacherForcedReLU(nn.Module):
    """ReLU of the teacher-forced BACKWARD gate: forward value = the other run's activation (so the next layer sees that
    run's operand), gradient path = x * mask with the mask pinned to that activation, and the gradient arriving at the
    output -- the oracle's own sum over this activation's consumers -- is RECORDED and then REPLACED by the other run's
    gradient, so that every comparison downstream sees one layer's backward arithmetic only."""

    def __init__(self, acts, grads_in, recorded):
        super().__init__()
        self.acts, self.grads_in, self.recorded = acts, grads_in, recorded

    def forward(self, x):
        a = self.acts.pop(0).to(x.dtype)
        forced = self.grads_in.pop(0)
        lin = x * (a > 0).to(x.dtype)
        out = lin + (a - lin).detach()
        slot = {}
        self.recorded.append(slot)

        def swap(g, slot=slot, forced=forced):
            slot["g"] = g.detach().clone()
            return forced.to(g.dtype)

        out.register_hook(swap)
        return out


def teacher_forced_backward(ref, x, grad_out, activations, act_grads, names):
    """float64 copy of `ref` (train mode) with BOTH passes teacher-forced by another run: in front of every layer the
    forward is handed that run's activation (`activations[unit]`), ReLU / max-pool decisions are pinned to it, and in
    the backward pass the gradient w.r.t. every unit's activation is replaced by that run's (`act_grads[unit]`) after the
    oracle's own value was recorded.  Each recorded activation gradient and each parameter gradient is therefore ONE
    layer's data gradient + BatchNorm backward + weight gradient applied to the other run's operands (a block's
    downsample branch: two layers).  Returns ({unit: oracle d loss / d activation}, {parameter name: gradient});
    restates autograd through d3f/train_denoiser/lit_module.py:117-119 for the bf16 engine plan."""
    model = copy.deepcopy(ref).double().train()
    acts = [activations[n] for n in names]
    gin = [act_grads[n] for n in names]
    recorded = []
    swap_relus(model, lambda: TeacherForcedReLU(acts, gin, recorded))
    _, idx = F.max_pool2d(activations["encoder.conv1"].float(), 3, 2, 1, return_indices=True)
    model.encoder.maxpool = PinnedMaxPool(idx)
    out = model(x.double())
    assert not acts and not gin and len(recorded) == len(names)
    out.backward(grad_out.double())
    dact = {n: slot["g"] for n, slot in zip(names, recorded) if "g" in slot}
    grads = {k: p.grad for k, p in model.named_parameters()}
    return dact, grads
