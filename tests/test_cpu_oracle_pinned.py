"""CPU: the oracle's gradient-gate helpers (oracle/pinned.py) checked against plain autograd, so that the GPU gates built
on them (tests/test_gpu_parity_layers.py, tests/test_gpu_bf16.py) measure the HIP engine and not the helper."""
import copy

import torch

from util import rel_l2


def _reference_run(B=2, S=64):
    import oracle
    from oracle.pinned import pinned_backward, unit_names
    torch.manual_seed(5)
    ref = oracle.Unet("resnet34", None, 3, 3, None).train()
    x = oracle.synthetic_face_crops(B, S, seed=2)
    gout = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(1)) / (B * 3 * S * S)
    names = unit_names()
    # a float64 run of the plain oracle supplies activations, activation gradients and parameter gradients
    from oracle.pinned import RecordingReLU, swap_relus
    m64 = copy.deepcopy(ref).double()
    acts = []
    swap_relus(m64, lambda: RecordingReLU(acts))
    out = m64(x.double())
    for a in acts:
        a.retain_grad()
    out.backward(gout.double())
    A = {n: a.detach() for n, a in zip(names, acts)}
    dA = {n: a.grad for n, a in zip(names, acts)}
    G = {k: p.grad for k, p in m64.named_parameters()}
    return ref, x, gout, names, A, dA, G, pinned_backward


def test_pinned_and_teacher_forced_backward_reproduce_autograd():
    """Fed a float64 run's OWN activations and activation gradients, both helpers must give that run's gradients back to
    float64 rounding: pinning / teacher forcing changes nothing when the teacher is the oracle itself."""
    from oracle.pinned import teacher_forced_backward
    ref, x, gout, names, A, dA, G, pinned_backward = _reference_run()
    _, dact, grads = pinned_backward(ref, x, gout, A, names)
    assert max(rel_l2(dact[n], dA[n]) for n in names) < 1e-12
    assert max(rel_l2(grads[k], G[k]) for k in G) < 1e-12
    dact, grads = teacher_forced_backward(ref, x, gout, A, dA, names)
    assert set(dact) == set(names) and set(grads) == set(G)
    assert max(rel_l2(dact[n], dA[n]) for n in names) < 1e-12
    assert max(rel_l2(grads[k], G[k]) for k in G) < 1e-12


def test_teacher_forced_backward_isolates_one_layer():
    """A gradient corrupted at ONE activation must show up only in the tensors one layer downstream of it (that is what
    makes the gate's per-tensor error a per-layer statement): perturb d loss / d a of layer2.1.conv1 by 10 % -- only that
    block's conv1 parameters, the gradient of its input activation (layer2.0.conv2) and nothing else may move."""
    from oracle.pinned import teacher_forced_backward
    ref, x, gout, names, A, dA, G, _ = _reference_run()
    bad = dict(dA)
    victim = "encoder.layer2.1.conv1"
    bad[victim] = dA[victim] * 1.1
    dact, grads = teacher_forced_backward(ref, x, gout, A, bad, names)
    moved_a = {n for n in names if rel_l2(dact[n], dA[n]) > 1e-9}
    moved_p = {k for k in G if rel_l2(grads[k], G[k]) > 1e-9}
    assert moved_a == {"encoder.layer2.0.conv2"}, moved_a
    assert moved_p == {"encoder.layer2.1.conv1.weight", "encoder.layer2.1.bn1.weight", "encoder.layer2.1.bn1.bias"}, moved_p
