"""The whole noisy -> clean optimiser step as ONE library call, replayed from a hipGraph.

What Lightning's automatic optimisation runs around the reference's `training_step`
(d3f/train_denoiser/lit_module.py:107-126; d3f/train_deep_fake/lit_module.py:162-181 for one net of the pair) --
zero_grad, blend noise, U-Net forward, MseStructuralSimilarityLoss, backward, Adam step -- is ~330 kernel launches over
three streams.  `GraphTrainStep` hands the step to `d3f_unet_train_step`, which captures the launch sequence once
(per set of buffers) and replays it with a single hipGraphLaunch: same kernels, same order, bit-identical parameters
(tests/test_gpu_training.py).

MEASURED (MI355X, ROCm 7.0, round 3 -- profiles/README.md): the replay is SLOWER than the eager step -- bf16 256x256
4.6 -> 10.9 ms, fp32 128x128 3.8 -> 11.5 ms, fp32 256x256 8.3 -> 13.7 ms per step: ~30 us per node for a graph whose
nodes span three captured streams; captured on ONE stream (D3F_SERIAL_BACKWARD=1 D3F_NO_ASYNC_PACK=1) it merely equals
the eager single-stream step (5.09 vs 5.11 ms), i.e. even the small configurations are bound by the GPU's own
kernel-to-kernel dispatch latency, not by the host's launch loop (untraced host loop 3.4 ms per step,
profiles/r03_b_hostprobe.txt).  The class therefore stays opt-in (`graph_step: true`, `bench.py --graph-step on`).

Random numbers stay outside the graph (torch's generator, the reference's order: randn for the noise first, then rand
for the blend ratios); so do Adam's per-step scalars, which travel as 8 floats in device memory.  Single GPU only: the
data-parallel path needs the bucket hooks between the backward segments (distributed.py) and runs eagerly.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import D3FError, check, ptr, stream_ptr
from .optim import FusedAdam
from .unet import Unet


class StepBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("params", "bnstats", "grads", "exp_avg", "exp_avg_sq", "image", "noise", "y_uniform", "noisy", "pred",
                 "grad_pred", "loss_out", "loss_workspace", "adam_coef")]


class GraphTrainStep:
    RING = 64  # pinned slots for Adam's coefficients; a slot is rewritten only after the copy that read it has run

    def __init__(self, model, optimizer, noise_lambda, input_min=-1.0, input_max=1.0, use_graph=True):
        if not isinstance(model, Unet) or not isinstance(optimizer, FusedAdam) or optimizer.module is not model:
            raise TypeError("GraphTrainStep(model=<d3f Unet>, optimizer=<the FusedAdam over its parameters>)")
        self.model, self.optimizer = model, optimizer
        self.lam, self.lo, self.hi = float(noise_lambda), float(input_min), float(input_max)
        self.use_graph = bool(use_graph)
        self._bufs = {}
        self._slot = 0

    def _buffers(self, shape, device):
        key = (tuple(shape), device.index)
        b = self._bufs.get(key)
        if b is None:
            B, Cc, H, W = shape
            L = _lib.lib()
            f32 = dict(dtype=torch.float32, device=device)
            b = dict(image=torch.empty(shape, **f32), noise=torch.empty(shape, **f32), y=torch.empty(B, **f32),
                     noisy=torch.empty(shape, **f32), pred=torch.empty(shape, **f32), gpred=torch.empty(shape, **f32),
                     loss=torch.zeros(3, **f32),
                     loss_ws=torch.empty(L.d3f_mse_ssim_loss_workspace_bytes(B, H, W), dtype=torch.uint8, device=device),
                     coef=torch.zeros(8, **f32), ring=torch.zeros((self.RING, 8), dtype=torch.float32).pin_memory(),
                     ring_events=[None] * self.RING)
            self._bufs[key] = b
        return b

    @torch.no_grad()
    def __call__(self, image):
        """one optimiser step on `image` ([B,3,H,W] f32 on the HIP device, already augmented); returns the loss as a
        fresh device scalar (a stream-ordered copy: callers such as Trainer / self.log keep it across later steps);
        `.parts` is a copy of the step's {loss, mse, ssim} buffer, which every replay overwrites"""
        m, opt = self.model, self.optimizer
        if m._rt["grad_sync"] is not None or m._rt.get("bn_sync") is not None:
            raise D3FError("GraphTrainStep is the single-GPU form: under data parallelism the gradient buckets are "
                           "all-reduced between the backward segments, and synchronised BatchNorm statistics call "
                           "back into Python inside the pass (use the eager step)")
        if not m.training:
            raise D3FError("GraphTrainStep needs the model in train mode (batch statistics, gradients)")
        if image.dim() != 4 or image.shape[1] != 3 or m.in_channels != 3 or m.classes != 3:
            raise RuntimeError(f"Expected input [B, 3, H, W] and a 3 -> 3 channel net, got {list(image.shape)}")
        m.check_input_shape(image)
        if image.device.type != "cuda":
            raise D3FError("d3f Unet runs on an MI355X (HIP) device only; there is no CPU fallback")
        dev = image.device
        m._ensure_flat(dev)
        rt = m._rt
        b = self._buffers(image.shape, dev)
        if image.data_ptr() != b["image"].data_ptr():
            b["image"].copy_(image)
        b["noise"].normal_()   # the reference's RNG order: randn_like(batch) first ...
        b["y"].uniform_()      # ... then rand(B, 1, 1, 1)
        if rt["flat_grad"] is None:
            rt["flat_grad"] = torch.empty_like(rt["flat"])
        params = m._param_list
        if params[0].grad is None or params[0].grad.data_ptr() != rt["flat_grad"].data_ptr():
            for (name, shape, off), p in zip(m._table()[0], params):  # .grad = views of the flat gradient, as eager
                p.grad = rt["flat_grad"][off:off + p.numel()].view(shape)
        opt._flat_state()  # moments allocated
        g = opt.param_groups[0]
        if g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False):
            raise NotImplementedError("FusedAdam implements plain Adam")
        opt._step += 1
        opt._opt_called = True  # the fused step IS this optimiser's step: torch's lr_scheduler.step() order check reads the flag
        L = _lib.lib()
        k = self._slot % self.RING
        self._slot += 1
        if b["ring_events"][k] is not None:
            b["ring_events"][k].synchronize()  # the copy that read this slot RING steps ago (normally long done)
        slot = b["ring"][k]
        check(L.d3f_adam_coefficients(float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                      int(opt._step), float(opt.grad_scale), C.cast(slot.data_ptr(), C.POINTER(C.c_float))))
        b["coef"].copy_(slot, non_blocking=True)
        if b["ring_events"][k] is None:
            b["ring_events"][k] = torch.cuda.Event()
        b["ring_events"][k].record()
        eng = m._engine(image.shape[0], image.shape[2], image.shape[3], dev)
        eng.serial += 1
        rt["last_engine"] = eng
        sb = StepBuffers(rt["flat"].data_ptr(), rt["flat_bn"].data_ptr(), rt["flat_grad"].data_ptr(),
                         opt.exp_avg.data_ptr(), opt.exp_avg_sq.data_ptr(), b["image"].data_ptr(), b["noise"].data_ptr(),
                         b["y"].data_ptr(), b["noisy"].data_ptr(), b["pred"].data_ptr(), b["gpred"].data_ptr(),
                         b["loss"].data_ptr(), b["loss_ws"].data_ptr(), b["coef"].data_ptr())
        check(L.d3f_unet_train_step(eng.h, C.byref(sb), self.lam, self.lo, self.hi, ptr(eng.workspace),
                                    1 if self.use_graph else 0, stream_ptr()))
        rt["flat_nbt"] += 1
        eng.packed_version = None  # the step packed BEFORE its Adam update: the layouts are one update behind
        m.mark_params_changed()
        self.parts = b["loss"].clone()  # {loss, mse, ssim}
        return self.parts[0]
