"""Fused optimiser-side updates over the Unet's flat parameter buffer.

FusedAdam : torch.optim.Adam(params, lr, betas) as the reference configures it
            (d3f/train_denoiser/lit_module.py:95; d3f/train_deep_fake/lit_module.py:116-120) in ONE
            kernel launch over all 24.4 M parameters (csrc/optim.hip) instead of ~140 per-tensor updates.
            overlap_tail=True (opt-in): the update of every gradient bucket but the last runs INSIDE backward, behind the
            chain's last kernel and next to the last bucket's weight gradients on the side stream; step() updates the
            rest.  Same values, bit for bit.
EMA       : ema_pytorch.EMA(model, beta, update_every, include_online_model=False) semantics
            (d3f/train_deep_fake/lit_module.py:62-70,185; defaults update_after_step=100, inv_gamma=1,
            power=2/3, min_value=0 -- SURVEY.md Appendix A.3) with the lerp as one launch per flat buffer.
"""
import copy

import torch

from . import ops
from .unet import Unet


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, module=None, grad_scale=1.0, overlap_tail=False):
        """overlap_tail: optimizer-in-backward for the leading gradient buckets (Unet.set_early_update).  The contract
        the caller accepts with it: every backward() is followed by exactly one step() with unchanged lr / betas / eps
        (Lightning's automatic optimisation, this package's Trainer and bench.py do that); the parameters of layer3,
        layer4, the decoder and the head already hold their updated values when backward() returns; .grad is complete
        and untouched as always -- but EDITING .grad between backward() and step() (clip_grad_norm_ / Lightning's
        gradient_clip_val, GradScaler.unscale_, hooks that rescale gradients) cannot reach the 94 % of the parameters
        whose update already ran: step() detects in-place TORCH operations on the flat gradient buffer or its .grad views
        (the buffer's autograd version counter) and RAISES instead of applying them to the last bucket only; writes that
        bypass torch's bookkeeping (raw-pointer kernels such as this package's ops.* on .grad, `.data` tricks) are NOT
        seen -- use overlap_tail=False with any step that edits gradients.  A data-parallel reducer, or gradients that do not land in the flat buffer directly
        (.grad not None before backward: zero_grad(set_to_none=False)), switch the early part off by themselves (the
        whole update then runs in step() as without the flag); a second backward() before step() (gradient accumulation),
        changed hyper-parameters or cleared gradients between backward() and step() raise in step()."""
        params = list(params)
        # the param-group keys of torch.optim.Adam, so that its load_state_dict accepts a state_dict saved here (and
        # the other way round); the variants behind the switches are not implemented by the one-launch update
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=False))
        if not isinstance(module, Unet):
            raise TypeError("FusedAdam needs module=<the d3f Unet that owns these parameters>")
        if [id(p) for p in params] != [id(p) for p in module.parameters()]:
            raise ValueError("FusedAdam must be given exactly module.parameters()")
        self.module = module
        self.grad_scale = grad_scale
        self.before_step = None  # e.g. DataParallel's reducer.wait
        self._step = 0
        self.exp_avg = None
        self.exp_avg_sq = None
        self.overlap_tail = bool(overlap_tail)
        self._early = None  # (lo, hi, hyper-parameters, gradient buffer) of an update already applied inside backward
        self.early_updates = 0  # steps whose leading buckets were really updated inside backward (what bench.py reports)
        # always (re)set the module's hook: a FusedAdam built earlier for the same module with overlap_tail=True must
        # not keep updating [lo, hi) with ITS moments inside backward next to this optimiser's step()
        module.set_early_update(self._early_update if self.overlap_tail else None)

    def zero_grad(self, set_to_none=True):
        """torch.optim.Optimizer.zero_grad; the set_to_none form without its per-parameter bookkeeping (143 parameters:
        ~0.1 ms of host time per step on the launch-bound configurations)"""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for p in self.module._param_list:
            p.grad = None

    def _hyper(self):
        g = self.param_groups[0]
        if g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False):
            raise NotImplementedError("FusedAdam implements plain Adam (weight_decay=0, amsgrad=False, maximize=False), "
                                      "which is what the reference configures")
        return (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(self.grad_scale))

    @torch.no_grad()
    def _early_update(self, grads, lo, hi):
        """Unet.set_early_update hook: gradients [lo, hi) are final on the current stream"""
        if self._early is not None:
            return  # (a second backward before step(): step() refuses the mixed update)
        flat = self.module.flat_params
        if self.exp_avg is None or self.exp_avg.data_ptr() == 0 or self.exp_avg.device != flat.device:
            self.exp_avg = torch.zeros_like(flat)
            self.exp_avg_sq = torch.zeros_like(flat)
        hyper = self._hyper()
        if lo % 4 or (hi % 4 and hi != flat.numel()):  # (every slice handed to d3f_adam_step must START on a 16-byte boundary)
            raise RuntimeError(f"FusedAdam(overlap_tail=True): bucket range [{lo}, {hi}) is not on 16-byte boundaries")
        ops.adam_step(flat[lo:hi], grads[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], hyper[0], hyper[1], hyper[2],
                      hyper[3], self._step + 1, hyper[4])
        self._early = (lo, hi, hyper, grads.data_ptr(), self.module._rt.get("backward_calls", 0), grads._version)
        self.early_updates += 1

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self.before_step is not None:
            self.before_step()
        m = self.module
        flat, grads = m.flat_params, m.flat_grads
        if flat is None or grads is None:
            raise RuntimeError("FusedAdam.step() before any backward pass")
        plist = m._param_list
        first = plist[0].grad
        if first is None:
            if self._early is not None:
                raise RuntimeError("FusedAdam(overlap_tail=True): the gradients were cleared between backward() and "
                                   "step(), but part of this step's update already ran inside backward()")
            return loss  # nothing to do (all grads cleared) -- same as torch.optim.Adam
        if first.data_ptr() != grads.data_ptr():
            if self._early is not None:
                raise RuntimeError("FusedAdam(overlap_tail=True): gradients were accumulated outside the flat buffer "
                                   "after part of this step's update already ran inside backward()")
            # gradients were accumulated outside the flat buffer: gather them
            off = 0
            for p in plist:
                grads[off:off + p.numel()].copy_(p.grad.reshape(-1))
                off += p.numel()
        if self.exp_avg is None or self.exp_avg.data_ptr() == 0 or self.exp_avg.device != flat.device:
            self.exp_avg = torch.zeros_like(flat)
            self.exp_avg_sq = torch.zeros_like(flat)
        hyper = self._hyper()
        self._step += 1
        early, self._early = self._early, None
        if early is None:
            ranges = [(0, flat.numel())]
        else:
            lo, hi, used, gptr, calls, gver = early
            if calls != m._rt.get("backward_calls", 0):
                raise RuntimeError("FusedAdam(overlap_tail=True): another backward() ran before step(), but part of this "
                                   "step's update already ran inside the first one (gradient accumulation needs "
                                   "overlap_tail=False)")
            if grads._version != gver:
                raise RuntimeError("FusedAdam(overlap_tail=True): the gradients were modified in place between backward() "
                                   "and step() (gradient clipping, GradScaler.unscale_, a hook that rescales .grad?), but "
                                   "the update of layer3 / layer4 / decoder / head already ran inside backward() with the "
                                   "unmodified values; build the optimiser with overlap_tail=False for such steps")
            if used != hyper or gptr != grads.data_ptr():
                raise RuntimeError("FusedAdam(overlap_tail=True): lr / betas / eps / grad_scale or the gradient buffer "
                                   f"changed between backward() and step() ({used} -> {hyper}); part of this step's "
                                   "update already ran inside backward() with the old values")
            ranges = [r for r in ((0, lo), (hi, flat.numel())) if r[1] > r[0]]
        for b, e in ranges:
            ops.adam_step(flat[b:e], grads[b:e], self.exp_avg[b:e], self.exp_avg_sq[b:e], hyper[0], hyper[1], hyper[2],
                          hyper[3], self._step, hyper[4])
        m.mark_params_changed()
        return loss

    # ---- checkpoints: torch.optim.Adam's own format, both ways --------------------------------------------------
    # A Lightning checkpoint written by the reference (d3f/train_deep_fake/start_training.py:19-23 resumes from one)
    # stores optimizer_states[i] = torch.optim.Adam.state_dict(): {"state": {k: {"step", "exp_avg", "exp_avg_sq"}},
    # "param_groups": [...]}, parameters numbered in model.parameters() order -- the order of the flat buffer.
    def _flat_state(self):
        flat = self.module.flat_params
        if flat is None:
            self.module.prepare()
            flat = self.module.flat_params
        if self.exp_avg is None or self.exp_avg.device != flat.device:
            self.exp_avg = torch.zeros_like(flat)
            self.exp_avg_sq = torch.zeros_like(flat)
        return flat

    def state_dict(self):
        """per-parameter torch.optim.Adam state (CPU copies), so that torch.optim.Adam -- i.e. the reference --
        can load what this optimiser saved"""
        sd = super().state_dict()
        state = {}
        if self._step > 0 and self.exp_avg is not None:
            off = 0
            for i, p in enumerate(self.module._param_list):
                n = p.numel()
                state[i] = {"step": torch.tensor(float(self._step)),
                            "exp_avg": self.exp_avg[off:off + n].reshape(p.shape).cpu(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].reshape(p.shape).cpu()}
                off += n
        sd["state"] = state
        return sd

    def load_state_dict(self, sd):
        """accepts torch.optim.Adam's per-parameter state (a reference checkpoint) and gathers it into the flat
        moments; also the round-1 private `d3f_flat` blob.  An optimizer state that names parameters but carries no
        moments for some of them is refused rather than silently zeroed."""
        sd = dict(sd)
        legacy = sd.pop("d3f_flat", None)
        state = sd.get("state", {})
        super().load_state_dict({"state": {}, "param_groups": sd["param_groups"]})
        params = self.module._param_list
        if state:
            if len(state) != len(params):
                raise ValueError(f"optimizer state covers {len(state)} parameters, the model has {len(params)}")
            flat = self._flat_state()
            steps = set()
            off = 0
            for i, p in enumerate(params):
                st = state[i] if i in state else state[str(i)]
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state {i}: exp_avg {tuple(st['exp_avg'].shape)} vs parameter "
                                     f"{tuple(p.shape)}")
                n = p.numel()
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1).to(flat.device, torch.float32))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(flat.device, torch.float32))
                steps.add(int(float(st["step"])))
                off += n
            if len(steps) != 1:
                raise ValueError(f"per-parameter Adam step counts differ ({sorted(steps)}): the flat update keeps one")
            self._step = steps.pop()
        elif legacy is not None and legacy.get("exp_avg") is not None:
            flat = self._flat_state()
            self.exp_avg.copy_(legacy["exp_avg"].to(flat.device))
            self.exp_avg_sq.copy_(legacy["exp_avg_sq"].to(flat.device))
            self._step = int(legacy["step"])
        else:
            self._step = 0
            self.exp_avg = self.exp_avg_sq = None


class EMA(torch.nn.Module):
    def __init__(self, model, beta=0.9999, update_every=1, update_after_step=100, inv_gamma=1.0,
                 power=2 / 3, min_value=0.0, include_online_model=False):
        super().__init__()
        if include_online_model:
            self.online_model = model
        else:
            self.__dict__["_online"] = [model]  # not a registered sub-module (kept out of state_dict)
        self.include_online_model = include_online_model
        self.ema_model = copy.deepcopy(model)
        self.ema_model.requires_grad_(False)
        self.beta, self.update_every, self.update_after_step = beta, update_every, update_after_step
        self.inv_gamma, self.power, self.min_value = inv_gamma, power, min_value
        self.register_buffer("initted", torch.tensor(False))
        self.register_buffer("step", torch.tensor(0))
        self._host_step = 0      # mirrors `step` without a device sync per update
        self._host_initted = False

    @property
    def model(self):
        return self.online_model if self.include_online_model else self.__dict__["_online"][0]

    def get_current_decay(self, step=None):
        step = self._host_step if step is None else step
        epoch = max(step - self.update_after_step - 1, 0.0)
        if epoch <= 0:
            return 0.0
        value = 1 - (1 + epoch / self.inv_gamma) ** -self.power
        return min(max(value, self.min_value), self.beta)

    def _flat_pairs(self):
        on, em = self.model, self.ema_model
        dev = next(on.parameters()).device
        on.prepare(dev)
        em.prepare(dev)
        return [(em.flat_params, on.flat_params), (em.flat_bn_stats, on.flat_bn_stats)]

    @torch.no_grad()
    def copy_params_from_model_to_ema(self):
        for e, o in self._flat_pairs():
            e.copy_(o)
        self.ema_model._rt["flat_nbt"].copy_(self.model._rt["flat_nbt"])
        self.ema_model.mark_params_changed()

    @torch.no_grad()
    def update(self):
        step = self._host_step
        self._host_step += 1
        self.step += 1
        if step % self.update_every != 0:
            return
        if step <= self.update_after_step:
            self.copy_params_from_model_to_ema()
            return
        if not self._host_initted:
            self.copy_params_from_model_to_ema()
            self._host_initted = True
            self.initted.fill_(True)
        w = 1.0 - self.get_current_decay()
        for e, o in self._flat_pairs():
            ops.ema_lerp(e, o, w)
        self.ema_model.mark_params_changed()

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self._host_step = int(self.step.item())
        self._host_initted = bool(self.initted.item())

    def forward(self, *args, **kwargs):
        return self.ema_model(*args, **kwargs)
