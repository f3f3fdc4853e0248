"""Minimal training loop with the pytorch_lightning 1.x semantics the reference relies on
 -- pytorch_lightning is not available on the MI355X image.

  * `Trainer(max_epochs, callbacks, ...).fit(module, ckpt_path=None)`
  * automatic optimisation with N optimisers: per batch, for optimizer_idx in 0..N-1:
        toggle (only that optimiser's parameters keep requires_grad) -> zero_grad ->
        training_step(batch, batch_idx[, optimizer_idx]) -> backward -> step -> untoggle;
    `global_step` counts optimiser steps
  * LR schedulers are stepped once per epoch
  * dict of dataloaders -> CombinedLoader(mode="max_size_cycle")
  * checkpoints are Lightning-style dicts {"epoch", "global_step", "state_dict",
    "optimizer_states", "lr_schedulers", "hyper_parameters"} under
    <default_root_dir>/version_N/checkpoints/
Data parallel: one process per GPU (torch.distributed.run); gradients are all-reduced by
distributed.DataParallel, overlapped with the backward pass; every rank iterates ITS shard of each
dataset (DistributedSampler, reshuffled per epoch) and draws its own noise stream (seed + rank).  `self.log` values stay on the device
and are flushed to metrics.csv every `flush_every` steps (no per-step host sync).
"""
import contextlib
import inspect
import os
import time
from pathlib import Path

import torch

from . import distributed as dist_utils


class Callback:
    def on_fit_start(self, trainer, module):
        pass

    def on_train_batch_end(self, trainer, module):
        pass

    def on_train_epoch_end(self, trainer, module):
        pass


class LearningRateMonitor(Callback):
    def __init__(self, logging_interval="step"):
        self.logging_interval = logging_interval

    def on_train_batch_end(self, trainer, module):
        for i, opt in enumerate(trainer.optimizers):
            name = "lr-Adam" if len(trainer.optimizers) == 1 else f"lr-Adam-{i}" if i else "lr-Adam"
            module._logged[name] = opt.param_groups[0]["lr"]


class ModelCheckpoint(Callback):
    def __init__(self, filename=None, save_top_k=1, monitor=None, mode="min", train_time_interval=None,
                 save_on_train_epoch_end=None, dirpath=None):
        self.filename, self.save_top_k, self.monitor, self.mode = filename, save_top_k, monitor, mode
        self.train_time_interval = train_time_interval
        self.save_on_train_epoch_end = save_on_train_epoch_end
        self.dirpath = dirpath
        self._last_time = time.monotonic()
        self._saved = []

    def _path(self, trainer):
        d = Path(self.dirpath) if self.dirpath else trainer.log_dir / "checkpoints"
        name = self.filename or f"epoch={trainer.current_epoch}-step={trainer.global_step}"
        return d / f"{name}.ckpt"

    def _save(self, trainer, module):
        if trainer.global_rank != 0:
            return
        path = self._path(trainer)
        trainer.save_checkpoint(path)
        if self.filename is None:
            self._saved.append(path)
            while self.save_top_k is not None and 0 <= self.save_top_k < len(self._saved):
                old = self._saved.pop(0)  # monitor="epoch", mode="max": newest wins
                if old.exists():
                    old.unlink()

    def on_train_batch_end(self, trainer, module):
        if self.train_time_interval is not None:
            if time.monotonic() - self._last_time >= self.train_time_interval.total_seconds():
                self._last_time = time.monotonic()
                self._save(trainer, module)

    def on_train_epoch_end(self, trainer, module):
        if self.train_time_interval is None:
            self._save(trainer, module)


class CombinedLoader:
    """dict of loaders, mode "max_size_cycle": epoch length = the longest loader; a shorter loader that runs out is
    iterated AGAIN (a fresh pass: reshuffled, re-augmented), not replayed from a cache of its first pass."""

    def __init__(self, loaders):
        self.loaders = loaders

    def __len__(self):
        return max(len(l) for l in self.loaders.values())

    def __iter__(self):
        its = {k: iter(l) for k, l in self.loaders.items()}
        for _ in range(len(self)):
            out = {}
            for k, l in self.loaders.items():
                try:
                    out[k] = next(its[k])
                except StopIteration:
                    its[k] = iter(l)
                    out[k] = next(its[k])
            yield out


def shard_loader(loader, world_size, rank, seed=0):
    """the same DataLoader over this rank's shard: DistributedSampler(shuffle as the original loader did,
    drop_last=False -> every rank sees ceil(N / world) samples, the tail padded by wrap-around as torch does)."""
    from torch.utils.data import DataLoader, RandomSampler, SequentialSampler
    from torch.utils.data.distributed import DistributedSampler
    if world_size <= 1:
        return loader
    if loader.batch_sampler is not None and type(loader.batch_sampler).__name__ != "BatchSampler":
        raise TypeError(f"shard_loader: a custom batch_sampler ({type(loader.batch_sampler).__name__}) cannot be "
                        f"sharded automatically; build the per-rank loader in train_dataloader()")
    if type(loader.sampler) not in (RandomSampler, SequentialSampler):
        # e.g. a WeightedRandomSampler over the `balance` difficulty classes: replacing it with a plain
        # DistributedSampler would silently change what the run trains on
        raise TypeError(f"shard_loader: sampler {type(loader.sampler).__name__} is neither sequential nor random; "
                        f"shard it yourself (per-rank sampler) in train_dataloader()")
    shuffle = isinstance(loader.sampler, RandomSampler)
    sampler = DistributedSampler(loader.dataset, num_replicas=world_size, rank=rank, shuffle=shuffle, seed=seed,
                                 drop_last=False)
    kw = dict(batch_size=loader.batch_size, sampler=sampler, num_workers=loader.num_workers,
              collate_fn=loader.collate_fn, pin_memory=loader.pin_memory, drop_last=loader.drop_last,
              timeout=loader.timeout, worker_init_fn=loader.worker_init_fn, generator=loader.generator,
              pin_memory_device=getattr(loader, "pin_memory_device", ""))
    if loader.num_workers > 0:
        kw.update(multiprocessing_context=loader.multiprocessing_context, persistent_workers=loader.persistent_workers,
                  prefetch_factor=loader.prefetch_factor)
    return DataLoader(loader.dataset, **kw)


def _set_epoch(loader, epoch, base_seed=0):
    """per-epoch shuffling that a resumed run can REPLAY: DistributedSampler.set_epoch on sharded loaders; on a single
    GPU a RandomSampler that has no generator of its own gets one seeded from (base_seed, epoch, loader index), so the
    first `done` batches a mid-epoch resume passes over are the ones trained before the checkpoint."""
    from torch.utils.data import RandomSampler
    loaders = list(loader.loaders.values()) if isinstance(loader, CombinedLoader) else [loader]
    for li, l in enumerate(loaders):
        sampler = getattr(l, "sampler", None)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)
        elif type(sampler) is RandomSampler and (sampler.generator is None or getattr(sampler, "_d3f_owned", False)):
            sampler.generator = torch.Generator().manual_seed((base_seed * 1000003 + epoch * 131 + li) % (1 << 63))
            sampler._d3f_owned = True


def lightning_batches_done(ckpt):
    """batches of the checkpoint's epoch that a pytorch_lightning 1.x run had completed when it saved mid-epoch
    (`ModelCheckpoint(train_time_interval=...)`, d3f/train_deep_fake/lit_module.py:127-140), from its `loops` record;
    None = saved at an epoch end (or no loop record)."""
    try:
        bp = ckpt["loops"]["fit_loop"]["epoch_loop.batch_progress"]
        if bp.get("is_last_batch", False):
            return None
        done = int(bp["current"]["completed"])
        return done if done > 0 else None
    except (KeyError, TypeError, ValueError):
        return None


def _to_device(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device, non_blocking=True)
    if isinstance(x, dict):
        return {k: _to_device(v, device) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_to_device(v, device) for v in x)
    return x


def optimizer_steps(model, optimizers, opt_params, batch, batch_idx, takes_idx, streams=None):
    """Lightning 1.x automatic optimisation for one batch: for every optimizer -- toggle (only its parameters keep
    requires_grad), zero_grad, training_step(batch, batch_idx[, optimizer_idx]), backward, step.

    streams: one HIP stream per optimizer (the module's `optimizer_streams()`), for modules whose optimizer steps are
    INDEPENDENT of each other (train_deep_fake's denoise mode: two nets, two domains).  Each step is enqueued on its
    own stream behind the batch's stream and the batch's stream is ordered behind all of them afterwards, so the steps
    of one batch overlap on the GPU -- at bs 8 per net a single step leaves more than half of the chip idle -- while
    every value stays what the sequential loop computes (the random draws keep their host order)."""
    multi = len(optimizers) > 1
    if multi and not streams and takes_idx and hasattr(model, "training_step_pair") and \
            model.pair_fused_active(batch, optimizers):
        # two INDEPENDENT optimizer steps (train_deep_fake's denoise mode) as one set of kernel launches: one forward and
        # one backward of the module's UnetPair, then both optimizer steps -- the values of the loop below, the launches
        # of one 16-image step instead of two 8-image ones.  (Every parameter already requires grad here: the toggles
        # of the loop are undone at its end.)
        for opt in optimizers:
            opt.zero_grad(set_to_none=True)
        losses = model.training_step_pair(batch, batch_idx)
        torch.autograd.backward(list(losses))
        for opt in optimizers:
            opt.step()
        return losses[-1]
    current = torch.cuda.current_stream() if streams else None
    for oi, opt in enumerate(optimizers):
        if streams:
            streams[oi].wait_stream(current)  # the batch (and the previous batch's joins) are ready
        with (torch.cuda.stream(streams[oi]) if streams else contextlib.nullcontext()):
            if multi:  # toggle_optimizer
                for oj, ps in enumerate(opt_params):
                    for p in ps:
                        p.requires_grad_(oj == oi)
            opt.zero_grad(set_to_none=True)
            loss = model.training_step(batch, batch_idx, oi) if takes_idx else model.training_step(batch, batch_idx)
            loss.backward()
            opt.step()
    if streams:
        for s in streams:
            current.wait_stream(s)
    if multi:
        for ps in opt_params:
            for p in ps:
                p.requires_grad_(True)
    return loss


class Trainer:
    def __init__(self, max_epochs=1, max_steps=-1, callbacks=None, log_every_n_steps=50, gpus=None,
                 accelerator=None, devices=None, default_root_dir="lightning_logs", flush_every=100,
                 enable_checkpointing=True, limit_train_batches=None, limit_val_batches=None, sync_batchnorm=False,
                 grad_buckets=None, grad_compress=None):
        self.max_epochs, self.max_steps = max_epochs, max_steps
        self.callbacks = list(callbacks or [])
        self.log_every_n_steps = log_every_n_steps
        self.default_root_dir = Path(default_root_dir)
        self.flush_every = flush_every
        self.enable_checkpointing = enable_checkpointing
        self.limit_train_batches = limit_train_batches
        self.limit_val_batches = limit_val_batches
        self.sync_batchnorm = sync_batchnorm  # Lightning's flag: BatchNorm statistics over all ranks (default: per GPU)
        # data parallel only (no Lightning counterpart; distributed.DataParallel): exchange buckets per backward pass
        # (None = DataParallel's default, 2) and "bf16" to send the gradient buckets as bfloat16 over xGMI
        self.grad_buckets, self.grad_compress = grad_buckets, grad_compress
        self.global_step = 0
        self.current_epoch = 0
        self.logger = None
        self.optimizers, self.lr_schedulers = [], []
        self.world_size, self.global_rank, self.local_rank = dist_utils.env_world()
        self.log_dir = None
        self._metric_rows = []
        self.module = None
        self._batches_done = None  # batches of the current epoch already trained (None between epochs)
        self._skip_batches = 0     # mid-epoch resume: batches of the first epoch to pass over
        self._base_seed = None     # rank-independent base of data shuffling (see fit)

    # ---- checkpoints ------------------------------------------------------------------------------
    def save_checkpoint(self, path):
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        ckpt = {
            "epoch": self.current_epoch,
            "global_step": self.global_step,
            # None: saved at the end of an epoch; n: saved mid-epoch after n batches (train_time_interval)
            "d3f_batches_done_in_epoch": self._batches_done,
            "d3f_loader_seed": self._base_seed,  # base of the per-epoch shuffling (mid-epoch resume replays it)
            "pytorch-lightning_version": "1.9.5+d3f-hip",
            "state_dict": {k: v.detach().cpu() for k, v in self.module.state_dict().items()},
            "optimizer_states": [o.state_dict() for o in self.optimizers],
            "lr_schedulers": [s.state_dict() for s in self.lr_schedulers],
            "hyper_parameters": dict(self.module.hparams),
        }
        tmp = path.with_suffix(".tmp")
        torch.save(ckpt, tmp)
        os.replace(tmp, path)

    def _restore(self, ckpt_path):
        """Lightning's fit(ckpt_path=...): strict state_dict (a key / encoder mismatch must not resume from random
        init silently), optimizer state in torch.optim.Adam's per-parameter form (reference-written checkpoints drop
        in: FusedAdam gathers exp_avg / exp_avg_sq / step into its flat moments), schedulers, loop position."""
        ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
        self.module.load_state_dict(ckpt["state_dict"], strict=True)
        opt_states = ckpt.get("optimizer_states", [])
        if opt_states and len(opt_states) != len(self.optimizers):
            raise RuntimeError(f"checkpoint holds {len(opt_states)} optimizer states, the module configures "
                               f"{len(self.optimizers)} optimizers")
        for o, sd in zip(self.optimizers, opt_states):
            o.load_state_dict(sd)
        for s, sd in zip(self.lr_schedulers, ckpt.get("lr_schedulers", [])):
            s.load_state_dict(sd)
        self.global_step = int(ckpt.get("global_step", 0))
        if ckpt.get("d3f_loader_seed") is not None:
            self._base_seed = int(ckpt["d3f_loader_seed"])  # replay the interrupted epoch's permutation
        done = ckpt.get("d3f_batches_done_in_epoch")
        if done is None and "d3f_batches_done_in_epoch" not in ckpt:
            done = lightning_batches_done(ckpt)  # a checkpoint the reference (pytorch_lightning 1.x) wrote mid-epoch
        if done is None:
            self.current_epoch = int(ckpt.get("epoch", 0)) + 1  # saved at an epoch end: resume with the next epoch
            self._skip_batches = 0
        else:
            self.current_epoch = int(ckpt.get("epoch", 0))      # saved mid-epoch: finish that epoch
            self._skip_batches = int(done)

    # ---- logging ------------------------------------------------------------------------------------
    def _new_version_dir(self):
        self.default_root_dir.mkdir(parents=True, exist_ok=True)
        n = 0
        while (self.default_root_dir / f"version_{n}").exists():
            n += 1
        d = self.default_root_dir / f"version_{n}"
        if self.global_rank == 0:
            d.mkdir(parents=True, exist_ok=True)
        return d

    def _flush_metrics(self):
        if not self._metric_rows or self.global_rank != 0:
            self._metric_rows = []
            return
        lines = []
        for step, row in self._metric_rows:
            vals = {k: (float(v) if not isinstance(v, (int, float)) else v) for k, v in row.items()}
            lines.append(f"{step}," + ";".join(f"{k}={v:.6g}" for k, v in sorted(vals.items())))
        with open(self.log_dir / "metrics.csv", "a") as f:
            f.write("\n".join(lines) + "\n")
        self._metric_rows = []

    # ---- fit ----------------------------------------------------------------------------------------
    def _run_validation(self, model, device):
        """Lightning's per-epoch validation loop for modules that define it (balance_training_images): eval mode,
        no grad, validation_step over val_dataloader(), then validation_epoch_end(outputs)."""
        if not (hasattr(model, "validation_step") and hasattr(model, "val_dataloader")):
            return
        model.eval()
        outputs = []
        with torch.no_grad():
            for batch_idx, batch in enumerate(model.val_dataloader()):
                if self.limit_val_batches is not None and batch_idx >= self.limit_val_batches:
                    break
                outputs.append(model.validation_step(_to_device(batch, device), batch_idx))
            if outputs and hasattr(model, "validation_epoch_end"):
                model.validation_epoch_end(outputs)
        model.train()

    def fit(self, model, ckpt_path=None):
        if not torch.cuda.is_available():
            raise RuntimeError("training needs an MI355X (HIP device); there is no CPU path")
        self.module = model
        dist_utils.init_process_group()
        torch.cuda.set_device(self.local_rank)
        device = torch.device("cuda", self.local_rank)
        model.to(device)
        model.train()
        model.__dict__["trainer"] = self
        self.log_dir = self._new_version_dir()

        conf = model.configure_optimizers()
        if isinstance(conf, tuple) and len(conf) == 2:
            self.optimizers, self.lr_schedulers = list(conf[0]), list(conf[1])
        elif isinstance(conf, (list, tuple)):
            self.optimizers, self.lr_schedulers = list(conf), []
        else:
            self.optimizers, self.lr_schedulers = [conf], []
        callbacks = list(self.callbacks)
        if hasattr(model, "configure_callbacks"):
            callbacks += list(model.configure_callbacks())
        if self.enable_checkpointing and not any(isinstance(c, ModelCheckpoint) for c in callbacks):
            callbacks.append(ModelCheckpoint())  # Lightning's default: one checkpoint per epoch
        if ckpt_path is not None:
            self._restore(ckpt_path)
        for opt in self.optimizers:
            mod = getattr(opt, "module", None)
            if mod is not None:
                dist_utils.DataParallel(mod, opt, sync_batchnorm=self.sync_batchnorm, grad_compress=self.grad_compress,
                                        **({} if self.grad_buckets is None else {"buckets": self.grad_buckets}))
        takes_idx = "optimizer_idx" in inspect.signature(model.training_step).parameters
        opt_params = [[p for g in o.param_groups for p in g["params"]] for o in self.optimizers]
        # independent optimizer steps on their own streams (the module decides: LitModule.optimizer_streams)
        opt_streams = model.optimizer_streams(device) if hasattr(model, "optimizer_streams") else None

        loaders = model.train_dataloader()
        if self._base_seed is None:  # (a resumed run keeps the checkpoint's)
            self._base_seed = dist_utils.shared_seed()
        if self.world_size > 1:
            # one shard per rank (otherwise every rank would train on all the data and the all-reduce would average
            # N copies of nearly the same gradient) and one noise / augmentation stream per rank.  The sampler seed
            # must be the SAME on every rank (each takes indices[rank::world] of one permutation): torch.initial_seed()
            # is per process, so it comes from PL_GLOBAL_SEED or is broadcast from rank 0 (distributed.shared_seed)
            seed = self._base_seed
            if isinstance(loaders, dict):
                loaders = {k: shard_loader(l, self.world_size, self.global_rank, seed) for k, l in loaders.items()}
            else:
                loaders = shard_loader(loaders, self.world_size, self.global_rank, seed)
            torch.manual_seed(seed + 1 + self.global_rank)
        loader = CombinedLoader(loaders) if isinstance(loaders, dict) else loaders
        for cb in callbacks:
            cb.on_fit_start(self, model)
        done = False
        while self.current_epoch < self.max_epochs and not done:
            _set_epoch(loader, self.current_epoch, self._base_seed)
            skip, self._skip_batches = self._skip_batches, 0
            self._batches_done = 0
            n_batches = len(loader) if self.limit_train_batches is None else min(len(loader), self.limit_train_batches)
            for batch_idx, batch in enumerate(loader):
                if self.limit_train_batches is not None and batch_idx >= self.limit_train_batches:
                    break
                if batch_idx < skip:  # mid-epoch resume: these batches were trained before the checkpoint
                    self._batches_done = batch_idx + 1
                    continue
                batch = _to_device(batch, device)
                if not getattr(model, "automatic_optimization", True):
                    # manual optimisation: the module's training_step runs backward and its optimiser step(s) itself
                    model.training_step(batch, batch_idx)
                    self.global_step += len(self.optimizers)
                else:
                    optimizer_steps(model, self.optimizers, opt_params, batch, batch_idx, takes_idx, opt_streams)
                    self.global_step += len(self.optimizers)
                self._batches_done = batch_idx + 1
                for cb in callbacks:
                    cb.on_train_batch_end(self, model)
                if self.global_step % max(self.log_every_n_steps, 1) == 0:
                    self._metric_rows.append((self.global_step, dict(model._logged)))
                if len(self._metric_rows) >= self.flush_every:
                    self._flush_metrics()
                if 0 < self.max_steps <= self.global_step:
                    done = True
                    break
            if done and self._batches_done < n_batches:
                # max_steps reached inside an epoch: the epoch is NOT complete -- no scheduler step, no validation, and
                # the checkpoint the callbacks write now is marked mid-epoch (resume finishes this epoch at this LR)
                for cb in callbacks:
                    cb.on_train_epoch_end(self, model)
                self._flush_metrics()
                break
            self._batches_done = None
            for s in self.lr_schedulers:
                s.step()
            self._run_validation(model, device)
            for cb in callbacks:
                cb.on_train_epoch_end(self, model)
            self._flush_metrics()
            self.current_epoch += 1
        torch.cuda.synchronize()
        return self
