"""Data-parallel training over RCCL/xGMI: one process per GPU, full replica per rank.

The only exchange step of the hot path is the gradient sum.  The backward pass of
the C engine is cut into 4 buckets (head+decoder, layer4, layer3, rest) that become final in that
order; as soon as bucket k's kernels are enqueued the flat gradient slice is handed to an
asynchronous all-reduce (torch.distributed "nccl" == RCCL), which runs on the process group's own
stream while the compute stream continues with bucket k+1.  `wait()` (called by FusedAdam.step)
joins the streams; averaging is folded into the Adam kernel (grad_scale = 1/world_size).
BatchNorm uses per-GPU batch statistics, as Lightning DDP would with the reference's Trainer flags
(no sync_batchnorm); running statistics stay local (rank 0's are checkpointed).

Nothing here exists in the reference (single device only, train_denoiser.py:43-48); it is the
multi-GPU row of BASELINE.json.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend=None, timeout_s=None):
    """timeout_s (default D3F_DIST_TIMEOUT or 600): the process group's collective timeout -- a rank that never shows up
    at the rendezvous or a collective that never completes raises on the others instead of hanging the job."""
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        import datetime
        if timeout_s is None:
            timeout_s = float(os.environ.get("D3F_DIST_TIMEOUT", "600"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # D3F_DIST_BACKEND: test hook (two ranks sharing one GPU need gloo; RCCL wants a device per rank)
            backend = os.environ.get("D3F_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=timeout_s), **kwargs)
    return world, rank, local


def shared_seed(group=None):
    """a data-shuffling seed that is the SAME on every rank: PL_GLOBAL_SEED when set (what Lightning's seed_everything
    exports), otherwise rank 0's torch.initial_seed() broadcast to the others.  torch.initial_seed() itself is drawn per
    process, so DistributedSampler(seed=initial_seed) would give every rank a different permutation: overlapping shards,
    images never seen in an epoch."""
    if os.environ.get("PL_GLOBAL_SEED") is not None:
        return int(os.environ["PL_GLOBAL_SEED"]) % (1 << 31)
    seed = int(torch.initial_seed() % (1 << 31))
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        box = [seed]
        if dist.get_backend(group) == "nccl":  # object collectives move through the device with RCCL
            t = torch.tensor(box, dtype=torch.int64, device=torch.device("cuda", torch.cuda.current_device()))
            dist.broadcast(t, src=0, group=group)
            box = [int(t.item())]
        else:
            dist.broadcast_object_list(box, src=0, group=group)
        seed = int(box[0])
    return seed


class BucketAllReducer:
    """sum-all-reduce of gradient buckets, launched as they become ready, joined by wait().

    compress="bf16" (opt-in): every bucket travels as bfloat16 -- 48.9 instead of 97.8 MB per net and step
    over xGMI, whose ring all-reduce is bound per link.  The fp32 bucket is rounded into a staging buffer that lives until
    wait(), summed there by the collective (bf16 sums, as torch DDP's bf16_compress_hook), and written back to the fp32
    gradient in wait(); the conversions are stream-ordered torch copies (plumbing, no host sync).  Default: fp32, exact."""

    def __init__(self, group=None, force=False, compress=None):
        if compress not in (None, "bf16"):
            raise ValueError(f"BucketAllReducer: compress must be None or 'bf16', got {compress!r}")
        self.group = group
        self.works = []
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = force  # issue the collective even for one rank (exercises RCCL on a one-GPU box)
        self.compress = compress
        self._staging = {}   # segment -> bf16 buffer, reused every step
        # timing = True: wait() measures what the collectives cost the optimiser's stream -- the stall between "this
        # stream has reached wait()" and "every bucket's all-reduce (and its write-back) is done", i.e. the part of the
        # exchange that the backward pass did NOT hide.  HIP events on the waiting stream (a work.wait() of the RCCL
        # backend only orders streams, it does not block the host); host wall time for host-blocking backends (gloo).
        self.timing = False
        self._timed = []     # (start event, end event) per wait(), or host seconds
        self.waits = 0

    def configure(self, compress):
        """switch the wire format between steps (bench.py's exchange autotune); never while buckets are in flight"""
        if compress not in (None, "bf16"):
            raise ValueError(f"BucketAllReducer: compress must be None or 'bf16', got {compress!r}")
        if self.works:
            raise RuntimeError("BucketAllReducer.configure() with collectives in flight: call wait() first")
        self.compress = compress

    def exposed_ms(self, reset=True):
        """(mean stall per wait() in ms since the last reset, waits measured); synchronises the device"""
        if not self._timed:
            return None, 0
        total = 0.0
        if torch.cuda.is_available() and not isinstance(self._timed[0], float):
            torch.cuda.synchronize()
        for t in self._timed:
            total += t * 1e3 if isinstance(t, float) else t[0].elapsed_time(t[1])
        n = len(self._timed)
        if reset:
            self._timed = []
        return total / n, n

    def __call__(self, segment, flat_slice):
        if not (self.world_size > 1 or self.force):
            return
        if self.compress is None:
            self.works.append((dist.all_reduce(flat_slice, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, None))
            return
        buf = self._staging.get(segment)
        if buf is None or buf.numel() != flat_slice.numel() or buf.device != flat_slice.device:
            buf = self._staging[segment] = torch.empty(flat_slice.numel(), dtype=torch.bfloat16, device=flat_slice.device)
        buf.copy_(flat_slice)  # on the caller's current stream (the engine's side stream): behind the bucket's gradients
        self.works.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), buf, flat_slice))

    def wait(self):
        timed = self.timing and bool(self.works)
        if timed:
            import time
            on_dev = self.works[0][0] is not None and torch.cuda.is_available() and \
                (dist.get_backend(self.group) == "nccl" if dist.is_initialized() else False)
            if on_dev:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            t0 = time.perf_counter()
        for w, buf, dst in self.works:
            w.wait()
            if buf is not None:
                dst.copy_(buf)  # on the stream that called wait() (the optimiser's), behind the collective
        if timed:
            if on_dev:
                e1.record()
                self._timed.append((e0, e1))
            else:
                self._timed.append(time.perf_counter() - t0)
        self.waits += 1 if self.works else 0
        self.works = []


def autotune_exchange(candidates, time_candidate, group=None, device=None):
    """Pick the data-parallel exchange configuration by MEASUREMENT, identically on every rank.

    candidates: a list of hashable settings (bench.py: (buckets, grad_compress)); time_candidate(c) -> seconds per step of
    THIS rank with setting c (every rank must run the same number of steps: the steps contain collectives).  Per candidate
    the ranks' times are combined with ONE all_reduce(MAX) -- the job's step time is its slowest rank's -- so every rank
    holds the same table and takes the same winner (first minimum: ties go to the earlier candidate).  No rank ever
    decides from its own clock.  Returns (winner, [(candidate, max-over-ranks seconds)])."""
    table = []
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    if multi and device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    for cand in candidates:
        t = float(time_candidate(cand))
        if multi:
            tt = torch.tensor([t], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=group)
            t = float(tt.item())
        table.append((cand, t))
    best = min(range(len(table)), key=lambda i: (table[i][1], i))
    return table[best][0], table


class DataParallel:
    """attach(model, optimizer): broadcast rank 0's parameters, overlap gradient all-reduce with backward."""

    def __init__(self, model, optimizer, group=None, sync_batchnorm=False, grad_compress=None, buckets=2):
        """buckets: exchange buckets per backward pass (2 = (head .. layer3: 92 MB) | (layer2 .. stem: 5.4 MB), the default;
        4 = one per engine segment; 1; or explicit segment ranges -- Unet.set_grad_sync).  2 and 4 both leave only the last
        5.4 MB behind the backward pass; 4 starts the first all-reduce earlier, but every bucket costs the dependent chain
        cross-stream event pairs whether or not bytes move: on one GPU over single-rank RCCL 4 buckets tax the step by 3.2 %,
        2 by 1.3 %, 1 by 0.6 % (bench.py --dp-selftest, profiles/r05_dp_selftest.json).  The 92 MB bucket is final at about
        two thirds of the backward pass and has ~1.5 ms of it left to hide in (0.54 ms at a ring bus bandwidth of 300 GB/s)."""
        self.model, self.optimizer = model, optimizer
        self.buckets = buckets
        self.reducer = BucketAllReducer(group, compress=grad_compress)
        self.world_size = self.reducer.world_size
        if self.world_size > 1 and sync_batchnorm:
            # pytorch_lightning's Trainer(sync_batchnorm=True): batch statistics over every rank's images
            model.set_sync_batchnorm(True if group is None else group, self.world_size)
        if self.world_size > 1:
            model.prepare()
            dist.broadcast(model.flat_params, src=0, group=group)
            dist.broadcast(model.flat_bn_stats, src=0, group=group)
            model.mark_params_changed()
            model.set_grad_sync(self.reducer, buckets)
            optimizer.grad_scale = 1.0 / self.world_size
            optimizer.before_step = self.reducer.wait

    def configure(self, buckets=None, grad_compress="keep"):
        """change the exchange between steps: buckets (as the constructor's) and / or the wire format (None | "bf16")"""
        if buckets is not None:
            self.buckets = buckets
            if self.world_size > 1:
                self.model.set_grad_sync(self.reducer, buckets)
        if grad_compress != "keep":
            self.reducer.configure(grad_compress)


def shard_indices(n, world_size, rank):
    """contiguous, near-equal shard of range(n) for this rank (images are independent)."""
    per = (n + world_size - 1) // world_size
    return range(min(rank * per, n), min((rank + 1) * per, n))
