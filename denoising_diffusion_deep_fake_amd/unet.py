"""`Unet(encoder_name, encoder_weights, in_channels, classes, activation)` -- drop-in for the
`segmentation_models_pytorch.Unet` object the reference builds at
d3f/train_denoiser/lit_module.py:46-52 and d3f/train_deep_fake/lit_module.py:53-59.

What callers rely on and what this class keeps:
  * constructor signature and error behaviour (unknown encoder -> KeyError, H/W not divisible
    by 32 -> RuntimeError);
  * `__call__(Tensor[B,C,H,W] f32) -> Tensor[B,classes,H,W]`, autograd through it;
  * `.parameters()` are ordinary leaf `nn.Parameter`s (torch.optim.Adam works on them),
    `state_dict()` keys are smp's (`encoder.layer1.0.conv1.weight`, `decoder.blocks.0.conv1.0.weight`,
    `segmentation_head.0.bias`, BatchNorm buffers incl. `num_batches_tracked`);
  * `copy.deepcopy`, `.train()/.eval()/.cuda()/.to()` (EMA wrapper, video script).

How it runs: every parameter is a view into ONE flat f32 buffer (likewise gradients and BN
running statistics), and forward/backward are single calls into libd3f_hip.so's whole-network
engine (csrc/engine.hip), which enqueues the hand-written gfx950 kernels on the current
stream.  There is no eager / CPU fallback: tensors must live on a HIP device.
"""
import ctypes as C
import os

import torch
import torch.nn as nn

from . import _lib
from ._lib import D3FError, check, ptr, ptr2, stream_ptr

_DTYPES = {"f32": _lib.F32, "fp32": _lib.F32, "float32": _lib.F32, torch.float32: _lib.F32,
           "bf16": _lib.BF16, "bfloat16": _lib.BF16, torch.bfloat16: _lib.BF16,
           "f32x3": _lib.F32X3}


# ---------------------------------------------------------------------------------------------
# parameter containers: same module tree (hence the same state_dict keys and initialisation)
# as torchvision ResNet + smp UnetDecoder.  Their forward() is never used.
# ---------------------------------------------------------------------------------------------
class _BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes))


class _Encoder(nn.Module):
    def __init__(self, in_channels, blocks):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inplanes = 64
        for li, n in enumerate(blocks, start=1):
            planes = 64 << (li - 1)
            layers = []
            for bi in range(n):
                layers.append(_BasicBlock(inplanes, planes, 2 if (bi == 0 and li > 1) else 1))
                inplanes = planes
            setattr(self, f"layer{li}", nn.Sequential(*layers))
        for m in self.modules():  # torchvision ResNet init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class _DecoderBlock(nn.Module):
    def __init__(self, cin, cskip, cout):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(cin + cskip, cout, 3, padding=1, bias=False),
                                   nn.BatchNorm2d(cout), nn.ReLU(inplace=True))
        self.attention1 = nn.Identity()
        self.conv2 = nn.Sequential(nn.Conv2d(cout, cout, 3, padding=1, bias=False),
                                   nn.BatchNorm2d(cout), nn.ReLU(inplace=True))
        self.attention2 = nn.Identity()


class _Decoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.center = nn.Identity()
        spec = [(512, 256, 256), (256, 128, 128), (128, 64, 64), (64, 64, 32), (32, 0, 16)]
        self.blocks = nn.ModuleList([_DecoderBlock(*s) for s in spec])
        for m in self.modules():  # smp initialize_decoder
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")


_ENCODERS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}


class _Engine:
    """one libd3f_hip whole-network plan + its workspace, for a fixed (B, H, W, dtype)."""

    def __init__(self, encoder_name, in_channels, classes, B, H, W, dtype, device, nets=1, plan_nets=1):
        L = _lib.lib()
        self.h = C.c_void_p()
        if nets == 1 and plan_nets == 1:
            check(L.d3f_unet_create(encoder_name.encode(), in_channels, classes, B, H, W, dtype, C.byref(self.h)))
        else:  # a pair (nets = 2: B images PER network), or one network planned like the pair (plan_nets = 2)
            check(L.d3f_unet_create_nets(encoder_name.encode(), in_channels, classes, B, H, W, dtype, nets, plan_nets,
                                         C.byref(self.h)))
        self.nets = nets
        self.net_stride = L.d3f_unet_net_workspace_stride(self.h)
        self.shape = (B, H, W)
        self.dtype = dtype
        self.workspace = torch.empty(L.d3f_unet_workspace_bytes(self.h), dtype=torch.uint8, device=device)
        if os.environ.get("D3F_POISON_WORKSPACE"):  # test hook: any read-before-write shows up as NaN
            self.workspace.fill_(0xFF)
        self.packed_version = None
        self._bn_cb = None
        self._side = False    # not asked yet
        self.serial = 0       # bumped by every forward that overwrites this workspace
        self.in_use = False   # a recorded autograd graph still needs this workspace's activations
        self.fwd_flops = L.d3f_unet_forward_flops(self.h)
        self.bwd_flops = L.d3f_unet_backward_flops(self.h)
        self.nseg = L.d3f_unet_num_segments(self.h)
        self.seg_ranges = []
        for s in range(self.nseg):
            b, e = C.c_int64(), C.c_int64()
            check(L.d3f_unet_segment_range(self.h, s, C.byref(b), C.byref(e)))
            self.seg_ranges.append((b.value, e.value))

    def set_bn_sync(self, group, world_size):
        """install (group given) or remove (None) the BatchNorm statistics all-reduce of this plan: the C engine calls
        back with a pointer into this workspace, the callback sums it over the ranks of `group` on the current stream"""
        L = _lib.lib()
        if group is None:
            check(L.d3f_unet_set_bn_sync(self.h, None, None, 1))
            self._bn_cb = None
            return
        import torch.distributed as dist
        ws = self.workspace
        base = ws.data_ptr()
        dev = ws.device
        err = self._bn_err = [None]  # the closure holds the workspace / group / this list, never the engine itself

        def allreduce(ctx, data, count, stream):
            try:
                off = int(data) - base
                view = ws[off:off + 4 * int(count)].view(torch.float32)
                # order on the stream the engine names (include/d3f_hip.h: d3f_allreduce_fn), whatever torch's current is
                cur = torch.cuda.current_stream(dev)
                s = cur if int(stream or 0) == cur.cuda_stream else torch.cuda.ExternalStream(int(stream or 0), device=dev)
                with torch.cuda.stream(s):
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=None if group is True else group)
                return 0
            except Exception as e:  # surfaces as the engine's error return (a raise cannot cross the C frame)
                err[0] = e
                return -3
        self._bn_cb = _lib.ALLREDUCE_FN(allreduce)  # kept alive with the engine
        check(L.d3f_unet_set_bn_sync(self.h, C.cast(self._bn_cb, C.c_void_p), None, int(world_size)))

    def side_stream(self, device):
        """the engine's weight-gradient stream as a torch stream (None when the engine runs everything on the caller's
        stream); owned by the engine, valid as long as it lives"""
        if self._side is False:
            sp = C.c_void_p()
            check(_lib.lib().d3f_unet_side_stream(self.h, C.byref(sp)))
            self._side = torch.cuda.ExternalStream(sp.value, device=device) if sp.value else None
        return self._side

    def __del__(self):
        try:
            if self.h:
                _lib.lib().d3f_unet_destroy(self.h)
        except Exception:
            pass


def param_table(encoder_name, in_channels, classes):
    """[(name, shape, offset)] and BN table [(prefix, C, rm_off, rv_off)] from the C engine (host only)."""
    L = _lib.lib()
    h = C.c_void_p()
    check(L.d3f_unet_create(encoder_name.encode(), in_channels, classes, 1, 32, 32, _lib.F32, C.byref(h)))
    try:
        params, bns = [], []
        buf = C.create_string_buffer(256)
        shape = (C.c_int32 * 4)()
        nd, off = C.c_int(), C.c_int64()
        for i in range(L.d3f_unet_num_params(h)):
            check(L.d3f_unet_param_info(h, i, buf, 256, shape, C.byref(nd), C.byref(off)))
            params.append((buf.value.decode(), tuple(shape[k] for k in range(nd.value)), off.value))
        c, rm, rv = C.c_int(), C.c_int64(), C.c_int64()
        for i in range(L.d3f_unet_num_bn(h)):
            check(L.d3f_unet_bn_info(h, i, buf, 256, C.byref(c), C.byref(rm), C.byref(rv)))
            bns.append((buf.value.decode(), c.value, rm.value, rv.value))
        return params, bns, L.d3f_unet_param_floats(h), L.d3f_unet_bnstat_floats(h)
    finally:
        L.d3f_unet_destroy(h)


class _Lease:
    """marks an engine's workspace as holding the activations of a live autograd graph; released by the backward
    pass, or when the graph is dropped without one (the node, and with it this object, is destroyed)."""

    def __init__(self, engine):
        self.engine = engine
        engine.in_use = True

    def release(self):
        if self.engine is not None:
            self.engine.in_use = False
            self.engine = None

    __del__ = release


class _UnetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, engine, anchor):
        # `anchor`: ONE parameter that requires grad -- enough for autograd to record the node and call backward, which
        # writes every parameter's .grad itself; handing all 143 parameters to apply() cost ~0.15 ms of host time per step
        ctx.module, ctx.engine = module, engine
        out = module._run_forward(engine, x, training=True)
        ctx.serial = engine.serial
        ctx.lease = _Lease(engine)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        eng = ctx.engine
        if ctx.lease.engine is None:
            raise D3FError("backward through the d3f Unet a second time: the activations were released by the first "
                           "backward (retain_graph is not supported by the HIP path)")
        if ctx.serial != eng.serial:
            # cannot happen through Unet.forward (a leased workspace is never handed out again); guards direct
            # _run_forward callers: silently using another forward's activations would give wrong gradients
            raise D3FError("the workspace of this forward pass was overwritten by a later forward before backward ran")
        ctx.module._run_backward(eng, grad_out)
        ctx.lease.release()
        return (None, None, None, None)  # every .grad was set by _run_backward (views of the flat gradient buffer)


def _views_unverified(module, incompatible_keys):
    module._rt["verified"] = False


class Unet(nn.Module):
    def __init__(self, encoder_name="resnet34", encoder_weights=None, in_channels=3, classes=3,
                 activation=None, compute_dtype="f32"):
        super().__init__()
        if encoder_name not in _ENCODERS:
            raise KeyError(f"Wrong encoder name `{encoder_name}`, supported encoders: {list(_ENCODERS)}")
        if encoder_weights is not None:
            raise KeyError(f"Wrong pretrained weights `{encoder_weights}` for encoder `{encoder_name}`. "
                           f"Available options are: [None] (no network access)")
        if activation is not None:
            raise ValueError(f"Activation should be None (the reference passes activation=None); got {activation}")
        if compute_dtype not in _DTYPES:
            raise ValueError(f"compute_dtype must be one of f32 / f32x3 / bf16, got {compute_dtype}")
        self.encoder_name, self.in_channels, self.classes = encoder_name, in_channels, classes
        self.compute_dtype = _DTYPES[compute_dtype]
        self.plan_nets = 1  # set_plan_nets(2): this network's kernels are chosen as for a UnetPair (bit-identity runs)
        self.encoder = _Encoder(in_channels, _ENCODERS[encoder_name])
        self.decoder = _Decoder()
        self.segmentation_head = nn.Sequential(nn.Conv2d(16, classes, 3, padding=1), nn.Identity(), nn.Identity())
        nn.init.xavier_uniform_(self.segmentation_head[0].weight)
        nn.init.constant_(self.segmentation_head[0].bias, 0)
        self._init_runtime_state()
        self.register_load_state_dict_post_hook(_views_unverified)  # (load_state_dict(assign=True) swaps Parameter objects)

    # -- runtime state that must not be deep-copied / pickled ------------------------------------
    def _init_runtime_state(self):
        self.__dict__["_rt"] = {"flat": None, "flat_grad": None, "flat_bn": None, "flat_nbt": None,
                                "engines": {}, "table": None, "gen": 0, "grad_sync": None, "params": None,
                                "bn_sync": None}

    def __deepcopy__(self, memo):
        rt = self.__dict__.pop("_rt")
        try:
            cls = self.__class__
            new = cls.__new__(cls)
            memo[id(self)] = new
            import copy
            for k, v in self.__dict__.items():
                new.__dict__[k] = copy.deepcopy(v, memo)
            new._init_runtime_state()
        finally:
            self.__dict__["_rt"] = rt
        return new

    def __getstate__(self):
        st = dict(self.__dict__)
        st.pop("_rt", None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._init_runtime_state()

    # -- flat buffers -----------------------------------------------------------------------------
    def _table(self):
        rt = self._rt
        if rt["table"] is None:
            rt["table"] = param_table(self.encoder_name, self.in_channels, self.classes)
            names = [n for n, _ in self.named_parameters()]
            tnames = [n for n, _, _ in rt["table"][0]]
            if names != tnames:
                raise D3FError("parameter order of the module tree and of the C engine differ")
        return rt["table"]

    @property
    def _param_list(self):
        rt = self._rt
        if rt.get("params") is None:
            rt["params"] = [p for _, p in self.named_parameters()]
        return rt["params"]

    def _ensure_flat(self, device):
        """(re)establish: every parameter / BN buffer is a view into the flat device buffers."""
        rt = self._rt
        flat = rt["flat"]
        if rt.get("verified") and flat is not None and flat.device == device:
            # fast path (every forward comes through here: the full walk below costs ~0.9 ms of host time): the views were
            # verified once and nothing that re-homes parameter storage has run since -- nn.Module._apply (.to / .cuda /
            # .float ...) clears the flag; the first and last parameter and statistics views are still re-checked
            ps, bn = rt["params"], rt["flat_bn"]
            if ps[0].data_ptr() == flat.data_ptr() and ps[-1].data_ptr() + 4 * ps[-1].numel() == flat.data_ptr() + 4 * flat.numel() \
                    and rt["bn_first"].running_mean.data_ptr() == bn.data_ptr():
                return
        ptable, btable, nparam, nbn = self._table()
        named = dict(self.named_parameters())
        ok = flat is not None and flat.device == device
        if ok:
            base = flat.data_ptr()
            for name, shape, off in ptable:
                p = named[name]
                if p.data_ptr() != base + 4 * off or p.device != device:
                    ok = False
                    break
        if ok:
            bufs = dict(self.named_buffers())
            bb = rt["flat_bn"].data_ptr()
            for prefix, c, rm, rv in btable:
                if bufs[prefix + ".running_mean"].data_ptr() != bb + 4 * rm:
                    ok = False
                    break
        if ok:
            rt["verified"] = True
            rt["bn_first"] = dict(self.named_modules())[btable[0][0]]
            return
        flat = torch.empty(nparam, dtype=torch.float32, device=device)
        flat_bn = torch.empty(nbn, dtype=torch.float32, device=device)
        flat_nbt = torch.zeros(len(btable), dtype=torch.int64, device=device)
        with torch.no_grad():
            for name, shape, off in ptable:
                p = named[name]
                if tuple(p.shape) != tuple(shape):
                    raise D3FError(f"parameter {name}: shape {tuple(p.shape)} != engine {shape}")
                view = flat[off:off + p.numel()].view(shape)
                view.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = view
            mods = dict(self.named_modules())
            for i, (prefix, c, rm, rv) in enumerate(btable):
                bn = mods[prefix]
                for attr, o in (("running_mean", rm), ("running_var", rv)):
                    view = flat_bn[o:o + c]
                    view.copy_(getattr(bn, attr).to(device=device, dtype=torch.float32))
                    setattr(bn, attr, view)
                flat_nbt[i] = bn.num_batches_tracked.to(device)
                bn.num_batches_tracked = flat_nbt[i]
        rt.update(flat=flat, flat_bn=flat_bn, flat_nbt=flat_nbt, flat_grad=None, grad_views=None, gen=rt.get("gen", 0) + 1,
                  params=[named[name] for name, _, _ in ptable], verified=True, bn_first=mods[btable[0][0]])
        for p in named.values():
            p.grad = None

    def set_plan_nets(self, plan_nets):
        """plan_nets = 2: every plan of this network makes the tile / split-K / slab / patch-kernel choices of a UnetPair
        (the workgroups of two networks counted together), so that stepping it alone computes bit for bit what the pair
        computes for it -- the reference side of the pair's bit-identity test.  Drops the existing plans."""
        if plan_nets not in (1, 2):
            raise ValueError("plan_nets must be 1 or 2")
        if plan_nets != getattr(self, "plan_nets", 1):
            self.plan_nets = plan_nets
            self._rt["engines"] = {}
            self._rt.pop("last_engine", None)
        return self

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .cpu() / .float() ...: torch replaces every parameter's storage -- the flat views are gone
        out = super()._apply(fn, *args, **kwargs)
        self._rt["verified"] = False
        return out

    def mark_params_changed(self):
        """call after writing the flat parameter buffer behind autograd's back (fused Adam / EMA)."""
        # a generation counter, not a flag: several plans (this network's own engines per shape, a UnetPair's engines)
        # hold packed copies of these weights and each compares its own stamp -- a flag cleared by whichever plan packed
        # first would leave the others on stale layouts
        self._rt["gen"] = self._rt.get("gen", 0) + 1

    @property
    def flat_params(self):
        return self._rt["flat"]

    @property
    def flat_grads(self):
        return self._rt["flat_grad"]

    @property
    def flat_bn_stats(self):
        return self._rt["flat_bn"]

    def prepare(self, device=None):
        """flatten parameters now (otherwise done lazily by the first forward); returns self."""
        device = torch.device(device) if device is not None else next(self.parameters()).device
        if device.type != "cuda":
            raise D3FError("d3f Unet runs on an MI355X (HIP) device only; there is no CPU fallback")
        self._ensure_flat(device)
        return self

    def set_sync_batchnorm(self, group=True, world_size=None):
        """Synchronised BatchNorm statistics over the ranks of `group` (True: the default process group; None: off = the
        default, per-GPU statistics -- what Lightning DDP does with the reference's Trainer flags).
        With it on, N ranks x bs are numerically one process with batch N*bs: every train-mode BatchNorm layer
        all-reduces its statistics in the forward pass and its two gradient sums in the backward pass (2 x 46 small
        collectives per step -- latency-bound, use it when batch statistics over bs/GPU images are too noisy)."""
        import torch.distributed as dist
        if group is not None:
            if not dist.is_initialized():
                raise D3FError("set_sync_batchnorm needs an initialised torch.distributed process group")
            world_size = world_size or dist.get_world_size(None if group is True else group)
        self._rt["bn_sync"] = None if group is None else (group, int(world_size))
        for pool in self._rt["engines"].values():
            for eng in pool:
                eng.set_bn_sync(*(self._rt["bn_sync"] or (None, 1)))

    def set_grad_sync(self, fn, buckets=2):
        """fn(bucket_index, flat_grad_slice) is called as soon as a gradient bucket is final
        (data-parallel all-reduce overlap); None disables.

        buckets: how the engine's backward segments (4: head + decoder | layer4 | layer3 | layer2 .. stem, final in that
        order) are grouped into exchange buckets -- None / 4: one bucket per segment; 2 (default, as DataParallel and
        bench.py: the machinery costs 1.3 % of a step against 3.2 % for 4): (head .. layer3: 92 MB, final at
        about two thirds of the backward pass) | (layer2 .. stem: 5.4 MB, the only bytes exchanged behind the backward
        pass); 1: one bucket = the whole gradient at the end of backward; or an explicit list of (first, end) segment
        ranges that tile range(nseg) in order.  Every bucket costs the chain a cross-stream event pair and RCCL a launch; fewer
        buckets expose more of the LAST bucket's all-reduce behind the backward pass."""
        self._rt["grad_sync"] = fn
        self._rt["grad_buckets"] = buckets

    @staticmethod
    def _bucket_groups(buckets, nseg):
        if isinstance(buckets, bool):
            raise D3FError(f"gradient buckets: {buckets!r} (a count 1 / 2 / {nseg}, None, or a list of segment ranges)")
        if buckets is None or buckets == nseg:
            return [(s, s + 1) for s in range(nseg)]
        if isinstance(buckets, int):
            if buckets == 1:
                return [(0, nseg)]
            if buckets == 2 and nseg >= 2:
                return [(0, nseg - 1), (nseg - 1, nseg)]
            raise D3FError(f"gradient buckets: {buckets} (the engine has {nseg} backward segments: 1, 2 or {nseg})")
        groups = [(int(b), int(e)) for b, e in buckets]
        if not groups or groups[0][0] != 0 or groups[-1][1] != nseg or any(b >= e for b, e in groups) or \
                any(groups[i][1] != groups[i + 1][0] for i in range(len(groups) - 1)):
            raise D3FError(f"gradient buckets {groups} do not tile the engine's {nseg} backward segments in order")
        return groups

    def set_early_update(self, fn):
        """fn(flat_grad, lo, hi) is called INSIDE backward, on the caller's stream behind the chain's last kernel and a
        wait for "the gradients of every bucket but the last are final" (flat range [lo, hi): layer3, layer4, decoder and
        head -- 94 % of the parameters), BEFORE the join with the side stream on which the last bucket's weight gradients
        (layer2, layer1, stem) are still running.  An optimiser that updates [lo, hi) in the hook runs next to that
        tail instead of behind it (FusedAdam(overlap_tail=True)).  Only
        taken when the gradients land in the flat buffer directly (every .grad None before backward) and no
        data-parallel reducer is attached; None disables."""
        self._rt["early_update"] = fn

    # -- engine -------------------------------------------------------------------------------------
    MAX_LIVE_GRAPHS = 4  # workspaces (2 GB each at bs 16, 256x256) per shape that may hold un-backwarded graphs

    def _engine(self, B, H, W, device):
        """a plan + workspace for this shape whose activations no live autograd graph still needs.  smp.Unet allows
        `crit(net(x1)) + crit(net(x2))` and a no_grad forward between forward and backward: each recorded forward
        leases its workspace until its backward has run, and a further forward of the same shape gets another one."""
        key = (B, H, W, self.compute_dtype, device.index, getattr(self, "plan_nets", 1))
        pool = self._rt["engines"].setdefault(key, [])
        for eng in pool:
            if not eng.in_use:
                return eng
        if len(pool) >= self.MAX_LIVE_GRAPHS:
            raise D3FError(f"{len(pool)} forward passes of shape {(B, H, W)} are waiting for their backward pass; "
                           f"run inference-only forwards under torch.no_grad()")
        eng = _Engine(self.encoder_name, self.in_channels, self.classes, B, H, W, self.compute_dtype, device,
                      plan_nets=getattr(self, "plan_nets", 1))
        if self._rt.get("bn_sync"):
            eng.set_bn_sync(*self._rt["bn_sync"])
        pool.append(eng)
        return eng

    def _pack_if_needed(self, eng):
        rt = self._rt
        # `p.data = view` keeps every parameter's OWN version counter, so in-place updates by
        # torch optimizers / load_state_dict show up on the parameters, not on the flat buffer
        ver = self._weights_version()
        if eng.packed_version != ver:
            check(_lib.lib().d3f_unet_pack_weights(eng.h, ptr(rt["flat"]), ptr(eng.workspace), stream_ptr()))
            eng.packed_version = ver

    def _weights_version(self):
        """stamp of the current parameter values: torch's version counters (in-place torch updates, load_state_dict) and the
        generation that mark_params_changed() bumps for raw-pointer updates (fused Adam, EMA lerp, re-homed buffers)"""
        rt = self._rt
        return (sum(p._version for p in rt["params"]) + rt["flat"]._version, rt.get("gen", 0))

    def _run_forward(self, eng, x, training):
        rt = self._rt
        self._pack_if_needed(eng)
        out = torch.empty((x.shape[0], self.classes, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device)
        eng.serial += 1
        rt["last_engine"] = eng
        check(_lib.lib().d3f_unet_forward(eng.h, ptr(rt["flat"]), ptr(rt["flat_bn"]), ptr(x), ptr(out),
                                          ptr(eng.workspace), 1 if training else 0, stream_ptr()))
        if training:
            rt["flat_nbt"] += 1
        return out

    def _run_backward(self, eng, grad_out):
        rt = self._rt
        L = _lib.lib()
        grad_out = grad_out.contiguous().float()
        target, direct = self._backward_target()
        sync = rt["grad_sync"]
        early = rt.get("early_update") if (sync is None and direct and eng.nseg > 1) else None
        side = eng.side_stream(grad_out.device) if early is not None else None
        if early is not None and side is not None:
            # every bucket is enqueued without a join; the leading buckets' gradients are final on the side stream at the
            # event, and the hook's update runs on THIS stream behind the chain's last kernel -- next to the last bucket's
            # weight gradients, which are still running on the side stream -- before the join that step() would wait for
            last = eng.nseg - 1
            lo, hi = min(b for b, _ in eng.seg_ranges[:last]), max(e for _, e in eng.seg_ranges[:last])
            if sum(e - b for b, e in eng.seg_ranges[:last]) != hi - lo:
                raise D3FError("early update: the leading gradient buckets are not one contiguous range")
            check(L.d3f_unet_backward_nojoin(eng.h, ptr(rt["flat"]), ptr(grad_out), ptr(target), ptr(eng.workspace),
                                             0, last, stream_ptr()))
            final = side.record_event()
            check(L.d3f_unet_backward_nojoin(eng.h, ptr(rt["flat"]), ptr(grad_out), ptr(target), ptr(eng.workspace),
                                             last, eng.nseg, stream_ptr()))
            torch.cuda.current_stream().wait_event(final)
            early(target, lo, hi)
            check(L.d3f_unet_backward_join(eng.h, stream_ptr()))
        elif sync is None:
            check(L.d3f_unet_backward(eng.h, ptr(rt["flat"]), ptr(grad_out), ptr(target), ptr(eng.workspace),
                                      0, eng.nseg, stream_ptr()))
        else:
            # Data parallel: bucket k's collective must wait for bucket k's gradients -- and nothing else may wait for
            # anything.  The gradients become final on the engine's SIDE stream (weight gradients run there; it also
            # waits for this stream at the end of each bucket), so the hook is called with the side stream current:
            # torch.distributed orders a collective behind the current stream at the time of the call.  This stream
            # (the dependent BatchNorm-backward -> data-gradient chain, the critical path) goes straight on with bucket
            # k+1; one join after the last bucket orders the optimiser behind the weight gradients.
            side = eng.side_stream(grad_out.device)
            for k, (s0, s1) in enumerate(self._bucket_groups(rt.get("grad_buckets"), eng.nseg)):
                check(L.d3f_unet_backward_nojoin(eng.h, ptr(rt["flat"]), ptr(grad_out), ptr(target),
                                                 ptr(eng.workspace), s0, s1, stream_ptr()))
                b, e = min(r[0] for r in eng.seg_ranges[s0:s1]), max(r[1] for r in eng.seg_ranges[s0:s1])
                if sum(r[1] - r[0] for r in eng.seg_ranges[s0:s1]) != e - b:
                    raise D3FError(f"gradient bucket {k}: segments {s0}..{s1 - 1} are not one contiguous flat range")
                if side is None:  # D3F_SERIAL_BACKWARD: everything is on this stream
                    sync(k, target[b:e])
                else:
                    with torch.cuda.stream(side):
                        sync(k, target[b:e])
            check(L.d3f_unet_backward_join(eng.h, stream_ptr()))
        self._publish_grads(target, direct)

    def _backward_target(self):
        """(flat buffer the engine writes this pass's gradients into, direct): direct = every .grad is None, the gradients
        land in the module's own flat gradient buffer and the parameters' .grad become views of it"""
        rt = self._rt
        direct = all(p.grad is None for p in self._param_list)
        if rt["flat_grad"] is None:
            rt["flat_grad"] = torch.empty_like(rt["flat"])
            rt["grad_views"] = None
        rt["backward_calls"] = rt.get("backward_calls", 0) + 1
        return (rt["flat_grad"] if direct else torch.empty_like(rt["flat"])), direct

    def _publish_grads(self, target, direct):
        rt = self._rt
        params = self._param_list
        ptable = self._table()[0]
        if direct:
            # the same 143 view tensors of the flat gradient buffer every step (making them anew cost ~0.3 ms of host time)
            views = rt.get("grad_views")
            if views is None or views[0].data_ptr() != target.data_ptr():
                views = rt["grad_views"] = [target[off:off + p.numel()].view(shape) for (_, shape, off), p in zip(ptable, params)]
            for p, view in zip(params, views):
                if p.requires_grad:
                    p.grad = view
            return
        for (name, shape, off), p in zip(ptable, params):
            if not p.requires_grad:
                continue
            view = target[off:off + p.numel()].view(shape)
            if p.grad is None:
                p.grad = view.clone()
            else:
                p.grad.add_(view)

    # -- nn.Module API ------------------------------------------------------------------------------
    def check_input_shape(self, x):
        h, w = x.shape[-2:]
        if h % 32 != 0 or w % 32 != 0:
            new_h = (h // 32 + 1) * 32 if h % 32 != 0 else h
            new_w = (w // 32 + 1) * 32 if w % 32 != 0 else w
            raise RuntimeError(
                f"Wrong input shape height={h}, width={w}. Expected image height and width divisible by 32. "
                f"Consider pad your images to shape ({new_h}, {new_w}).")

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise RuntimeError(f"Expected input [B, {self.in_channels}, H, W], got {list(x.shape)}")
        self.check_input_shape(x)
        if x.device.type != "cuda":
            raise D3FError("d3f Unet runs on an MI355X (HIP) device only; there is no CPU fallback "
                           "(input tensor is on %s)" % x.device)
        self._ensure_flat(x.device)
        xin = x.detach().contiguous().float()
        eng = self._engine(x.shape[0], x.shape[2], x.shape[3], x.device)
        params = self._param_list
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        if need_grad:
            if not self.training:
                raise D3FError("backward through an eval-mode Unet is not supported (BatchNorm is folded)")
            if x.requires_grad:
                raise D3FError("gradient w.r.t. the network input is not computed by the HIP path")
            return _UnetFunction.apply(xin, self, eng, next(p for p in params if p.requires_grad))
        return self._run_forward(eng, xin, training=self.training)

    @torch.no_grad()
    def forward_graph(self, x, out=None):
        """Eval-mode forward replayed from a hipGraph (BASELINE.json configs[4]: "hipGraph-captured denoise step"; the
        reference's frame loop is d3f/script_tools/put_video_through_fake_model.py:111-119).  The graph is captured on
        the first call per (x, out) buffer pair and bakes those pointers in, so a sampling loop passes the SAME two
        buffers every iteration (write the next input into `x` in place); results are bit-identical to `self(x)` in
        eval mode.  Parameter updates between calls are picked up (values are read at replay time)."""
        if self.training:
            raise D3FError("forward_graph is the eval-mode (BatchNorm folded) forward: call .eval() first")
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise RuntimeError(f"Expected input [B, {self.in_channels}, H, W], got {list(x.shape)}")
        self.check_input_shape(x)
        if x.device.type != "cuda":
            raise D3FError("d3f Unet runs on an MI355X (HIP) device only; there is no CPU fallback")
        if x.dtype != torch.float32 or not x.is_contiguous():
            raise ValueError("forward_graph bakes the input pointer into the graph: pass a contiguous float32 tensor")
        shape = (x.shape[0], self.classes, x.shape[2], x.shape[3])
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=x.device)
        elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
            raise ValueError(f"out must be a contiguous float32 tensor of shape {shape} on the input's device")
        self._ensure_flat(x.device)
        eng = self._engine(x.shape[0], x.shape[2], x.shape[3], x.device)
        self._pack_if_needed(eng)
        rt = self._rt
        eng.serial += 1
        rt["last_engine"] = eng
        check(_lib.lib().d3f_unet_forward_graph(eng.h, ptr(rt["flat"]), ptr(rt["flat_bn"]), ptr(x), ptr(out),
                                                ptr(eng.workspace), stream_ptr()))
        eng.keep_graph = (x, out)  # the captured graph holds these pointers: keep them alive with the engine
        return out

    @torch.no_grad()
    def predict_u8(self, frames_bgr, mean, std, graph=True, out=None):
        """Inference on uint8 BGR frames ([H,W,3] or [B,H,W,3] on the HIP device) -> uint8 BGR frames.

        Eval-mode forward (BatchNorm running statistics folded into the conv epilogues) with the reference's
        `cv2_to_tensor_normalised` / `tensor_cv2_to_denormalised` (d3f/train_deep_fake/lit_module.py:272-300) fused
        into the first and last kernel; `graph=True` replays a hipGraph captured per (input, output) buffer pair,
        so pass the same buffers again (`out=`) in a frame loop."""
        if frames_bgr.dtype != torch.uint8 or frames_bgr.shape[-1] != 3 or frames_bgr.dim() not in (3, 4):
            raise ValueError("predict_u8 expects uint8 frames [H, W, 3] or [B, H, W, 3] in BGR order")
        if frames_bgr.device.type != "cuda":
            raise D3FError("predict_u8 needs frames on the HIP device (no CPU fallback)")
        single = frames_bgr.dim() == 3
        x = frames_bgr.unsqueeze(0) if single else frames_bgr
        x = x.contiguous()
        B, H, W, _ = x.shape
        if H % 32 or W % 32:
            raise RuntimeError(f"Wrong input shape height={H}, width={W}. Expected image height and width "
                               f"divisible by 32.")
        self._ensure_flat(x.device)
        eng = self._engine(B, H, W, x.device)
        self._pack_if_needed(eng)
        if out is None:
            out = torch.empty_like(x)
        elif out.shape != x.shape or out.dtype != torch.uint8 or not out.is_contiguous():
            raise ValueError("out must be a contiguous uint8 tensor shaped like the input batch")
        rt = self._rt
        m = (C.c_float * 3)(*[float(v) for v in mean])
        sd = (C.c_float * 3)(*[float(v) for v in std])
        check(_lib.lib().d3f_unet_predict_u8(eng.h, ptr(rt["flat"]), ptr(rt["flat_bn"]), ptr(x), ptr(out), m, sd,
                                             ptr(eng.workspace), 1 if graph else 0, stream_ptr()))
        eng.keep = (x, out)  # the captured graph bakes these pointers in: keep them alive with the engine
        return out[0] if single else out

    def export_activation(self, name):
        """debugging / parity tests: an internal tensor of the MOST RECENT forward (+ backward) as NCHW f32 --
        "<conv name>:y" raw conv output, ":a" post BatchNorm(+residual)+ReLU activation, ":da" gradient w.r.t. that
        activation (d3f_unet_export).  Channels beyond the real count (vector padding) are cut off by the caller."""
        eng = self._rt.get("last_engine")
        if eng is None:
            raise D3FError("export_activation() before any forward pass")
        out = torch.empty(self.export_activation_shape(name), dtype=torch.float32, device=eng.workspace.device)
        check(_lib.lib().d3f_unet_export(eng.h, name.encode(), ptr(eng.workspace), ptr(out), stream_ptr()))
        return out

    def export_activation_shape(self, name):
        eng = self._rt.get("last_engine")
        if eng is None:
            raise D3FError("export_activation_shape() before any forward pass")
        dims = (C.c_int32 * 3)()
        check(_lib.lib().d3f_unet_export_shape(eng.h, name.encode(), dims))
        return (eng.shape[0], dims[0], dims[1], dims[2])

    # flops of the conv contractions of one call (2*MAC), for roofline reporting
    def conv_flops(self, B, H, W, device=None):
        device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        eng = self._engine(B, H, W, device)
        return eng.fwd_flops, eng.bwd_flops


# ---------------------------------------------------------------------------------------------
# two networks, one set of launches
# ---------------------------------------------------------------------------------------------
class _UnetPairFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xa, xb, pair, engine, anchor_a, anchor_b):
        ctx.pair, ctx.engine = pair, engine
        out = pair._run_forward(engine, xa, xb)
        ctx.serial = engine.serial
        ctx.lease = _Lease(engine)
        return out

    @staticmethod
    def backward(ctx, ga, gb):
        eng = ctx.engine
        if ctx.lease.engine is None:
            raise D3FError("backward through the d3f UnetPair a second time: the activations were released by the first "
                           "backward (retain_graph is not supported by the HIP path)")
        if ctx.serial != eng.serial:
            raise D3FError("the workspace of this pair forward was overwritten by a later forward before backward ran")
        # (autograd materialises the gradient of an output the loss does not depend on as zeros: that network's
        # parameter gradients then come out as exact zeros)
        ctx.pair._run_backward(eng, ga, gb)
        ctx.lease.release()
        return (None,) * 6


class UnetPair:
    """Two `Unet`s of identical architecture stepped as ONE set of kernel launches.

    train_deep_fake's `mode: "denoise"` trains model_a on domain a and model_b on domain b with nothing shared
    (d3f/train_deep_fake/lit_module.py:142-181): two independent forward / backward passes of 8 images each per batch, every
    launch half-filling the chip.  `UnetPair(model_a, model_b)(xa, xb)` runs both networks' layer k as one launch
    (libd3f_hip.so: d3f_unet_pair_forward / d3f_unet_pair_backward -- gridDim.z carries the network) with the tile choices
    of a 16-image batch, and returns (prediction_a, prediction_b) wired into autograd: ONE backward pass fed both output
    gradients, `torch.autograd.backward([loss_a, loss_b])`, writes both networks' .grad.  The two modules keep their own
    parameters, gradients, BatchNorm statistics and optimisers; values are bit for bit those of each network stepped alone
    with `Unet.set_plan_nets(2)`.  Train mode, per-GPU BatchNorm statistics."""

    def __init__(self, net_a, net_b):
        if not (isinstance(net_a, Unet) and isinstance(net_b, Unet)) or net_a is net_b:
            raise TypeError("UnetPair takes two distinct d3f Unet modules")
        for attr in ("encoder_name", "in_channels", "classes", "compute_dtype"):
            if getattr(net_a, attr) != getattr(net_b, attr):
                raise ValueError(f"UnetPair: the two networks differ in {attr}")
        self.nets = (net_a, net_b)
        self._engines = {}
        self.last_engine = None

    # plans and workspaces are runtime state (C handles): a copy / pickle of whatever holds the pair starts without them
    def __deepcopy__(self, memo):
        import copy
        return UnetPair(copy.deepcopy(self.nets[0], memo), copy.deepcopy(self.nets[1], memo))

    def __getstate__(self):
        return {"nets": self.nets}

    def __setstate__(self, st):
        self.nets = st["nets"]
        self._engines = {}
        self.last_engine = None

    def _engine(self, B, H, W, device):
        a = self.nets[0]
        key = (B, H, W, a.compute_dtype, device.index)
        pool = self._engines.setdefault(key, [])
        for eng in pool:
            if not eng.in_use:
                return eng
        if len(pool) >= Unet.MAX_LIVE_GRAPHS:
            raise D3FError(f"{len(pool)} pair forward passes of shape {(B, H, W)} are waiting for their backward pass")
        eng = _Engine(a.encoder_name, a.in_channels, a.classes, B, H, W, a.compute_dtype, device, nets=2, plan_nets=2)
        pool.append(eng)
        return eng

    def _pack_if_needed(self, eng):
        ver = (self.nets[0]._weights_version(), self.nets[1]._weights_version())
        if eng.packed_version != ver:
            a, b = self.nets
            check(_lib.lib().d3f_unet_pair_pack_weights(eng.h, ptr2(a._rt["flat"], b._rt["flat"]), ptr(eng.workspace),
                                                        stream_ptr()))
            eng.packed_version = ver

    def _run_forward(self, eng, xa, xb):
        a, b = self.nets
        self._pack_if_needed(eng)
        shape = (xa.shape[0], a.classes, xa.shape[2], xa.shape[3])
        oa = torch.empty(shape, dtype=torch.float32, device=xa.device)
        ob = torch.empty(shape, dtype=torch.float32, device=xa.device)
        eng.serial += 1
        self.last_engine = eng
        check(_lib.lib().d3f_unet_pair_forward(eng.h, ptr2(a._rt["flat"], b._rt["flat"]),
                                               ptr2(a._rt["flat_bn"], b._rt["flat_bn"]), ptr2(xa, xb), ptr2(oa, ob),
                                               ptr(eng.workspace), stream_ptr()))
        for net in self.nets:
            net._rt["flat_nbt"] += 1
        return oa, ob

    def _run_backward(self, eng, ga, gb):
        a, b = self.nets
        L = _lib.lib()
        ga, gb = ga.contiguous().float(), gb.contiguous().float()
        (ta, da), (tb, db) = a._backward_target(), b._backward_target()
        params, douts, grads = ptr2(a._rt["flat"], b._rt["flat"]), ptr2(ga, gb), ptr2(ta, tb)
        syncs = (a._rt["grad_sync"], b._rt["grad_sync"])
        if syncs[0] is None and syncs[1] is None:
            check(L.d3f_unet_pair_backward(eng.h, params, douts, grads, ptr(eng.workspace), 0, eng.nseg, 1, stream_ptr()))
        else:
            # data parallel (as Unet._run_backward): bucket k's collectives -- one per network, each module's own reducer
            # -- wait for bucket k's gradients on the engine's side stream, the chain goes straight on with bucket k + 1
            if syncs[0] is None or syncs[1] is None:
                raise D3FError("UnetPair: attach the data-parallel reducer to both networks or to neither")
            if a._rt.get("grad_buckets") != b._rt.get("grad_buckets"):
                raise D3FError("UnetPair: the two networks' gradient bucket settings differ")
            side = eng.side_stream(ga.device)
            for k, (s0, s1) in enumerate(Unet._bucket_groups(a._rt.get("grad_buckets"), eng.nseg)):
                check(L.d3f_unet_pair_backward(eng.h, params, douts, grads, ptr(eng.workspace), s0, s1, 0, stream_ptr()))
                lo, hi = min(r[0] for r in eng.seg_ranges[s0:s1]), max(r[1] for r in eng.seg_ranges[s0:s1])
                if sum(r[1] - r[0] for r in eng.seg_ranges[s0:s1]) != hi - lo:
                    raise D3FError(f"gradient bucket {k}: segments {s0}..{s1 - 1} are not one contiguous flat range")
                with (torch.cuda.stream(side) if side is not None else _nullcontext()):
                    syncs[0](k, ta[lo:hi])
                    syncs[1](k, tb[lo:hi])
            check(L.d3f_unet_backward_join(eng.h, stream_ptr()))
        a._publish_grads(ta, da)
        b._publish_grads(tb, db)

    def usable(self, xa, xb):
        """can this batch go through the pair?  Same shape, both networks training on one HIP device, and neither with
        synchronised BatchNorm statistics (the pair engine keeps per-GPU statistics)"""
        a, b = self.nets
        return (xa.shape == xb.shape and xa.device == xb.device and xa.device.type == "cuda" and a.training and b.training
                and not a._rt.get("bn_sync") and not b._rt.get("bn_sync"))

    def __call__(self, xa, xb):
        a, b = self.nets
        if xa.shape != xb.shape or xa.dim() != 4 or xa.shape[1] != a.in_channels:
            raise RuntimeError(f"UnetPair expects two inputs of one shape [B, {a.in_channels}, H, W], got "
                               f"{list(xa.shape)} and {list(xb.shape)}")
        a.check_input_shape(xa)
        if xa.device.type != "cuda" or xb.device != xa.device:
            raise D3FError("d3f UnetPair runs on one MI355X (HIP) device; there is no CPU fallback")
        if not (a.training and b.training):
            raise D3FError("UnetPair runs the train-mode passes of train_deep_fake's denoise step; eval-mode inference "
                           "goes through each Unet")
        if a._rt.get("bn_sync") or b._rt.get("bn_sync"):
            raise D3FError("UnetPair keeps per-GPU BatchNorm statistics; switch set_sync_batchnorm off or step the networks alone")
        for net in self.nets:
            net._ensure_flat(xa.device)
        xa, xb = xa.detach().contiguous().float(), xb.detach().contiguous().float()
        eng = self._engine(xa.shape[0], xa.shape[2], xa.shape[3], xa.device)
        need = [any(p.requires_grad for p in net._param_list) for net in self.nets]
        if torch.is_grad_enabled() and any(need):
            if not all(need):
                raise D3FError("UnetPair: both networks (or neither) must require gradients")
            if xa.requires_grad or xb.requires_grad:
                raise D3FError("gradient w.r.t. the network input is not computed by the HIP path")
            anchors = [next(p for p in net._param_list if p.requires_grad) for net in self.nets]
            return _UnetPairFunction.apply(xa, xb, self, eng, anchors[0], anchors[1])
        return self._run_forward(eng, xa, xb)

    def export_activation(self, net_index, name):
        """`Unet.export_activation` for network 0 / 1 of the most recent pair pass"""
        eng = self.last_engine
        if eng is None:
            raise D3FError("export_activation() before any pair forward pass")
        dims = (C.c_int32 * 3)()
        check(_lib.lib().d3f_unet_export_shape(eng.h, name.encode(), dims))
        out = torch.empty((eng.shape[0], dims[0], dims[1], dims[2]), dtype=torch.float32, device=eng.workspace.device)
        ws = C.c_void_p(eng.workspace.data_ptr() + int(net_index) * eng.net_stride)
        check(_lib.lib().d3f_unet_export(eng.h, name.encode(), ws, ptr(out), stream_ptr()))
        return out


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False
