#!/usr/bin/env python3
"""Headline benchmark: training images/sec of the d3f noisy->clean training step (noise blend ->
U-Net forward -> (MSE + 1-SSIM)/2 -> backward -> Adam) on synthetic 256x256 face crops, bs=16 per GPU.

    python bench.py --gpus N --steps K --warmup W
N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
(one rank per GPU over RCCL), or -- when WORLD_SIZE is not set -- this process starts exactly that command as a
CHILD (before any HIP call of its own) and relays rank 0's JSON line and the exit code.
Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_START = time.perf_counter()
# dense MFMA peaks, MI355X_MICROARCH.md; f32x3 forms every fp32 product from six bf16 MFMAs, so its
# fp32-equivalent ceiling is the bf16 peak / 6
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f32x3": 2500.0 / 6}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f32x3", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frame-size", type=int, default=448, help="--workload predict: frame height = width")
    ap.add_argument("--no-alt", action="store_true", help="skip the secondary f32x3 / bf16 measurements of the default run")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket the contraction kernels with HIP events in the timed region")
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--overlap-tail", default="off", choices=["on", "off"],
                    help="FusedAdam(overlap_tail=...): Adam over every gradient bucket but the last inside backward (single "
                         "process; a data-parallel reducer keeps the whole update in step()).  Bit-identical values.")
    ap.add_argument("--fit-source", default="synthetic", choices=["synthetic", "jpeg", "png"],
                    help="--workload fit: `synthetic: true` dataset, or an on-disk image list written to --fit-dir and "
                         "decoded by ImageDataset (PIL)")
    ap.add_argument("--fit-workers", type=int, default=8, help="--workload fit: DataLoader workers (the YAML's num_workers)")
    ap.add_argument("--fit-files", type=int, default=2048, help="--workload fit: images written to disk (epochs cycle)")
    ap.add_argument("--fit-dir", default="/tmp/d3f_fit_data")
    ap.add_argument("--fit-uint8", default="off", choices=["on", "off"],
                    help="--workload fit with an on-disk list: `uint8_batches: true` -- workers hand over HWC uint8 images, "
                         "Normalize + ToTensor run on the GPU (bit-identical)")
    ap.add_argument("--fit-pin", default="on", choices=["on", "off"], help="--workload fit: DataLoader(pin_memory=...)")
    ap.add_argument("--fit-module", default="denoiser", choices=["denoiser", "deepfake"],
                    help="--workload fit: train_denoiser (one net, bs = --batch) or train_deep_fake's denoise mode (two "
                         "nets, two image lists behind a CombinedLoader, bs = --batch per domain: BASELINE configs[3])")
    ap.add_argument("--fit-augment", default="on", choices=["on", "off"],
                    help="--workload fit: the random affine warp of training_step (the reference always augments)")
    ap.add_argument("--workload", default="denoiser", choices=["denoiser", "deepfake", "sample50", "predict", "fit"],
                    help="denoiser: headline (train_denoiser step); deepfake: paired-domain train_deep_fake "
                         "step (BASELINE config 3, bs 8 per domain); sample50: 50 eval-mode forwards of a "
                         "batch of 64 (BASELINE config 4)")
    ap.add_argument("--mode", default="denoise", choices=["denoise", "swap"],
                    help="--workload deepfake: train_deep_fake's `mode` -- denoise (denoise_config.yml, 50 epochs) or swap "
                         "(swap_config.yml, the reference's 200-epoch phase: EMA update + teacher forward under no_grad + "
                         "student step per optimizer, d3f/train_deep_fake/lit_module.py:183-206)")
    ap.add_argument("--pair-fused", default="auto", choices=["auto", "on", "off"],
                    help="--workload deepfake, denoise mode: the two nets' steps as ONE set of launches (trainer.optimizer_steps "
                         "takes the fused route when the module offers it); off = Lightning's one-after-the-other loop")
    ap.add_argument("--pair-batch", type=int, default=8,
                    help="--workload deepfake: images per domain and step (BASELINE configs[3]: 8; the authors' "
                         "denoise_config.yml / swap_config.yml: 14 at --size 448)")
    ap.add_argument("--pair-plan", default="off", choices=["on", "off"],
                    help="--workload deepfake: every network plans its kernels as for a UnetPair (hparam pair_plan: the "
                         "tile / split-K / slab choices of a 16-image batch on 8-image launches) -- the A/B of the plan "
                         "heuristics at 8 images per launch")
    ap.add_argument("--graph-step", default="off", choices=["on", "off"],
                    help="on: the whole optimiser step replayed from one captured hipGraph (graph_step.py; single GPU).  "
                         "Bit-identical to the eager step, and SLOWER on this ROCm (measured r03: bf16 4.6 -> 10.9 ms, "
                         "128x128 3.8 -> 11.5 ms per step: a replayed graph whose nodes span three captured streams costs "
                         "~30 us per node; captured on one stream it equals the eager single-stream step) -- off by default")
    ap.add_argument("--dp-buckets", type=int, default=None, choices=[1, 2, 4],
                    help="gradient exchange buckets per backward pass at N > 1 (DataParallel(buckets=...)): 2 = (head .. "
                         "layer3) | (layer2 .. stem) (1.3 %% machinery tax on one GPU against 3.2 %% for 4), 4 = one "
                         "per engine segment, 1 = one all-reduce at the end.  Default: MEASURED -- the ranks time "
                         "--dp-autotune-steps steps of 2 and of 4 buckets behind the warm-up, agree on the faster one "
                         "(max over ranks) and run the timed region with it (config.dp_autotune holds the table)")
    ap.add_argument("--dp-compress", default=None, choices=["none", "bf16"],
                    help="N > 1: gradient buckets travel as bfloat16 (DataParallel(grad_compress='bf16'): half the bytes on "
                         "xGMI, bf16 sums like torch DDP's compression hook).  Default: fp32 (exact) for --dtype f32 / f32x3; "
                         "for --dtype bf16 both forms are measured like the buckets and the faster one is taken")
    ap.add_argument("--dp-autotune-steps", type=int, default=5,
                    help="N > 1: timed steps per exchange candidate (after 1 untimed step); 0 = no measurement, the defaults "
                         "(2 buckets, fp32 on the wire)")
    ap.add_argument("--dist-timeout", type=float, default=float(os.environ.get("D3F_DIST_TIMEOUT", "600")),
                    help="N > 1: seconds a rank waits for the rendezvous / the first barrier / any collective before it "
                         "exits non-zero with its rank and the stage it was stuck in (never hangs the job)")
    ap.add_argument("--dp-selftest", action="store_true",
                    help="N=1: price the data-parallel machinery on ONE GPU -- the same step plain and with the 4 gradient "
                         "buckets all-reduced over a single-rank RCCL group (BucketAllReducer(force=True)); one JSON line")
    return ap.parse_args()


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def usable_cores():
    """host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(size, batch, steps):
    """the CPU oracle (pure torch.nn restatement, oracle/) timed on this box's host cores."""
    import oracle
    cores = usable_cores()
    torch.set_num_threads(cores)
    log(f"cpu baseline: {cores} threads (affinity {len(os.sched_getaffinity(0))})")
    torch.manual_seed(0)
    model = oracle.Unet("resnet34", None, 3, 3, None).train()
    crit = oracle.MseStructuralSimilarityLoss(-1.0, 1.0)
    opt = torch.optim.Adam(model.parameters(), lr=0.02)
    x = oracle.synthetic_face_crops(batch, size, seed=1234)
    g = torch.Generator().manual_seed(1)
    noise = torch.randn(x.shape, generator=g)
    r = torch.rand(batch, generator=g) * 0.5 + 0.05
    t0 = time.perf_counter()
    oracle.training_step(model, crit, opt, x, noise, r)  # warm-up
    log(f"cpu baseline: warm-up step {time.perf_counter() - t0:.1f}s")
    t0 = time.perf_counter()
    done = 0
    for _ in range(steps):
        oracle.training_step(model, crit, opt, x, noise, r)
        done += 1
        log(f"cpu baseline: step {done}/{steps}")
        if time.perf_counter() - t0 > 45.0:  # bounded sample
            break
    steps = done
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{steps} training steps (after 1 warm-up) of the oracle at bs={batch}, {size}x{size}, fp32, "
                      f"torch.set_num_threads({cores})"}


ALT_NOTES = {
    "f32x3": "fp32 storage/accumulation, conv products from 6 bf16 MFMAs over an exact 3-way operand split; "
             "opt-in (--dtype f32x3), same parity gates as the fp32-MFMA path; not the headline",
    "bf16": "bf16 activations / packed weights on the bf16 MFMA, fp32 accumulation, statistics and master weights "
            "(BASELINE.json config 2's dtype); opt-in (--dtype bf16), gated by tests/test_gpu_bf16.py; not the headline",
}


def wants_graph_step(args, dtype, world):
    return world == 1 and args.graph_step == "on"


def alt_dtype(args, dev, dtype):
    """Secondary figures of the default N=1 run, NOT the headline: the same training step with
    compute_dtype="f32x3" (fp32 tensors and fp32 accumulation, every conv product formed from six bf16 MFMAs over an
    exact 3-way split of both fp32 operands) or "bf16".  `value` of the JSON line is always the true fp32 MFMA."""
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    from denoising_diffusion_deep_fake_amd.train_denoiser.lit_module import LitModule
    torch.manual_seed(0)
    lit = LitModule(batch_size=args.batch, learning_rate=0.02, max_epochs=100, cosine_scheduler_max_epoch=100,
                    num_workers=0, encoder_name="resnet34", noise_exponential_sampling_lambda=5,
                    mean=[128, 128, 128], std=[128, 128, 128], synthetic=True, image_size=args.size,
                    augment=False, precision=dtype, graph_step=wants_graph_step(args, dtype, 1),
                    optimizer_overlap_tail=args.overlap_tail == "on").to(dev).train()
    (opt,), _ = lit.configure_optimizers()
    lit.attach_optimizers([opt])
    nb = 4
    data = [synthetic_face_crops(args.batch, args.size, seed=1234 + i, device=dev) for i in range(nb)]

    def step(i):
        if not lit.automatic_optimization:  # the whole step is one captured graph (manual optimisation)
            return lit.training_step({"image": data[i % nb], "index": None}, i)
        opt.zero_grad(set_to_none=True)
        loss = lit.training_step({"image": data[i % nb], "index": None}, i)
        loss.backward()
        opt.step()
        return loss
    for i in range(args.warmup):
        loss = step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": round(args.batch * args.steps / dt, 2), "unit": "images/sec",
            "ms_per_step": round(1e3 * dt / args.steps, 3), "final_loss": round(float(loss.item()), 5),
            "graph_step": not lit.automatic_optimization, "note": ALT_NOTES[dtype]}


def write_image_list(args):
    """`--fit-files` synthetic face crops as uint8 RGB files + images.txt (the reference's dataset format,
    d3f/dataset/image_dataset.py:19-31: paths relative to the list's own directory)"""
    from PIL import Image
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    ext = "jpg" if args.fit_source == "jpeg" else "png"
    root = os.path.join(args.fit_dir, f"{args.fit_source}_{args.size}_{args.fit_files}")
    lst = os.path.join(root, "images.txt")
    if os.path.exists(lst):
        return lst
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    names = []
    for i0 in range(0, args.fit_files, 64):
        x = synthetic_face_crops(min(64, args.fit_files - i0), args.size, seed=5000 + i0)
        u8 = ((x * 0.5 + 0.5) * 255.0).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).numpy()
        for j in range(u8.shape[0]):
            name = f"images/{i0 + j:06d}.{ext}"
            Image.fromarray(u8[j]).save(os.path.join(root, name), **({"quality": 92} if ext == "jpg" else {}))
            names.append(name)
    with open(lst + ".tmp", "w") as f:
        f.write("\n".join(names) + "\n")
    os.replace(lst + ".tmp", lst)
    return lst


def fit_workload_deepfake(args):
    """END TO END for the paired-domain trainer (BASELINE configs[3]): Trainer.fit over train_deep_fake's LitModule in
    denoise mode -- two on-disk image lists (or two synthetic sets) behind the CombinedLoader (max_size_cycle), the
    albumentations-style ShiftScaleRotate(p=0.7) on the GPU, two nets stepped one after the other -- next to the same
    combined step on resident batches.  Reference: d3f/train_deep_fake/lit_module.py:72-111 (loaders), :142-206 (steps)."""
    import tempfile
    from denoising_diffusion_deep_fake_amd.trainer import Callback, Trainer, optimizer_steps
    from denoising_diffusion_deep_fake_amd.train_deep_fake.lit_module import LitModule
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    total = args.warmup + args.steps
    hp = dict(mode="denoise", batch_size=args.batch, learning_rate=0.01, adam_b1=0.5, adam_b2=0.999, max_epochs=10 ** 6,
              cosine_scheduler_max_epoch=50, num_workers=args.fit_workers, encoder_name="resnet34",
              noise_exponential_sampling_lambda=3, mean_a=[0.5] * 3, std_a=[0.5] * 3, mean_b=[0.5] * 3, std_b=[0.5] * 3,
              image_size=args.size, precision=args.dtype, augment=args.fit_augment == "on",
              uint8_batches=args.fit_uint8 == "on", pin_memory=args.fit_pin == "on")
    if args.fit_source == "synthetic":
        hp.update(synthetic=True, synthetic_length=args.fit_files)
    else:
        lst = write_image_list(args)
        hp.update(synthetic=False, data_path_a=lst, data_path_b=lst)  # (two loaders over the same files: two shuffles)
    lit = LitModule(**hp)

    class Clock(Callback):
        def __init__(self):
            self.t0 = self.t1 = None
            self.n = 0

        def on_train_batch_end(self, trainer, module):
            self.n += 1
            if self.n == args.warmup:
                torch.cuda.synchronize()
                self.t0 = time.perf_counter()
            elif self.n == total:
                torch.cuda.synchronize()
                self.t1 = time.perf_counter()
    clock = Clock()
    with tempfile.TemporaryDirectory() as tmp:
        tr = Trainer(max_epochs=10 ** 6, max_steps=2 * total, callbacks=[clock], enable_checkpointing=False,
                     default_root_dir=tmp, log_every_n_steps=50)   # (global_step counts optimiser steps: 2 per batch)
        tr.fit(lit)
    fit_s = (clock.t1 - clock.t0) / args.steps
    # the same combined step on resident batches
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    opts = tr.optimizers
    opt_params = [[p for g in o.param_groups for p in g["params"]] for o in opts]
    batch = {k: {"image": synthetic_face_crops(args.batch, args.size, seed=7 + i, device=dev), "index": None}
             for i, k in enumerate("ab")}
    for i in range(5):
        optimizer_steps(lit, opts, opt_params, batch, i, True, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = min(args.steps, 50)
    for i in range(n):
        optimizer_steps(lit, opts, opt_params, batch, i, True, None)
    torch.cuda.synchronize()
    res_s = (time.perf_counter() - t0) / n
    out = {"workload": f"Trainer.fit end to end: d3f train_deep_fake (denoise mode, two nets), {args.size}x{args.size}, "
                       f"bs={args.batch} per domain, {args.fit_source} dataset x 2 behind the CombinedLoader, "
                       f"{args.fit_workers} DataLoader workers per loader, augment {args.fit_augment}, uint8 batches "
                       f"{args.fit_uint8}, pinned {args.fit_pin}",
           "dtype": args.dtype, "steps": args.steps, "warmup": args.warmup,
           "images_per_sec_fit": round(2 * args.batch / fit_s, 1), "ms_per_combined_batch_fit": round(1e3 * fit_s, 3),
           "images_per_sec_resident_same_process": round(2 * args.batch / res_s, 1),
           "ms_per_combined_batch_resident": round(1e3 * res_s, 3), "fit_over_resident": round(res_s / fit_s, 4),
           "host_cores": usable_cores()}
    print(json.dumps(out), flush=True)


def fit_workload(args):
    """END TO END: images/s through Trainer.fit -> DataLoader (spawned workers) -> _to_device -> (affine warp) -> noise
    blend -> U-Net step, next to the same step on resident batches (what `python bench.py` times) and to the loader and
    the host->device copy on their own.  Replaces the reference's fit loop around d3f/train_denoiser/lit_module.py:72-126
    (dataloader :72-90, training_step :107-126) and d3f/dataset/image_dataset.py:33-44."""
    import tempfile
    from denoising_diffusion_deep_fake_amd.trainer import Callback, Trainer, _to_device
    from denoising_diffusion_deep_fake_amd.train_denoiser.lit_module import LitModule
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    total = args.warmup + args.steps
    hp = dict(batch_size=args.batch, learning_rate=0.02, max_epochs=10 ** 6, cosine_scheduler_max_epoch=10 ** 6,
              num_workers=args.fit_workers, encoder_name="resnet34", noise_exponential_sampling_lambda=5,
              mean=[128, 128, 128], std=[128, 128, 128], image_size=args.size, augment=args.fit_augment == "on",
              precision=args.dtype, uint8_batches=args.fit_uint8 == "on", pin_memory=args.fit_pin == "on")
    if args.fit_source == "synthetic":
        hp.update(synthetic=True, synthetic_length=args.fit_files)
    else:
        t0 = time.perf_counter()
        hp.update(synthetic=False, input_image_list_path=write_image_list(args))
        log(f"image list ready in {time.perf_counter() - t0:.1f}s: {hp['input_image_list_path']}")
    lit = LitModule(**hp)

    # (1) the loader alone: batches/s the workers can deliver to the main process (no GPU work)
    loader = lit.train_dataloader()
    it = iter(loader)
    first = next(it)
    nb, t0 = 0, time.perf_counter()
    for batch in it:
        nb += 1
        if nb >= 40:
            break
    loader_s = (time.perf_counter() - t0) / max(nb, 1)
    img = first["image"]
    batch_bytes = img.numel() * img.element_size()
    # (2) host -> device copy of one batch as the trainer does it
    _to_device(first, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        _to_device(first, dev)
    torch.cuda.synchronize()
    h2d_s = (time.perf_counter() - t0) / 10
    del it, loader
    log(f"loader alone: {1e3 * loader_s:.2f} ms/batch ({args.batch / loader_s:.0f} images/s), H2D {1e3 * h2d_s:.2f} ms/batch")

    # (3) Trainer.fit
    class Clock(Callback):
        def __init__(self):
            self.t0 = self.t1 = None
            self.n = 0

        def on_train_batch_end(self, trainer, module):
            self.n += 1
            if self.n == args.warmup:
                torch.cuda.synchronize()
                self.t0 = time.perf_counter()
            elif self.n == total:
                torch.cuda.synchronize()
                self.t1 = time.perf_counter()
    clock = Clock()
    with tempfile.TemporaryDirectory() as tmp:
        tr = Trainer(max_epochs=10 ** 6, max_steps=total, callbacks=[clock], enable_checkpointing=False,
                     default_root_dir=tmp, log_every_n_steps=50)
        tr.fit(lit)
    fit_s = (clock.t1 - clock.t0) / args.steps
    lossv = float(lit._logged["loss"])

    # (4) the same step on resident batches in the same process (what the headline bench times)
    (opt,) = tr.optimizers
    data = [first["image"].to(dev) if first["image"].dtype == torch.float32 else None for _ in range(1)]
    if data[0] is None:
        data = [lit.normalise_on_device(first["image"].to(dev))]

    def step():
        opt.zero_grad(set_to_none=True)
        loss = lit.training_step({"image": data[0], "index": None}, 0)
        loss.backward()
        opt.step()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(min(args.steps, 50)):
        step()
    torch.cuda.synchronize()
    res_s = (time.perf_counter() - t0) / min(args.steps, 50)
    out = {"workload": f"Trainer.fit end to end: d3f train_denoiser, {args.size}x{args.size}, bs={args.batch}, "
                       f"{args.fit_source} dataset, {args.fit_workers} DataLoader workers, augment {args.fit_augment}, "
                       f"uint8 batches {args.fit_uint8}, pinned {args.fit_pin}",
           "dtype": args.dtype, "steps": args.steps, "warmup": args.warmup,
           "images_per_sec_fit": round(args.batch / fit_s, 1), "ms_per_step_fit": round(1e3 * fit_s, 3),
           "images_per_sec_resident_same_process": round(args.batch / res_s, 1), "ms_per_step_resident": round(1e3 * res_s, 3),
           "fit_over_resident": round(res_s / fit_s, 4),
           "loader_alone_images_per_sec": round(args.batch / loader_s, 1), "loader_alone_ms_per_batch": round(1e3 * loader_s, 3),
           "loader_share_of_fit_step": round(min(1.0, loader_s / fit_s), 3),
           "batch_dtype_from_loader": str(img.dtype).replace("torch.", ""), "batch_bytes": batch_bytes,
           "h2d_ms_per_batch": round(1e3 * h2d_s, 3), "h2d_GBps": round(batch_bytes / h2d_s / 1e9, 2),
           "host_cores": usable_cores(), "final_loss": round(lossv, 5)}
    print(json.dumps(out), flush=True)


def extra_workload(args):
    """secondary workloads of BASELINE.json (not the headline metric): one JSON line each."""
    if args.workload == "fit":
        return fit_workload_deepfake(args) if args.fit_module == "deepfake" else fit_workload(args)
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    world, rank, local = 1, 0, 0
    if args.workload == "deepfake" and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # data parallel (one process per GPU, launched as the headline is): each rank its own shard of both domains and its
        # own noise stream, both nets' gradients all-reduced bucket by bucket from inside the ONE backward pass of the pair
        from denoising_diffusion_deep_fake_amd.distributed import init_process_group
        world, rank, local = init_process_group(timeout_s=args.dist_timeout)
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        if os.environ.get("D3F_FORCE_DEVICE") is not None:  # test hook: several ranks on one GPU (gloo)
            local = int(os.environ["D3F_FORCE_DEVICE"])
    dev = torch.device("cuda", local)
    torch.cuda.set_device(local)
    torch.manual_seed(0)
    if args.workload == "predict":
        # single-frame inference: uint8 BGR frame -> uint8 BGR frame, B=1, at the
        # authors' 448x448; device-resident frames (the PCIe-inclusive figure is reported next to it)
        import numpy as np
        from denoising_diffusion_deep_fake_amd import Unet
        size = args.frame_size
        net = Unet("resnet34", None, 3, 3, None, compute_dtype=args.dtype).to(dev).eval()
        mean, std = [0.5] * 3, [0.5] * 3
        mt, st = torch.tensor(mean, device=dev), torch.tensor(std, device=dev)
        rng = np.random.default_rng(0)
        host = rng.integers(0, 256, size=(size, size, 3), dtype=np.uint8)
        fin = torch.from_numpy(host).to(dev)
        fout = torch.empty((1, size, size, 3), dtype=torch.uint8, device=dev)

        def unfused():
            t = fin.flip(-1).float().permute(2, 0, 1)
            t = ((t - mt.reshape(3, 1, 1) * 255) / (st.reshape(3, 1, 1) * 255)).unsqueeze(0).contiguous()
            with torch.no_grad():
                y = net(t)
            y = (y.squeeze(0) * (st.reshape(3, 1, 1) * 255) + mt.reshape(3, 1, 1) * 255).permute(1, 2, 0)
            return y.int().clamp(0, 255).to(torch.uint8).flip(-1)

        def timed(fn, n):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t0) / n

        def pcie():
            fin.copy_(torch.from_numpy(host))
            return net.predict_u8(fin, mean, std, graph=True, out=fout).cpu()
        n = max(args.steps, 50)
        res = {"workload": f"predict_fake single frame, B=1, {size}x{size}, uint8 BGR in/out", "dtype": args.dtype,
               "ms_per_frame_unfused_torch_pre_post": round(timed(unfused, n), 4),
               "ms_per_frame_fused_eager": round(timed(lambda: net.predict_u8(fin, mean, std, graph=False, out=fout), n), 4),
               "ms_per_frame_fused_hipgraph": round(timed(lambda: net.predict_u8(fin, mean, std, graph=True, out=fout), n), 4),
               "ms_per_frame_fused_hipgraph_pcie_inclusive": round(timed(pcie, n), 4)}
        res["frames_per_sec"] = round(1e3 / min(res["ms_per_frame_fused_eager"], res["ms_per_frame_fused_hipgraph"]), 1)
        best = min(res["ms_per_frame_fused_eager"], res["ms_per_frame_fused_hipgraph"])
        tf = net.conv_flops(1, size, size, dev)[0] / (best * 1e-3) / 1e12
        res["roofline"] = {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                           "frac": round(tf / PEAK_TFLOPS[args.dtype], 4), "traffic": None,
                           "basis": "one eval forward of B = 1 (launch-bound: ~50 dependent launches of a few us each), conv FLOPs "
                                    "counted as the direct convolution's"}
        print(json.dumps(res), flush=True)
        return
    if args.workload == "deepfake":
        from denoising_diffusion_deep_fake_amd.train_deep_fake.lit_module import LitModule
        bs = args.pair_batch
        swap = args.mode == "swap"
        # hyper-parameters of denoise_config.yml / swap_config.yml (lambda 3 / 8, ema_beta 0.9999, ema_update_every 1)
        lit = LitModule(mode=args.mode, batch_size=bs, learning_rate=0.01, adam_b1=0.5, adam_b2=0.999, max_epochs=50,
                        cosine_scheduler_max_epoch=50, num_workers=0, encoder_name="resnet34",
                        noise_exponential_sampling_lambda=8 if swap else 3, mean_a=[0.5] * 3, std_a=[0.5] * 3,
                        mean_b=[0.5] * 3, std_b=[0.5] * 3, synthetic=True, image_size=args.size, precision=args.dtype,
                        ema_beta=0.9999, ema_update_every=1, augment=False, pair_plan=args.pair_plan == "on",
                        pair_fused={"auto": None, "on": True, "off": False}[args.pair_fused]).to(dev).train()
        opts, _ = lit.configure_optimizers()
        if world > 1:
            import torch.distributed as dist
            from denoising_diffusion_deep_fake_amd.distributed import DataParallel
            for opt in opts:  # rank 0's parameters to everyone; each net its own reducer (trainer.Trainer.fit does the same)
                DataParallel(opt.module, opt, buckets=args.dp_buckets or 2,
                             grad_compress=None if args.dp_compress in (None, "none") else args.dp_compress)
            dist.barrier()
            torch.manual_seed(1000 + rank)
        if swap:
            # steady state of the 200-epoch phase: past ema_pytorch's update_after_step = 100 warm-up copies, every
            # update() is the lerp (one launch over the flat parameters + one over the BatchNorm statistics per net)
            for ema in (lit.ema_model_a, lit.ema_model_b):
                for _ in range(ema.update_after_step + 2):
                    ema.update()
        batch = {k: {"image": synthetic_face_crops(bs, args.size, seed=7 + i + 97 * rank, device=dev), "index": None}
                 for i, k in enumerate("ab")}
        from denoising_diffusion_deep_fake_amd.trainer import optimizer_steps
        opt_params = [[p for g in o.param_groups for p in g["params"]] for o in opts]
        # the trainer's own per-batch loop; D3F_CONCURRENT_NETS=1: the two nets' steps on two streams (opt-in: measured
        # slower, see LitModule.optimizer_streams)
        if os.environ.get("D3F_CONCURRENT_NETS"):
            lit.hparams["concurrent_optimizers"] = True
        streams = lit.optimizer_streams(dev)

        def step(i):
            return optimizer_steps(lit, opts, opt_params, batch, i, True, streams)
        images_per_step = 2 * bs
        fwd_fl, bwd_fl = lit.model_a.conv_flops(bs, args.size, args.size, dev)
        # two nets, one training step each per combined batch; swap mode adds the EMA teacher's forward (SURVEY.md 8d:
        # 62.36 GFLOP per image at 256x256)
        flops_per_step = 2 * (fwd_fl + bwd_fl + (fwd_fl if swap else 0))
        fused = bool(getattr(lit, "pair_fused_active", lambda: False)())
        name = (f"d3f train_deep_fake {args.mode}-mode step, two nets, bs={bs} per domain, the two optimizer steps "
                + ("as ONE set of launches (two-net plan)" if fused else
                   "overlapped on two streams" if streams else "one after the other")
                + ("; per optimizer: EMA lerp of the other net's teacher, teacher forward (no_grad, train-mode BatchNorm), "
                   "noise blend, student forward / loss / backward / Adam" if swap else ""))
    else:
        # BASELINE.json configs[4]: 50 sequential eval-mode forwards of a batch of 64 at 256x256, the output fed back
        # as the next input (clamped: the "re-noise" stand-in of SURVEY.md 8d), the denoise step replayed from a
        # hipGraph captured once (Unet.forward_graph); the eager loop is timed next to it and must agree bitwise
        from denoising_diffusion_deep_fake_amd import Unet
        net = Unet("resnet34", None, 3, 3, None, compute_dtype=args.dtype).to(dev).eval()
        x0 = synthetic_face_crops(64, args.size, seed=3, device=dev)
        xbuf, ybuf = torch.empty_like(x0), torch.empty_like(x0)

        @torch.no_grad()
        def eager(i):
            y = x0
            for _ in range(50):
                y = net(y).clamp_(-1.0, 1.0)  # keep the fed-back "image" in range (random-init weights)
            return y

        @torch.no_grad()
        def step(i):
            xbuf.copy_(x0)
            for _ in range(50):
                net.forward_graph(xbuf, out=ybuf)
                torch.clamp(ybuf, -1.0, 1.0, out=xbuf)
            return xbuf.mean()
        for i in range(max(args.warmup, 1)):
            ref = eager(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            ref = eager(i)
        torch.cuda.synchronize()
        eager_ms = 1e3 * (time.perf_counter() - t0) / args.steps
        step(0)
        bitwise = bool(torch.equal(ref, xbuf))
        images_per_step, name = 64, ("50 eval-mode forwards (BatchNorm folded) of a batch of 64, output fed back; "
                                     "denoise step replayed from a hipGraph")
        flops_per_step = 50 * net.conv_flops(64, args.size, args.size, dev)[0]
        extra = {"ms_per_step_eager": round(eager_ms, 3), "replay_equals_eager_bitwise": bitwise}
    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    for i in range(args.warmup):
        out = step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:  # the job's time is its slowest rank's; the replicas of BOTH nets must agree bit for bit afterwards
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        same = True
        for net in (lit.model_a, lit.model_b):
            cs = net.flat_params.view(torch.int32).to(torch.int64).sum().reshape(1)
            lo, hi = cs.clone(), cs.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            same = same and bool((lo == hi).item())
        if rank != 0:
            dist.destroy_process_group()
            return
    res = {"workload": name, "dtype": args.dtype, "image_size": args.size, "steps": args.steps,
           "ms_per_step": round(1e3 * dt / args.steps, 3),
           "images_per_sec": round(images_per_step * world * args.steps / dt, 2), "last": float(out.item())}
    if world > 1:
        res.update(n_gpus=world, scaling="weak", backend=dist.get_backend(), replicas_bit_identical=same,
                   dp_buckets=args.dp_buckets or 2)
    # whole-workload roofline: algorithmic conv FLOPs (2 x MACs of the ORIGINAL convolutions, SURVEY.md 8d -- Winograd /
    # folded / class-form kernels are credited the direct count) over the wall time of the timed steps, against the
    # dense MFMA peak of the arithmetic type; everything that is not a contraction (BatchNorm, loss, Adam) is inside `dt`
    tf = flops_per_step * args.steps / dt / 1e12   # per GPU (weak scaling: every rank steps its own batch)
    res["roofline"] = {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                       "frac": round(tf / PEAK_TFLOPS[args.dtype], 4), "traffic": None,
                       "basis": "whole timed region (all kernels), conv FLOPs counted as the direct convolution's",
                       "conv_gflop_per_step": round(flops_per_step / 1e9, 2)}
    if args.workload == "sample50":
        res.update(extra)
    if args.workload == "deepfake" and not swap and fused and args.dtype == "f32" and args.size == 256:
        # HBM bytes per launch of the contraction kernels of the FUSED step (every launch carries both networks), from the
        # builder's PMC passes over this command -- attached by source digest like the headline's figure
        traffic, tsrc = pmc_from_file("traffic.json", lambda d: d.get("hbm_bytes_per_launch"), args, tag="deepfake_")
        res["roofline"]["traffic"] = traffic
        if tsrc is not None:
            res["roofline"]["traffic_source"] = tsrc
    print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()
        if not same:
            raise SystemExit("the replicas' parameters differ after the run: the gradient exchange is broken")
    if args.workload == "sample50" and not extra["replay_equals_eager_bitwise"]:
        raise SystemExit("sample50: hipGraph replay differs from the eager loop")


def dp_selftest(args):
    """What the data-parallel path costs BEFORE any second GPU is involved (one rank, the real "nccl" == RCCL backend):
    the backward pass runs bucket by bucket (d3f_unet_backward_nojoin), each bucket's flat gradient slice goes to an
    asynchronous all_reduce ordered behind the engine's weight-gradient stream, Adam joins.  With one rank the sum is the
    identity, so (a) the gradients must be bit-identical to the plain pass and (b) any slowdown is pure machinery:
    per-bucket launches, RCCL's kernels taking wave slots, the joins.  A/B/A/B on one box, same process."""
    import socket
    import torch.distributed as dist
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    from denoising_diffusion_deep_fake_amd.distributed import BucketAllReducer
    from denoising_diffusion_deep_fake_amd.train_denoiser.lit_module import LitModule
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    torch.manual_seed(0)
    lit = LitModule(batch_size=args.batch, learning_rate=0.02, max_epochs=100, cosine_scheduler_max_epoch=100,
                    num_workers=0, encoder_name="resnet34", noise_exponential_sampling_lambda=5,
                    mean=[128, 128, 128], std=[128, 128, 128], synthetic=True, image_size=args.size,
                    augment=False, precision=args.dtype).to(dev).train()
    (opt,), _ = lit.configure_optimizers()
    red = BucketAllReducer(force=True)
    nb = 4
    data = [synthetic_face_crops(args.batch, args.size, seed=1234 + i, device=dev) for i in range(nb)]

    def mode(buckets):  # None: the plain pass; 4 / 2 / 1: that many exchange buckets over the single-rank RCCL group
        lit.model.set_grad_sync(red if buckets else None, buckets)
        opt.before_step = red.wait if buckets else None

    def step(i):
        opt.zero_grad(set_to_none=True)
        loss = lit.training_step({"image": data[i % nb], "index": None}, i)
        loss.backward()
        opt.step()
        return loss

    def grads_of(buckets):  # one backward on a fixed batch and fixed noise, no optimiser step
        mode(buckets)
        opt.zero_grad(set_to_none=True)
        torch.manual_seed(99)
        lit.training_step({"image": data[0], "index": None}, 0).backward()
        if buckets:
            red.wait()
        torch.cuda.synchronize()
        return lit.model.flat_grads.clone()

    variants = (None, 4, 2, ((0, 2), (2, 4)), 1)  # 2 = (head .. layer3) | (layer2 .. stem); explicit: the halves by segment count
    for i in range(args.warmup):
        mode(variants[i % len(variants)])
        step(i)
    g_plain = grads_of(None)
    identical = all(bool(torch.equal(g_plain, grads_of(b))) for b in variants[1:])
    rounds, times = 3, {b: [] for b in variants}
    for r in range(rounds):
        for b in variants:
            mode(b)
            step(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                loss = step(i)
            torch.cuda.synchronize()
            times[b].append(1e3 * (time.perf_counter() - t0) / args.steps)
            log(f"dp-selftest round {r} {str(b) + ' buckets + RCCL' if b else 'plain'}: {times[b][-1]:.3f} ms/step")
    # the exposed-exchange clock of `bench.py --gpus N` (HIP events around reducer.wait()) on the real RCCL backend: with one
    # rank nothing moves, so this is the floor of config.exposed_allreduce_ms -- what the two event records and the
    # stream-order waits themselves cost
    mode(2)
    red.timing = True
    for i in range(5):
        step(i)
    red.timing = False
    exposed, nwaits = red.exposed_ms()

    def label(b):
        return "plain" if b is None else f"buckets{b}" if isinstance(b, int) else "buckets2_by_segments"
    plain, buck = min(times[None]), min(times[4])
    segs = lit.model._rt["last_engine"].seg_ranges
    res = {"workload": f"dp-selftest: d3f train_denoiser step at N=1, plain vs 4 / 2 / 1 gradient buckets all-reduced over a "
                       f"single-rank RCCL group, {args.size}x{args.size}, bs={args.batch}, {args.dtype}",
           "backend": dist.get_backend(), "ranks_seen": dist.get_world_size(), "steps": args.steps, "rounds": rounds,
           "ms_per_step_plain": round(plain, 3), "ms_per_step_bucketed_rccl": round(buck, 3),
           "dp_tax": round(buck / plain - 1.0, 4),
           "dp_tax_by_buckets": {label(b): round(min(times[b]) / plain - 1.0, 4) for b in variants[1:]},
           "all_rounds_ms": {label(b): [round(t, 3) for t in times[b]] for b in variants},
           "bucket_mb": [round(4e-6 * (e - b), 1) for b, e in segs],
           "exposed_allreduce_ms_floor_single_rank": None if exposed is None else round(exposed, 4), "exposed_waits": nwaits,
           "gradients_bit_identical_to_plain": identical, "final_loss": round(float(loss.item()), 5),
           "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("D3F_")}}
    print(json.dumps(res), flush=True)
    dist.destroy_process_group()
    if not identical:
        raise SystemExit("dp-selftest: bucketed gradients differ from the plain pass")


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) with no launcher around it: start
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a CHILD process
    (never os.exec*), before this process has made any HIP call, and hand back its exit code; rank 0 of the child job
    prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    log("no launcher around --gpus %d: starting %s" % (args.gpus, " ".join(cmd[1:8])))
    return subprocess.run(cmd, env=env).returncode


def pmc_from_file(name, key, args, tag=None):
    """PMC counters cannot be read from inside this process.  Figures collected by the builder with rocprofv3 (separate
    --pmc passes, gfx950 corrections; profiles/tools/collect_*.sh) are attached WITH their provenance, and only for the
    configuration they were collected on -- they are not measurements of this run."""
    # the newest round's file first (profiles/rNN_<name>); bf16 / 128 / 448 have their own files
    if tag is None:
        tag = {("f32", 256, 16): "", ("bf16", 256, 16): "bf16_", ("f32", 128, 16): "128_", ("f32", 448, 14): "448_"}.get(
            (args.dtype, args.size, args.batch))
    if tag is None:
        return None, None
    path = None
    for rnd in ("r06", "r05", "r04"):
        cand = os.path.join(ROOT, "profiles", f"{rnd}_{tag}{name}")
        if os.path.exists(cand):
            path, name = cand, f"{rnd}_{tag}{name}"
            break
    if path is None:
        return None, None
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    src = {"file": "profiles/" + name, "collected_at_git_head": d.get("git_head"), "date": d.get("date"),
           "csrc_digest": d.get("csrc_digest"),
           "note": "builder's rocprofv3 --pmc run of the same command, NOT measured by this run"}
    from denoising_diffusion_deep_fake_amd import _lib
    if d.get("csrc_digest") != _lib.source_digest():
        # the kernels changed after the counters were collected: the figure would describe other code -> dropped
        src["stale"] = "csrc/ sources differ from the ones the counters were collected on: fields dropped"
        return None, src
    return key(d), src


def main():
    args = parse()
    if args.workload == "deepfake" and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))  # (the paired-domain step data parallel: BASELINE configs[3] "on 8 x MI355X")
    if args.workload != "denoiser":
        return extra_workload(args)
    if args.dp_selftest:
        if args.gpus != 1:
            raise SystemExit("--dp-selftest is the N=1 measurement")
        return dp_selftest(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))  # before anything in this process touches the GPU
    from denoising_diffusion_deep_fake_amd import _lib
    from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops
    from denoising_diffusion_deep_fake_amd.distributed import DataParallel, init_process_group
    from denoising_diffusion_deep_fake_amd.train_denoiser.lit_module import LitModule
    import ctypes as C
    import torch.distributed as dist

    # N > 1 must never hang the driver's scaling run: a watchdog thread ends THIS rank with a non-zero code, its rank and
    # the stage it is stuck in (os._exit -- no exec, no retry; torch.distributed.run then tears the other ranks down)
    # if the rendezvous + first barrier take longer than --dist-timeout; the process group carries the same timeout for
    # every later collective (RCCL's watchdog raises with the failing collective's name).
    stage = {"name": "rendezvous (init_process_group)"}
    watchdog = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import threading

        def _stuck():
            sys.stderr.write(f"[bench] rank {os.environ.get('RANK')} (local {os.environ.get('LOCAL_RANK')}): no progress "
                             f"for {args.dist_timeout:.0f} s in stage '{stage['name']}' (MASTER {os.environ.get('MASTER_ADDR')}:"
                             f"{os.environ.get('MASTER_PORT')}, backend {os.environ.get('D3F_DIST_BACKEND') or 'nccl=RCCL'}); "
                             f"giving up instead of hanging\n")
            sys.stderr.flush()
            os._exit(3)
        watchdog = threading.Timer(args.dist_timeout, _stuck)
        watchdog.daemon = True
        watchdog.start()
    try:
        world, rank, local = init_process_group(timeout_s=args.dist_timeout)
    except Exception as e:  # RCCL / gloo rendezvous errors: say which rank, then fail
        sys.stderr.write(f"[bench] rank {os.environ.get('RANK')}: init_process_group failed: {type(e).__name__}: {e}\n")
        raise
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("D3F_FORCE_DEVICE") is not None:  # test hook: several ranks on one GPU (gloo)
        local = int(os.environ["D3F_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(0)  # identical init on every rank (and broadcast from rank 0 anyway)
    lit = LitModule(batch_size=args.batch, learning_rate=0.02, max_epochs=100, cosine_scheduler_max_epoch=100,
                    num_workers=0, encoder_name="resnet34", noise_exponential_sampling_lambda=5,
                    mean=[128, 128, 128], std=[128, 128, 128], synthetic=True, image_size=args.size,
                    augment=False, precision=args.dtype,
                    graph_step=wants_graph_step(args, args.dtype, world),
                    optimizer_overlap_tail=args.overlap_tail == "on").to(dev).train()
    (opt,), _ = lit.configure_optimizers()
    lit.attach_optimizers([opt])
    stage["name"] = "parameter broadcast (DataParallel)"
    # exchange candidates (buckets, wire format): what the command line fixed stays fixed, the rest is measured
    cand_b = [args.dp_buckets] if args.dp_buckets is not None else [2, 4]
    cand_c = ([None if args.dp_compress == "none" else args.dp_compress] if args.dp_compress is not None
              else ([None, "bf16"] if args.dtype == "bf16" else [None]))
    candidates = [(b, c) for c in cand_c for b in cand_b]
    if args.dp_autotune_steps <= 0:
        candidates = candidates[:1]
    dp = DataParallel(lit.model, opt, buckets=candidates[0][0], grad_compress=candidates[0][1])
    if world > 1:
        stage["name"] = "first barrier"
        dist.barrier()
        torch.cuda.synchronize()
    if watchdog is not None:
        watchdog.cancel()
    torch.manual_seed(1000 + rank)  # distinct noise stream per rank
    nb = 4  # resident synthetic batches, distinct per rank
    data = [synthetic_face_crops(args.batch, args.size, seed=1234 + 97 * rank + i, device=dev) for i in range(nb)]

    def step(i):
        if not lit.automatic_optimization:  # the whole step is one captured graph (manual optimisation)
            return lit.training_step({"image": data[i % nb], "index": None}, i)
        opt.zero_grad(set_to_none=True)
        loss = lit.training_step({"image": data[i % nb], "index": None}, i)
        loss.backward()
        opt.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    L = _lib.lib()
    log("model built, starting warm-up")
    for i in range(args.warmup):
        loss = step(i)
        if i == 0:
            torch.cuda.synchronize()
            log("first step done")
    use_events = not args.no_kernel_events
    # N > 1: the exchange configuration is MEASURED on this job, not assumed (no N > 1 box existed while this was
    # written): every rank times every candidate over the same steps, one all_reduce(MAX) per candidate gives all ranks
    # the same table, the first minimum runs the timed region.  N = 1: nothing happens here.
    chosen, dp_table = candidates[0], None
    if world > 1 and len(candidates) > 1:
        from denoising_diffusion_deep_fake_amd.distributed import autotune_exchange
        tune_i = [0]

        def time_candidate(cand):
            dp.configure(buckets=cand[0], grad_compress=cand[1])
            step(args.warmup + tune_i[0])  # untimed: staging buffers, RCCL channels of this bucket size
            fence()
            t0 = time.perf_counter()
            for k in range(args.dp_autotune_steps):
                step(args.warmup + tune_i[0] + 1 + k)
            fence()
            tune_i[0] += 1 + args.dp_autotune_steps
            return (time.perf_counter() - t0) / args.dp_autotune_steps
        chosen, table = autotune_exchange(candidates, time_candidate)
        dp.configure(buckets=chosen[0], grad_compress=chosen[1])
        step(args.warmup)  # one settling step with the winner
        dp_table = [{"buckets": c[0], "grad_compress": c[1] or "none", "ms_per_step_max_over_ranks": round(1e3 * t, 3)}
                    for c, t in table]
        log(f"dp autotune: {dp_table} -> {chosen}")
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    log(f"timed region done: {dt:.3f}s for {args.steps} steps")
    # N > 1: what the exchange still costs the optimiser's stream -- the stall inside reducer.wait(), bracketed by HIP
    # events on that stream, in a pass of its own right after the timed region (an event pair taxes the stream)
    exposed_ms = None
    if world > 1:
        dp.reducer.timing = True
        for i in range(min(args.steps, 5)):
            loss = step(args.warmup + args.steps + i)
        dp.reducer.timing = False
        mine, nwaits = dp.reducer.exposed_ms()
        ex = torch.tensor([mine if mine is not None else -1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(ex, op=dist.ReduceOp.MAX)
        exposed_ms = round(float(ex.item()), 4) if nwaits else None
    # Kernel-level roofline figures: HIP events around every launch of a kernel class, in two passes of `diag` steps
    # RIGHT AFTER the timed region (same process, same resident batches, same clocks) -- not inside it: an event pair
    # costs ~8 us of stream time (its marker packets drain the queue), 97 pairs per step would tax `value` by ~2 %.
    # Pass 1: the roofline kernel conv_igemm_kernel (+ its LDS-patch forms), 47 forward + ~50 data-gradient launches
    # per step, on the caller's stream; pass 2: the weight-gradient kernels on the engine's side stream.
    ms, n, fl = (C.c_double * 3)(), (C.c_int64 * 3)(), (C.c_double * 3)()
    diag = min(args.steps, 5)
    sampled = diag
    if use_events:
        if lit._graph_step is not None:
            lit._graph_step.use_graph = False  # per-launch events need the launches themselves: same step, eagerly
        for classes in (3, 4):
            _lib.check(L.d3f_profile_classes(classes))
            _lib.check(L.d3f_profile_enable(diag * 128 + 64))
            for i in range(diag):
                loss = step(args.warmup + args.steps + i)
            torch.cuda.synchronize()
            ms2, n2, fl2 = (C.c_double * 3)(), (C.c_int64 * 3)(), (C.c_double * 3)()
            L.d3f_profile_collect(ms2, n2, fl2)
            for k in ((0, 1) if classes == 3 else (2,)):
                ms[k], n[k], fl[k] = ms2[k], n2[k], fl2[k]
        L.d3f_profile_classes(7)
        L.d3f_profile_enable(0)
    lossv = float(loss.item())
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    replicas_identical = None
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # every rank applied the same summed gradients to the same broadcast parameters: the replicas must agree BIT FOR
        # BIT after the run (an integer checksum of the parameter bits, min == max over ranks)
        cs = lit.model.flat_params.view(torch.int32).to(torch.int64).sum().reshape(1)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool((lo == hi).item())
    dt = float(tmax.item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    if not (lossv == lossv) or abs(lossv) > 1e6:
        raise SystemExit(f"non-finite loss {lossv}")

    images = args.batch * args.steps * world
    fwd_fl, bwd_fl = lit.model.conv_flops(args.batch, args.size, args.size, dev)
    step_flops = fwd_fl + bwd_fl
    peak = PEAK_TFLOPS[args.dtype]
    whole = step_flops / (dt / args.steps) / 1e12
    out = {
        "metric": "training images/sec (256x256 U-Net, bs=16/GPU)",
        "value": round(images / dt, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"d3f train_denoiser step: noise blend -> Unet(resnet34) fwd -> (MSE+1-SSIM)/2 -> bwd "
                               f"-> Adam, {args.size}x{args.size} synthetic face crops, bs={args.batch}/GPU, "
                               f"random-init weights, train-mode BatchNorm",
                   "image_size": args.size, "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                   "parallelism": f"dp{world}", "final_loss": round(lossv, 5),
                   # True: the optimiser step is replayed from one captured hipGraph (--graph-step; never at N > 1)
                   "graph_step": not lit.automatic_optimization,
                   # True: Adam over every gradient bucket but the last ran inside backward (FusedAdam(overlap_tail=True));
                   # False at N > 1 (the reducer keeps the whole update in step()) and under --overlap-tail off
                   # (counted, not inferred: steps whose early update really ran -- a reducer, a missing side stream or
                   # gradients outside the flat buffer switch it off silently)
                   "adam_overlap_tail": bool(opt.early_updates > 0),
                   "adam_overlap_tail_steps": int(opt.early_updates),
                   # N > 1: gradient exchange buckets per backward pass; parameter bits equal on every rank after the run
                   "dp_buckets": chosen[0] if world > 1 else None,
                   "dp_grad_compress": ((chosen[1] or "none") if world > 1 else None),
                   "replicas_bit_identical": replicas_identical,
                   # what the collective layer saw (None at N=1: no process group, no exchange step)
                   "backend": dist.get_backend() if world > 1 else None,
                   "ranks_seen": dist.get_world_size() if world > 1 else 1,
                   # every tuning / debugging knob of libd3f_hip.so that was set in the environment of this run
                   "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("D3F_")},
                   # digest of the sources baked into the loaded libd3f_hip.so (= the sources next to it unless D3F_LIB)
                   "csrc_digest": _lib.built_digest(),
                   "conv_gflop_per_image_step": round(step_flops / args.batch / 1e9, 3),
                   "whole_step_conv_tflops": round(whole, 2),
                   "whole_step_frac_of_peak": round(whole / peak, 4)},
    }
    if world > 1:
        # the table the ranks agreed on (None: the command line fixed the exchange, or --dp-autotune-steps 0) and the stall
        # of the optimiser's stream inside reducer.wait() per step, max over ranks (5 steps right after the timed region)
        out["config"]["dp_autotune"] = dp_table
        out["config"]["exposed_allreduce_ms"] = exposed_ms
    if use_events and n[0] > 0 and n[1] > 0:
        names = ["conv_igemm_kernel + conv_patch_kernel + conv_winograd_kernel (forward launches)",
                 "conv_igemm_kernel + conv_patch_kernel (data-gradient launches)",
                 "conv_wgrad_* (weight-gradient launches)"]
        nsteps = [sampled, sampled, diag]
        per = [{"kernel": names[k], "launches": int(n[k]), "avg_us": round(1e3 * ms[k] / max(n[k], 1), 2),
                "tflops": round(fl[k] / max(ms[k], 1e-9) / 1e9, 2),
                "ms_per_step": round(ms[k] / nsteps[k], 3)} for k in range(3)]
        # the data-gradient launches share the machine with the weight-gradient kernels of the second stream, so
        # their durations include that contention -- exactly what rocprofv3's kernel trace of this command reports
        per[1]["concurrent_with_weight_gradient_stream"] = True
        per[2]["concurrent_with_data_gradient_stream"] = True
        for q in per:
            q["timed_region"] = False  # measured in separate passes right after the timed steps (no event tax on `value`)
        # ROOFLINE FIGURE: every launch of the dominant kernel (forward AND data gradient) of the sampled steps
        t_all, f_all, n_all = ms[0] + ms[1], fl[0] + fl[1], n[0] + n[1]
        ach = f_all / t_all / 1e9
        traffic, tsrc = pmc_from_file("traffic.json", lambda d: d.get("hbm_bytes_per_launch"), args)
        busy, bsrc = pmc_from_file("mfma_util.json", lambda d: next(
            (k["mfma_pipe_busy"] for k in d.get("kernels", []) if k["kernel"] == "conv_igemm"), None), args)
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                           "frac": round(ach / peak, 4), "traffic": traffic,
                           # forward launches alone (uncontended): comparable with round 1's forward-only figure
                           "frac_forward": round(fl[0] / max(ms[0], 1e-9) / 1e9 / peak, 4),
                           "kernel": f"conv_igemm_kernel (+ conv_patch_kernel, its LDS-patch form for the 16-channel "
                                     f"full-resolution layers, + conv_winograd_kernel, the F(2x2,3x3) form of the wide "
                                     f"stride-1 3x3 forward layers; FLOPs counted as the direct convolution's), "
                                     f"ALL launches ({int(n[0]) // sampled} forward + "
                                     f"{int(n[1]) // sampled} data-gradient per step)",
                           "launches": int(n_all),
                           "sampled_steps": f"{sampled} steps right after the {args.steps} timed steps (HIP events on the "
                                            f"launch stream; the timed region itself carries no events)",
                           "avg_launch_us": round(1e3 * t_all / n_all, 2),
                           "flop_per_launch": round(f_all / n_all, 1),
                           "ms_per_step": round(t_all / sampled, 3),
                           "per_kernel": per}
        if tsrc is not None:
            out["roofline"]["traffic_source"] = tsrc
        if bsrc is not None:
            out["roofline"]["pmc_from_file"] = dict(bsrc, mfma_pipe_busy=busy)
    else:
        out["roofline"] = {"bound": "mfma", "achieved": None, "peak": peak, "unit": "TFLOP/s", "frac": None,
                           "traffic": None}
    if world == 1 and args.dtype == "f32" and not args.no_alt:
        del lit, opt, data, step  # free the fp32 run's 2 GB workspace before building the next model
        for alt in ("f32x3", "bf16"):
            torch.cuda.empty_cache()
            out["alt_" + alt] = alt_dtype(args, dev, alt)
            log(f"alt {alt}: {out['alt_' + alt]['value']} images/s")
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.size, args.batch, args.cpu_steps)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if replicas_identical is False:
        raise SystemExit("the replicas' parameters differ after the run: the gradient exchange is broken")


if __name__ == "__main__":
    main()
