#!/usr/bin/env python3
"""Where does the host spend a swap-mode combined batch (train_deep_fake, bs 8 x 2, 256x256)?  cProfile over the trainer's
own per-batch loop + wall time per batch at several step counts + the GPU-side busy time from HIP events.
   python3 profiles/tools/swap_hostprobe.py [denoise|swap]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops  # noqa: E402
from denoising_diffusion_deep_fake_amd.train_deep_fake.lit_module import LitModule  # noqa: E402
from denoising_diffusion_deep_fake_amd.trainer import optimizer_steps  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "swap"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
bs = 8
lit = LitModule(mode=mode, batch_size=bs, learning_rate=0.01, adam_b1=0.5, adam_b2=0.999, max_epochs=50,
                cosine_scheduler_max_epoch=50, num_workers=0, encoder_name="resnet34",
                noise_exponential_sampling_lambda=8, mean_a=[0.5] * 3, std_a=[0.5] * 3, mean_b=[0.5] * 3, std_b=[0.5] * 3,
                synthetic=True, image_size=256, ema_beta=0.9999, ema_update_every=1, augment=False).to(dev).train()
opts, _ = lit.configure_optimizers()
if mode == "swap":
    for ema in (lit.ema_model_a, lit.ema_model_b):
        for _ in range(ema.update_after_step + 2):
            ema.update()
batch = {k: {"image": synthetic_face_crops(bs, 256, seed=7 + i, device=dev), "index": None} for i, k in enumerate("ab")}
opt_params = [[p for g in o.param_groups for p in g["params"]] for o in opts]


def step(i):
    return optimizer_steps(lit, opts, opt_params, batch, i, True, None)


for i in range(10):
    step(i)
torch.cuda.synchronize()
for n in (3, 10, 30, 30):
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n:3d} batches: {1e3 * dt / n:.3f} ms per combined batch (host loop alone {1e3 * t_host / n:.3f} ms)", flush=True)
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("cumulative").print_stats(45)
