"""Where does the host time of one training step go?  (at 128x128 the host enqueue loop, 3.53 ms, is as
long as the GPU-paced step.)  Times, with the GPU idle at the start of every phase (so nothing blocks on a full queue):
the C calls alone (d3f_unet_forward / d3f_unet_backward through Unet._run_forward / _run_backward), the Python around
them (autograd Function, loss, noise draws), and the optimiser.
    python3 profiles/tools/hostprobe2.py [dtype=f32] [size=128] [batch=16]"""
import os
import sys
import time

import torch

DTYPE = sys.argv[1] if len(sys.argv) > 1 else "f32"
SIZE = int(sys.argv[2]) if len(sys.argv) > 2 else 128
BATCH = int(sys.argv[3]) if len(sys.argv) > 3 else 16
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from denoising_diffusion_deep_fake_amd import ops  # noqa: E402
from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops  # noqa: E402
from denoising_diffusion_deep_fake_amd.train_denoiser.lit_module import LitModule  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
lit = LitModule(batch_size=BATCH, learning_rate=0.02, max_epochs=100, cosine_scheduler_max_epoch=100, num_workers=0,
                encoder_name="resnet34", noise_exponential_sampling_lambda=5, mean=[128] * 3, std=[128] * 3,
                synthetic=True, image_size=SIZE, augment=False, precision=DTYPE).to(dev).train()
(opt,), _ = lit.configure_optimizers()
x = synthetic_face_crops(BATCH, SIZE, seed=1, device=dev)
net = lit.model


def full_step(i):
    opt.zero_grad(set_to_none=True)
    loss = lit.training_step({"image": x, "index": None}, i)
    loss.backward()
    opt.step()


for i in range(5):
    full_step(i)
torch.cuda.synchronize()
N = 30
T = {}


def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    return r


for i in range(N):
    opt.zero_grad(set_to_none=True)
    noisy = timed("py: noise draws + blend", lambda: lit.blend_random_amount_of_noise_with_each_sample(x))
    eng = net._engine(BATCH, SIZE, SIZE, dev)
    timed("py: weight-version check + pack call", lambda: net._pack_if_needed(eng))
    pred = timed("C: d3f_unet_forward (one call)", lambda: net._run_forward(eng, noisy, True))
    lossv, gout = timed("py+C: loss op", lambda: ops.mse_ssim_loss(pred, x))
    timed("C: d3f_unet_backward (one call) + .grad views", lambda: net._run_backward(eng, gout))
    timed("py+C: optimizer step", lambda: opt.step())
    timed("py: whole training_step + backward + step, for comparison", lambda: full_step(i))
print(f"[{DTYPE} {SIZE}x{SIZE} bs{BATCH}] host ms per step, GPU idle at the start of each phase:")
for k, v in T.items():
    print(f"  {k:62s} {1e3 * v / N:7.3f}")
