"""One conv shape launched N times (for PMC passes): DT=f32|f32x3, K=3|5 (filter), SHAPE=l1|l2|l3."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from denoising_diffusion_deep_fake_amd import ops
DT = {"f32": ops.F32, "f32x3": ops.F32X3}[os.environ.get("DT", "f32")]
k = int(os.environ.get("K", 5))
B, H, W, C, Co = {"l1": (16, 64, 64, 64, 64), "l2": (16, 32, 32, 128, 128), "l3": (16, 16, 16, 256, 256)}[os.environ.get("SHAPE", "l1")]
d = ops.make_desc(B, H, W, C, 0, Co, k, 1, k // 2, False)
s0 = torch.randn(B, H, W, C, device="cuda")
w = torch.randn(Co, C, k, k, device="cuda") * 0.05
wf, wd = ops.pack_weights(d, w, DT)
for _ in range(int(os.environ.get("N", 10))):
    ops.conv_forward(d, s0, None, wf, DT, splitk=True)
torch.cuda.synchronize()
