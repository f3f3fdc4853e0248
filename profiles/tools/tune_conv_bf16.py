"""Stand-alone timing of the wide 3x3 stride-1 layer shapes of Unet(resnet34) at B=16, 256x256 in bf16 storage: forward and
data gradient, 20 launches back to back (uncontended):  python profiles/tools/tune_conv_bf16.py
D3F_NO_PRES_CONV=1 selects the implicit GEMM (conv_igemm_kernel) instead of the patch-resident kernel (conv_pres.hip)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from denoising_diffusion_deep_fake_amd import ops

B = int(os.environ.get("B", 16))
shapes = {
    "L1  64->64  @64": (B, 64, 64, 64, 0, 64, 3, 1, 1, False),
    "L2 128->128 @32": (B, 32, 32, 128, 0, 128, 3, 1, 1, False),
    "L3 256->256 @16": (B, 16, 16, 256, 0, 256, 3, 1, 1, False),
    "L4 512->512 @8": (B, 8, 8, 512, 0, 512, 3, 1, 1, False),
}
tag = "igemm" if os.environ.get("D3F_NO_PRES_CONV") else "pres"
DT = ops.BF16
for name, (b, H, W, C0, C1, Co, k, s, pd, up) in shapes.items():
    d = ops.make_desc(b, H, W, C0, C1, Co, k, s, pd, up)
    s0 = torch.randn(b, H, W, C0, device="cuda").bfloat16()
    dy = torch.randn(b, H, W, Co, device="cuda").bfloat16()
    w = torch.randn(Co, C0, k, k, device="cuda") * 0.05
    wf, wd = ops.pack_weights(d, w, DT)
    fl = 2.0 * b * H * W * Co * k * k * C0
    for what, fn in (("fwd  ", lambda: ops.conv_forward(d, s0, None, wf, DT, splitk=True)),
                     ("dgrad", lambda: ops.conv_backward_data(d, dy, wd, DT, splitk=True))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{tag:6s} {name} {what}: {us:7.1f} us  {fl / us / 1e6:7.1f} TF")
