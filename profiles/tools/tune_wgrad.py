"""Per-layer weight-gradient timing (uncontended) over the distinct conv shapes of Unet(resnet34) at
B=16, 256x256: python profiles/tools/tune_wgrad.py  -> one line per shape with us and TFLOP/s."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from denoising_diffusion_deep_fake_amd import ops

DT = {"f32": ops.F32, "f32x3": ops.F32X3}[os.environ.get("DT", "f32")]
B = int(os.environ.get("B", 16))
S = int(os.environ.get("S", 256))
# name: (H, W (input), C0, C1, Cout, k, stride, pad, upsample0, count in the network)
shapes = {
    "stem 7x7 s2 3->64":      (S, S, 4, 0, 64, 7, 2, 3, False, 1, 3),
    "l1 64->64 @/4":          (S // 4, S // 4, 64, 0, 64, 3, 1, 1, False, 6, None),
    "l2.0 64->128 s2":        (S // 4, S // 4, 64, 0, 128, 3, 2, 1, False, 1, None),
    "l2 ds 1x1 s2":           (S // 4, S // 4, 64, 0, 128, 1, 2, 0, False, 1, None),
    "l2 128->128 @/8":        (S // 8, S // 8, 128, 0, 128, 3, 1, 1, False, 7, None),
    "l3.0 128->256 s2":       (S // 8, S // 8, 128, 0, 256, 3, 2, 1, False, 1, None),
    "l3 256->256 @/16":       (S // 16, S // 16, 256, 0, 256, 3, 1, 1, False, 11, None),
    "l4.0 256->512 s2":       (S // 16, S // 16, 256, 0, 512, 3, 2, 1, False, 1, None),
    "l4 512->512 @/32":       (S // 32, S // 32, 512, 0, 512, 3, 1, 1, False, 5, None),
    "d0.1 768->256 @/16":     (S // 16, S // 16, 512, 256, 256, 3, 1, 1, True, 1, None),
    "d0.2 256->256 @/16":     (S // 16, S // 16, 256, 0, 256, 3, 1, 1, False, 1, None),
    "d1.1 384->128 @/8":      (S // 8, S // 8, 256, 128, 128, 3, 1, 1, True, 1, None),
    "d1.2 128->128 @/8":      (S // 8, S // 8, 128, 0, 128, 3, 1, 1, False, 1, None),
    "d2.1 192->64 @/4":       (S // 4, S // 4, 128, 64, 64, 3, 1, 1, True, 1, None),
    "d2.2 64->64 @/4":        (S // 4, S // 4, 64, 0, 64, 3, 1, 1, False, 1, None),
    "d3.1 128->32 @/2":       (S // 2, S // 2, 64, 64, 32, 3, 1, 1, True, 1, None),
    "d3.2 32->32 @/2":        (S // 2, S // 2, 32, 0, 32, 3, 1, 1, False, 1, None),
    "d4.1 32->16 @/1":        (S, S, 32, 0, 16, 3, 1, 1, True, 1, None),
    "d4.2 16->16 @/1":        (S, S, 16, 0, 16, 3, 1, 1, False, 1, None),
    "head 16->3 @/1":         (S, S, 16, 0, 3, 3, 1, 1, False, 1, None),
}
tot = 0.0
totfl = 0.0
for name, (H, W, C0, C1, Co, k, st, pd, up, cnt, creal) in shapes.items():
    d = ops.make_desc(B, H, W, C0, C1, Co, k, st, pd, up, cin_real=creal)
    h0, w0 = (H // 2, W // 2) if up else (H, W)
    s0 = torch.randn(B, h0, w0, C0, device="cuda")
    s1 = torch.randn(B, H, W, C1, device="cuda") if C1 else None
    ho, wo = ops.out_hw(d)
    cop = (Co + 3) // 4 * 4
    dy = torch.randn(B, ho, wo, cop, device="cuda")
    try:
        for _ in range(3):
            ops.conv_backward_weight(d, dy, s0, s1, DT)
    except Exception as e:
        print(f"{name:24s} ERR {str(e)[:80]}")
        continue
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv_backward_weight(d, dy, s0, s1, DT)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    fl = 2.0 * B * ho * wo * Co * k * k * (creal or (C0 + C1))
    tot += us * cnt
    totfl += fl * cnt
    print(f"{name:24s} x{cnt:2d} {us:8.1f} us  {fl / us / 1e6:6.1f} TF   step share {us * cnt:8.1f} us")
print(f"total {tot:.1f} us/step  {totfl / tot / 1e6:.1f} TF")
