"""Forward and data-gradient conv timing (uncontended) over the distinct conv shapes of Unet(resnet34) at
B=16, 256x256, for the fp32-MFMA and the x3 (bf16 3-way split) contraction modes."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from denoising_diffusion_deep_fake_amd import ops

B = int(os.environ.get("B", 16))
S = int(os.environ.get("S", 256))
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
}


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {}
print(f"{'layer':24s}      fwd f32   fwd x3   dgrad f32  dgrad x3   (us)")
for name, (H, W, C0, C1, Co, k, st, pd, up, cnt, creal) in shapes.items():
    d = ops.make_desc(B, H, W, C0, C1, Co, k, st, pd, up, cin_real=creal)
    h0, w0 = (H // 2, W // 2) if up else (H, W)
    s0 = torch.randn(B, h0, w0, C0, device="cuda")
    s1 = torch.randn(B, H, W, C1, device="cuda") if C1 else None
    ho, wo = ops.out_hw(d)
    dy = torch.randn(B, ho, wo, Co, device="cuda")
    w = torch.randn(Co, creal or (C0 + C1), k, k, device="cuda") * 0.05
    row = []
    for dt in (ops.F32, ops.F32X3):
        wf, wd = ops.pack_weights(d, w, dt)
        row.append(timeit(lambda: ops.conv_forward(d, s0, s1, wf, dt, splitk=True)))
    for dt in (ops.F32, ops.F32X3):
        wf, wd = ops.pack_weights(d, w, dt)
        if name.startswith("stem"):
            row.append(0.0)
            continue
        dx0 = torch.empty(B, h0 * (2 if up else 1) // (2 if up else 1), w0, C0, device="cuda") if not up else None
        row.append(timeit(lambda: ops.conv_backward_data(d, dy, wd, dt, splitk=True)))
    for i, v in enumerate(row):
        tot[i] = tot.get(i, 0.0) + v * cnt
    print(f"{name:24s} x{cnt:2d} {row[0]:8.1f} {row[1]:8.1f}   {row[2]:8.1f} {row[3]:8.1f}")
print(f"{'total per step':24s}     {tot[0]:8.1f} {tot[1]:8.1f}   {tot[2]:8.1f} {tot[3]:8.1f}")
