import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from denoising_diffusion_deep_fake_amd import ops
B = int(os.environ.get("PB", "16"))
shapes = {
  "L2 128->128 @32":  (B, 32, 32, 128, 0, 128, 3, 1, 1, False),
  "L3 256->256 @16":  (B, 16, 16, 256, 0, 256, 3, 1, 1, False),
  "L4 512->512 @8":   (B, 8, 8, 512, 0, 512, 3, 1, 1, False),
  "D0 768->256 @16":  (B, 16, 16, 512, 256, 256, 3, 1, 1, True),
}
tag = os.environ.get("D3F_FORCE_TILE", "default") + f" B={B}"
for name, (B_,H,W,C0,C1,Co,k,s,pd,up) in shapes.items():
    d = ops.make_desc(B_,H,W,C0,C1,Co,k,s,pd,up)
    h0,w0 = (H//2,W//2) if up else (H,W)
    s0 = torch.randn(B_,h0,w0,C0, device="cuda"); s1 = torch.randn(B_,H,W,C1, device="cuda") if C1 else None
    w = torch.randn(Co, C0+C1, k, k, device="cuda")*0.05
    wf, wd = ops.pack_weights(d, w, ops.F32)
    dy = torch.randn(B_, H, W, Co, device="cuda")
    for what in ("fwd", "dgrad"):
        fn = (lambda: ops.conv_forward(d, s0, s1, wf, ops.F32, splitk=True)) if what == "fwd" else (lambda: ops.conv_backward_data(d, dy, wd, ops.F32, splitk=True))
        try:
            for _ in range(3): fn()
        except Exception as e:
            print(f"{tag:16s} {name} {what}: ERR {str(e)[:60]}"); continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1)/30*1e3
        fl = 2.0*B_*H*W*Co*k*k*(C0+C1)
        print(f"{tag:16s} {name} {what}: {us:7.1f} us  {fl/us/1e6:6.1f} TF  {fl/us/1e6/157.3:.2f}")
