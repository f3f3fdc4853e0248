import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from denoising_diffusion_deep_fake_amd import ops
shapes = {
  "L1  64->64  @64":  (16, 64, 64, 64, 0, 64, 3, 1, 1, False),
  "L2 128->128 @32":  (16, 32, 32, 128, 0, 128, 3, 1, 1, False),
  "L3 256->256 @16":  (16, 16, 16, 256, 0, 256, 3, 1, 1, False),
  "L4 512->512 @8":   (16, 8, 8, 512, 0, 512, 3, 1, 1, False),
  "D0 768->256 @16":  (16, 16, 16, 512, 256, 256, 3, 1, 1, True),
  "D1 384->128 @32":  (16, 32, 32, 256, 128, 128, 3, 1, 1, True),
  "D2 192->64  @64":  (16, 64, 64, 128, 64, 64, 3, 1, 1, True),
}
tag = os.environ.get("D3F_FORCE_TILE", "default")
DT = {"f32": ops.F32, "f32x3": ops.F32X3}[os.environ.get("DT", "f32")]
tag += "/" + os.environ.get("DT", "f32")
for name, (B,H,W,C0,C1,Co,k,s,pd,up) in shapes.items():
    d = ops.make_desc(B,H,W,C0,C1,Co,k,s,pd,up)
    h0,w0 = (H//2,W//2) if up else (H,W)
    s0 = torch.randn(B,h0,w0,C0, device="cuda"); s1 = torch.randn(B,H,W,C1, device="cuda") if C1 else None
    w = torch.randn(Co, C0+C1, k, k, device="cuda")*0.05
    wf, wd = ops.pack_weights(d, w, DT)
    try:
        for _ in range(3): ops.conv_forward(d, s0, s1, wf, DT, splitk=True)
    except Exception as e:
        print(f"{tag:12s} {name}: ERR {str(e)[:60]}"); continue
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.conv_forward(d, s0, s1, wf, DT, splitk=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)/20*1e3
    fl = 2.0*B*H*W*Co*k*k*(C0+C1)
    print(f"{tag:12s} {name}: {us:7.1f} us  {fl/us/1e6:6.1f} TF")
