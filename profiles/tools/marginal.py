"""Marginal cost per k-tile and fixed cost per launch of the conv kernel (3x3 vs 5x5 filters at the same M x N), f32 vs f32x3."""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from denoising_diffusion_deep_fake_amd import ops
def t(dt, B,H,W,C,Co,k):
    d = ops.make_desc(B,H,W,C,0,Co,k,1,k//2,False)
    s0 = torch.randn(B,H,W,C, device="cuda"); w = torch.randn(Co,C,k,k, device="cuda")*0.05
    wf, wd = ops.pack_weights(d, w, dt)
    for _ in range(3): ops.conv_forward(d, s0, None, wf, dt, splitk=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.conv_forward(d, s0, None, wf, dt, splitk=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/20*1e3
for name,(B,H,W,C,Co) in {"64->64 @64 (128x64 tile)":(16,64,64,64,64), "128->128 @32 (64x64)":(16,32,32,128,128), "256->256 @16":(16,16,16,256,256)}.items():
    for nm, dt in (("f32", ops.F32), ("f32x3", ops.F32X3)):
        t1, t3, t5 = t(dt,B,H,W,C,Co,1), t(dt,B,H,W,C,Co,3), t(dt,B,H,W,C,Co,5)
        kt1, kt3, kt5 = C//32, 9*C//32, 25*C//32
        m = (t5-t3)/(kt5-kt3)
        print(f"{name:28s} {nm:6s} 1x1 {t1:6.1f} 3x3 {t3:6.1f} 5x5 {t5:6.1f} us  marginal {m*1e3:6.1f} ns/k-tile  fixed ~{t3-kt3*m:5.1f} us  loop-only rate {2.0*B*H*W*Co*32/m/1e6:6.1f} TF")
