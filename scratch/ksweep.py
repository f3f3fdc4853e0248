import sys, os; sys.path.insert(0, '/root/repo')
import torch
from denoising_diffusion_deep_fake_amd import ops
def t(B,H,W,C0,Co,splitk=True):
    d = ops.make_desc(B,H,W,C0,0,Co,3,1,1,False)
    s0 = torch.randn(B,H,W,C0, device="cuda"); w = torch.randn(Co, C0, 3, 3, device="cuda")*0.05
    wf, wd = ops.pack_weights(d, w)
    for _ in range(3): ops.conv_forward(d, s0, None, wf, splitk=splitk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.conv_forward(d, s0, None, wf, splitk=splitk)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/20*1e3
for (B,H,W,Co) in [(16,32,32,128),(16,64,64,64)]:
    for C0 in (32, 64, 128, 256, 512):
        us = t(B,H,W,C0,Co,splitk=False)
        nk = 9*C0//32
        fl = 2.0*B*H*W*Co*9*C0
        print(f"M={B*H*W} N={Co} Cin={C0:4d} nk={nk:4d}: {us:7.1f} us  {fl/us/1e6:6.1f} TF   us/ktile {us/nk:.3f}")
