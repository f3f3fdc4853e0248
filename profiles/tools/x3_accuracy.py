"""Forward conv rel-L2 error against a float64 conv, fp32 MFMA vs f32x3, with and without split-K, over 8 layer shapes."""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from denoising_diffusion_deep_fake_amd import ops
torch.manual_seed(0)
def run(name, B,H,W,C0,C1,Co,k,s,pd,up):
    d = ops.make_desc(B,H,W,C0,C1,Co,k,s,pd,up)
    h0,w0 = (H//2,W//2) if up else (H,W)
    s0 = torch.randn(B,h0,w0,C0, device="cuda"); s1 = torch.randn(B,H,W,C1, device="cuda") if C1 else None
    w = torch.randn(Co, C0+C1, k, k, device="cuda")*0.05
    x0 = s0.permute(0,3,1,2).double()
    if up: x0 = torch.nn.functional.interpolate(x0, scale_factor=2, mode="nearest")
    x = torch.cat([x0, s1.permute(0,3,1,2).double()], 1) if C1 else x0
    ref = torch.nn.functional.conv2d(x, w.double(), stride=s, padding=pd).permute(0,2,3,1)
    out = {}
    for nm, dt in (("f32", ops.F32), ("f32x3", ops.F32X3)):
        wf, wd = ops.pack_weights(d, w, dt)
        for sk in (False, True):
            y, st, tiles = ops.conv_forward(d, s0, s1, wf, dt, splitk=sk)
            e = ((y.double()-ref).norm()/ref.norm()).item()
            out[(nm, sk)] = e
    print(name, {f"{k[0]}{'/sk' if k[1] else ''}": f"{v:.2e}" for k,v in out.items()})
run("64->64 @32", 4,32,32,64,0,64,3,1,1,False)
run("128->128 @16 s2", 4,32,32,128,0,128,3,2,1,False)
run("up+cat 256+128->128", 2,32,32,256,128,128,3,1,1,True)
run("512->512 @8", 4,8,8,512,0,512,3,1,1,False)
run("32->32 @64", 2,64,64,32,0,32,3,1,1,False)
run("16->16 @64", 2,64,64,16,0,16,3,1,1,False)
run("stem 4->64 7x7 s2", 2,64,64,4,0,64,7,2,3,False)
run("1x1 64->128 s2", 2,32,32,64,0,128,1,2,0,False)
