import sys, os; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, torch.nn.functional as F
from denoising_diffusion_deep_fake_amd import ops
from util import rel_l2, to_nchw, to_nhwc
BF=ops.BF16
def case(B,H,W,C0,C1,Co,k,s,pd,up, check=True):
    g=torch.Generator().manual_seed(0)
    h0,w0=(H//2,W//2) if up else (H,W)
    x0=torch.randn(B,C0,h0,w0,generator=g).bfloat16().float()
    x1=torch.randn(B,C1,H,W,generator=g).bfloat16().float() if C1 else None
    w=(torch.randn(Co,C0+C1,k,k,generator=g)/((C0+C1)*k*k)**0.5).bfloat16().float()
    d=ops.make_desc(B,H,W,C0,C1,Co,k,s,pd,up)
    s0=to_nhwc(x0).bfloat16().cuda(); s1=to_nhwc(x1).bfloat16().cuda() if C1 else None
    wf,wd=ops.pack_weights(d,w.cuda(),dtype=BF)
    y,stats,tiles=ops.conv_forward(d,s0,s1,wf,dtype=BF,splitk=True)
    torch.cuda.synchronize()
    if check:
        xin=F.interpolate(x0,scale_factor=2,mode="nearest") if up else x0
        if C1: xin=torch.cat([xin,x1],1)
        yr=F.conv2d(xin,w,None,s,pd)
        e=rel_l2(to_nchw(y.float().cpu()),yr)
        dy=torch.randn(yr.shape,generator=g).bfloat16().float()
        xr=xin.clone().requires_grad_(True); F.conv2d(xr,w,None,s,pd).backward(dy)
        dx0,dx1=ops.conv_backward_data(d,to_nhwc(dy).bfloat16().cuda(),wd,dtype=BF,splitk=True)
        e2=rel_l2(to_nchw(dx0.float().cpu()),xr.grad[:,:C0])
        print("  parity fwd %.2e dgrad %.2e"%(e,e2))
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.conv_forward(d,s0,s1,wf,dtype=BF,splitk=True)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/20*1e3; fl=2.0*B*H*W*Co*k*k*(C0+C1)
    print(f"bf16 {B}x{H}x{W} {C0}+{C1}->{Co} up={up}: {us:7.1f} us {fl/us/1e6:7.1f} TF")
case(2,16,16,64,0,64,3,1,1,False)
case(2,16,16,64,64,32,3,1,1,True)
case(2,16,16,64,0,128,3,2,1,False)
case(2,16,16,16,0,16,3,1,1,False)
for sh in [(16,64,64,64,0,64,3,1,1,False),(16,32,32,128,0,128,3,1,1,False),(16,16,16,256,0,256,3,1,1,False),(16,8,8,512,0,512,3,1,1,False),(16,32,32,256,128,128,3,1,1,True),(16,64,64,128,64,64,3,1,1,True),(16,256,256,32,0,16,3,1,1,True),(16,256,256,16,0,16,3,1,1,False)]:
    case(*sh, check=False)
