import sys, copy; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, torch.nn.functional as F, oracle
from denoising_diffusion_deep_fake_amd import Unet, ops
from util import rel_l2, to_nchw, to_nhwc
BF=ops.BF16
# per-op wgrad in bf16
for (B,H,W,C0,C1,Co,k,s,pd,up) in [(2,16,16,64,0,64,3,1,1,False),(2,16,16,64,64,32,3,1,1,True),(2,16,16,32,0,16,3,1,1,True),(2,16,16,16,0,16,3,1,1,False),(1,8,8,256,0,256,3,1,1,False),(2,16,16,64,0,128,3,2,1,False),(2,32,32,8,0,64,7,2,3,False)]:
    g=torch.Generator().manual_seed(0)
    cr = 3 if k==7 else C0+C1
    h0,w0=(H//2,W//2) if up else (H,W)
    x0=torch.randn(B,(cr if k==7 else C0),h0,w0,generator=g).bfloat16().float()
    x1=torch.randn(B,C1,H,W,generator=g).bfloat16().float() if C1 else None
    w=torch.randn(Co,cr,k,k,generator=g)
    xin=F.interpolate(x0,scale_factor=2,mode="nearest") if up else x0
    if C1: xin=torch.cat([xin,x1],1)
    wr=w.clone().requires_grad_(True)
    yr=F.conv2d(xin,wr,None,s,pd); dy=torch.randn(yr.shape,generator=g).bfloat16().float(); yr.backward(dy)
    d=ops.make_desc(B,H,W,C0,C1,Co,k,s,pd,up,cr)
    copad=(Co+7)//8*8
    dw=ops.conv_backward_weight(d,to_nhwc(dy,copad).bfloat16().cuda(),to_nhwc(x0,C0).bfloat16().cuda(),to_nhwc(x1).bfloat16().cuda() if C1 else None,dtype=BF)
    print("wgrad bf16", (B,H,W,C0,C1,Co,k,s), "rel %.2e"%rel_l2(dw.cpu(),wr.grad))
# whole net
torch.manual_seed(1)
ref=oracle.Unet("resnet34",None,3,3,None).train()
net=Unet("resnet34",None,3,3,None,compute_dtype="bf16"); net.load_state_dict(ref.state_dict()); net=net.cuda().train()
net32=Unet("resnet34",None,3,3,None); net32.load_state_dict(ref.state_dict()); net32=net32.cuda().train()
for (Bn,S) in [(4,64),(8,128)]:
    x=oracle.synthetic_face_crops(Bn,S,seed=11); g=torch.Generator().manual_seed(5)
    noise=torch.randn(x.shape,generator=g); r=torch.rand(Bn,generator=g)*0.5+0.05
    noisy=oracle.step_oracle.blend_with_given_noise(x,noise,r)
    crit=oracle.MseStructuralSimilarityLoss(-1.,1.)
    for p in ref.parameters(): p.grad=None
    pr=ref(noisy); crit(pr,x).backward()
    outs={}
    for name,n in (("bf16",net),("f32",net32)):
        for p in n.parameters(): p.grad=None
        pred=n(noisy.cuda()); lossv,gp=ops.mse_ssim_loss(pred.detach(),x.cuda()); pred.backward(gp)
        g32=torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
        cos=F.cosine_similarity(n.flat_grads.cpu().double(),g32.double(),dim=0).item()
        print(name,"B",Bn,"S",S,"fwd rel %.3e loss %.5f (ref %.5f) grad rel %.3e cos %.5f"%(rel_l2(pred,pr),lossv[0].item(),crit(pr,x).item(),rel_l2(n.flat_grads,g32),cos))
