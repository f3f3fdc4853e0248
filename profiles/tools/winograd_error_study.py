"""fp32 error of Winograd F(2x2,3x3) vs direct convolution (both against float64), for layer-like operands"""
import torch, math
torch.manual_seed(0)
torch.set_num_threads(8)
G = torch.tensor([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], dtype=torch.float64)
Bt = torch.tensor([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], dtype=torch.float64)
At = torch.tensor([[1,1,1,0],[0,1,-1,-1]], dtype=torch.float64)
def winograd(x, w, dt):
    # x [B,C,H,W] (H, W even), w [K,C,3,3]; 'same' conv, pad 1
    B,C,H,W = x.shape; K = w.shape[0]
    xp = torch.nn.functional.pad(x, (1,1,1,1)).to(dt)
    U = torch.einsum('ij,kcjl,ml->kcim', G.to(dt), w.to(dt), G.to(dt))          # [K,C,4,4]
    # tiles: 4x4 with stride 2
    t = xp.unfold(2,4,2).unfold(3,4,2)                                           # [B,C,H/2,W/2,4,4]
    V = torch.einsum('ij,bcyxjl,ml->bcyxim', Bt.to(dt), t, Bt.to(dt))            # [B,C,ty,tx,4,4]
    M = torch.einsum('kcim,bcyxim->bkyxim', U, V)                                # 16 GEMMs (fp32 accumulate in dt)
    Y = torch.einsum('ij,bkyxjl,ml->bkyxim', At.to(dt), M, At.to(dt))            # [B,K,ty,tx,2,2]
    return Y.permute(0,1,2,4,3,5).reshape(B,K,H,W)
for (C,K,H) in [(64,64,32),(128,128,16),(256,256,16),(512,512,8)]:
    x = torch.relu(torch.randn(4,C,H,H))*1.2 + 0.0
    w = torch.randn(K,C,3,3)/math.sqrt(9*C)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    d32 = torch.nn.functional.conv2d(x, w, padding=1)
    w32 = winograd(x, w, torch.float32)
    w64 = winograd(x, w, torch.float64)
    e = lambda a: float((a.double()-ref).norm()/ref.norm())
    m = lambda a: float((a.double()-ref).abs().max()/ref.abs().max())
    print(f"C={C:3d} K={K:3d} H={H:2d}: direct fp32 rel {e(d32):.2e} max {m(d32):.2e} | winograd fp32 rel {e(w32):.2e} max {m(w32):.2e}  ratio {e(w32)/e(d32):.1f}x | winograd fp64 rel {e(w64):.1e}")
