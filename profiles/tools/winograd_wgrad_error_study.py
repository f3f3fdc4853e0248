"""fp32 error of the Winograd-domain WEIGHT gradient, F(2x2, 3x3) transposed:
    dg = G^T [ sum over tiles (B^T d B) (.) (A dY A^T) ] G        (16 [Cout x Cin] contractions over tiles instead of 9 taps
                                                                   over pixels: 2.25x fewer multiplies)
against the direct fp32 weight gradient, both measured against float64, for layer-like operands at the headline batch (16):
post-ReLU inputs, small dense output gradients.  Groundwork for DESIGN.md section 8 item 1: does the form fit the 2e-4
mask-pinned gate of tests/test_gpu_parity_layers.py?   python3 profiles/tools/winograd_wgrad_error_study.py   (CPU only)"""
import math

import torch

torch.manual_seed(0)
torch.set_num_threads(8)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def winograd_wgrad(x, dy, dt):
    """x [B,C,H,W], dy [B,K,H,W] ('same' 3x3 conv, pad 1) -> dw [K,C,3,3]; transforms and accumulation in dt"""
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1)).to(dt)
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                           # [B,C,ty,tx,4,4]
    V = torch.einsum('ij,bcyxjl,ml->bcyxim', Bt.to(dt), t, Bt.to(dt))                # B^T d B
    d = dy.to(dt).unfold(2, 2, 2).unfold(3, 2, 2)                                     # [B,K,ty,tx,2,2]
    D = torch.einsum('ji,bkyxjl,lm->bkyxim', At.to(dt), d, At.to(dt))                # A dY A^T  (A = At^T: 4x2)
    # 16 contractions over (b, ty, tx): [K x tiles] . [tiles x C] per position, accumulated in dt
    Bn, Cc = V.shape[0], V.shape[1]
    Kk = D.shape[1]
    Vm = V.permute(4, 5, 0, 2, 3, 1).reshape(16, -1, Cc)                              # [pos][tiles][C]
    Dm = D.permute(4, 5, 1, 0, 2, 3).reshape(16, Kk, -1)                              # [pos][K][tiles]
    M = torch.bmm(Dm, Vm).reshape(4, 4, Kk, Cc)                                       # [i][m][K][C]
    return torch.einsum('ia,imkc,mb->kcab', G.to(dt), M, G.to(dt))                    # G^T M G


rows = []
for (C, K, H, Bn) in [(64, 64, 64, 16), (128, 128, 32, 16), (256, 256, 16, 16), (512, 512, 8, 16)]:
    x = torch.relu(torch.randn(Bn, C, H, H)) * 1.2
    dy = torch.randn(Bn, K, H, H) * 1e-3
    xr = x.double().requires_grad_(False)
    w64 = torch.zeros(K, C, 3, 3, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(xr, w64, padding=1).backward(dy.double())
    ref = w64.grad
    w32 = torch.zeros(K, C, 3, 3, requires_grad=True)
    torch.nn.functional.conv2d(x, w32, padding=1).backward(dy)
    d32 = w32.grad
    g32 = winograd_wgrad(x, dy, torch.float32)
    g64 = winograd_wgrad(x, dy, torch.float64)
    e = lambda a: float((a.double() - ref).norm() / ref.norm())
    m = lambda a: float((a.double() - ref).abs().max() / ref.abs().max())
    print(f"C={C:3d} K={K:3d} {H:2d}x{H:<2d} bs {Bn}: direct fp32 rel {e(d32):.2e} max {m(d32):.2e} | winograd-domain fp32 rel "
          f"{e(g32):.2e} max {m(g32):.2e}  ratio {e(g32) / e(d32):.1f}x | winograd-domain fp64 rel {e(g64):.1e}")
