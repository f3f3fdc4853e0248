import sys, os; sys.path.insert(0, '/root/repo')
import torch
from denoising_diffusion_deep_fake_amd import ops
B,H,W,C0,C1,Co,k,s,pd,up = 16, 32, 32, 128, 0, 128, 3, 1, 1, False
d = ops.make_desc(B,H,W,C0,C1,Co,k,s,pd,up)
s0 = torch.randn(B,H,W,C0, device="cuda"); w = torch.randn(Co, C0, k, k, device="cuda")*0.05
wf, wd = ops.pack_weights(d, w)
for _ in range(30): ops.conv_forward(d, s0, None, wf, splitk=True)
torch.cuda.synchronize()
