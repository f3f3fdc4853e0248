"""Per-workgroup phase times of the conv kernel (prologue / main loop / epilogue) for every layer shape.

Needs a profiling build of the library with -DD3F_PHASE_TIMING (see conv_igemm.hip):
    cd denoising_diffusion_deep_fake_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off \
        -DD3F_PHASE_TIMING -c conv_igemm.hip -o /tmp/ci_phase.o && hipcc --offload-arch=gfx950 -shared -fPIC /tmp/ci_phase.o \
        build/{conv_wgrad,conv_wgrad_patch,pointwise,loss,optim,engine,c_api}.o -o ../../scratch/libd3f_phase.so
    D3F_LIB=$PWD/scratch/libd3f_phase.so python profiles/tools/phase_timing.py
The clock is the 100 MHz wall clock (10 ns ticks); numbers are averages over workgroups of thread 0's timestamps."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from denoising_diffusion_deep_fake_amd import _lib, ops

B, S = 16, int(os.environ.get("SIZE", 256))
DTN = os.environ.get("DT", "f32")   # f32 | bf16
DT = {"f32": ops.F32, "bf16": ops.BF16}[DTN]
TDT = torch.bfloat16 if DTN == "bf16" else torch.float32
shapes = {
    "l1 64->64 @/4":      (S // 4, S // 4, 64, 0, 64, 3, 1, 1, False),
    "l2 128->128 @/8":    (S // 8, S // 8, 128, 0, 128, 3, 1, 1, False),
    "l3 256->256 @/16":   (S // 16, S // 16, 256, 0, 256, 3, 1, 1, False),
    "l4 512->512 @/32":   (S // 32, S // 32, 512, 0, 512, 3, 1, 1, False),
    "d2.1 192->64 @/4":   (S // 4, S // 4, 128, 64, 64, 3, 1, 1, True),
    "d3.2 32->32 @/2":    (S // 2, S // 2, 32, 0, 32, 3, 1, 1, False),
    "d4.1 32->16 @/1":    (S, S, 32, 0, 16, 3, 1, 1, True),
    "d4.2 16->16 @/1":    (S, S, 16, 0, 16, 3, 1, 1, False),
}
L = _lib.lib()
L.d3f_debug_phase_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 8)()
print(f"{'layer':20s} {'kernel us':>9s} {'wg':>6s} | per workgroup (us): prologue  loop  epilogue  lifetime | shader clock in the loop (MHz)")
for name, (H, W, C0, C1, Co, k, st, pd, up) in shapes.items():
    d = ops.make_desc(B, H, W, C0, C1, Co, k, st, pd, up)
    h0, w0 = (H // 2, W // 2) if up else (H, W)
    s0 = torch.randn(B, h0, w0, C0, device="cuda").to(TDT)
    s1 = torch.randn(B, H, W, C1, device="cuda").to(TDT) if C1 else None
    w = torch.randn(Co, C0 + C1, k, k, device="cuda") * 0.05
    wf, wd = ops.pack_weights(d, w, DT)
    for _ in range(3):
        ops.conv_forward(d, s0, s1, wf, DT, splitk=False)
    torch.cuda.synchronize()
    L.d3f_debug_phase_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        ops.conv_forward(d, s0, s1, wf, DT, splitk=False)
    e1.record()
    torch.cuda.synchronize()
    L.d3f_debug_phase_read(buf, 1)
    wg = buf[3] / n
    t = [buf[i] / max(buf[3], 1) * 0.01 for i in (0, 1, 2, 4)]
    mhz = buf[5] / max(buf[1], 1) * 100.0  # shader cycles per 100 MHz tick of the loop phase
    print(f"{name:20s} {e0.elapsed_time(e1) / n * 1e3:9.1f} {wg:6.0f} | {t[0]:8.2f} {t[1]:8.2f} {t[2]:8.2f} {t[3]:8.2f} | {mhz:6.0f}")
