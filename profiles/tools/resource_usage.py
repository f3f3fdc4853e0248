#!/usr/bin/env python3
"""Registers / LDS / occupancy of every kernel of one .hip file:
   python3 profiles/tools/resource_usage.py denoising_diffusion_deep_fake_amd/csrc/conv_igemm.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-ffp-contract=off",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split(" ")[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("void d3f::", "").split("(")[0]
    if flt not in dem:
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)  # noqa: E731
    occ, lds, scr = g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"), g(r"ScratchSize \[bytes/lane\]")
    print(f"{dem[:86]:86s} VGPR {g('VGPRs'):>3} AGPR {g('AGPRs'):>3} SGPR {g('TotalSGPRs'):>3} waves/SIMD {occ} "
          f"LDS {lds:>6} scratch {scr}")
