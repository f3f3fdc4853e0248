#!/usr/bin/env python3
"""Instruction mix of the hottest loop (most MFMAs, shortest) of every kernel in a hipcc -S listing:
   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -S --cuda-device-only -Iinclude <file>.hip -o x.s
   python3 profiles/tools/loop_mix.py x.s [name filter]
Columns: MFMA / vector-ALU / v_accvgpr moves / scalar / LDS / memory instructions, barriers, s_nop, branches.  What
round 2 found with it: wave-uniform state kept in vector registers, per-iteration AGPR<->VGPR accumulator copies, and
(via ScratchSize in the same listing) closures demoted to scratch memory."""
import re,sys,subprocess
txt=open(sys.argv[1]).read()
flt=sys.argv[2] if len(sys.argv)>2 else ''
for m in re.finditer(r'^(_ZN3d3f\w+):.*?s_endpgm', txt, re.S|re.M):
    body=m.group(0).split('\n'); name=m.group(1)
    dem=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip().replace('void d3f::','').split('(')[0]
    if flt not in dem: continue
    labels={l.split(':')[0]:i for i,l in enumerate(body) if l.startswith('.LBB')}
    best=None
    for i,l in enumerate(body):
        mm=re.search(r's_cbranch_\w+ (\.LBB\w+)',l)
        if mm and mm.group(1) in labels and labels[mm.group(1)]<i:
            a=labels[mm.group(1)]; seg=body[a:i+1]
            nm=sum('v_mfma' in x for x in seg)
            if nm and (best is None or nm>best[0] or (nm==best[0] and len(seg)<best[2])): best=(nm,a,len(seg),seg)
    if not best: continue
    seg=[x.strip() for x in best[3] if x.strip() and not x.strip().startswith(';') and not x.strip().startswith('.')]
    ops=[x.split()[0] for x in seg]
    valu=sum(o.startswith('v_') and 'mfma' not in o and 'accvgpr' not in o for o in ops)
    acc=sum('accvgpr' in o for o in ops); salu=sum(o.startswith('s_') and not o.startswith('s_waitcnt') and not o.startswith('s_barrier') and not o.startswith('s_nop') for o in ops)
    print(f"{dem[:66]:66s} mfma {best[0]:3d} valu {valu:3d} acc {acc:3d} salu {salu:3d} lds {sum(o.startswith('ds_') for o in ops):3d} vmem {sum(o.startswith('buffer_') or o.startswith('global_') for o in ops):3d} bar {sum(o=='s_barrier' for o in ops)} nop {sum(o=='s_nop' for o in ops)} br {sum(o.startswith('s_cbranch') for o in ops)}")
