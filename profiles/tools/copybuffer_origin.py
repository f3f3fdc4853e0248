#!/usr/bin/env python3
"""Where do the __amd_rocclr_copyBuffer / fillBuffer launches of a traced bench.py run sit?
   python3 profiles/tools/copybuffer_origin.py <kernel_trace.csv>
Step boundaries = launches of adam_kernel.  Prints the blit kernels before the first step (model construction), per
steady-state step (with stream and the kernels right before / after each one), and after the last step."""
import csv
import sys


def short(n):
    return n.replace("void d3f::", "").replace("d3f::", "").split("(")[0].split("<")[0]


def main(path):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id")), short(r["Kernel_Name"]))
                for r in rows)
    adam = [i for i, e in enumerate(ev) if e[3].startswith("adam")]
    blit = [i for i, e in enumerate(ev) if "rocclr" in e[3]]
    print(f"{len(ev)} launches, {len(adam)} adam launches, {len(blit)} blit launches")
    if not adam:
        return
    before = [i for i in blit if i < adam[0]]
    after = [i for i in blit if i > adam[-1]]
    print(f"before the first optimiser step: {len(before)}; after the last: {len(after)}")
    for k in range(len(adam) - 1):
        inside = [i for i in blit if adam[k] < i < adam[k + 1]]
        print(f"step {k}: {len(inside)} blit launches")
        if k == len(adam) - 2 or k == 1:
            for i in inside:
                prev = ev[i - 1][3] if i > 0 else "-"
                nxt = ev[i + 1][3] if i + 1 < len(ev) else "-"
                print(f"    {ev[i][3]:36s} stream {ev[i][2]} {1e-3 * (ev[i][1] - ev[i][0]):7.1f} us   after {prev}   before {nxt}")


if __name__ == "__main__":
    main(sys.argv[1])
