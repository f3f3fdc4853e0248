#!/usr/bin/env python3
"""One training step as seen by rocprofv3's kernel trace: where the two streams of the backward pass are busy / idle.
   python3 profiles/tools/step_timeline.py <trace_kernel_trace.csv> [step index from the end, default: the step of median span among the last six]
Step boundaries = launches of adam_kernel.  Prints, for one steady-state step: forward / backward windows, busy time per
stream, the idle intervals of the weight-gradient (side) stream inside the backward window, and the overhang (side
stream still running after the main chain's last backward kernel)."""
import csv
import sys


def short(n):
    return n.replace("void d3f::", "").replace("d3f::", "").split("(")[0].split("<")[0]


def main(path, back=None):
    rows = list(csv.DictReader(open(path)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id")), short(r["Kernel_Name"]))
                 for r in rows))
    adam = [i for i, e in enumerate(ev) if e[3] == "adam_kernel"]
    if back is None:
        # the step of MEDIAN span among the last six (a host hiccup under the tracer -- one 92 ms "step" in the round-6
        # trace -- says nothing about the schedule)
        cand = [(ev[adam[-k]][1] - ev[adam[-k - 1]][1], k) for k in range(1, min(7, len(adam)))]
        back = sorted(cand)[len(cand) // 2][1]
    a0, a1 = adam[-back - 1], adam[-back]
    step = ev[a0 + 1:a1 + 1]
    t0 = ev[a0][1]
    by = {}
    for s, e, sid, n in step:
        by.setdefault(sid, []).append((s - t0, e - t0, n))
    main_sid = max(by, key=lambda k: len(by[k]))
    print(f"step span {1e-3 * (step[-1][1] - t0):.1f} us, {len(step)} launches, streams: " +
          ", ".join(f"{k}: {len(v)} launches, busy {1e-3 * sum(e - s for s, e, _ in v):.0f} us" for k, v in by.items()))
    m = by[main_sid]
    gaps = sum(max(0, m[i][0] - m[i - 1][1]) for i in range(1, len(m)))
    print(f"main stream: inter-kernel gaps {1e-3 * gaps:.0f} us")
    fam = {}
    for s, e, n in m:
        fam[n] = fam.get(n, 0) + e - s
    for n, t in sorted(fam.items(), key=lambda kv: -kv[1])[:14]:
        print(f"    main {n:34s} {1e-3 * t:8.0f} us")
    for sid, v in by.items():
        if sid == main_sid:
            continue
        first, last = v[0][0], max(e for _, e, _ in v)
        # main-chain kernels of the backward pass: from the first bn_bwd / nchw_to_nhwc after ssim to the last conv/bn before adam
        bwd_main = [x for x in m if x[0] >= first - 200000 and x[2] != "adam_kernel"]
        main_end = max(e for _, e, n in bwd_main)
        idle, cur = 0, first
        holes = []
        for s, e, n in v:
            if s > cur:
                idle += s - cur
                if s - cur > 20000:
                    holes.append((cur, s))
            cur = max(cur, e)
        print(f"side stream {sid}: first kernel at {1e-3 * first:.0f} us, last ends {1e-3 * last:.0f} us; idle inside "
              f"{1e-3 * idle:.0f} us; main chain's last non-adam kernel ends {1e-3 * main_end:.0f} us -> overhang "
              f"{1e-3 * (last - main_end):.0f} us")
        for a, b in holes[:12]:
            print(f"    side idle {1e-3 * a:8.0f} .. {1e-3 * b:8.0f} us ({1e-3 * (b - a):.0f})")
        fam = {}
        for s, e, n in v:
            fam[n] = fam.get(n, 0) + e - s
        for n, t in sorted(fam.items(), key=lambda kv: -kv[1])[:6]:
            print(f"    side {n:34s} {1e-3 * t:8.0f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
