#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace CSV: busy time per stream, overlap between streams and idle gaps per step."""
import csv
import sys


def main(path, steps):
    rows = list(csv.DictReader(open(path)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], r["Kernel_Name"]) for r in rows))
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    streams = {}
    for s, e, sid, _ in ev:
        streams.setdefault(sid, 0)
        streams[sid] += e - s
    # union of busy intervals
    busy = 0
    cur_s, cur_e = ev[0][0], ev[0][1]
    for s, e, _, _ in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    total = sum(streams.values())
    print(f"span {1e-3 * (t1 - t0) / steps:.1f} us/step; union busy {1e-3 * busy / steps:.1f}; "
          f"sum of kernel times {1e-3 * total / steps:.1f}; overlapped {1e-3 * (total - busy) / steps:.1f}; "
          f"idle {1e-3 * ((t1 - t0) - busy) / steps:.1f}")
    for sid, t in sorted(streams.items(), key=lambda kv: -kv[1]):
        print(f"  stream {sid}: {1e-3 * t / steps:.1f} us/step")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))


def critical(path, steps):
    """Main-stream (the busiest stream) time by kernel family, plus the gaps between its kernels."""
    rows = list(csv.DictReader(open(path)))
    by = {}
    for r in rows:
        by.setdefault(r["Stream_Id"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    main_sid = max(by, key=lambda k: sum(e - s for s, e, _ in by[k]))
    ev = sorted(by[main_sid])
    fam, gaps, big = {}, 0, 0
    for i, (s, e, n) in enumerate(ev):
        key = n.split("<")[0].split("(")[0].replace("void d3f::", "").replace("(anonymous namespace)::", "")[:40]
        if "conv_igemm" in n:
            key = "conv_igemm"
        fam[key] = fam.get(key, 0) + e - s
        if i:
            g = s - ev[i - 1][1]
            if 0 < g < 200000:
                gaps += g
            elif g >= 200000:
                big += g
    print(f"main stream {main_sid}: inter-kernel gaps {1e-3 * gaps / steps:.1f} us/step (host-side pauses {1e-3 * big / steps:.1f})")
    for k, t in sorted(fam.items(), key=lambda kv: -kv[1])[:16]:
        print(f"  {k:42s} {1e-3 * t / steps:9.1f} us/step")


if __name__ == "__main__" and len(sys.argv) > 3:
    critical(sys.argv[1], int(sys.argv[2]))
