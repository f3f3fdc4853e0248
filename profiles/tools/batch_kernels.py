#!/usr/bin/env python3
"""Kernel time by name over the last N optimiser steps of a rocprofv3 kernel trace (step boundary = a launch of adam_kernel).
   python3 profiles/tools/batch_kernels.py <kernel_trace.csv> [optimiser steps per batch, default 1] [rows, default 40]"""
import collections
import csv
import sys


def short(n):
    return n.replace("void ", "").replace("d3f::", "").split("(")[0][:76]


def main(path, per_batch=1, top=40):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id")), short(r["Kernel_Name"]))
                for r in rows)
    adam = [i for i, e in enumerate(ev) if e[3].startswith("adam")]
    if len(adam) < per_batch + 1:
        print(f"only {len(adam)} adam launches in the trace")
        return
    a0, a1 = adam[-per_batch - 1], adam[-1]
    c, t = collections.Counter(), collections.Counter()
    streams = collections.Counter()
    for s, e, sid, n in ev[a0 + 1:a1 + 1]:
        c[n] += 1
        t[n] += e - s
        streams[sid] += e - s
    span = ev[a1][1] - ev[a0][1]
    print(f"last batch ({per_batch} optimiser steps): span {1e-3 * span:.1f} us, {a1 - a0} launches, kernel time "
          f"{1e-3 * sum(t.values()):.1f} us; busy per stream: " + ", ".join(f"{k}: {1e-3 * v:.0f}" for k, v in streams.items()))
    for n, us in sorted(t.items(), key=lambda kv: -kv[1])[:top]:
        print(f"{n:78s} {c[n]:4d} {1e-3 * us:9.1f} us  avg {1e-3 * us / c[n]:8.1f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1, int(sys.argv[3]) if len(sys.argv) > 3 else 40)
