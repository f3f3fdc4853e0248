#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace CSV per (kernel, grid): launches/step, avg us, us/step."""
import collections
import csv
import sys


def main(path, steps):
    rows = list(csv.DictReader(open(path)))
    agg = collections.OrderedDict()
    total = 0.0
    for r in rows:
        name = r["Kernel_Name"].replace("void d3f::", "").replace("d3f::", "")
        name = name.split("(")[0]
        key = (name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]))
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += d
        total += d
    print(f"total kernel time per step: {total / steps:.1f} us")
    byname = collections.Counter()
    for (name, gx, gy), (n, t) in agg.items():
        byname[name] += t
    print("-- by kernel --")
    for name, t in byname.most_common(16):
        print(f"{name[:70]:70s} {t / steps:9.1f} us/step {100 * t / total:5.1f}%")
    print("-- by (kernel, grid) --")
    for (name, gx, gy), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:34]:
        print(f"{name[:58]:58s} wg {gx:6d} x{gy:4d} n/step {n / steps:5.1f} avg {t / n:8.1f} us  step {t / steps:8.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
