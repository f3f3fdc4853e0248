#!/usr/bin/env python3
"""Every kernel launch of one steady-state training step, in start order, from rocprofv3's kernel trace:
   python3 profiles/tools/step_launches.py <trace_kernel_trace.csv> [step index from the end, default 2]
columns: start (us from the previous step's Adam), stream, duration, gap to the previous launch of the same stream,
workgroups, kernel (template arguments kept)."""
import csv
import sys


def name(n):
    return n.replace("void d3f::", "").replace("d3f::", "").split("(")[0][:70]


def main(path, back=2):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id")),
                 name(r["Kernel_Name"]),
                 int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) * max(int(r["Grid_Size_Y"]), 1) *
                 max(int(r["Grid_Size_Z"]), 1)) for r in rows)
    adam = [i for i, e in enumerate(ev) if e[3].startswith("adam_kernel")]
    a0, a1 = adam[-back - 1], adam[-back]
    t0 = ev[a0][1]
    last_end = {}
    for s, e, sid, n, wg in ev[a0 + 1:a1 + 1]:
        gap = (s - last_end[sid]) / 1e3 if sid in last_end else 0.0
        last_end[sid] = e
        print(f"{(s - t0) / 1e3:9.1f} s{sid} {(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  wg {wg:6d}  {n}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
