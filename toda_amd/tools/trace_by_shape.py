#!/usr/bin/env python
"""Per (kernel, grid size) averages of the hand-written sparse-conv kernels over the timed steps of a rocprofv3
--kernel-trace CSV of bench.py:  trace_by_shape.py <kernel_trace.csv> <timed_steps> [out.csv]"""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    from trace_summary import timed_window
    t0, t_end = timed_window(rows, steps)
    agg = collections.defaultdict(list)
    for r in rows:
        if not (t0 < int(r["End_Timestamp"]) <= t_end):
            continue
        name = r["Kernel_Name"]
        if "gather_gemm" in name or "wgrad_kernel" in name:
            grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))
            agg[(name.split("(")[0], grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lines = ["kernel,grid_threads,launches_per_step,avg_us,min_us,max_us"]
    for (name, grid), us in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f'"{name}",{grid},{len(us) / steps:.1f},{sum(us) / len(us):.1f},{min(us):.1f},{max(us):.1f}')
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text + "\n")


if __name__ == "__main__":
    main()
