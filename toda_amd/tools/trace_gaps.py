#!/usr/bin/env python
"""Idle gaps of the GPU inside the timed steps of a rocprofv3 --kernel-trace CSV of bench.py:
    trace_gaps.py <kernel_trace.csv> <timed_steps> [min_gap_us]
Lists, per (previous kernel -> next kernel) pair, how often and for how long the device sat idle between them."""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.split("(")[0]
    return name[-70:]


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    from trace_summary import timed_window
    t0, t_end = timed_window(rows, steps)
    sel = [r for r in rows if t0 < int(r["End_Timestamp"]) <= t_end]
    agg = collections.defaultdict(lambda: [0.0, 0])
    end = int(sel[0]["End_Timestamp"])
    total = 0.0
    small = 0.0
    for prev, cur in zip(sel, sel[1:]):
        gap = (int(cur["Start_Timestamp"]) - end) / 1e3
        if gap > 0:
            total += gap
            if gap >= min_gap:
                key = (short(prev["Kernel_Name"]), short(cur["Kernel_Name"]))
                agg[key][0] += gap
                agg[key][1] += 1
            else:
                small += gap
        end = max(end, int(cur["End_Timestamp"]))
    print(f"# idle {total / steps / 1e3:.3f} ms/step in total; {small / steps / 1e3:.3f} ms/step in gaps < {min_gap} us")
    for (a, b), (us, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
        print(f"{us / steps / 1e3:7.3f} ms/step  x{n / steps:5.1f}  avg {us / n:7.1f} us   {a}  ->  {b}")


if __name__ == "__main__":
    main()
