#!/usr/bin/env python
"""Kernel launch order of the LAST training step in a rocprofv3 --kernel-trace CSV of bench.py:
    trace_sequence.py <kernel_trace.csv> [out.txt]
One line per dispatch: start offset (us), duration (us), idle gap in front of it (us), short kernel name.  Used to find
which module a run of small torch kernels belongs to (the hand-written kernels on either side name the place)."""
import csv
import re
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"at::native::", "", name)
    m = re.match(r"([\w:]+(?:<[^(]{0,90})?)", name)
    return (m.group(1) if m else name)[:120]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    from trace_summary import timed_window
    lo, hi = timed_window(rows, 1)          # the last step: from the end of the optimizer step before it
    sel = [r for r in rows if lo < int(r["End_Timestamp"]) <= hi]
    t0 = int(sel[0]["Start_Timestamp"])
    end = t0
    out = []
    for r in sel:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        out.append(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {max(0, s - end) / 1e3:7.1f}  {short(r['Kernel_Name'])}")
        end = max(end, e)
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
