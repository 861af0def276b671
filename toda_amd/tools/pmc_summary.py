#!/usr/bin/env python
"""Fold rocprofv3 --pmc counter_collection CSVs into the per-launch-shape summary bench.py reads for `roofline.traffic`.
    pmc_summary.py <kernel substring> <out.json> <counter_collection.csv> [more csv ...]
The summary records the SHA-256 of the kernel sources at collection time; bench.py refuses it once they change.
Every CSV comes from its own pass (one --pmc set per run, as MI355X_MICROARCH.md prescribes).  Counters are averaged
per (kernel, grid size); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction of the guide's HBM section)."""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SOURCES = ["toda_amd/csrc/spconv.hip", "toda_amd/csrc/spconv_split.cuh", "toda_amd/csrc/split_common.cuh", "toda_amd/csrc/conv2d.hip"]


def main():
    pattern, out = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    names = set()
    for path in sys.argv[3:]:
        for r in csv.DictReader(open(path)):
            if pattern not in r["Kernel_Name"]:
                continue
            kname = r["Kernel_Name"].split("(")[0]
            names.add(kname)
            acc[(kname, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):      # the dispatch's duration under the counter pass
                acc[(kname, int(r["Grid_Size"]))]["_duration_ns:" + r["Counter_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    shapes = []
    for (kname, grid), counters in sorted(acc.items()):
        row = {"kernel": kname, "grid_threads": grid, "dispatches": max(len(v) for v in counters.values())}
        durations = {}
        for name, vals in sorted(counters.items()):
            if name.startswith("_duration_ns:"):
                durations[name.split(":", 1)[1]] = sum(vals) / len(vals)
            else:
                row[name] = sum(vals) / len(vals)
        # shader clock the kernel actually ran at: busy cycles of the pass that counted them over that pass's own duration
        if "GRBM_GUI_ACTIVE" in row and durations.get("GRBM_GUI_ACTIVE"):
            row["duration_ns_in_clock_pass"] = durations["GRBM_GUI_ACTIVE"]
            row["sclk_ghz"] = row["GRBM_GUI_ACTIVE"] / durations["GRBM_GUI_ACTIVE"]
        if "FETCH_SIZE" in row and "WRITE_SIZE" in row:
            row["hbm_bytes"] = int((2 * row["FETCH_SIZE"] + row["WRITE_SIZE"]) * 1024)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in row and "SQ_BUSY_CU_CYCLES" in row:
            row["mfma_pipe_busy"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * row["SQ_BUSY_CU_CYCLES"])
        shapes.append(row)
    sha = {p: hashlib.sha256(open(os.path.join(ROOT, p), "rb").read()).hexdigest() for p in SOURCES if os.path.exists(os.path.join(ROOT, p))}
    json.dump({"kernel": sorted(names), "source_sha256": sha, "note": "one rocprofv3 --pmc pass per counter set over `python bench.py --steps 2 --warmup 2 "
               "--no-cpu-baseline`; FETCH_SIZE / WRITE_SIZE in KiB per dispatch; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024",
               "launch_shapes": shapes}, open(out, "w"), indent=1)
    for s in shapes:
        print(s)


if __name__ == "__main__":
    main()
