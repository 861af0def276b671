#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV of bench.py over its last <timed_steps> training steps only (the warm-up
holds MIOpen find-mode kernels).  Steps are delimited by their optimizer kernels (timed_window).
Usage: trace_summary.py <kernel_trace.csv> <timed_steps> [out.csv]"""
import collections
import csv
import sys


def timed_window(rows, steps):
    """[t0, t_end] of the last `steps` training steps of a bench.py kernel trace (rows sorted by start time).  A step ends with
    the last kernel of its optimizer launch cluster (toda::clip_adam_update_kernel, or torch's FusedOptimizer multi-tensor kernels less than 1 ms apart); the window
    runs from the end of the step before the first counted one to the end of the last - whatever the step does in between
    (its own voxelisation, or with the input pipeline the NEXT step's on the side stream) is inside."""
    adam = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows
                  if "FusedOptimizerTensorListMetadata" in r["Kernel_Name"] or "fused_adam" in r["Kernel_Name"].lower()
                  or "clip_adam_update_kernel" in r["Kernel_Name"])
    if not adam:
        # forward-only workload (BASELINE config 2): a step starts with its MeanVFE launch (one per forward, on the training stream)
        vfe = sorted(int(r["Start_Timestamp"]) for r in rows if "mean_vfe_fwd_kernel" in r["Kernel_Name"])
        if len(vfe) <= steps:
            raise SystemExit("neither optimizer nor MeanVFE kernels delimit enough steps in the trace")
        return vfe[-steps - 1] - 1, vfe[-1] - 1
    ends = []
    for s, e in adam:
        if ends and s - ends[-1] < 1_000_000:
            ends[-1] = max(ends[-1], e)
        else:
            ends.append(e)
    if len(ends) <= steps:
        raise SystemExit(f"only {len(ends)} optimizer steps in the trace, need more than {steps}")
    return ends[-steps - 1], ends[-1]


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0, t_end = timed_window(rows, steps)
    sel = [r for r in rows if t0 < int(r["End_Timestamp"]) <= t_end]
    agg = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        agg[r["Kernel_Name"]][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[r["Kernel_Name"]][1] += 1
    busy = sum(v[0] for v in agg.values()) / 1e6
    span = (t_end - t0) / 1e6
    lines = [f"# timed steps={steps} span_ms={span:.2f} busy_ms={busy:.2f} kernels_per_step={len(sel) / steps:.0f}",
             "kernel,calls_per_step,avg_us,ms_per_step"]
    groups = collections.defaultdict(float)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        lines.append(f'"{k}",{v[1] / steps:.1f},{v[0] / v[1] / 1e3:.2f},{v[0] / 1e6 / steps:.4f}')
        name = k
        if "toda::" in name:
            key = "toda: " + name.split("toda::")[1].split("(")[0].split("<")[0]
        elif any(t in name for t in ("miopenSp3", "igemm", "Cijk", "conv", "Conv", "Im2d", "Col2Im", "transpose", "SubTensor", "gemm")):
            key = "miopen/blas conv"
        elif "atch" in name and "orm" in name:
            key = "batchnorm (torch/miopen)"
        else:
            key = "torch elementwise/reduce/other"
        groups[key] += v[0] / 1e6 / steps
    print(lines[0])
    for k, v in sorted(groups.items(), key=lambda kv: -kv[1]):
        print(f"{v:8.3f} ms/step  {k}")
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
