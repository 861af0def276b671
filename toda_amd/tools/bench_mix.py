#!/usr/bin/env python
"""Timing of the point-table kernels behind the mix processors (HIP events, 180k-point cloud, 40 boxes):
    python -m toda_amd.tools.bench_mix
Prints per primitive the mean time and the algorithmic HBM rate (bytes = point columns read + flags / rows written)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from toda_amd import ops  # noqa: E402
from toda_amd.pcdet.datasets.processor import point_mix  # noqa: E402
from toda_amd.pcdet.datasets.synthetic import synth_cloud  # noqa: E402


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    pts, bx, _ = synth_cloud("waymo_toda", 1, n_boxes=40)
    tpts, tbx, _ = synth_cloud("nuscenes_toda", 2, n_boxes=30)
    p = torch.from_numpy(pts[:, :4].copy()).cuda()
    b = torch.from_numpy(bx).cuda()
    n, c = p.shape
    flags = ops.points_sector(p, -1.0, 0.5)
    rows = []
    rows.append(("points_sector", timed(lambda: ops.points_sector(p, -1.0, 0.5)), n * (8 + 4)))
    rows.append(("points_rect", timed(lambda: ops.points_rect(p, [-20, -20], [20, 20])), n * (8 + 4)))
    ye, de = np.linspace(-np.pi, np.pi, 3), np.linspace(0, 54, 4)
    rows.append(("points_polar_cell", timed(lambda: ops.points_polar_cell(p, 0.3, ye, de, 1e-5, 54.0)), n * (8 + 4)))
    rows.append(("points_in_boxes k=40", timed(lambda: ops.points_in_boxes(p, b, 0)), n * (12 + 4)))
    rows.append(("rows_select_append", timed(lambda: ops.RowBuffer(n, c, "cuda").append(p, flags, 1)),
                 n * (4 + 4 * c) + int(flags.sum()) * 4 * c))
    rows.append(("points_rotate_z", timed(lambda: ops.points_rotate_z(p, 0.8, 0.6)), n * 8 * c))
    for name, ms, nbytes in rows:
        print(f"{name:24s} {ms * 1e3:8.1f} us   {nbytes / (ms * 1e-3) / 1e9:8.1f} GB/s algorithmic")
    src = {"points": p, "gt_boxes": np.concatenate([bx, np.ones((len(bx), 1), np.float32)], 1)}
    tgt = {"points": torch.from_numpy(tpts[:, :4].copy()).cuda(), "gt_boxes": np.concatenate([tbx, np.ones((len(tbx), 1), np.float32)], 1)}
    for name, fn in [("polarmix center", lambda: point_mix.polarmix(src, tgt, 1, 1.570796, 0.0, ["FIX", "FIX", "FIX"], "center")),
                     ("polarmix corner_del", lambda: point_mix.polarmix(src, tgt, 2, 1.570796, 0.0, ["FIX", "FIX", "FIX"], "corner_del")),
                     ("lasermix 3x2", lambda: point_mix.lasermix_cyc(src, tgt, 3, 2, [-54, -54, -5, 54, 54, 4.8], "center")),
                     ("mixup_cd", lambda: point_mix.mixup(src, tgt, 2.0, collision=True))]:
        np.random.seed(0)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        print(f"{name:24s} {(time.perf_counter() - t0) / 20 * 1e3:8.2f} ms wall per mixed scene (host box logic + launches + 1 sync)")


if __name__ == "__main__":
    main()
