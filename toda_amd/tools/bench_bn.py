#!/usr/bin/env python
"""HBM rate of the fused sparse-BatchNorm kernels on the C3 level shapes:  python -m toda_amd.tools.bench_bn"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from toda_amd import lib as L  # noqa: E402

SHAPES = [(300000, 16), (682284, 32), (389533, 64), (116982, 64), (111493, 128)]


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    lib = L.load()
    tot = {"moments": 0.0, "affine": 0.0, "bwd": 0.0}
    for n, c in SHAPES:
        x = torch.randn((n, c), device="cuda")
        dy = torch.randn((n, c), device="cuda")
        y = torch.empty_like(x)
        sums = torch.zeros((lib.toda_rows_reduce_doubles(n, c),), dtype=torch.float64, device="cuda")
        stats = torch.rand((4, c), device="cuda") + 0.5
        gamma = torch.ones(c, device="cuda")
        s = L.stream()
        t_m = timed(lambda: lib.toda_rows_moments(L.ptr(x), n, c, L.ptr(sums), s))
        t_a = timed(lambda: lib.toda_rows_affine_act(L.ptr(x), L.ptr(stats[2]), L.ptr(stats[3]), None, n, c, 1, L.ptr(y), s))
        t_b = timed(lambda: lib.toda_rows_bn_bwd(L.ptr(dy), L.ptr(x), L.ptr(stats), L.ptr(gamma), n, c, 1, L.ptr(sums), L.ptr(y), s))
        e = n * c
        print(f"[{n:7d} x {c:3d}] moments {t_m * 1e3:6.1f} us {4 * e / t_m / 1e6:7.0f} GB/s | affine+relu {t_a * 1e3:6.1f} us {8 * e / t_a / 1e6:7.0f} GB/s"
              f" | bwd (reduce+apply) {t_b * 1e3:6.1f} us {20 * e / t_b / 1e6:7.0f} GB/s")
        tot["moments"] += t_m
        tot["affine"] += t_a
        tot["bwd"] += t_b
    print("sum over the 5 shapes (ms):", {k: round(v, 4) for k, v in tot.items()})


if __name__ == "__main__":
    main()
