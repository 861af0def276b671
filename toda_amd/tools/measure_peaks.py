#!/usr/bin/env python
"""Measured ceilings of this box next to the vendor peaks the rooflines use (SURVEY.md §8d asks for both):
    python -m toda_amd.tools.measure_peaks [out.json]
HBM: device-to-device copy and a read-modify-write stream (babelstream 'copy' / 'scale' shapes) over 4 GiB;
fp32 matrix: square torch.matmul (rocBLAS / hipBLASLt) with TF32-like shortcuts off."""
import json
import sys

import torch


def timed(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    torch.backends.cuda.matmul.allow_tf32 = False
    n = 1 << 30                                             # 4 GiB of fp32
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    t_copy = timed(lambda: b.copy_(a), 10)
    t_scale = timed(lambda: torch.mul(a, 1.0001, out=b), 10)
    out = {"hbm_copy_GBs": round(2 * 4 * n / t_copy / 1e9, 1), "hbm_scale_GBs": round(2 * 4 * n / t_scale / 1e9, 1),
           "hbm_vendor_GBs": 8000.0}
    del a, b
    for m in (4096, 8192, 16384):
        x = torch.randn((m, m), device="cuda")
        y = torch.randn((m, m), device="cuda")
        t = timed(lambda: torch.matmul(x, y), 10 if m < 16384 else 4)
        out[f"fp32_gemm_{m}_TFs"] = round(2 * m ** 3 / t / 1e12, 1)
    out["fp32_mfma_vendor_TFs"] = 157.3
    out["device"] = torch.cuda.get_device_name(0)
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
