"""Per-layer timing of the dense 3x3 convolutions of the BEV neck / heads: this repo's Winograd F(4x4,3x3) MFMA kernels against
torch's library path (MIOpen) on the same tensors, forward / data gradient / weight gradient.
    python -m toda_amd.tools.bench_conv2d [--config c3|c5] [--iters 20]
TFLOP/s are DIRECT-convolution FLOPs (2 * 9 * Cin * Cout * B * H * W) so both columns are comparable; the matrix-core work
of the Winograd kernels is 4x smaller."""
import argparse
import json

import torch
import torch.nn.functional as F

from toda_amd import ops

LAYERS = {
    # (name, batch, cin, cout, H, W, count per step)
    "c3": [("neck0.in 256->128 @188", 2, 256, 128, 188, 188, 1), ("neck0 128->128 @188", 2, 128, 128, 188, 188, 5),
           ("neck1 256->256 @94", 2, 256, 256, 94, 94, 5), ("head.shared 512->64 @188", 2, 512, 64, 188, 188, 1),
           ("head 64->64 @188", 2, 64, 64, 188, 188, 5)],
    "c5": [("neck0.in 256->128 @180", 2, 256, 128, 180, 180, 1), ("neck0 128->128 @180", 2, 128, 128, 180, 180, 5),
           ("neck1 256->256 @90", 2, 256, 256, 90, 90, 5), ("head.shared 512->64 @180", 2, 512, 64, 180, 180, 1),
           ("head 64->64 @180", 2, 64, 64, 180, 180, 5)],
}


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3", choices=sorted(LAYERS))
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    rows = []
    for name, b, cin, cout, h, w, count in LAYERS[args.config]:
        x = torch.relu(torch.randn(b, cin, h, w, device="cuda"))
        wt = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
        gy = torch.randn(b, cout, h, w, device="cuda")
        flop = 2.0 * 9 * cin * cout * b * h * w
        u0, u1 = ops.conv3x3_transform_weight(wt, 0), ops.conv3x3_transform_weight(wt, 1)
        r = {"layer": name, "count": count, "gflop": round(flop / 1e9, 2)}
        r["fwd_ms"] = timed(lambda: ops.conv3x3_run(x, u0, None, cout), args.iters)
        r["fwd_with_transform_ms"] = timed(lambda: ops.conv3x3_run(x, ops.conv3x3_transform_weight(wt, 0), None, cout), args.iters)
        r["dgrad_ms"] = timed(lambda: ops.conv3x3_run(gy, u1, None, cin), args.iters)
        r["wgrad_ms"] = timed(lambda: ops.conv3x3_wgrad(x, gy, wt.shape), args.iters)
        r["miopen_fwd_ms"] = timed(lambda: F.conv2d(x, wt, padding=1), args.iters)
        r["miopen_dgrad_ms"] = timed(lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                                                 (True, False, False)), args.iters)
        r["miopen_wgrad_ms"] = timed(lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                                                 (False, True, False)), args.iters)
        for k in ("fwd", "dgrad", "wgrad", "miopen_fwd", "miopen_dgrad", "miopen_wgrad"):
            r[k + "_tflops_direct"] = round(flop / (r[k + "_ms"] * 1e-3) / 1e12, 1)
        y = ops.conv3x3_run(x, u0, None, cout)
        ref = F.conv2d(x.double(), wt.double(), padding=1)
        r["fwd_max_rel_err_vs_fp64"] = float((y.double() - ref).abs().max() / ref.abs().max())
        r["miopen_max_rel_err_vs_fp64"] = float((F.conv2d(x, wt, padding=1).double() - ref).abs().max() / ref.abs().max())
        rows.append({k: (round(v, 4) if isinstance(v, float) and k.endswith("_ms") else v) for k, v in r.items()})
        print(json.dumps(rows[-1]), flush=True)
    tot = {k: round(sum(r[k] * r["count"] for r in rows), 3) for k in ("fwd_ms", "dgrad_ms", "wgrad_ms", "miopen_fwd_ms", "miopen_dgrad_ms", "miopen_wgrad_ms")}
    print(json.dumps({"per_step_totals_ms": tot}))


if __name__ == "__main__":
    main()
