#!/usr/bin/env python
"""Per-kernel timing of the sparse backbone on the C3 workload (HIP events around every C-ABI call).

    python -m toda_amd.tools.bench_kernels [--workload c3] [--iters 5]

Prints one line per (operator, shape): mean ms, algorithmic GB/s and TFLOP/s.  Used to pick kernel
variants (environment knobs TODA_GG_RT / TODA_GG_PF / TODA_GG_LDS / TODA_GG_LDS88 / TODA_WG_SUB are read once per process by
libtoda_hip.so; TODA_HIP_LIB points at an alternative build for A/B runs inside one gpurun call)."""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from toda_amd import lib as L  # noqa: E402
from toda_amd import ops  # noqa: E402
from toda_amd.pcdet.datasets import SyntheticLidarDataset  # noqa: E402
from toda_amd.pcdet.models import build_network, voxelize_on_gpu  # noqa: E402


class Recorder:
    """Wraps entry points of libtoda_hip.so with HIP events on the current stream."""

    def __init__(self, names):
        self.lib = L.load()
        self.events = collections.defaultdict(list)
        self.enabled = False
        for name in names:
            self._wrap(name)

    def _wrap(self, name):
        fn = getattr(self.lib, name)
        rec = self

        def timed(*args):
            if not rec.enabled:
                return fn(*args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            rec.events[(name,) + rec.label(name, args)].append((e0, e1))
            return rc

        setattr(self.lib, name, timed)

    @staticmethod
    def label(name, a):
        if name in ("toda_spconv_gather_gemm", "toda_spconv_gather_gemm_ordered"):
            return (a[5], a[6], a[2], a[7])          # rows, K, c_gather, c_produce
        if name == "toda_spconv_wgrad":
            return (a[4], a[5], a[6], a[7])          # rows, K, cin, cout
        if name in ("toda_rulebook_subm", "toda_rulebook_conv", "toda_gridindex_from_coords", "toda_gridindex_from_conv"):
            return (a[1],)
        if name == "toda_voxelize_hard":
            return (a[1], a[2])
        if name in ("toda_sparse_to_dense_fwd", "toda_sparse_to_dense_bwd"):
            return (a[2], a[3])
        return ()

    def report(self):
        torch.cuda.synchronize()
        rows = []
        for key, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            rows.append((sum(ms) / len(ms) * (len(ms)), key, sum(ms) / len(ms), len(ms)))
        rows.sort(reverse=True)
        return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    yaml_path, per_gpu, _ = bench.WORKLOADS[args.workload]
    cfg = bench.load_cfg(yaml_path)
    dataset = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
    torch.manual_seed(0)
    device = torch.device("cuda", 0)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), dataset).to(device).train()
    batch0 = bench.make_device_batches(dataset, per_gpu, 1, 0, device)[0]
    rec = Recorder(["toda_spconv_gather_gemm_ordered", "toda_spconv_wgrad", "toda_rulebook_subm", "toda_rulebook_conv",
                    "toda_gridindex_from_coords", "toda_gridindex_from_conv", "toda_voxelize_hard", "toda_mean_vfe_fwd",
                    "toda_sparse_to_dense_fwd", "toda_sparse_to_dense_bwd", "toda_center_assign",
                    "toda_spconv_pack_weight"])

    def step():
        batch = dict(batch0)
        voxelize_on_gpu(batch, dataset.voxel_cfg)
        ret, _, _ = model(batch)
        ret["loss"].backward()
        model.zero_grad(set_to_none=True)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    rec.enabled = True
    for _ in range(args.iters):
        step()
    rows = rec.report()
    total = sum(r[0] for r in rows) / args.iters
    print(f"# {args.workload}: {total:.3f} ms/step inside libtoda_hip.so "
          f"(TODA_GG_RT={os.environ.get('TODA_GG_RT', '0')} TODA_GG_PF={os.environ.get('TODA_GG_PF', '0')} "
          f"TODA_GG_LDS={os.environ.get('TODA_GG_LDS', '1')} TODA_WG_SUB={os.environ.get('TODA_WG_SUB', '7')})")
    for tot, key, mean, n in rows:
        extra = ""
        if key[0] in ("toda_spconv_gather_gemm", "toda_spconv_gather_gemm_ordered", "toda_spconv_wgrad"):
            rows_, K, ci, co = key[1:]
            dense = 2.0 * rows_ * K * ci * co
            extra = f"  dense-equiv {dense / (mean * 1e-3) / 1e12:6.1f} TF/s"
        print(f"{tot / args.iters:8.3f} ms/step  {mean:8.4f} ms x{n / args.iters:4.1f}  {key}{extra}")


if __name__ == "__main__":
    main()
