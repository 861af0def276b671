#!/usr/bin/env python
"""Timing of the dense 2-D part (BaseBEVBackbone + CenterHead convs, fwd+bwd) under MIOpen options:
    python -m toda_amd.tools.bench_dense [--benchmark] [--channels-last]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from toda_amd.pcdet.datasets import SyntheticLidarDataset  # noqa: E402
from toda_amd.pcdet.models import build_network  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--benchmark", action="store_true")
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = args.benchmark
    cfg = bench.load_cfg(bench.WORKLOADS["c3"][0])
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES)
    model = build_network(cfg.MODEL, 3, ds).cuda().train()
    x = torch.randn(2, 256, 188, 188, device="cuda").relu_()
    if args.channels_last:
        model.backbone_2d = model.backbone_2d.to(memory_format=torch.channels_last)
        model.dense_head = model.dense_head.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)

    def step():
        d = model.backbone_2d({"spatial_features": x})
        y = model.dense_head.shared_conv(d["spatial_features_2d"])
        outs = [h(y) for h in model.dense_head.heads_list]
        loss = sum(v.square().mean() for o in outs for v in o.values())
        loss.backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        step()
    torch.cuda.synchronize()
    print(f"benchmark={args.benchmark} channels_last={args.channels_last}: {(time.perf_counter() - t0) / args.iters * 1e3:.2f} ms/iter")


if __name__ == "__main__":
    main()
