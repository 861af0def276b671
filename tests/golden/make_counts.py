#!/usr/bin/env python
"""Freezes the synthetic generator and the index arithmetic at the FULL BASELINE sizes (SURVEY.md §8d:
"freeze the generator's SHA + per-config counts (N', M, N_l, Pairs per layer)").

    python tests/golden/make_counts.py            # writes tests/golden/counts.json (oracle, CPU, ~1 min)

For C2 / C3 / C5 it takes dataset samples 0 and 1 as one batch, runs the CPU oracle (voxelize ->
rulebooks of the VoxelBackBone8x index plan) and stores counts plus order-sensitive checksums of every
index table.  tests/test_oracle_counts.py re-derives one config on the CPU; tests/test_gpu_counts.py
derives all of them through libtoda_hip.so and must agree bit for bit.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CONFIGS = {
    "c2": "toda_amd/tools/cfgs/models/second_backbone_nuscenes.yaml",
    "c3": "toda_amd/tools/cfgs/models/centerpoint_voxel_waymo.yaml",
    "c5": "toda_amd/tools/cfgs/models/toda_stage1_centerpoint_res.yaml",
}

# VoxelBackBone8x / VoxelResBackBone8x index plan (reference spconv_backbone.py:77-117, 191-232)
PLAN = [
    ("subm1", "subm", dict(ksize=3)),
    ("spconv2", "conv", dict(ksize=3, stride=2, pad=1)),
    ("subm2", "subm", dict(ksize=3)),
    ("spconv3", "conv", dict(ksize=3, stride=2, pad=1)),
    ("subm3", "subm", dict(ksize=3)),
    ("spconv4", "conv", dict(ksize=3, stride=2, pad=(0, 1, 1))),
    ("subm4", "subm", dict(ksize=3)),
    ("spconv_down2", "conv", dict(ksize=(3, 1, 1), stride=(2, 1, 1), pad=0)),
]

MOD = (1 << 61) - 1


def table_checksum(t):
    """Order-sensitive checksum of an int table: sum((v + 2) * (flat position + 1)) mod (2^61 - 1)."""
    v = np.asarray(t).astype(np.int64).ravel() + 2
    pos = np.arange(1, v.size + 1, dtype=np.int64)
    # 128-bit safe: split the position weight
    lo = (v * (pos & 0xFFFFF)) % MOD
    hi = ((v * (pos >> 20)) % MOD) * ((1 << 20) % MOD) % MOD
    return int((lo.sum() % MOD + hi.sum() % MOD) % MOD)


def load_dataset(name):
    from toda_amd.pcdet.config import AttrDict, cfg_from_yaml_file
    from toda_amd.pcdet.datasets import SyntheticLidarDataset

    cfg = AttrDict()
    cfg_from_yaml_file(os.path.join(ROOT, CONFIGS[name]), cfg)
    return SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)


def derive(name, voxelize, subm, conv):
    """voxelize(points) -> (coords_zyx [M,3], num [M]); subm(idx, batch, shape, ksize) -> (nbr, cnt);
    conv(idx, batch, shape, ksize, stride, pad) -> (idx_out, shape_out, o2i, i2o, cnt).  All numpy."""
    ds = load_dataset(name)
    rec = {"samples": [], "levels": {}}
    coords_all = []
    for i in range(2):
        pts = ds[i]["points"]
        zyx, num = voxelize(pts)
        rec["samples"].append({
            "points_sha256": hashlib.sha256(np.ascontiguousarray(pts).tobytes()).hexdigest(),
            "n_points": int(pts.shape[0]), "n_voxels": int(zyx.shape[0]),
            "coords_checksum": table_checksum(zyx), "num_points_checksum": table_checksum(num),
        })
        coords_all.append(np.concatenate([np.full((zyx.shape[0], 1), i, np.int32), zyx.astype(np.int32)], 1))
    idx = np.concatenate(coords_all, 0)
    gx, gy, gz = (int(v) for v in ds.grid_size)
    shape = [gz + 1, gy, gx]
    for key, kind, kw in PLAN:
        if kind == "subm":
            nbr, cnt = subm(idx, 2, shape, kw["ksize"])
            rec["levels"][key] = {"rows": int(idx.shape[0]), "shape": list(shape), "pairs": int(np.sum(cnt)),
                                  "pair_cnt": [int(c) for c in cnt], "nbr_checksum": table_checksum(nbr)}
        else:
            idx_out, shape_out, o2i, i2o, cnt = conv(idx, 2, shape, kw["ksize"], kw["stride"], kw["pad"])
            rec["levels"][key] = {"rows_in": int(idx.shape[0]), "rows_out": int(idx_out.shape[0]), "shape_out": list(shape_out),
                                  "pairs": int(np.sum(cnt)), "pair_cnt": [int(c) for c in cnt],
                                  "out_indices_checksum": table_checksum(idx_out),
                                  "o2i_checksum": table_checksum(o2i), "i2o_checksum": table_checksum(i2o)}
            idx, shape = idx_out, list(shape_out)
    return rec


def derive_oracle(name):
    from oracle import oracle as O

    ds = load_dataset(name)
    vc = ds.voxel_cfg

    def voxelize(pts):
        _, zyx, num = O.voxelize_hard(pts, vc["point_cloud_range"], vc["voxel_size"], vc["max_points_per_voxel"], vc["max_num_voxels"])
        return zyx, num

    return derive(name, voxelize, lambda idx, b, sh, ks: O.rulebook_subm(idx, b, sh, ks),
                  lambda idx, b, sh, ks, st, pd: O.rulebook_conv(idx, b, sh, ks, st, pd))


def main():
    out = {name: derive_oracle(name) for name in CONFIGS}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "counts.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for name, rec in out.items():
        print(name, [s["n_voxels"] for s in rec["samples"]], {k: v["pairs"] for k, v in rec["levels"].items()})


if __name__ == "__main__":
    main()
