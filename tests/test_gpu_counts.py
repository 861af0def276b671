"""Bit-exact index parity at the FULL BASELINE sizes: voxelization and every rulebook of the
VoxelBackBone8x plan, derived through libtoda_hip.so on C2 / C3 / C5 clouds, must reproduce the counts
and order-sensitive table checksums the CPU oracle froze in tests/golden/counts.json."""
import numpy as np
import pytest
import torch

from tests.golden import make_counts
from tests.test_oracle_counts import load_frozen

pytestmark = pytest.mark.gpu


def derive_gpu(name):
    from toda_amd import ops

    ds = make_counts.load_dataset(name)
    vc = ds.voxel_cfg

    def voxelize(pts):
        _, zyx, num = ops.voxelize(torch.from_numpy(pts).cuda(), vc["point_cloud_range"], vc["voxel_size"],
                                   vc["max_points_per_voxel"], vc["max_num_voxels"])
        return zyx.cpu().numpy(), num.cpu().numpy()

    def subm(idx, batch, shape, ksize):
        rb, _ = ops.build_subm_rulebook(torch.from_numpy(idx).cuda(), batch, shape, ksize)
        return rb.nbr_fwd.cpu().numpy(), rb.pair_cnt.cpu().numpy()

    def conv(idx, batch, shape, ksize, stride, pad):
        idx_out, shape_out, rb, _ = ops.build_conv_rulebook(torch.from_numpy(idx).cuda(), batch, shape, ksize, stride, pad)
        return idx_out.cpu().numpy(), [int(v) for v in shape_out], rb.nbr_fwd.cpu().numpy(), rb.nbr_bwd.cpu().numpy(), rb.pair_cnt.cpu().numpy()

    return make_counts.derive(name, voxelize, subm, conv)


@pytest.mark.parametrize("name", ["c2", "c3", "c5"])
def test_gpu_reproduces_frozen_counts(name):
    frozen = load_frozen()[name]
    got = derive_gpu(name)
    assert got["samples"] == frozen["samples"]
    for key in frozen["levels"]:
        assert got["levels"][key] == frozen["levels"][key], key


def test_one_sync_index_plan_matches_frozen_counts():
    """The product path (build_index_plan: all levels back-to-back, one host sync) against the same fixture."""
    from toda_amd import ops

    frozen = load_frozen()["c3"]
    ds = make_counts.load_dataset("c3")
    vc = ds.voxel_cfg
    clouds = [torch.from_numpy(ds[i]["points"]).cuda() for i in range(2)]
    _, coords, _ = ops.voxelize_batch(clouds, vc["point_cloud_range"], vc["voxel_size"], vc["max_points_per_voxel"], vc["max_num_voxels"])
    gx, gy, gz = (int(v) for v in ds.grid_size)
    steps = []
    for key, kind, kw in make_counts.PLAN:
        if kind == "subm":
            steps.append({"kind": "subm", "key": key, "ksize": kw["ksize"], "dilation": 1})
        else:
            steps.append({"kind": "conv", "key": key, "ksize": kw["ksize"], "stride": kw["stride"], "padding": kw["pad"]})
    plan = ops.build_index_plan(coords, 2, [gz + 1, gy, gx], steps)
    for key, kind, _ in make_counts.PLAN:
        rb, want = plan[key]["rb"], frozen["levels"][key]
        assert int(rb.pair_cnt.sum()) == want["pairs"]
        if kind == "subm":
            assert make_counts.table_checksum(rb.nbr_fwd.cpu().numpy()) == want["nbr_checksum"]
        else:
            assert make_counts.table_checksum(plan[key]["out_indices"].cpu().numpy()) == want["out_indices_checksum"]
            assert make_counts.table_checksum(rb.nbr_fwd.cpu().numpy()) == want["o2i_checksum"]
            assert make_counts.table_checksum(rb.nbr_bwd.cpu().numpy()) == want["i2o_checksum"]
