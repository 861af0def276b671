"""Round-4 input pipeline on the GPU against the CPU oracle, bit for bit: the three-launch batch voxeliser (toda_voxelize_batch),
the O(sites) voxel-level grid index, strided output sets derived from bitmaps (toda_gridindex_from_bitmap), the output-stationary
o2i table, the one-sync input plan (ops.build_input_plan) and the rotating arena slots the prefetcher builds it into.
Reference: pcdet/datasets/processor/data_processor.py:44-60,115-143 (voxelisation in the workers), pcdet/datasets/dataset.py:161-178
(collate_batch), pcdet/models/backbones_3d/spconv_backbone.py:77-125 (the indice_keys of VoxelBackBone8x)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import helpers as H
from tests.test_gpu_parity import dev, lidar_points

pytestmark = pytest.mark.gpu

STEPS = [
    {"kind": "subm", "key": "subm1", "ksize": [3, 3, 3], "dilation": [1, 1, 1]},
    {"kind": "conv", "key": "spconv2", "ksize": [3, 3, 3], "stride": [2, 2, 2], "padding": [1, 1, 1]},
    {"kind": "subm", "key": "subm2", "ksize": [3, 3, 3], "dilation": [1, 1, 1]},
    {"kind": "conv", "key": "spconv3", "ksize": [3, 3, 3], "stride": [2, 2, 2], "padding": [1, 1, 1]},
    {"kind": "subm", "key": "subm3", "ksize": [3, 3, 3], "dilation": [1, 1, 1]},
    {"kind": "conv", "key": "spconv4", "ksize": [3, 3, 3], "stride": [2, 2, 2], "padding": [0, 1, 1]},
    {"kind": "subm", "key": "subm4", "ksize": [3, 3, 3], "dilation": [1, 1, 1]},
    {"kind": "conv", "key": "spconv_down2", "ksize": [3, 1, 1], "stride": [2, 1, 1], "padding": [0, 0, 0]},
]


def oracle_batch(clouds, rng, vs, P, cap):
    vox, coords, nums = [], [], []
    for b, p in enumerate(clouds):
        v, c, n = O.voxelize_hard(p, rng, vs, P, cap)
        vox.append(v), nums.append(n)
        coords.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
    return np.concatenate(vox), np.concatenate(coords), np.concatenate(nums)


def check_plan(plan, idx, batch, shape, steps=STEPS):
    cur_idx, cur_shape = idx, shape
    for st in steps:
        e = plan[st["key"]]
        if st["kind"] == "subm":
            nbr0, cnt0 = O.rulebook_subm(cur_idx, batch, cur_shape)
            assert np.array_equal(e["rb"].nbr_fwd.cpu().numpy(), nbr0), st["key"]
            assert np.array_equal(e["rb"].pair_cnt.cpu().numpy(), cnt0), st["key"]
        else:
            io0, sho0, o2i0, i2o0, cnt0 = O.rulebook_conv(cur_idx, batch, cur_shape, st["ksize"], st["stride"], st["padding"])
            assert e["out_shape"] == sho0
            assert np.array_equal(e["out_indices"].cpu().numpy(), io0), st["key"]
            assert np.array_equal(e["rb"].nbr_fwd.cpu().numpy(), o2i0), st["key"]
            assert np.array_equal(e["rb"].nbr_bwd.cpu().numpy(), i2o0), st["key"]
            assert np.array_equal(e["rb"].pair_cnt.cpu().numpy(), cnt0), st["key"]
            cur_idx, cur_shape = io0, sho0


BATCH_CASES = [
    # sizes, c, range, voxel, P, cap
    ([5000, 7000, 0, 3000], 5, [-51.2, -51.2, -5, 51.2, 51.2, 3], [0.1, 0.1, 0.2], 10, 60000),
    ([30000, 9000], 5, [-75.2, -75.2, -2, 75.2, 75.2, 4], [0.1, 0.1, 0.15], 5, 3000),          # the cap bites in both samples
    ([20000, 100, 20000], 4, [0, -39.68, -3, 69.12, 39.68, 1], [0.16, 0.16, 4], 32, 16000),     # pillars: long per-cell lists
    ([4097, 1, 1023, 1025], 4, [0, 0, 0, 4, 4, 2], [1, 1, 1], 1, 3),                             # 32 cells in all, P = 1, block edges
    ([6000], 6, [0, 0, 0, 8, 8, 2], [0.5, 0.5, 1], 64, 100),                                     # P = 64, hundreds of points per cell
]


@pytest.mark.parametrize("sizes,c,rng,vs,P,cap", BATCH_CASES)
def test_voxelize_batch_bit_exact(sizes, c, rng, vs, P, cap):
    from toda_amd import ops

    clouds = [lidar_points(n, c, rng, seed=100 + 7 * b + n) for b, n in enumerate(sizes)]
    v0, c0, n0 = oracle_batch(clouds, rng, vs, P, cap)
    for _ in range(2):       # twice: same bits
        v1, c1, n1 = ops.voxelize_batch([dev(p) for p in clouds], rng, vs, P, cap)
        assert np.array_equal(c1.cpu().numpy(), c0)
        assert np.array_equal(n1.cpu().numpy(), n0)
        assert np.array_equal(v1.cpu().numpy(), v0)


def test_voxelize_batch_reads_a_collated_points_tensor_in_place():
    """[sum N, 1 + C] with the batch index in column 0 (collate_batch's `points`): column 1 on, rows 1 + C floats apart."""
    from toda_amd import ops

    rng, vs, P, cap = [-51.2, -51.2, -5, 51.2, 51.2, 3], [0.1, 0.1, 0.2], 10, 60000
    clouds = [lidar_points(4000 + 500 * b, 5, rng, seed=40 + b) for b in range(3)]
    v0, c0, n0 = oracle_batch(clouds, rng, vs, P, cap)
    flat = np.concatenate([np.concatenate([np.full((len(p), 1), b, np.float32), p], 1) for b, p in enumerate(clouds)])
    t = dev(flat)
    rows, start = [], 0
    for p in clouds:
        rows.append((t, start * 6 + 1, len(p), 5, 6))
        start += len(p)
    vox, coords, num, counts = ops.voxelize_enqueue(rows, rng, vs, P, cap)
    got = counts.cpu().numpy()
    assert got[-1] == len(c0) and list(got[:-1]) == [int((c0[:, 0] == b).sum()) for b in range(3)]
    m = int(got[-1])
    assert np.array_equal(coords[:m].cpu().numpy(), c0) and np.array_equal(num[:m].cpu().numpy(), n0)
    assert np.array_equal(vox[:m].cpu().numpy(), v0)


def test_voxeliser_leaves_its_table_clean_inside_an_arena_slot():
    """Consecutive batches of different sizes through ONE slot: the hash table is initialised once (ws_clean thereafter) and every
    result still equals the oracle; a later, larger batch re-lays the workspace out."""
    from toda_amd import arena, ops

    rng, vs, P, cap = [-51.2, -51.2, -5, 51.2, 51.2, 3], [0.1, 0.1, 0.2], 10, 60000
    slot = arena.ArenaSlot("cuda")
    grown = []
    for it, sizes in enumerate([[9000, 3000], [2000, 12000], [8000, 8000], [40000, 100]]):
        clouds = [lidar_points(n, 5, rng, seed=300 + 10 * it + b) for b, n in enumerate(sizes)]
        v0, c0, n0 = oracle_batch(clouds, rng, vs, P, cap)
        slot.reset()
        with arena.use_slot(slot):
            v1, c1, n1 = ops.voxelize_batch([dev(p) for p in clouds], rng, vs, P, cap)
        assert np.array_equal(c1.cpu().numpy(), c0) and np.array_equal(n1.cpu().numpy(), n0) and np.array_equal(v1.cpu().numpy(), v0)
        grown.append(slot.grown)
    assert grown[1] == grown[2] == grown[0]      # same layout: nothing allocated after the first use
    assert grown[3] == grown[0] + 1              # 40 k points per sample: one new workspace


def test_unordered_grid_index_and_bitmap_levels_on_ragged_grids():
    """Widths that are no multiple of 32 (bitmap words straddle rows), a lattice smaller than one word, batch > 1."""
    from toda_amd import ops

    for shape, batch, npb, seed in (([5, 9, 11], 3, 150, 3), ([7, 33, 65], 2, 900, 4), ([3, 3, 3], 1, 20, 5), ([41, 200, 176], 2, 20000, 21)):
        idx, _ = H.clustered_sparse(batch, shape, npb, 1, seed=seed)
        steps = [s for s in STEPS if s["key"] in ("subm1", "spconv2", "subm2", "spconv3", "subm3")] if min(shape) >= 5 else STEPS[:3]
        plan = ops.build_index_plan(dev(idx), batch, shape, steps)
        check_plan(plan, idx, batch, shape, steps)


def test_input_plan_one_sync_equals_the_oracle_chain():
    from toda_amd import ops

    rng, vs, P, cap = [0, -20, -2, 40, 20, 2], [0.1, 0.1, 0.1], 5, 20000
    shape = [41, 400, 400]
    clouds = [lidar_points(n, 5, rng, seed=500 + n) for n in (30000, 12000)]
    v0, c0, n0 = oracle_batch(clouds, rng, vs, P, cap)
    cfg = {"point_cloud_range": rng, "voxel_size": vs, "max_points_per_voxel": P, "max_num_voxels": cap}
    vox, coords, num, plan = ops.build_input_plan(ops._cloud_list([dev(p) for p in clouds]), cfg, 2, shape, STEPS, training=True)
    assert np.array_equal(coords.cpu().numpy(), c0) and np.array_equal(num.cpu().numpy(), n0) and np.array_equal(vox.cpu().numpy(), v0)
    check_plan(plan, c0, 2, shape)
    for key in ("spconv2", "spconv3", "spconv4"):      # the data gradients' class orders were built with the plan
        assert plan[key]["rb"]._class_order is not None


def test_arena_slots_rotate_without_growing_and_keep_the_voxel_bitmap_clean():
    """What the prefetcher does: plan after plan into rotating slots, different clouds every time.  Every plan equals the oracle,
    the slots stop allocating after their first use, and a slot's voxel-level bitmap is all zero again once it is re-acquired."""
    from toda_amd import arena, ops

    rng, vs, P, cap = [0, -20, -2, 40, 20, 2], [0.1, 0.1, 0.1], 5, 20000
    shape = [41, 400, 400]
    cfg = {"point_cloud_range": rng, "voxel_size": vs, "max_points_per_voxel": P, "max_num_voxels": cap}
    ar = arena.IndexArena("cuda", slots=2)
    stream = torch.cuda.current_stream()
    grown = []
    for it in range(6):
        clouds = [lidar_points(n, 5, rng, seed=700 + 13 * it + n) for n in (20000 + 1500 * (it % 3), 9000)]
        _, c0, _ = oracle_batch(clouds, rng, vs, P, cap)
        slot = ar.acquire(stream)
        if it >= 2:
            buf = slot.persist[("gi0", 0)][1]
            cells = 2 * shape[0] * shape[1] * shape[2] // 32
            assert int(buf[:8 * cells].view(torch.int32).abs().sum().item()) == 0, "voxel-level bitmap not clean on re-acquire"
        with arena.use_slot(slot):
            vox, coords, num, plan = ops.build_input_plan(ops._cloud_list([dev(p) for p in clouds]), cfg, 2, shape, STEPS, training=True)
        assert np.array_equal(coords.cpu().numpy(), c0)
        check_plan(plan, c0, 2, shape)
        ar.release(slot, stream)
        grown.append(ar.grown)
    assert grown[2:] == [grown[1]] * 4, grown


def test_prefetcher_hands_out_arena_backed_batches_and_the_detector_trains_on_them():
    """InputPrefetcher -> prepare_batch_on_gpu -> backbone.plan_input: the batch of every step comes with voxels and rulebooks from an
    arena slot; loss and gradients equal the path without the prefetcher (plain allocator, two-step voxelise + plan)."""
    from toda_amd.pcdet.config import AttrDict, cfg_from_yaml_file
    from toda_amd.pcdet.datasets import SyntheticLidarDataset
    from toda_amd.pcdet.models import InputPrefetcher, build_network, load_data_to_gpu, voxelize_on_gpu
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = AttrDict()
    cfg_from_yaml_file(os.path.join(root, "toda_amd/tools/cfgs/models/centerpoint_voxel_waymo.yaml"), cfg)
    cfg.DATA_CONFIG.SYNTHETIC.NUM_POINTS = 20000
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
    torch.manual_seed(3)
    net = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds).cuda()
    net.train()
    for m in net.modules():      # frozen statistics: the two routes see the batches in the same state
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    raw = [ds.collate_batch([ds[2 * k], ds[2 * k + 1]]) for k in range(3)]

    def losses(batches):
        out = []
        for b in batches:
            net.zero_grad()
            ret, _, _ = net(b)
            loss = ret["loss"].mean()
            loss.backward()
            g = torch.cat([p.grad.flatten() for p in net.parameters() if p.grad is not None])
            out.append((float(loss), g.clone()))
        return out

    def plain():
        for r in raw:
            b = dict(r)
            load_data_to_gpu(b)
            voxelize_on_gpu(b, ds.voxel_cfg)
            yield b

    ref = losses(plain())
    pre = InputPrefetcher(iter([dict(r) for r in raw]), net, torch.device("cuda", 0))
    got = []
    for _ in raw:
        b = pre.next()
        assert "sparse_index_plan" in b and b["voxels"].is_cuda
        pre.kick()
        got += losses([b])
    assert pre.arena is not None and pre.arena.grown > 0
    for (l0, g0), (l1, g1) in zip(ref, got):
        assert l0 == l1, (l0, l1)                      # same kernels on the same bits
        assert torch.equal(g0, g1)



def test_a_batch_kept_across_next_needs_keep_and_the_pipeline_survives_a_failed_preparation():
    """(a) tensors of batch t live in an arena slot that next() hands back - InputPrefetcher.keep() copies them out, and the copy
    still holds batch t's tables after two more batches went through the slots; (b) a preparation that raises gives its slot back and the
    original error reaches the caller instead of a ZeroDivisionError from the next acquire."""
    import bench
    from toda_amd.pcdet.datasets import SyntheticLidarDataset
    from toda_amd.pcdet.models import InputPrefetcher, build_network

    cfg = bench.load_cfg(bench.WORKLOADS["c2"][0])
    cfg.DATA_CONFIG.SYNTHETIC.NUM_POINTS = 20000
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
    dev_ = torch.device("cuda", 0)
    net = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds).to(dev_).eval()
    batches = bench.make_device_batches(ds, 1, 3, 0, dev_, host=True)

    def stream():
        for b in batches:
            yield dict(b)
        yield {"batch_size": 1, "points": torch.zeros(5, 3).pin_memory()}      # malformed: 3 columns, no batch index / features
        for b in batches:
            yield dict(b)

    with torch.no_grad(), InputPrefetcher(stream(), net, dev_, eager=False) as pre:
        first = pre.next()
        kept = InputPrefetcher.keep(first)
        ref_coords = first["voxel_coords"].clone()
        assert kept["voxel_coords"].data_ptr() != first["voxel_coords"].data_ptr()
        second = pre.next()
        third = pre.next()
        torch.cuda.synchronize()
        assert torch.equal(kept["voxel_coords"], ref_coords)                     # the copy is batch 1 whatever the slots hold by now
        assert not (second["voxel_coords"].shape == third["voxel_coords"].shape and torch.equal(second["voxel_coords"], third["voxel_coords"]))
        with pytest.raises(Exception) as err:
            pre.next()                                                           # the malformed batch
        assert "ZeroDivision" not in type(err.value).__name__
        again = pre.next()                                                       # the pipeline goes on with the slots it has
        torch.cuda.synchronize()
        assert torch.equal(again["voxel_coords"], ref_coords)
