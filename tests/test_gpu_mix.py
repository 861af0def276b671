"""TODA mixing processors on the MI355X (csrc/points.hip through the C ABI) - bit-exact against
  (i) the golden vectors captured from the reference's own Python processors (tests/golden/mix_*.npz),
  (ii) the CPU oracle (oracle/mix.py, oracle_points_in_boxes) on seeded inputs up to the full C5 cloud sizes."""
import numpy as np
import pytest
import torch

from oracle import mix as OM
from oracle import oracle as O
from tests.test_oracle_mix import CASES, PC_RANGE, run_case

pytestmark = pytest.mark.gpu


def cloud(seed, n, c=4, span=50.0):
    rng = np.random.default_rng(seed)
    p = rng.uniform(-span, span, (n, c)).astype(np.float32)
    p[:, 2] = rng.uniform(-3, 3, n)
    return p


def boxes(seed, k, span=45.0):
    rng = np.random.default_rng(seed)
    b = np.concatenate([rng.uniform(-span, span, (k, 2)), rng.uniform(-1, 1, (k, 1)), rng.uniform(1.5, 6, (k, 2)),
                        rng.uniform(1, 3, (k, 1)), rng.uniform(-np.pi, np.pi, (k, 1)), rng.integers(1, 4, (k, 1))], 1)
    return b.astype(np.float32)


def dev(a):
    return torch.from_numpy(a).cuda()


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("n,k", [(50000, 40), (1000, 1), (7, 300), (180000, 64)])
def test_points_in_boxes_matches_oracle(mode, n, k):
    from toda_amd import ops
    p, b = cloud(1, n), boxes(2, k)
    p[: min(n, 4 * k), :3] = np.repeat(b[:, :3], 4, 0)[: min(n, 4 * k)] + np.random.default_rng(3).normal(0, 1.0, (min(n, 4 * k), 3)).astype(np.float32)
    want = O.points_in_boxes(p[:, :3].copy(), b[:, :7].copy(), mode).sum(0) != 0
    got = ops.points_in_boxes(dev(p), dev(b), mode).cpu().numpy()
    assert want.sum() > 0
    np.testing.assert_array_equal(got != 0, want)


def test_points_in_boxes_empty_inputs():
    from toda_amd import ops
    p = dev(cloud(1, 100))
    assert int(ops.points_in_boxes(p, torch.zeros((0, 7), device="cuda")).sum()) == 0
    assert ops.points_in_boxes(torch.zeros((0, 4), device="cuda"), dev(boxes(1, 3))).numel() == 0


def test_sector_rect_and_cell_flags_match_numpy_restatement():
    from toda_amd import ops
    p = cloud(5, 200000)
    p[:8, :2] = [[0, 0], [1, 0], [-1, 0], [0, 1], [0, -1], [-1, -0.0], [3, 3], [-3, 3]]      # axis / branch-cut cases
    yaw = OM.yaw32(p[:, 0], p[:, 1])
    d = dev(p)
    for lo, hi in [(-0.5, 1.2), (2.0, np.pi), (-np.pi, -2.5), (0.3, 0.3)]:
        want = (yaw > np.float32(lo)) & (yaw < np.float32(hi))
        np.testing.assert_array_equal(ops.points_sector(d, np.float32(lo), np.float32(hi)).cpu().numpy() != 0, want)
    lo, hi = np.array([-10.25, 3.5]), np.array([21.125, 30.0])
    want = (p[:, 0] > lo[0]) & (p[:, 0] < hi[0]) & (p[:, 1] > lo[1]) & (p[:, 1] < hi[1])
    np.testing.assert_array_equal(ops.points_rect(d, lo, hi, closed=False).cpu().numpy() != 0, want)
    p[:3, 0], p[:3, 1] = lo[0], hi[1]                                                         # exactly on the boundary
    d = dev(p)
    want = (p[:, 0] >= lo[0]) & (p[:, 0] <= hi[0]) & (p[:, 1] >= lo[1]) & (p[:, 1] <= hi[1])
    got = ops.points_rect(d, lo, hi, closed=True).cpu().numpy() != 0
    np.testing.assert_array_equal(got, want)
    assert got[:3].all() and not (ops.points_rect(d, lo, hi, closed=False).cpu().numpy()[:3] != 0).any()
    # LaserMix cells
    for phase, n_ang, n_dis in [(0.7, 2, 3), (-2.9, 4, 2), (3.1, 6, 5)]:
        yaw_e, dis_e = np.linspace(-np.pi, np.pi, n_ang + 1), np.linspace(0, np.float32(54.0), n_dis + 1)
        yw = OM._wrap(OM.yaw32(p[:, 0], p[:, 1]), phase)
        ds = OM._clip_range(p[:, 0], p[:, 1], np.float32(54.0))
        want = np.full(len(p), -1, np.int32)
        for i in range(n_ang):
            for j in range(n_dis):
                m = (yw > yaw_e[i]) & (yw <= yaw_e[i + 1]) & (ds > dis_e[j]) & (ds <= dis_e[j + 1])
                want[m] = i * n_dis + j
        got = ops.points_polar_cell(d, np.float32(phase), yaw_e, dis_e, np.float32(1e-05), np.float32(54.0) - np.float32(1e-05))
        np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_range_cut_elevation_test_and_elevation_bands_match_numpy_restatement():
    """toda_points_polar_select / _pitch_range / _pitch_band against the oracle's fp32 expressions on 200k points."""
    from toda_amd import ops
    p = cloud(15, 200000)
    p[:6, :3] = [[0, 0, 1], [1, 0, 0], [0.6, 0.8, -2], [0.6, 0.8000001, 2], [-30, 0.0, 1.8], [3, 4, -1.7]]   # range 0, exactly 1, just beyond 1
    d = dev(p)
    yaw, dis = OM.yaw32(p[:, 0], p[:, 1]), OM._range32(p[:, 0], p[:, 1])
    pitch = OM.pitch32(p[:, 2], dis)
    for lo, hi, th in [(-0.5, 1.2, 21.5), (2.0, np.pi, 40.25), (-np.pi, -2.5, 3.0)]:
        lo32, hi32, th32 = np.float32(lo), np.float32(hi), np.float32(th)
        inside = (yaw > lo32) & (yaw < hi32)
        np.testing.assert_array_equal(ops.points_polar_select(d, lo32, hi32, dis_mode=1, dis_th=th32).cpu().numpy() != 0, inside & (dis < th32))
        np.testing.assert_array_equal(ops.points_polar_select(d, lo32, hi32, dis_mode=2, dis_th=th32).cpu().numpy() != 0, inside & (dis > th32))
        np.testing.assert_array_equal(ops.points_polar_select(d, lo32, hi32).cpu().numpy() != 0, inside)
        np.testing.assert_array_equal(ops.points_polar_select(d, lo32, hi32, outside=True).cpu().numpy() != 0, (yaw < lo32) | (yaw > hi32))
    # elevation span of another cloud (beyond 1 m), read on the device by the select kernel
    q = cloud(16, 70001)
    q[:, 2] *= 0.05
    qd = OM._range32(q[:, 0], q[:, 1])
    qp = OM.pitch32(q[:, 2], qd)[qd > 1]
    span = ops.points_pitch_range(dev(q))
    np.testing.assert_array_equal(span.cpu().numpy(), np.array([qp.min(), qp.max()], np.float32))
    want = ((yaw < np.float32(-0.5)) | (yaw > np.float32(1.2))) & ((pitch < qp.min()) | (pitch > qp.max())) & (dis > 1)
    got = ops.points_polar_select(d, np.float32(-0.5), np.float32(1.2), outside=True, pitch_range=span).cpu().numpy() != 0
    assert want.sum() > 1000
    np.testing.assert_array_equal(got, want)
    # a device-side row count, and a cloud with nothing beyond 1 m
    n_dev = torch.tensor([12345], dtype=torch.int32, device="cuda")
    qp2 = OM.pitch32(q[:12345, 2], qd[:12345])[qd[:12345] > 1]
    np.testing.assert_array_equal(ops.points_pitch_range(dev(q), n_dev).cpu().numpy(), np.array([qp2.min(), qp2.max()], np.float32))
    near = ops.points_pitch_range(dev(np.zeros((10, 4), np.float32))).cpu().numpy()
    assert near[0] == np.inf and near[1] == -np.inf
    # spherical LaserMix bands (degrees in, radians compared in fp64; clip bounds in degrees as the reference has them)
    for pa, nb in [([-20, 0], 5), ([-25, 3], 6), ([-10, 10], 1)]:
        lo, hi = np.float32(pa[0] + 1e-5), np.float32(pa[1] - 1e-5)
        edges = np.linspace(pa[1], pa[0], nb + 1) / 180 * np.pi
        ev = np.clip(OM.pitch32(np.float32(-1.8) + p[:, 2], dis, sign=+1.0), lo, hi)
        want = np.full(len(p), -1, np.int32)
        for i in range(nb):
            want[(ev > edges[i + 1]) & (ev <= edges[i])] = i
        assert (want >= 0).sum() > 1000 and (want < 0).sum() > 1000
        np.testing.assert_array_equal(ops.points_pitch_band(d, np.float32(-1.8), lo, hi, edges).cpu().numpy(), want)


def test_select_append_is_a_stable_compaction_with_device_side_counts():
    from toda_amd import ops
    a, b = cloud(7, 70001, c=5), cloud(8, 33333, c=5)
    ka = (np.random.default_rng(9).integers(0, 4, len(a))).astype(np.int32)
    kb = (np.random.default_rng(10).integers(0, 2, len(b))).astype(np.int32)
    buf = ops.RowBuffer(len(a) + len(b), 5, "cuda")
    buf.append(dev(a), dev(ka), 2).append(dev(b), dev(kb), 1, invert=True).append(dev(a[:10]))
    want = np.concatenate([a[ka == 2], b[kb != 1], a[:10]], 0)
    got = buf.finish().cpu().numpy()
    np.testing.assert_array_equal(got, want)
    # chained: the second stage reads its row count from the first stage's cursor (no host sync in between)
    stage1 = ops.RowBuffer(len(a), 5, "cuda").append(dev(a), dev(ka), 0, invert=True)
    sel = a[ka != 0]
    flags = ops.points_sector(stage1.data, -1.0, 1.0, stage1.cursor)
    stage2 = ops.RowBuffer(len(a), 5, "cuda").append(stage1.data, flags, 1, n_dev=stage1.cursor)
    yaw = OM.yaw32(sel[:, 0], sel[:, 1])
    np.testing.assert_array_equal(stage2.finish().cpu().numpy(), sel[(yaw > -1.0) & (yaw < 1.0)])
    # nothing selected / everything selected / overflow is reported
    assert ops.RowBuffer(10, 5, "cuda").append(dev(a), dev(ka), 99).finish().shape[0] == 0
    with pytest.raises(RuntimeError, match="overflow"):
        ops.RowBuffer(10, 5, "cuda").append(dev(a)).finish()


def test_rotate_z_rounds_like_the_fp64_product():
    from toda_amd import ops
    p = cloud(11, 30000, c=5)
    om = 1.2345
    rot = np.array([[np.cos(om), np.sin(om), 0], [-np.sin(om), np.cos(om), 0], [0, 0, 1]])
    want = np.zeros_like(p)
    want[:, :3] = np.dot(p[:, :3], rot)
    want[:, 3] = p[:, 3]
    got = ops.points_rotate_z(dev(p), np.cos(om), np.sin(om)).cpu().numpy()
    np.testing.assert_array_equal(got[:, 2:], want[:, 2:])
    np.testing.assert_array_equal(got[:, :2], want[:, :2])


@pytest.mark.parametrize("name", CASES)
def test_device_mixers_reproduce_reference_outputs(name):
    from toda_amd.pcdet.datasets.processor import point_mix
    z, out = run_case(name, point_mix)
    np.testing.assert_array_equal(out["gt_boxes"], z["out_boxes"])
    assert out["points"].shape == z["out_points"].shape
    np.testing.assert_array_equal(out["points"], z["out_points"])


def full_scene(kind, seed, n_boxes):
    from toda_amd.pcdet.datasets.synthetic import synth_cloud
    pts, bx, _ = synth_cloud(kind, seed, n_boxes=n_boxes)
    cls = (1 + np.arange(len(bx)) % 3).astype(np.float32)[:, None]
    return {"points": np.ascontiguousarray(pts[:, :4]), "gt_boxes": np.concatenate([bx, cls], 1).astype(np.float32)}


@pytest.mark.parametrize("which", ["cutmix", "polarmix", "lasermix", "mixup_cd", "polarmix_pitch", "polarmix_rand", "lasermix_sph",
                                   "pseudobbox", "pseudobackground"])
def test_device_mixers_match_oracle_at_full_c5_size(which):
    """180k-point Waymo-shape source x 35k-point nuScenes-shape target (config C5), same seed on both sides."""
    from toda_amd.pcdet.datasets.processor import point_mix
    src, tgt = full_scene("waymo_toda", 501, 40), full_scene("nuscenes_toda", 502, 30)
    if which == "cutmix":
        tgt = full_scene("waymo_toda", 503, 30)                                    # needs > 10 000 target points in the crop
        args = lambda e, r: e.cutmix(src, tgt, PC_RANGE, rng=r)
    elif which == "polarmix":
        args = lambda e, r: e.polarmix(src, tgt, 2, [1.0, 2.0], 0.4, ["FIX", "RAND", "ASC_SIG"], "corner_del", rng=r)
    elif which == "lasermix":
        args = lambda e, r: e.lasermix_cyc(src, tgt, 3, 4, PC_RANGE, "corner_del", rng=r)
    elif which == "polarmix_pitch":
        args = lambda e, r: e.polarmix(src, tgt, 2, [1.0, 2.0], 0.4, ["FIX", "RAND", "ASC_SIG"], "corner_del", rng=r, use_pitch=True)
    elif which == "polarmix_rand":
        args = lambda e, r: e.polarmix(src, tgt, 1, [0.8, 1.6], 0.7, ["DESC", "RAND", "FIX"], "center", rng=r, polar_dis="RAND", pc_range=PC_RANGE)
    elif which == "lasermix_sph":
        args = lambda e, r: e.lasermix_sph(src, tgt, [-20, 0], [4, 5, 6], 1, rng=r)
    elif which == "pseudobbox":
        tgt = dict(tgt, gt_boxes=np.concatenate([tgt["gt_boxes"], src["gt_boxes"][:6] + np.float32([0.3, -0.2, 0, 0, 0, 0, 0.1, 0])], 0))
        args = lambda e, r: e.pseudobbox(src, tgt)
    elif which == "pseudobackground":
        args = lambda e, r: e.pseudobackground(src, tgt)
    else:
        args = lambda e, r: e.mixup(src, tgt, 2.0, collision=True, rng=r)
    want = args(OM, np.random.RandomState(77))
    got = args(point_mix, np.random.RandomState(77))
    np.testing.assert_array_equal(got["gt_boxes"], want["gt_boxes"])
    assert got["points"].shape == want["points"].shape and got["points"].shape[0] > (25000 if which == "lasermix_sph" else 30000)
    np.testing.assert_array_equal(got["points"], want["points"])


@pytest.mark.parametrize("empty", ["source", "target", "both"])
def test_round4_mixers_handle_scenes_without_boxes(empty):
    """No boxes on one or both sides (an unlabeled target frame before pseudo-labelling, a source frame whose objects were all
    filtered): pseudo mixes, spherical LaserMix, PolarMix with use_pitch / RAND range - device against the oracle."""
    from toda_amd.pcdet.datasets.processor import point_mix
    src, tgt = full_scene("nuscenes_toda", 701, 8), full_scene("nuscenes_toda", 702, 9)
    none = np.zeros((0, 8), np.float32)
    if empty in ("source", "both"):
        src = dict(src, gt_boxes=none)
    if empty in ("target", "both"):
        tgt = dict(tgt, gt_boxes=none)
    calls = [lambda e, r: e.pseudobbox(src, tgt), lambda e, r: e.pseudobackground(src, tgt),
             lambda e, r: e.lasermix_sph(src, tgt, [-20, 0], [5], 0, rng=r),
             lambda e, r: e.polarmix(src, tgt, 1, 1.2, 0.5, ["FIX", "RAND"], "center", rng=r, polar_dis="RAND", pc_range=PC_RANGE),
             lambda e, r: e.polarmix(src, tgt, 2, 1.2, 0.5, ["FIX"], "corner_del", rng=r, use_pitch=True)]
    for k, call in enumerate(calls):
        want, got = call(OM, np.random.RandomState(31 + k)), call(point_mix, np.random.RandomState(31 + k))
        np.testing.assert_array_equal(got["gt_boxes"], want["gt_boxes"])
        np.testing.assert_array_equal(got["points"], want["points"])


def test_mixers_keep_cuda_tensors_on_the_device_and_wrappers_are_drop_in():
    from toda_amd.pcdet.datasets.processor.inter_domain_point_polarmix import inter_domain_point_polarmix
    from toda_amd.pcdet.datasets.processor.intra_domain_point_mixup import intra_domain_point_mixup
    src, tgt = full_scene("nuscenes_toda", 601, 10), full_scene("nuscenes_toda", 602, 10)
    src_d = {"points": dev(src["points"]), "gt_boxes": src["gt_boxes"], "frame_id": "a"}
    tgt_d = {"points": dev(tgt["points"]), "gt_boxes": tgt["gt_boxes"], "frame_id": "b"}
    np.random.seed(3)
    out = inter_domain_point_polarmix(src_d, tgt_d, 1, 1.570796, 0.0, ["FIX", "FIX", "FIX"], PC_RANGE, "FULL", "center", False)
    assert out["points"].is_cuda and out["frame_id"] == "b" and out["gt_boxes"].shape[1] == 8
    np.random.seed(3)
    want = OM.polarmix(src, tgt, 1, 1.570796, 0.0, ["FIX", "FIX", "FIX"], "center")
    np.testing.assert_array_equal(out["points"].cpu().numpy(), want["points"])
    np.random.seed(4)
    out = intra_domain_point_mixup(dict(src, frame_id="a"), dict(tgt), alpha=2)
    assert isinstance(out["points"], np.ndarray) and out["frame_id"] == "a"
    # POLARMIX_DIS = RAND (the reference's call raises a TypeError; its swap_with_range is what runs here) and use_pitch
    np.random.seed(5)
    out = inter_domain_point_polarmix(src_d, tgt_d, 1, 1.5, 0.0, ["FIX"], PC_RANGE, "RAND", "center", False)
    np.random.seed(5)
    want = OM.polarmix(src, tgt, 1, 1.5, 0.0, ["FIX"], "center", polar_dis="RAND", pc_range=PC_RANGE)
    np.testing.assert_array_equal(out["points"].cpu().numpy(), want["points"])
    np.testing.assert_array_equal(out["gt_boxes"], want["gt_boxes"])
    np.random.seed(6)
    out = inter_domain_point_polarmix(src_d, tgt_d, 2, 1.5, 0.0, ["FIX", "FIX"], PC_RANGE, "FULL", "corner", True)
    np.random.seed(6)
    want = OM.polarmix(src, tgt, 2, 1.5, 0.0, ["FIX", "FIX"], "corner", use_pitch=True)
    np.testing.assert_array_equal(out["points"].cpu().numpy(), want["points"])
    with pytest.raises(NotImplementedError):
        inter_domain_point_polarmix(src_d, tgt_d, 1, 1.5, 0.0, ["FIX"], PC_RANGE, "HALF", "center", False)
    # the pseudo mixes write into the target dict and return it, like the reference; spherical LaserMix through its entry point
    from toda_amd.pcdet.datasets.processor.inter_domain_point_lasermix import inter_domain_point_lasermix
    from toda_amd.pcdet.datasets.processor.inter_domain_point_pseudomix import inter_domain_point_pseudobackground, inter_domain_point_pseudobbox
    t2 = dict(tgt_d)
    out = inter_domain_point_pseudobbox(src_d, t2)
    assert out is t2 and out["points"].is_cuda and out["frame_id"] == "b"
    want = OM.pseudobbox(src, tgt)
    np.testing.assert_array_equal(out["points"].cpu().numpy(), want["points"])
    np.testing.assert_array_equal(out["gt_boxes"], want["gt_boxes"])
    out = inter_domain_point_pseudobackground(src_d, dict(tgt_d))
    np.testing.assert_array_equal(out["points"].cpu().numpy(), OM.pseudobackground(src, tgt)["points"])
    np.random.seed(7)
    out = inter_domain_point_lasermix(src_d, tgt_d, [-20, 0], [5], None, PC_RANGE, "center")
    np.random.seed(7)
    want = OM.lasermix_sph(src, tgt, [-20, 0], [5], "center")
    assert out["points"].is_cuda and out["frame_id"] == "b"
    np.testing.assert_array_equal(out["points"].cpu().numpy(), want["points"])
    np.testing.assert_array_equal(out["gt_boxes"], want["gt_boxes"])


def test_mix_dataset_item_matches_the_host_pipeline():
    """SyntheticMixDataset.__getitem__ (encode -> PolarMix -> range mask -> shuffle, all on the device) against the same
    chain on the host: oracle mix + numpy mask + numpy permutation, same numpy seed."""
    import os
    from toda_amd.pcdet.config import AttrDict, cfg_from_yaml_file
    from toda_amd.pcdet.datasets import SyntheticMixDataset
    from toda_amd.pcdet.datasets.synthetic import synth_cloud

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = AttrDict()
    cfg_from_yaml_file(os.path.join(root, "toda_amd/tools/cfgs/models/toda_stage1_polarmix.yaml"), cfg)
    cfg.DATA_CONFIG.POLARMIX_PROB = 1.0
    cfg.DATA_CONFIG.MIX_INC_METHOD = "corner_del"
    cfg.DATA_CONFIG.POLARMIX_RC_NUM = 2
    ds = SyntheticMixDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
    index = 3
    np.random.seed(99)
    item = ds[index]
    assert item["points"].is_cuda and item["points"].shape[1] == 4

    np.random.seed(99)
    assert np.random.random(1) < 1.0
    frames = []
    for kind, idx in ((ds.source_kind, index % ds.num_source), (ds.target_kind, 100_000 + index % ds.num_target)):
        pts, bx, _ = synth_cloud(kind, ds.seed + idx, class_count=1)
        pts = pts[:, :4].copy()
        pts[:, 3:4] = pts[:, 3:4] / max(pts[:, 3:4].max(), 1e-12)                    # normalize_intensity
        frames.append({"points": pts, "gt_boxes": np.concatenate([bx, np.ones((len(bx), 1), np.float32)], 1)})
    mixed = OM.polarmix(frames[0], frames[1], 2, 1.570796, 0.0, ["FIX", "FIX", "FIX"], "corner_del")
    r = ds.point_cloud_range
    p = mixed["points"]
    p = p[(p[:, 0] >= r[0]) & (p[:, 0] <= r[3]) & (p[:, 1] >= r[1]) & (p[:, 1] <= r[4])]
    keep_b = OM.boxes_with_corners_in_range(mixed["gt_boxes"], r, 1)
    p = p[np.random.permutation(p.shape[0])]
    np.testing.assert_array_equal(item["gt_boxes"], mixed["gt_boxes"][keep_b])
    assert p.shape[0] > 50000
    np.testing.assert_array_equal(item["points"].cpu().numpy(), p)
    # and the collated batch feeds the voxeliser
    from toda_amd.pcdet.models import voxelize_on_gpu
    batch = ds.collate_batch([item, ds[index + 1]])
    assert batch["points"].is_cuda and batch["points"].shape[1] == 5 and batch["points_per_sample"][0] == p.shape[0]
    voxelize_on_gpu(batch, ds.voxel_cfg)
    assert batch["voxel_coords"].shape[0] > 20000 and int(batch["voxel_coords"][:, 0].max()) == 1


@pytest.mark.parametrize("mix_type", ["pseudobbox", "pseudobackground", "lasermix_sph"])
def test_mix_dataset_serves_the_pseudo_mixes_and_spherical_lasermix(mix_type):
    """MIX_TYPE pseudobbox / pseudobackground (tools/cfgs/stage1_pseudomix) and lasermix without LASERMIX_NUM_ANGLES
    (tools/cfgs/stage1_lasermix/*_pp01.yaml) through SyntheticMixDataset, against the oracle mix on the same frames."""
    import os
    from toda_amd.pcdet.config import AttrDict, cfg_from_yaml_file
    from toda_amd.pcdet.datasets import SyntheticMixDataset
    from toda_amd.pcdet.datasets.synthetic import synth_cloud

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = AttrDict()
    cfg_from_yaml_file(os.path.join(root, "toda_amd/tools/cfgs/models/toda_stage1_polarmix.yaml"), cfg)
    cfg.DATA_CONFIG.POLARMIX_PROB = 1.0
    if mix_type == "lasermix_sph":
        cfg.DATA_CONFIG.MIX_TYPE = "lasermix"
        cfg.DATA_CONFIG.pop("LASERMIX_NUM_ANGLES", None)
        cfg.DATA_CONFIG.LASERMIX_NUM_AREAS = [5]
        cfg.DATA_CONFIG.LASERMIX_PITCH_ANGLE = [-20, 0]
    else:
        cfg.DATA_CONFIG.MIX_TYPE = mix_type
    ds = SyntheticMixDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
    assert ds.mix_prob == 1.0 and (mix_type != "lasermix_sph" or ds.laser_num_angles is None)
    index = 5
    np.random.seed(123)
    item = ds[index]
    np.random.seed(123)
    assert np.random.random(1) < 1.0
    frames = []
    for kind, idx in ((ds.source_kind, index % ds.num_source), (ds.target_kind, 100_000 + index % ds.num_target)):
        pts, bx, _ = synth_cloud(kind, ds.seed + idx, class_count=1)
        pts = pts[:, :4].copy()
        pts[:, 3:4] = pts[:, 3:4] / max(pts[:, 3:4].max(), 1e-12)
        frames.append({"points": pts, "gt_boxes": np.concatenate([bx, np.ones((len(bx), 1), np.float32)], 1)})
    if mix_type == "lasermix_sph":
        mixed = OM.lasermix_sph(frames[0], frames[1], [-20, 0], [5], ds.mix_inc_method)
    else:
        mixed = getattr(OM, mix_type)(frames[0], frames[1])
    r = ds.point_cloud_range
    p = mixed["points"]
    p = p[(p[:, 0] >= r[0]) & (p[:, 0] <= r[3]) & (p[:, 1] >= r[1]) & (p[:, 1] <= r[4])]
    keep_b = OM.boxes_with_corners_in_range(mixed["gt_boxes"], r, 1)
    p = p[np.random.permutation(p.shape[0])]
    np.testing.assert_array_equal(item["gt_boxes"], mixed["gt_boxes"][keep_b])
    assert p.shape[0] > 10000 and item["points"].is_cuda
    np.testing.assert_array_equal(item["points"].cpu().numpy(), p)


def test_world_augmentations_on_the_device_match_reference():
    """Flips are exact; rotation / scaling agree with the reference's fp32 torch / numpy result to 1 ulp-level tolerance
    (the CPU matmul's summation order is not observable)."""
    from tests.test_eval_host import ROOT, _run_world_augs
    import os
    z = np.load(os.path.join(ROOT, "tests/golden/aug_world.npz"))
    z, out = _run_world_augs(dev(z["in_points"].copy()))
    for tag, (boxes, pts) in out.items():
        assert pts.is_cuda
        np.testing.assert_array_equal(boxes, z[f"boxes_{tag}"])
        got, want = pts.cpu().numpy(), z[f"points_{tag}"]
        if tag.startswith("flip"):
            np.testing.assert_array_equal(got, want)
        else:
            np.testing.assert_allclose(got, want, rtol=3e-7, atol=2e-6)
        np.testing.assert_array_equal(got[:, 3], z["in_points"][:, 3])


def _materialise_db(tmp_path, z, packed):
    import pickle
    infos = {}
    (tmp_path / "gt_database").mkdir(exist_ok=True)
    for k in range(len(z["db_names"])):
        lo, hi = (int(v) for v in z["db_offsets"][k])
        name = str(z["db_names"][k])
        rel = f"gt_database/{int(z['db_frame'][k])}_{name}_{int(z['db_gt_idx'][k])}.bin"
        z["db_points"][lo:hi].tofile(str(tmp_path / rel))
        infos.setdefault(name, []).append({"name": name, "path": rel, "image_idx": int(z["db_frame"][k]), "gt_idx": int(z["db_gt_idx"][k]),
                                           "box3d_lidar": z["db_boxes"][k], "num_points_in_gt": hi - lo, "difficulty": 0,
                                           "global_data_offset": [lo, hi]})
    with open(tmp_path / "dbinfos.pkl", "wb") as f:
        pickle.dump(infos, f)
    if packed:
        np.save(str(tmp_path / "gt_database_global.npy"), z["db_points"])


@pytest.mark.parametrize("variant", ["files_numpy", "packed_cuda"])
def test_gt_sampling_reproduces_reference_sampler(tmp_path, variant):
    """DataBaseSampler on the device against two consecutive calls of the reference's sampler (tests/golden/gt_sampling.npz):
    same picks, same collision filtering, same points in the same order."""
    import os
    from tests.test_eval_host import ROOT
    from toda_amd.pcdet.config import AttrDict
    from toda_amd.pcdet.datasets.augmentor.database_sampler import DataBaseSampler
    z = dict(np.load(os.path.join(ROOT, "tests/golden/gt_sampling.npz")))
    packed = variant == "packed_cuda"
    _materialise_db(tmp_path, z, packed)
    cfg = AttrDict({"DB_INFO_PATH": ["dbinfos.pkl"], "PREPARE": {"filter_by_min_points": ["cls1:5", "cls2:5", "cls3:1000"]},
                    "SAMPLE_GROUPS": ["cls1:5", "cls2:4", "cls3:2"], "NUM_POINT_FEATURES": 4, "REMOVE_EXTRA_WIDTH": [0.1, 0.1, 0.0],
                    "LIMIT_WHOLE_SCENE": True, "USE_SHARED_MEMORY": packed, "DB_DATA_PATH": ["gt_database_global.npy"]})
    sampler = DataBaseSampler(tmp_path, cfg, ["cls1", "cls2", "cls3"])
    names = np.array([f"cls{1 + i % 3}" for i in range(6)])
    np.random.seed(int(z["seed"]))
    for call in range(2):
        pts = z["scene_points"].copy()
        d = sampler({"points": dev(pts) if packed else pts, "gt_boxes": z["scene_boxes"].copy(), "gt_names": names.copy(),
                     "gt_boxes_mask": np.array([True, True, False, True, True, True])})
        got = d["points"].cpu().numpy() if packed else d["points"]
        assert (torch.is_tensor(d["points"]) and d["points"].is_cuda) == packed
        np.testing.assert_array_equal(d["gt_boxes"], z[f"boxes_{call}"])
        assert d["gt_names"].astype(str).tolist() == z[f"names_{call}"].tolist()
        np.testing.assert_array_equal(got, z[f"points_{call}"])
        assert "gt_boxes_mask" not in d


def test_groundtruth_database_cut_and_first_box_membership(tmp_path):
    """create_groundtruth_database on synthetic frames: every object's points are those whose FIRST containing box is the
    object's box (oracle membership, margin 1e-5), stored relative to the box centre; the packed array matches the files."""
    import pickle
    from toda_amd import ops
    from toda_amd.pcdet.datasets.augmentor.database_sampler import create_groundtruth_database
    from tests.test_eval_host import toda_cfg
    from toda_amd.pcdet.datasets import SyntheticLidarDataset
    cfg = toda_cfg(n_points=6000, samples=2)
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
    infos = create_groundtruth_database(ds, tmp_path, used_classes=["car"])
    assert set(infos) == {"car"} and len(infos["car"]) == 60
    packed = np.load(tmp_path / "gt_database_global.npy")
    assert pickle.load(open(tmp_path / "dbinfos.pkl", "rb"))["car"][3]["path"] == infos["car"][3]["path"]
    pts, boxes, _ = ds.raw_sample(1)
    owner = O.points_in_boxes(pts[:, :3].copy(), boxes[:, :7].copy(), 2)
    first = np.where(owner.any(0), owner.argmax(0), -1)
    np.testing.assert_array_equal(ops.points_in_boxes(dev(pts), dev(boxes), mode=2).cpu().numpy(), first)
    for info in infos["car"][30:40]:
        i = info["gt_idx"]
        want = pts[first == i].copy()
        want[:, :3] -= boxes[i, :3]
        got = np.fromfile(str(tmp_path / info["path"]), dtype=np.float32).reshape(-1, 4)
        np.testing.assert_array_equal(got, want)
        lo, hi = info["global_data_offset"]
        np.testing.assert_array_equal(packed[lo:hi], want)
        assert info["num_points_in_gt"] == len(want) > 0


def test_device_data_processor_matches_reference_fixture():
    """Range mask (toda_points_rect + compaction) and shuffle on a CUDA cloud against the reference DataProcessor's output."""
    import os
    from tests.test_eval_host import ROOT
    from toda_amd.pcdet.config import AttrDict
    from toda_amd.pcdet.datasets.processor.data_processor import DataProcessor
    z = np.load(os.path.join(ROOT, "tests/golden/data_processor.npz"))
    cfgs = [AttrDict({"NAME": "mask_points_and_boxes_outside_range", "REMOVE_OUTSIDE_BOXES": True}),
            AttrDict({"NAME": "shuffle_points", "SHUFFLE_ENABLED": {"train": True, "test": False}})]
    proc = DataProcessor(cfgs, point_cloud_range=z["range"], training=True, num_point_features=4)
    np.random.seed(int(z["seed"]))
    out = proc.forward({"points": dev(z["in_points"].copy()), "gt_boxes": z["in_boxes"].copy()})
    assert out["points"].is_cuda
    np.testing.assert_array_equal(out["gt_boxes"], z["out_boxes"])
    np.testing.assert_array_equal(out["points"].cpu().numpy(), z["out_points"])
