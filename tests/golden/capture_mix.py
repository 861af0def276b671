#!/usr/bin/env python
"""Capture golden vectors from the REFERENCE's own data-path Python (build container only).

    python tests/golden/capture_mix.py
    python tests/golden/capture_mix.py --only-round4      (the fixtures added in round 4, the others untouched)
writes tests/golden/mix_*.npz (CutMix / PolarMix incl. use_pitch and swap_with_range / LaserMix cylindrical and spherical /
pseudo-box and pseudo-background mixes / MixUp), aug_world.npz (global flip / rotation / scaling),
gt_sampling.npz (DataBaseSampler), collate_batch.npz, data_processor.npz (range mask + shuffle), decode_bbox.npz
(decode_bbox_from_heatmap), consistency.npz (reverse_transform / get_consistency_loss), small_utils.npz
(PointFeatureEncoder, box / angle helpers).

The four processor files (inter_domain_point_{cutmix,polarmix,lasermix}.py, intra_domain_point_mixup.py)
are loaded by path under a stub package named `pcdet` (they use absolute `pcdet.…` imports,
SURVEY.md Appendix D step 4).  Their two compiled helpers cannot be built in this image
(`roiaware_pool3d_cuda.points_in_boxes_cpu` links CUDA launchers, `iou3d_nms_cuda.boxes_iou_bev_cpu`
includes <cuda.h>), so those two symbols are served by the oracle's C restatements
(oracle_points_in_boxes / oracle_boxes_iou_bev).  What the fixtures pin is therefore the reference's
Python logic: order of the random draws, masks, ordering of output rows, box bookkeeping.
Only inputs, parameters, the numpy seed and outputs are stored — no reference source.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import capture_reference as CR  # noqa: E402
from oracle import oracle as O  # noqa: E402
from toda_amd.pcdet.datasets.synthetic import synth_cloud  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
PC_RANGE = np.array([-54.0, -54.0, -5.0, 54.0, 54.0, 4.8], np.float32)


def setup():
    CR.ALIAS = "pcdet"
    CR.setup()
    cu = sys.modules["pcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda"]
    iu = sys.modules["pcdet.ops.iou3d_nms.iou3d_nms_cuda"]

    def points_in_boxes_cpu(boxes, pts, out):
        out.copy_(torch.from_numpy(O.points_in_boxes(pts.numpy(), boxes.numpy(), 0)))
        return 1

    def boxes_iou_bev_cpu(a, b, out):
        out.copy_(torch.from_numpy(O.boxes_iou_bev(a.numpy(), b.numpy())))
        return 1

    cu.points_in_boxes_cpu = points_in_boxes_cpu
    iu.boxes_iou_bev_cpu = boxes_iou_bev_cpu
    CR._load("pcdet.datasets.augmentor.augmentor_utils", "pcdet/datasets/augmentor/augmentor_utils.py")
    M = {"augmentor_utils": sys.modules["pcdet.datasets.augmentor.augmentor_utils"]}
    for name in ("inter_domain_point_cutmix", "inter_domain_point_polarmix", "inter_domain_point_lasermix",
                 "intra_domain_point_mixup", "inter_domain_point_pseudomix"):
        M[name] = CR._load(f"pcdet.datasets.processor.{name}", f"pcdet/datasets/processor/{name}.py")
    return M


def scene(kind, seed, n_points, n_boxes):
    """points [N, 4] fp32, gt_boxes [K, 8] fp32 (7 + class id), clipped to the TODA range like the data pipeline."""
    pts, boxes, _ = synth_cloud(kind, seed, n_points=n_points, n_boxes=n_boxes)
    pts = pts[:, :4].copy()
    keep = (np.abs(pts[:, 0]) < 54) & (np.abs(pts[:, 1]) < 54)
    cls = (1 + np.arange(len(boxes)) % 3).astype(np.float32)[:, None]
    return {"points": np.ascontiguousarray(pts[keep]), "gt_boxes": np.concatenate([boxes, cls], 1).astype(np.float32)}


def save(name, src, tgt, out, **params):
    np.savez_compressed(os.path.join(OUT, f"mix_{name}.npz"), src_points=src["points"], src_boxes=src["gt_boxes"],
                        tgt_points=tgt["points"], tgt_boxes=tgt["gt_boxes"], out_points=np.asarray(out["points"], np.float32),
                        out_boxes=np.asarray(out["gt_boxes"], np.float32), **params)
    print(name, src["points"].shape, tgt["points"].shape, "->", np.asarray(out["points"]).shape, np.asarray(out["gt_boxes"]).shape)


def copy(d):
    return {k: v.copy() for k, v in d.items()}


def cap_gt_sampling():
    """Reference DataBaseSampler (database_sampler.py:13-252) on a small object database cut (by the oracle's first-box
    membership, the points_in_boxes_gpu rule) out of three synthetic frames; two consecutive calls on one scene."""
    import pickle
    import tempfile
    from pathlib import Path
    ds = CR._load("pcdet.datasets.augmentor.database_sampler", "pcdet/datasets/augmentor/database_sampler.py")
    names_of = lambda k: np.array([f"cls{1 + i % 3}" for i in range(k)])
    tmp = Path(tempfile.mkdtemp())
    (tmp / "gt_database").mkdir()
    infos, packed, off = {}, [], 0
    for f in range(3):
        sc = scene("nuscenes_toda", 60 + f, 6000, 12)
        boxes, names = sc["gt_boxes"][:, :7], names_of(12)
        owner = O.points_in_boxes(sc["points"][:, :3].copy(), boxes.copy(), 2)
        first = np.where(owner.any(0), owner.argmax(0), -1)
        for i in range(len(boxes)):
            obj = sc["points"][first == i].copy()
            obj[:, :3] -= boxes[i, :3]
            rel = f"gt_database/{f}_{names[i]}_{i}.bin"
            obj.tofile(str(tmp / rel))
            infos.setdefault(names[i], []).append({"name": names[i], "path": rel, "image_idx": f, "gt_idx": i, "box3d_lidar": boxes[i],
                                                   "num_points_in_gt": len(obj), "difficulty": 0, "global_data_offset": [off, off + len(obj)]})
            packed.append(obj)
            off += len(obj)
    with open(tmp / "dbinfos.pkl", "wb") as fh:
        pickle.dump(infos, fh)
    cfg = CR.EasyDict({"DB_INFO_PATH": ["dbinfos.pkl"], "PREPARE": {"filter_by_min_points": ["cls1:5", "cls2:5", "cls3:1000"]},
                       "SAMPLE_GROUPS": ["cls1:5", "cls2:4", "cls3:2"], "NUM_POINT_FEATURES": 4, "REMOVE_EXTRA_WIDTH": [0.1, 0.1, 0.0],
                       "LIMIT_WHOLE_SCENE": True, "USE_SHARED_MEMORY": False})
    sampler = ds.DataBaseSampler(tmp, cfg, ["cls1", "cls2", "cls3"], logger=None)
    sc = scene("nuscenes_toda", 70, 8000, 6)
    out = {}
    np.random.seed(606)
    for call in range(2):
        d = sampler({"points": sc["points"].copy(), "gt_boxes": sc["gt_boxes"][:, :7].copy(), "gt_names": names_of(6),
                     "gt_boxes_mask": np.array([True, True, False, True, True, True])})
        out[f"points_{call}"], out[f"boxes_{call}"], out[f"names_{call}"] = d["points"], d["gt_boxes"], d["gt_names"].astype(str)
        print("gt_sampling call", call, sc["points"].shape, "->", d["points"].shape, d["gt_boxes"].shape)
    flat = [i for name in ("cls1", "cls2", "cls3") for i in infos[name]]
    np.savez_compressed(os.path.join(OUT, "gt_sampling.npz"), scene_points=sc["points"], scene_boxes=sc["gt_boxes"][:, :7], seed=606,
                        db_points=np.concatenate(packed, 0), db_offsets=np.array([i["global_data_offset"] for i in flat]),
                        db_boxes=np.stack([i["box3d_lidar"] for i in flat]), db_names=np.array([i["name"] for i in flat]),
                        db_frame=np.array([i["image_idx"] for i in flat]), db_gt_idx=np.array([i["gt_idx"] for i in flat]), **out)


def cap_collate():
    """Reference DatasetTemplate.collate_batch (pcdet/datasets/dataset.py:161-233) on two tiny voxelised samples."""
    for name, rel in [("pcdet.datasets.augmentor.database_sampler", "pcdet/datasets/augmentor/database_sampler.py"),
                      ("pcdet.datasets.augmentor.data_augmentor", "pcdet/datasets/augmentor/data_augmentor.py"),
                      ("pcdet.datasets.processor.point_feature_encoder", "pcdet/datasets/processor/point_feature_encoder.py"),
                      ("pcdet.datasets.processor.data_processor", "pcdet/datasets/processor/data_processor.py")]:
        if name not in sys.modules:
            CR._load(name, rel)
    dsm = CR._load("pcdet.datasets.dataset", "pcdet/datasets/dataset.py")
    rng = np.random.default_rng(808)
    samples = []
    for k, (m, n_pts, n_gt) in enumerate([(5, 11, 2), (3, 6, 4)]):
        samples.append({"points": rng.standard_normal((n_pts, 4)).astype(np.float32),
                        "voxels": rng.standard_normal((m, 5, 4)).astype(np.float32),
                        "voxel_coords": rng.integers(0, 40, (m, 3)).astype(np.int32),
                        "voxel_num_points": rng.integers(1, 6, (m,)).astype(np.int32),
                        "gt_boxes": rng.standard_normal((n_gt, 8)).astype(np.float32),
                        "frame_id": f"f{k}", "use_lead_xyz": True})
    out = dsm.DatasetTemplate.collate_batch([dict(s) for s in samples])
    flat = {}
    for k, s in enumerate(samples):
        for key, val in s.items():
            flat[f"in{k}_{key}"] = np.asarray(val)
    for key, val in out.items():
        flat[f"out_{key}"] = np.asarray(val)
    np.savez_compressed(os.path.join(OUT, "collate_batch.npz"), **flat)
    print("collate", {k: np.asarray(v).shape for k, v in out.items()})


def cap_data_processor():
    """Reference DataProcessor.mask_points_and_boxes_outside_range + shuffle_points (data_processor.py:78-103), seeded."""
    dp = sys.modules["pcdet.datasets.processor.data_processor"]
    cfgs = [CR.EasyDict({"NAME": "mask_points_and_boxes_outside_range", "REMOVE_OUTSIDE_BOXES": True}),
            CR.EasyDict({"NAME": "shuffle_points", "SHUFFLE_ENABLED": {"train": True, "test": False}})]
    rng_pc = np.array([-20.0, -20.0, -5.0, 20.0, 20.0, 4.8], np.float32)
    proc = dp.DataProcessor(cfgs, point_cloud_range=rng_pc, training=True, num_point_features=4)
    sc = scene("nuscenes_toda", 90, 4000, 14)
    sc["points"][:3, 0] = [20.0, -20.0, 20.0000019]          # on and just beyond the inclusive boundary
    np.random.seed(909)
    out = proc.forward({"points": sc["points"].copy(), "gt_boxes": sc["gt_boxes"].copy()})
    np.savez_compressed(os.path.join(OUT, "data_processor.npz"), in_points=sc["points"], in_boxes=sc["gt_boxes"], range=rng_pc, seed=909,
                        out_points=out["points"], out_boxes=out["gt_boxes"])
    print("data_processor", sc["points"].shape, "->", out["points"].shape, sc["gt_boxes"].shape, "->", out["gt_boxes"].shape)


def cap_decode():
    """Reference centernet_utils.decode_bbox_from_heatmap (centernet_utils.py:154-216) on random head outputs."""
    cu = sys.modules["pcdet.models.model_utils.centernet_utils"]
    g = torch.Generator().manual_seed(77)
    b, ncls, h, w = 2, 3, 24, 20
    heat = torch.rand((b, ncls, h, w), generator=g) ** 3
    ins = {"heatmap": heat, "rot_cos": torch.randn((b, 1, h, w), generator=g), "rot_sin": torch.randn((b, 1, h, w), generator=g),
           "center": torch.rand((b, 2, h, w), generator=g), "center_z": torch.randn((b, 1, h, w), generator=g),
           "dim": torch.rand((b, 3, h, w), generator=g) * 3 + 0.5}
    out = cu.decode_bbox_from_heatmap(point_cloud_range=[-9.6, -12.0, -5.0, 9.6, 12.0, 3.0], voxel_size=[0.1, 0.125, 0.2],
                                      feature_map_stride=8, K=40, circle_nms=False, score_thresh=0.2,
                                      post_center_limit_range=torch.tensor([-9.0, -11.0, -6.0, 9.0, 11.0, 4.0]), **ins)
    flat = {f"in_{k}": v.numpy() for k, v in ins.items()}
    for i, d in enumerate(out):
        for k, v in d.items():
            flat[f"out{i}_{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "decode_bbox.npz"), **flat)
    print("decode", [tuple(d["pred_boxes"].shape) for d in out])


def cap_consistency():
    """Reference reverse_transform / get_consistency_loss of the stage-2 step (pcdet/models/__init__.py:127-260).  The file
    imports `.detectors` (the whole detector zoo) - served by a stub with a dummy build_detector - and uses `F` without
    importing torch.nn.functional (a defect of the reference as shipped); the name is injected."""
    import types
    det = types.ModuleType("pcdet.models.detectors")
    det.build_detector = lambda **kw: None
    sys.modules["pcdet.models.detectors"] = det
    sys.modules["pcdet.models._ref_init.detectors"] = det       # the file is an __init__.py: it resolves `.detectors` under itself
    m = CR._load("pcdet.models._ref_init", "pcdet/models/__init__.py")
    m.F = torch.nn.functional
    g = torch.Generator().manual_seed(99)

    def boxes(n):
        b = torch.cat([torch.rand((n, 3), generator=g) * 20 - 10, torch.rand((n, 3), generator=g) * 3 + 1, torch.rand((n, 1), generator=g) * 6 - 3], 1)
        return b

    org = [boxes(9), boxes(0), boxes(6)]
    adv = [org[0][:7] + torch.randn((7, 7), generator=g) * 0.2, boxes(4), boxes(5)]
    aug_list = [["random_world_flip", "random_world_rotation", "random_world_scaling"], ["random_world_rotation"], ["gt_sampling", "random_world_flip"]]
    aug_params = [{"random_world_flip": ["x", "y"], "random_world_rotation": 0.3, "random_world_scaling": 1.04},
                  {"random_world_rotation": -0.2}, {"gt_sampling": None, "random_world_flip": ["y"]}]
    fwd = m.forward_transform([{"pred_boxes": b.clone()} for b in org], {"augmentation_list": aug_list, "augmentation_params": aug_params})
    back = m.reverse_transform([{"pred_boxes": d["pred_boxes"].clone()} for d in fwd], {"augmentation_list": aug_list, "augmentation_params": aug_params})
    closs, sloss = m.get_consistency_loss([{"pred_boxes": b.clone()} for b in adv], [{"pred_boxes": b.clone()} for b in org])
    flat = {"center_loss": closs.numpy(), "size_loss": sloss.numpy()}
    for i in range(3):
        flat[f"org{i}"], flat[f"adv{i}"] = org[i].numpy(), adv[i].numpy()
        flat[f"fwd{i}"], flat[f"back{i}"] = fwd[i]["pred_boxes"].numpy(), back[i]["pred_boxes"].numpy()
    np.savez_compressed(os.path.join(OUT, "consistency.npz"), **flat)
    print("consistency", float(closs), float(sloss))


def cap_small_utils():
    """PointFeatureEncoder (point_feature_encoder.py:4-61) and the box / angle helpers the data path leans on
    (box_utils.py:28-72,145-158, common_utils.py:24-57)."""
    pfe = sys.modules["pcdet.datasets.processor.point_feature_encoder"]
    bu = sys.modules["pcdet.utils.box_utils"]
    cu = sys.modules["pcdet.utils.common_utils"]
    rng = np.random.default_rng(4242)
    pts = rng.standard_normal((500, 5)).astype(np.float32)
    pts[:, 3] = rng.uniform(0, 255, 500)
    enc = pfe.PointFeatureEncoder(CR.EasyDict({"encoding_type": "absolute_coordinates_encoding", "used_feature_list": ["x", "y", "z", "intensity"],
                                               "src_feature_list": ["x", "y", "z", "intensity", "timestamp"], "normalize_intensity": True}),
                                  point_cloud_range=PC_RANGE)
    out = enc.forward({"points": pts.copy()})
    boxes = np.concatenate([rng.uniform(-30, 30, (40, 3)), rng.uniform(0.5, 6, (40, 3)), rng.uniform(-7, 7, (40, 1))], 1).astype(np.float32)
    limit = np.array([-20, -25, -3, 22, 18, 2.5], np.float32)
    np.savez_compressed(os.path.join(OUT, "small_utils.npz"), points=pts, enc_points=out["points"], enc_use_lead_xyz=out["use_lead_xyz"],
                        enc_num_features=enc.num_point_features, boxes=boxes, limit=limit,
                        corners=bu.boxes_to_corners_3d(boxes), mask_c1=bu.mask_boxes_outside_range_numpy(boxes, limit, 1),
                        mask_c8=bu.mask_boxes_outside_range_numpy(boxes, limit, 8), enlarged=bu.enlarge_box3d(boxes, (0.2, 0.3, 0.1)).numpy(),
                        limited_pi=cu.limit_period(boxes[:, 6], offset=0.5, period=np.pi), limited_2pi=cu.limit_period(boxes[:, 6], offset=0.5, period=2 * np.pi),
                        rotated=cu.rotate_points_along_z(pts[None, :50, :], np.array([0.77]))[0])
    print("small_utils", out["points"].shape)


def cap_round4(M):
    """The mixes round 4 added: pseudo-box / pseudo-background (inter_domain_point_pseudomix.py:19-68), PolarMix's swap with
    use_pitch (through the entry point) and swap_with_range (called directly: the entry point's own call of it raises a
    TypeError, inter_domain_point_polarmix.py:215-220), spherical LaserMix (through the entry point, which hands inc_method
    to `order`, and directly with order 0 / 1)."""
    pm, lm, ps = M["inter_domain_point_polarmix"], M["inter_domain_point_lasermix"], M["inter_domain_point_pseudomix"]
    src, tgt = scene("waymo_toda", 71, 6000, 14), scene("nuscenes_toda", 72, 5000, 12)
    # three target boxes that sit on source boxes (they must not be pasted) next to the target's own
    clash = src["gt_boxes"][[1, 4, 9]].copy()
    clash[:, 0:2] += np.float32(0.4)
    tgt_b = dict(tgt, gt_boxes=np.concatenate([tgt["gt_boxes"][:5], clash, tgt["gt_boxes"][5:]], 0))
    save("pseudobbox", src, tgt_b, ps.inter_domain_point_pseudobbox(copy(src), copy(tgt_b)))
    save("pseudobackground", src, tgt, ps.inter_domain_point_pseudobackground(copy(src), copy(tgt)))

    src, tgt = scene("waymo_toda", 21, 6000, 14), scene("nuscenes_toda", 22, 4000, 12)
    for name, seed, p in [("polarmix_pitch_center", 204, dict(rc=1, degree=1.570796, pct=0.3, methods=["FIX", "FIX"], inc="center")),
                          ("polarmix_pitch_corner_del", 205, dict(rc=2, degree=[0.8, 1.9], pct=0.6, methods=["RAND", "ASC", "DESC"], inc="corner_del"))]:
        np.random.seed(seed)
        out = pm.inter_domain_point_polarmix(copy(src), copy(tgt), p["rc"], p["degree"], p["pct"], p["methods"], PC_RANGE, "FULL", p["inc"], True)
        deg = np.atleast_1d(np.asarray(p["degree"], np.float64))
        save(name, src, tgt, out, seed=seed, rc=p["rc"], degree=deg, degree_is_float=isinstance(p["degree"], float),
             pct=p["pct"], methods=np.array(p["methods"]), inc=p["inc"])
    for name, seed, lo, hi in [("polar_range_near", 206, -0.7, 1.1), ("polar_range_far", 218, 1.9, 3.0)]:
        np.random.seed(seed)
        pts, boxes = pm.swap_with_range(src["points"].copy(), tgt["points"].copy(), lo, hi, src["gt_boxes"].copy(), tgt["gt_boxes"].copy(), PC_RANGE)
        save(name, src, tgt, {"points": pts, "gt_boxes": boxes}, seed=seed, lo=lo, hi=hi)

    src, tgt = scene("waymo_toda", 31, 6000, 14), scene("nuscenes_toda", 32, 5000, 12)
    np.random.seed(303)
    out = lm.inter_domain_point_lasermix(copy(src), copy(tgt), [-20, 0], [3, 4, 5, 6], None, PC_RANGE, "center")
    save("lasermix_sph_entry", src, tgt, out, seed=303, pitch=np.array([-20, 0]), num_areas=np.array([3, 4, 5, 6]))
    for name, seed, order, pitch, areas in [("lasermix_sph_order0", 304, 0, [-25, 3], [4, 5, 6]), ("lasermix_sph_order1", 305, 1, [-20, 0], [5])]:
        np.random.seed(seed)
        out = lm.laser_mix_transform_sph(copy(src), copy(tgt), pitch, areas, order)
        save(name, src, tgt, out, seed=seed, pitch=np.array(pitch), num_areas=np.array(areas), order=order)


def main():
    M = setup()
    if "--only-round4" in sys.argv:
        cap_round4(M)
        return
    cap_round4(M)
    # CutMix needs > 10 000 target points inside the crop (inter_domain_point_cutmix.py:57)
    src, tgt = scene("waymo_toda", 11, 14000, 12), scene("nuscenes_toda", 12, 26000, 10)
    np.random.seed(101)
    out = M["inter_domain_point_cutmix"].inter_domain_point_cutmix(copy(src), copy(tgt), PC_RANGE, "center")
    save("cutmix", src, tgt, out, seed=101)

    src, tgt = scene("waymo_toda", 21, 6000, 14), scene("nuscenes_toda", 22, 4000, 12)
    cases = [("polarmix_center", 201, dict(rc=1, degree=1.570796, pct=0.3, methods=["FIX", "FIX", "FIX"], inc="center")),
             ("polarmix_corner", 202, dict(rc=2, degree=[0.8, 1.9], pct=0.6, methods=["RAND", "ASC"], inc="corner")),
             ("polarmix_corner_del", 203, dict(rc=3, degree=[1.0, 2.5], pct=0.5, methods=["ASC_SIG", "DESC", "RAND"], inc="corner_del"))]
    for name, seed, p in cases:
        np.random.seed(seed)
        out = M["inter_domain_point_polarmix"].inter_domain_point_polarmix(
            copy(src), copy(tgt), p["rc"], p["degree"], p["pct"], p["methods"], PC_RANGE, "FULL", p["inc"], False)
        deg = np.atleast_1d(np.asarray(p["degree"], np.float64))
        save(name, src, tgt, out, seed=seed, rc=p["rc"], degree=deg, degree_is_float=isinstance(p["degree"], float),
             pct=p["pct"], methods=np.array(p["methods"]), inc=p["inc"])

    src, tgt = scene("waymo_toda", 31, 6000, 14), scene("nuscenes_toda", 32, 5000, 12)
    for name, seed, areas, angles, inc in [("lasermix_center", 301, 3, 2, "center"), ("lasermix_corner_del", 302, 2, 4, "corner_del")]:
        np.random.seed(seed)
        out = M["inter_domain_point_lasermix"].inter_domain_point_lasermix(copy(src), copy(tgt), None, areas, angles, PC_RANGE, inc)
        save(name, src, tgt, out, seed=seed, num_areas=areas, num_angles=angles, inc=inc)

    # global augmentations (augmentor_utils.py:8-81): flip x, flip y, rotation, scaling on one scene, seeded
    sc = scene("nuscenes_toda", 51, 3000, 10)
    au = M["augmentor_utils"]
    np.random.seed(505)
    boxes, pts = sc["gt_boxes"][:, :7].copy(), sc["points"].copy()
    stages = {}
    for tag, fn in [("flip_x", lambda b, p: au.random_flip_along_x(b, p)), ("flip_y", lambda b, p: au.random_flip_along_y(b, p)),
                    ("rot", lambda b, p: au.global_rotation(b, p, [-0.78539816, 0.78539816])),
                    ("scale", lambda b, p: au.global_scaling(b, p, [0.95, 1.05]))]:
        boxes, pts = fn(boxes, pts)
        stages[f"boxes_{tag}"], stages[f"points_{tag}"] = boxes.copy(), np.asarray(pts, np.float32).copy()
    np.savez_compressed(os.path.join(OUT, "aug_world.npz"), in_points=sc["points"], in_boxes=sc["gt_boxes"][:, :7], seed=505, **stages)
    print("aug_world", {k: v.shape for k, v in stages.items() if k.startswith("points")})

    cap_gt_sampling()
    cap_collate()
    cap_data_processor()
    cap_decode()
    cap_consistency()
    cap_small_utils()

    d1, d2 = scene("nuscenes_toda", 41, 5000, 12), scene("nuscenes_toda", 42, 4000, 40)
    for name, seed, fn in [("mixup", 401, "intra_domain_point_mixup"), ("mixup_cd", 402, "intra_domain_point_mixup_cd")]:
        np.random.seed(seed)
        out = getattr(M["intra_domain_point_mixup"], fn)(copy(d1), copy(d2), alpha=2)
        save(name, d1, d2, out, seed=seed, alpha=2)


if __name__ == "__main__":
    main()
