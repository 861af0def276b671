"""Eval / pseudo-label round trip on the MI355X: 3-D IoU through the C ABI against the oracle, the
recall record, and the CLI chain train -> checkpoint -> test -> generate_pseudo_labels -> stage-2 training on the
pseudo infos."""
import os
import pickle

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rand_boxes(seed, k, span=20.0):
    rng = np.random.default_rng(seed)
    return np.concatenate([rng.uniform(-span, span, (k, 2)), rng.uniform(-1, 1, (k, 1)), rng.uniform(1.5, 6, (k, 2)),
                           rng.uniform(1, 3, (k, 1)), rng.uniform(-np.pi, np.pi, (k, 1))], 1).astype(np.float32)


def test_overlap_area_and_iou3d_match_oracle():
    from toda_amd import ops
    from toda_amd.pcdet.ops.iou3d_nms import iou3d_nms_utils
    a, b = rand_boxes(1, 150), rand_boxes(2, 220)
    b[:40] = a[:40] + np.random.default_rng(3).normal(0, 0.3, (40, 7)).astype(np.float32)         # real overlaps
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    area = ops.boxes_overlap_bev(da, db).cpu().numpy()
    want = O.boxes_overlap_bev(a, b)
    assert (want > 0).sum() > 40
    np.testing.assert_allclose(area, want, rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(area == 0, want == 0)
    np.testing.assert_allclose(iou3d_nms_utils.boxes_iou3d_gpu(da, db).cpu().numpy(), O.boxes_iou3d(a, b), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(iou3d_nms_utils.boxes_bev_iou_cpu(a, b), O.boxes_iou_bev(a, b), rtol=1e-4, atol=1e-5)
    assert iou3d_nms_utils.boxes_bev_iou_cpu(a[:0], b).shape == (0, 220)


def test_recall_record_counts_recalled_ground_truth():
    from toda_amd.pcdet.models.detectors.detector3d_template import Detector3DTemplate
    gt = torch.from_numpy(rand_boxes(5, 12)).cuda()
    padded = torch.zeros((2, 15, 8), device="cuda")
    padded[0, :12, :7], padded[0, :12, 7] = gt, 1
    padded[1, :5, :7], padded[1, :5, 7] = gt[:5], 1
    preds = gt.clone()
    preds[6:, 0] += 100.0                                                      # half of the predictions are far away
    rec = Detector3DTemplate.generate_recall_record(preds, {}, 0, {"gt_boxes": padded}, [0.3, 0.5, 0.7])
    assert rec == {"gt": 12, "roi_0.3": 0, "rcnn_0.3": 6, "roi_0.5": 0, "rcnn_0.5": 6, "roi_0.7": 0, "rcnn_0.7": 6}
    rec = Detector3DTemplate.generate_recall_record(preds[:0], rec, 1, {"gt_boxes": padded}, [0.3, 0.5, 0.7])
    assert rec["gt"] == 17 and rec["rcnn_0.5"] == 6


def test_train_test_pseudo_label_stage2_round_trip(tmp_path):
    from toda_amd.tools import generate_pseudo_labels as gen
    from toda_amd.tools import test as tester
    from toda_amd.tools import train as trainer

    cfg_file = os.path.join(ROOT, "toda_amd/tools/cfgs/models/toda_stage1_centerpoint_res.yaml")
    small = ["DATA_CONFIG.SYNTHETIC.NUM_SAMPLES", "4", "DATA_CONFIG.SYNTHETIC.NUM_POINTS", "20000",
             "DATA_CONFIG.POINT_CLOUD_RANGE", "[-21.6,-21.6,-5.0,21.6,21.6,4.8]",
             "MODEL.DENSE_HEAD.POST_PROCESSING.SCORE_THRESH", "0.0"]
    out = str(tmp_path / "out")
    trainer.main(["--cfg_file", cfg_file, "--epochs", "1", "--batch_size", "2", "--output_dir", out, "--fix_random_seed", "--set"] + small)
    ckpts = sorted((tmp_path / "out").rglob("checkpoint_epoch_1.pth"))
    assert len(ckpts) == 1
    state = torch.load(ckpts[0], map_location="cpu")
    assert {"epoch", "it", "model_state", "optimizer_state", "version"} <= set(state) and state["it"] == 2
    assert state["model_state"]["backbone_3d.conv_input.0.weight"].shape == (16, 3, 3, 3, 4)      # spconv-2 layout

    ret = tester.main(["--cfg_file", cfg_file, "--ckpt", str(ckpts[0]), "--batch_size", "2", "--output_dir", out, "--set"] + small)
    assert "recall/rcnn_0.3" in ret and "car/recall_2m" in ret and 0.0 <= ret["recall/rcnn_0.3"] <= 1.0
    result = sorted((tmp_path / "out").rglob("result.pkl"))
    annos = pickle.load(open(result[0], "rb"))
    assert len(annos) == 4 and {"name", "score", "boxes_lidar", "pred_labels", "frame_id"} <= set(annos[0])
    assert [a["frame_id"] for a in annos] == [f"syn_{i:06d}" for i in range(4)]
    assert annos[0]["boxes_lidar"].shape[1] == 7 and len(annos[0]["score"]) == len(annos[0]["name"]) <= 83   # NMS_POST_MAXSIZE

    pseudo = gen.main(["--cfg_file", cfg_file, "--ckpt", str(ckpts[0]), "--pseudo_thresh", "0.05", "--batch_size", "2", "--output_dir", out,
                       "--set"] + small)
    infos = pickle.load(open(pseudo, "rb"))
    assert len(infos) == 4 and all(i["gt_boxes"].shape[1:] == (7,) or i["gt_boxes"].shape == (0, 7) for i in infos)
    kept = sum(len(i["gt_names"]) for i in infos)
    total = sum(int((a["score"] > 0.05).sum()) for a in annos)
    assert kept == total                                                      # same model, same frames, same threshold
    if kept == 0:
        pytest.skip("random-init model produced no detection above the threshold")
    # stage 2 trains on the pseudo labels
    trainer.main(["--cfg_file", cfg_file, "--epochs", "1", "--batch_size", "2", "--output_dir", str(tmp_path / "stage2"),
                  "--pretrained_model", str(ckpts[0]), "--set"] + small + ["DATA_CONFIG.PSEUDO_INFO_PATH", str(pseudo)])
    assert len(sorted((tmp_path / "stage2").rglob("checkpoint_epoch_1.pth"))) == 1


def test_stage2_perturb_labels_and_mixup_pair_training(tmp_path):
    """Teacher pass with stored voxel gradients -> SyntheticMixupPairDataset (MixUp-cd on the device, adversarial frames,
    recorded augmentations) -> the 2-forward / 1-backward consistency trainer."""
    from toda_amd.pcdet.config import AttrDict, cfg_from_yaml_file
    from toda_amd.pcdet.datasets import SyntheticMixupPairDataset
    from toda_amd.tools import generate_pseudo_labels_perturb as gen
    from toda_amd.tools import stage2_mixup_train_cl as stage2

    cfg_file = os.path.join(ROOT, "toda_amd/tools/cfgs/models/toda_stage2_mixup_cl.yaml")
    small = ["DATA_CONFIG.SYNTHETIC.NUM_SAMPLES", "4", "DATA_CONFIG.SYNTHETIC.NUM_GT", "2", "DATA_CONFIG.SYNTHETIC.NUM_POINTS", "20000",
             "DATA_CONFIG.POINT_CLOUD_RANGE", "[-21.6,-21.6,-5.0,21.6,21.6,4.8]", "MODEL.DENSE_HEAD.POST_PROCESSING.SCORE_THRESH", "0.0"]
    out = str(tmp_path / "out")
    # the teacher: a random-init model is enough to exercise the path (scores ~ sigmoid(-2.19) = 0.1)
    pseudo = gen.main(["--cfg_file", cfg_file, "--pseudo_thresh", "0.05", "--batch_size", "2", "--output_dir", out, "--set"] + small
                      + ["DATA_CONFIG.DATASET", "SyntheticLidarDataset"])
    infos = pickle.load(open(pseudo, "rb"))
    assert len(infos) == 4
    for info in infos:
        assert {"gt_boxes", "gt_names", "p_score", "p_voxel_perturb", "p_voxel_coords"} <= set(info)
        assert info["p_voxel_perturb"].shape == (info["p_voxel_coords"].shape[0], 3) and info["p_voxel_coords"].shape[0] > 1000
        assert len(info["p_score"]) == len(info["gt_boxes"]) and np.isfinite(info["p_voxel_perturb"]).all()
    assert any(np.abs(i["p_voxel_perturb"]).max() > 0 for i in infos)

    cfg = AttrDict()
    cfg_from_yaml_file(cfg_file, cfg)
    cfg.DATA_CONFIG.SYNTHETIC.NUM_SAMPLES, cfg.DATA_CONFIG.SYNTHETIC.NUM_GT, cfg.DATA_CONFIG.SYNTHETIC.NUM_POINTS = 4, 2, 20000
    cfg.DATA_CONFIG.POINT_CLOUD_RANGE = [-21.6, -21.6, -5.0, 21.6, 21.6, 4.8]
    cfg.DATA_CONFIG.PSEUDO_INFO_PATH = str(pseudo)
    cfg.DATA_CONFIG.PSEUDO_THRESH = 0.05
    ds = SyntheticMixupPairDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
    kinds = set()
    np.random.seed(11)
    for i in range(6):
        adv, org = ds[i]
        assert adv["points"].is_cuda and org["points"].is_cuda and adv["gt_boxes"].shape[1] == 8
        assert adv["augmentation_list"] == ["random_world_flip", "random_world_rotation", "random_world_scaling"]
        assert set(adv["augmentation_params"]) == set(adv["augmentation_list"])
        kinds.add("mixed" if adv["points"].shape[0] < 15000 or adv["gt_boxes"].shape[0] > 30 else "single")
    batch = ds.collate_batch([ds[0], ds[1]])
    assert batch[0]["points"].is_cuda and len(batch[0]["augmentation_params"]) == 2 and batch[1]["gt_boxes"].shape[0] == 2

    steps = stage2.main(["--cfg_file", cfg_file, "--epochs", "1", "--batch_size", "2", "--set"] + small
                        + ["DATA_CONFIG.PSEUDO_INFO_PATH", str(pseudo), "DATA_CONFIG.PSEUDO_THRESH", "0.05"])
    assert steps == 2


def test_stage1_trainer_runs_on_the_device_mix_dataset(tmp_path):
    """tools/stage1_cutmix_train (= train.py) on SyntheticMixDataset: samples arrive as CUDA tensors from __getitem__."""
    from toda_amd.tools import train as trainer
    cfg_file = os.path.join(ROOT, "toda_amd/tools/cfgs/models/toda_stage1_polarmix.yaml")
    small = ["DATA_CONFIG.SYNTHETIC.NUM_SOURCE", "2", "DATA_CONFIG.SYNTHETIC.NUM_TARGET", "2",
             "DATA_CONFIG.SYNTHETIC.NUM_POINTS_SOURCE", "30000", "DATA_CONFIG.SYNTHETIC.NUM_POINTS_TARGET", "12000",
             "DATA_CONFIG.POLARMIX_PROB", "1.0"]
    trainer.main(["--cfg_file", cfg_file, "--epochs", "1", "--batch_size", "2", "--output_dir", str(tmp_path / "out"), "--set"] + small)
    assert len(sorted((tmp_path / "out").rglob("checkpoint_epoch_1.pth"))) == 1


def test_training_with_gt_sampling_and_world_augmentations(tmp_path):
    """The reference's usual DATA_AUGMENTOR list (gt_sampling + flip + rotation + scaling) inside tools/train.py: build the
    object database with the CLI, then train an epoch with it."""
    import yaml
    from toda_amd.tools import create_gt_database as dbtool
    from toda_amd.tools import train as trainer

    base = os.path.join(ROOT, "toda_amd/tools/cfgs/models/toda_stage1_centerpoint_res.yaml")
    small = ["DATA_CONFIG.SYNTHETIC.NUM_SAMPLES", "4", "DATA_CONFIG.SYNTHETIC.NUM_POINTS", "20000",
             "DATA_CONFIG.POINT_CLOUD_RANGE", "[-21.6,-21.6,-5.0,21.6,21.6,4.8]"]
    # database from the same synthetic frames
    from toda_amd.pcdet.config import cfg, cfg_from_list, cfg_from_yaml_file
    dbtool.main(["--cfg_file", base, "--out", str(tmp_path / "db")])
    assert (tmp_path / "db" / "dbinfos.pkl").exists()
    # a config that adds the augmentor on top of the stage-1 model config
    aug_cfg = {"_BASE_CONFIG_": os.path.relpath(os.path.join(ROOT, "toda_amd/tools/cfgs/dataset_configs/synthetic_toda.yaml"), os.getcwd()),
               "DATA_PATH": str(tmp_path / "db"),
               "DATA_AUGMENTOR": {"DISABLE_AUG_LIST": ["placeholder"], "AUG_CONFIG_LIST": [
                   {"NAME": "gt_sampling", "DB_INFO_PATH": ["dbinfos.pkl"], "PREPARE": {"filter_by_min_points": ["car:5"]},
                    "SAMPLE_GROUPS": ["car:6"], "NUM_POINT_FEATURES": 4, "REMOVE_EXTRA_WIDTH": [0.0, 0.0, 0.0], "LIMIT_WHOLE_SCENE": False,
                    "USE_SHARED_MEMORY": True, "DB_DATA_PATH": ["gt_database_global.npy"]},
                   {"NAME": "random_world_flip", "ALONG_AXIS_LIST": ["x", "y"]},
                   {"NAME": "random_world_rotation", "WORLD_ROT_ANGLE": [-0.3925, 0.3925]},
                   {"NAME": "random_world_scaling", "WORLD_SCALE_RANGE": [0.95, 1.05]}]}}
    data_yaml = tmp_path / "data_aug.yaml"
    data_yaml.write_text(yaml.safe_dump(aug_cfg))
    model = yaml.safe_load(open(base))
    model["DATA_CONFIG"] = {"_BASE_CONFIG_": str(data_yaml)}
    model_yaml = tmp_path / "model_aug.yaml"
    model_yaml.write_text(yaml.safe_dump(model))
    trainer.main(["--cfg_file", str(model_yaml), "--epochs", "1", "--batch_size", "2", "--output_dir", str(tmp_path / "out"), "--set"] + small)
    assert len(sorted((tmp_path / "out").rglob("checkpoint_epoch_1.pth"))) == 1
    # the dataset really pastes objects: more boxes than the 30 of a plain frame is possible only with gt_sampling
    from toda_amd.pcdet.config import AttrDict
    from toda_amd.pcdet.datasets import SyntheticLidarDataset
    c = AttrDict()
    cfg_from_yaml_file(str(model_yaml), c)
    cfg_from_list(small, c)
    ds = SyntheticLidarDataset(c.DATA_CONFIG, c.CLASS_NAMES, training=True)
    np.random.seed(0)
    item = ds[0]
    assert item["augmentation_list"][-3:] == ["random_world_flip", "random_world_rotation", "random_world_scaling"] or "augmentation_list" in item


def test_stage2_consistency_helpers_on_the_gpu_match_the_cpu_path():
    """The GPU forms of the stage-2 helpers - candidates padded to K rows behind a mask instead of boolean-mask indexing, the
    world rotation with host-built scalars instead of a device tensor (no host synchronisation per sample) - against the CPU
    forms pinned by the reference fixture (tests/golden/consistency.npz): reverse_transform and both consistency terms, with
    unselected rows of arbitrary content, an empty sample and a sample with nothing selected on one side."""
    import os

    from toda_amd.pcdet import models as M

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "consistency.npz"))
    aug_list = [["random_world_flip", "random_world_rotation", "random_world_scaling"], ["random_world_rotation"], ["gt_sampling", "random_world_flip"]]
    aug_params = [{"random_world_flip": ["x", "y"], "random_world_rotation": 0.3, "random_world_scaling": 1.04},
                  {"random_world_rotation": -0.2}, {"gt_sampling": None, "random_world_flip": ["y"]}]
    meta = {"augmentation_list": aug_list, "augmentation_params": aug_params}
    fwd_cpu = [{"pred_boxes": torch.from_numpy(z[f"fwd{i}"].copy())} for i in range(3)]
    fwd_gpu = [{"pred_boxes": torch.from_numpy(z[f"fwd{i}"].copy()).cuda()} for i in range(3)]
    back_cpu, back_gpu = M.reverse_transform(fwd_cpu, meta), M.reverse_transform(fwd_gpu, meta)
    for c, g in zip(back_cpu, back_gpu):
        assert g["pred_boxes"].is_cuda and torch.allclose(g["pred_boxes"].cpu(), c["pred_boxes"], rtol=0, atol=2e-6)

    rng = np.random.default_rng(0)

    def padded(arr, k=24):
        n = len(arr)
        rows = rng.uniform(-50, 50, (k, arr.shape[1])).astype(np.float32)      # unselected rows: arbitrary content
        pos = np.sort(rng.choice(k, n, replace=False))
        rows[pos] = arr
        m = np.zeros(k, bool)
        m[pos] = True
        return {"pred_boxes": torch.from_numpy(rows).cuda(), "mask": torch.from_numpy(m).cuda()}

    adv = [z[f"adv{i}"] for i in range(3)]
    org = [z[f"org{i}"] for i in range(3)]
    ref = M.get_consistency_loss([{"pred_boxes": torch.from_numpy(a.copy())} for a in adv], [{"pred_boxes": torch.from_numpy(o.copy())} for o in org])
    got = M.get_consistency_loss([padded(a) for a in adv], [padded(o) for o in org])
    for r, g in zip(ref, got):
        assert abs(float(g) - float(r)) <= 1e-6 * max(1.0, abs(float(r))), (float(g), float(r))
    # nothing selected on one side of a sample: that sample contributes nothing, the normalisation still counts it
    adv2 = adv + [adv[0]]
    org2 = org + [np.zeros((0, 7), np.float32)]
    ref = M.get_consistency_loss([{"pred_boxes": torch.from_numpy(a.copy())} for a in adv2], [{"pred_boxes": torch.from_numpy(o.copy())} for o in org2])
    got = M.get_consistency_loss([padded(a) for a in adv2], [padded(o) for o in org2])
    for r, g in zip(ref, got):
        assert abs(float(g) - float(r)) <= 1e-6 * max(1.0, abs(float(r))), (float(g), float(r))
