"""Golden vectors captured from the reference's own torch-only modules (tests/golden/*.npz, made by
tests/golden/capture_reference.py in the build container) against (a) the CPU oracle and (b) this
repo's host-side mirror of the reference interface.  CPU-only; the GPU twins live in
tests/test_gpu_golden.py."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from oracle.cpu_backend import oracle_backend
from toda_amd.pcdet.config import AttrDict

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(G, name + ".npz")))


def load_weights(module, g, prefix="w."):
    sd = {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}
    missing, unexpected = module.load_state_dict(sd, strict=True)
    assert not missing and not unexpected


def test_mean_vfe_oracle_matches_reference():
    g = load("mean_vfe")
    np.testing.assert_allclose(O.mean_vfe_fwd(g["voxels"], g["num"]), g["out"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(O.mean_vfe_bwd(g["gout"], g["num"], 5), g["gvoxels"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("name", ["center_assign_waymo", "center_head"])
def test_center_assign_oracle_matches_reference(name):
    g = load(name)
    fm = g["heatmap"].shape[-1]
    hm, rb, inds, mask = O.center_assign(g["gt"], 3, fm, fm, g["pc_range"], g["voxel_size"], 8, 500, 0.1, 2)
    assert np.array_equal(inds, g["inds"])
    assert np.array_equal(mask, g["masks"])
    np.testing.assert_allclose(hm, g["heatmap"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(rb, g["target_boxes"], rtol=1e-6, atol=1e-6)


def test_center_assign_two_heads_oracle_matches_reference():
    g = load("center_assign_two_heads")
    for head, classes in enumerate([[1], [2, 3]]):
        gt = g["gt"].copy()
        lut = np.zeros(4, np.float32)
        for local, c in enumerate(classes):
            lut[c] = local + 1
        gt[..., 7] = lut[gt[..., 7].astype(int)]
        hm, rb, inds, mask = O.center_assign(gt, len(classes), 188, 188, g["pc_range"], g["voxel_size"], 8, 500, 0.1, 2)
        assert np.array_equal(inds, g[f"inds{head}"])
        assert np.array_equal(mask, g[f"masks{head}"])
        np.testing.assert_allclose(hm, g[f"heatmap{head}"], rtol=0, atol=1e-7)
        np.testing.assert_allclose(rb, g[f"target_boxes{head}"], rtol=1e-6, atol=1e-6)


BEV_CFG = dict(LAYER_NUMS=[1, 2], LAYER_STRIDES=[1, 2], NUM_FILTERS=[8, 16], UPSAMPLE_STRIDES=[1, 2],
               NUM_UPSAMPLE_FILTERS=[16, 16])


BEV_WIDE_CFG = dict(LAYER_NUMS=[1, 2], LAYER_STRIDES=[1, 2], NUM_FILTERS=[32, 64], UPSAMPLE_STRIDES=[1, 2],
                    NUM_UPSAMPLE_FILTERS=[32, 32])


def rel_err(got, want):
    """max |got - want| relative to max |want|: the scale-relative max-norm error the wide fixtures are judged by."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30))


def check_bev_backbone(device, tol=1.0, name="bev_backbone", cfg=None, cin=12):
    """BaseBEVBackbone forward / input-grad / weight-grads / running statistics vs the reference's module (tol scales the
    tolerances: 1 on the CPU; the GPU twin allows the library convolutions' different summation order)."""
    from toda_amd.pcdet.models.backbones_2d import BaseBEVBackbone

    g = load(name)
    m = BaseBEVBackbone(AttrDict(cfg or BEV_CFG), cin).train()
    load_weights(m, g)
    m = m.to(device)
    x = torch.from_numpy(g["x"]).to(device).requires_grad_(True)
    y = m({"spatial_features": x})["spatial_features_2d"]
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["y"], rtol=1e-5 * tol, atol=1e-5 * tol)
    y.backward(torch.from_numpy(g["gy"]).to(device))
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["gx"], rtol=1e-4 * tol, atol=1e-5 * tol)
    for n, p in m.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), g["g." + n], rtol=1e-4 * tol, atol=1e-4 * tol)
    for k, v in m.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), g["after." + k], rtol=1e-5 * tol, atol=1e-6 * tol)


def test_bev_backbone_matches_reference():
    check_bev_backbone("cpu")


def test_bev_backbone_wide_matches_reference():
    """32 / 64-channel neck on a 16 x 24 map (reference base_bev_backbone.py:81-112): the fixture whose GPU twin reaches the
    hand-written Winograd and bn2d kernels."""
    check_bev_backbone("cpu", tol=3.0, name="bev_backbone_wide", cfg=BEV_WIDE_CFG, cin=32)


def check_bev_backbone_wide_scaled(device, tol_fwd, tol_grad):
    """The wide fixture by scale-relative max-norm error (no per-element rtol / atol games): forward, input gradient, every
    parameter gradient, BN running statistics.  Returns the measured errors."""
    from toda_amd.pcdet.models.backbones_2d import BaseBEVBackbone

    g = load("bev_backbone_wide")
    m = BaseBEVBackbone(AttrDict(BEV_WIDE_CFG), 32).train()
    load_weights(m, g)
    m = m.to(device)
    x = torch.from_numpy(g["x"]).to(device).requires_grad_(True)
    y = m({"spatial_features": x})["spatial_features_2d"]
    errs = {"y": rel_err(y.detach().cpu().numpy(), g["y"])}
    y.backward(torch.from_numpy(g["gy"]).to(device))
    errs["gx"] = rel_err(x.grad.cpu().numpy(), g["gx"])
    for n, p in m.named_parameters():
        errs["g." + n] = rel_err(p.grad.cpu().numpy(), g["g." + n])
    for k, v in m.state_dict().items():
        if "running" in k:
            errs["after." + k] = rel_err(v.cpu().numpy(), g["after." + k])
    assert errs["y"] <= tol_fwd, errs
    bad = {k: v for k, v in errs.items() if k != "y" and v > (tol_fwd if k.startswith("after.") else tol_grad)}
    assert not bad, bad
    return errs


def test_bev_backbone_wide_scaled_error_on_cpu():
    check_bev_backbone_wide_scaled("cpu", 1e-5, 1e-4)


HEAD_CFG = dict(
    CLASS_AGNOSTIC=False, CLASS_NAMES_EACH_HEAD=[["Vehicle", "Pedestrian", "Cyclist"]], SHARED_CONV_CHANNEL=16,
    USE_BIAS_BEFORE_NORM=True, NUM_HM_CONV=2,
    SEPARATE_HEAD_CFG=dict(HEAD_ORDER=["center", "center_z", "dim", "rot"],
                           HEAD_DICT=dict(center=dict(out_channels=2, num_conv=2), center_z=dict(out_channels=1, num_conv=2),
                                          dim=dict(out_channels=3, num_conv=2), rot=dict(out_channels=2, num_conv=2))),
    TARGET_ASSIGNER_CONFIG=dict(FEATURE_MAP_STRIDE=8, NUM_MAX_OBJS=500, GAUSSIAN_OVERLAP=0.1, MIN_RADIUS=2),
    LOSS_CONFIG=dict(LOSS_WEIGHTS=dict(cls_weight=1.0, loc_weight=2.0, code_weights=[1.0] * 8)),
    POST_PROCESSING=dict(SCORE_THRESH=0.1, POST_CENTER_LIMIT_RANGE=[-75.2, -75.2, -2, 75.2, 75.2, 4], MAX_OBJ_PER_SAMPLE=500,
                         NMS_CONFIG=dict(NMS_TYPE="nms_gpu", NMS_THRESH=0.7, NMS_PRE_MAXSIZE=4096, NMS_POST_MAXSIZE=500)),
)


def build_head(g, shared=16, cin=24):
    from toda_amd.pcdet.models.dense_heads import CenterHead

    cfg = AttrDict(HEAD_CFG)
    cfg.SHARED_CONV_CHANNEL = shared
    head = CenterHead(cfg, cin, 3, ["Vehicle", "Pedestrian", "Cyclist"], np.array([128, 128, 40]),
                      g["pc_range"], list(g["voxel_size"]), predict_boxes_when_training=False).train()
    load_weights(head, g)
    return head


def check_center_head_wide(device, tol_fwd, tol_grad, backend=None):
    """center_head_wide.npz (SHARED_CONV_CHANNEL 64 on a 32-channel 16 x 16 map; reference center_head.py:11-45,73-80,221-272):
    predictions, targets, loss terms, input and parameter gradients, BN running statistics by scale-relative max-norm error."""
    import contextlib

    g = load("center_head_wide")
    head = build_head(g, shared=64, cin=32).to(device)
    x = torch.from_numpy(g["x"]).to(device).requires_grad_(True)
    with (backend() if backend else contextlib.nullcontext()):
        head({"spatial_features_2d": x, "gt_boxes": torch.from_numpy(g["gt"].copy()).to(device), "batch_size": 2})
    td = head.forward_ret_dict["target_dicts"]
    assert np.array_equal(td["inds"][0].cpu().numpy(), g["inds"]) and np.array_equal(td["masks"][0].cpu().numpy(), g["masks"])
    errs = {"heatmap": rel_err(td["heatmaps"][0].cpu().numpy(), g["heatmap"])}
    for k, v in head.forward_ret_dict["pred_dicts"][0].items():
        errs["pred." + k] = rel_err(v.detach().cpu().numpy(), g["pred." + k])
    loss, tb = head.get_loss()
    errs["loss"] = abs(float(loss) - float(g["loss"])) / max(1.0, abs(float(g["loss"])))
    errs["hm_loss"] = abs(float(tb["hm_loss_head_0"]) - float(g["hm_loss"])) / max(1.0, abs(float(g["hm_loss"])))
    errs["loc_loss"] = abs(float(tb["loc_loss_head_0"]) - float(g["loc_loss"])) / max(1.0, abs(float(g["loc_loss"])))
    loss.backward()
    errs["gx"] = rel_err(x.grad.cpu().numpy(), g["gx"])
    for n, p in head.named_parameters():
        if "g." + n not in g:
            continue
        want = g["g." + n]
        scale = np.abs(want).max()
        sibling = "g." + n[:-len("bias")] + "weight"
        if n.endswith(".bias") and sibling in g:
            # a conv bias in front of a train-mode BatchNorm (USE_BIAS_BEFORE_NORM) has a mathematically ZERO gradient: what the
            # reference stores there is its own rounding noise (1e-9), so such a bias is judged on the scale of its layer's weights
            scale = max(scale, np.abs(g[sibling]).max())
        errs["g." + n] = float(np.abs(p.grad.cpu().numpy().astype(np.float64) - want).max() / (scale + 1e-30))
    for k, v in head.state_dict().items():
        if "running" in k:
            errs["after." + k] = rel_err(v.cpu().numpy(), g["after." + k])
    fwd_keys = [k for k in errs if k.startswith(("pred.", "after.", "heatmap")) or k in ("loss", "hm_loss", "loc_loss")]
    bad = {k: v for k, v in errs.items() if v > (tol_fwd if k in fwd_keys else tol_grad)}
    assert not bad, bad
    return errs


def test_center_head_wide_matches_reference():
    check_center_head_wide("cpu", 2e-5, 2e-4, backend=oracle_backend)


def test_center_head_forward_targets_loss_backward_match_reference():
    g = load("center_head")
    head = build_head(g)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    with oracle_backend():  # target assignment through the CPU oracle (the product uses the HIP kernel)
        head({"spatial_features_2d": x, "gt_boxes": torch.from_numpy(g["gt"].copy()), "batch_size": 2})
    td = head.forward_ret_dict["target_dicts"]
    assert np.array_equal(td["inds"][0].numpy(), g["inds"])
    assert np.array_equal(td["masks"][0].numpy(), g["masks"])
    np.testing.assert_allclose(td["heatmaps"][0].numpy(), g["heatmap"], atol=1e-7)
    for k, v in head.forward_ret_dict["pred_dicts"][0].items():
        np.testing.assert_allclose(v.detach().numpy(), g["pred." + k], rtol=1e-4, atol=1e-5)
    loss, tb = head.get_loss()
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * max(1, float(g["loss"]))
    assert abs(float(tb["hm_loss_head_0"]) - float(g["hm_loss"])) < 1e-4 * max(1, float(g["hm_loss"]))
    assert abs(float(tb["loc_loss_head_0"]) - float(g["loc_loss"])) < 1e-4 * max(1, float(g["loc_loss"]))
    loss.backward()
    np.testing.assert_allclose(x.grad.numpy(), g["gx"], rtol=1e-3, atol=1e-5)
    for n, p in head.named_parameters():
        if "g." + n in g:
            np.testing.assert_allclose(p.grad.numpy(), g["g." + n], rtol=1e-3, atol=1e-4, err_msg=n)


def test_focal_loss_branch_free_form_equals_reference_branches():
    from toda_amd.pcdet.utils.loss_utils import neg_loss_cornernet

    pred = torch.rand(2, 3, 8, 8).clamp(1e-4, 1 - 1e-4)
    gt = torch.rand(2, 3, 8, 8) * 0.9
    ref_no_pos = -(torch.log(1 - pred) * pred ** 2 * (1 - gt) ** 4).sum()
    assert torch.allclose(neg_loss_cornernet(pred, gt), ref_no_pos)  # num_pos == 0 branch


def test_onecycle_and_decoupled_adam_match_reference():
    import torch.nn as nn
    from toda_amd.tools.train_utils.optimization import OneCycle, OneCycleAdam

    g = load("optim_onecycle")
    model = nn.Sequential(nn.Linear(6, 8), nn.BatchNorm1d(8), nn.ReLU(), nn.Linear(8, 3))
    load_weights(model, g)
    opt = OneCycleAdam(model, wd=0.01, fused=False)
    assert [len(gr["params"]) for gr in opt.param_groups] == list(g["groups"])
    sched = OneCycle(opt, 100, 1e-3, [0.95, 0.85], 10, 0.4)
    for it in range(100):
        sched.step(it)
        assert abs(opt.lr - g["lrs"][it]) < 1e-12 and abs(opt.mom - g["moms"][it]) < 1e-12
    sched2 = OneCycle(opt, 10, 3e-3, [0.95, 0.85], 10, 0.4)
    x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["t"])
    for it in range(3):
        sched2.step(it)
        opt.zero_grad()
        ((model(x) - t) ** 2).mean().backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10)
        opt.step()
        for k, v in model.state_dict().items():
            np.testing.assert_allclose(v.numpy(), g[f"s{it}.{k}"], rtol=1e-5, atol=1e-6, err_msg=f"step {it} {k}")


def check_anchor_losses(device):
    from toda_amd.pcdet.utils import loss_utils as lu

    g = load("anchor_losses")

    def T(a):
        return torch.from_numpy(a).to(device)

    focal = lu.SigmoidFocalClassificationLoss(alpha=0.25, gamma=2.0)(T(g["logits"]), T(g["onehot"]), T(g["w"]))
    np.testing.assert_allclose(focal.cpu().numpy(), g["focal"], rtol=1e-5, atol=1e-7)
    sl1 = lu.WeightedSmoothL1Loss(code_weights=[1, 1, 1, 1, 1, 1, 0.5])(T(g["a"]), T(g["b"]), T(g["w"]))
    np.testing.assert_allclose(sl1.cpu().numpy(), g["sl1"], rtol=1e-5, atol=1e-7)
    ce = lu.WeightedCrossEntropyLoss()(T(g["d"]), T(g["dt"]), T(g["w"]))
    np.testing.assert_allclose(ce.cpu().numpy(), g["ce"], rtol=1e-5, atol=1e-7)


def test_anchor_head_losses_match_reference():
    check_anchor_losses("cpu")


C1_VOXEL = [0.16, 0.16, 4]
C1_VFE = dict(WITH_DISTANCE=False, USE_ABSLOTE_XYZ=True, USE_NORM=True, NUM_FILTERS=[32])
C1_BEV = dict(LAYER_NUMS=[1, 1, 1], LAYER_STRIDES=[2, 2, 2], NUM_FILTERS=[16, 16, 32], UPSAMPLE_STRIDES=[1, 2, 4],
              NUM_UPSAMPLE_FILTERS=[16, 16, 16])
C1_HEAD = dict(
    CLASS_AGNOSTIC=False, USE_DIRECTION_CLASSIFIER=True, DIR_OFFSET=0.78539, DIR_LIMIT_OFFSET=0.0, NUM_DIR_BINS=2,
    ANCHOR_GENERATOR_CONFIG=[
        dict(class_name="Car", anchor_sizes=[[3.9, 1.6, 1.56]], anchor_rotations=[0, 1.57], anchor_bottom_heights=[-1.78],
             align_center=False, feature_map_stride=2, matched_threshold=0.6, unmatched_threshold=0.45),
        dict(class_name="Pedestrian", anchor_sizes=[[0.8, 0.6, 1.73]], anchor_rotations=[0, 1.57], anchor_bottom_heights=[-0.6],
             align_center=False, feature_map_stride=2, matched_threshold=0.5, unmatched_threshold=0.35),
        dict(class_name="Cyclist", anchor_sizes=[[1.76, 0.6, 1.73]], anchor_rotations=[0, 1.57], anchor_bottom_heights=[-0.6],
             align_center=False, feature_map_stride=2, matched_threshold=0.5, unmatched_threshold=0.35)],
    TARGET_ASSIGNER_CONFIG=dict(NAME="AxisAlignedTargetAssigner", POS_FRACTION=-1.0, SAMPLE_SIZE=512,
                                NORM_BY_NUM_EXAMPLES=False, MATCH_HEIGHT=False, BOX_CODER="ResidualCoder"),
    LOSS_CONFIG=dict(LOSS_WEIGHTS=dict(cls_weight=1.0, loc_weight=2.0, dir_weight=0.2, code_weights=[1.0] * 7)),
)


def check_c1_pointpillar_chain(device, tol=1.0):
    """BASELINE config 1: PillarVFE -> PointPillarScatter -> BaseBEVBackbone -> AnchorHeadSingle incl. anchors, target
    assignment, the three losses and backward, against the vectors captured from the reference's own modules.  On the
    CPU this is the "plumbing" configuration; on the GPU the same modules run with the HIP pillar scatter."""
    from toda_amd.pcdet.models.backbones_2d import BaseBEVBackbone
    from toda_amd.pcdet.models.backbones_2d.map_to_bev import PointPillarScatter
    from toda_amd.pcdet.models.backbones_3d.vfe import PillarVFE
    from toda_amd.pcdet.models.dense_heads import AnchorHeadSingle

    def N(t):
        return t.detach().cpu().numpy()

    g = load("c1_pointpillar_chain")
    grid = np.array([48, 48, 1])
    vfe = PillarVFE(AttrDict(C1_VFE), 4, C1_VOXEL, g["pc_range"]).train()
    scatter = PointPillarScatter(AttrDict(NUM_BEV_FEATURES=32), grid)
    bev = BaseBEVBackbone(AttrDict(C1_BEV), 32).train()
    head = AnchorHeadSingle(AttrDict(C1_HEAD), 48, 3, ["Car", "Pedestrian", "Cyclist"], grid, g["pc_range"],
                            predict_boxes_when_training=False).train()
    load_weights(vfe, g, "vfe.")
    load_weights(bev, g, "bev.")
    load_weights(head, g, "head.")
    vfe, scatter, bev, head = vfe.to(device), scatter.to(device), bev.to(device), head.to(device)
    if device != "cpu":
        head.anchors = [a.to(device) for a in head.anchors]
    np.testing.assert_allclose(N(torch.cat(head.anchors, dim=-3)), g["anchors"], rtol=0, atol=1e-6)
    voxels = torch.from_numpy(g["voxels"]).to(device).requires_grad_(True)
    d = {"voxels": voxels, "voxel_num_points": torch.from_numpy(g["num"]).to(device),
         "voxel_coords": torch.from_numpy(g["coords"]).to(device), "gt_boxes": torch.from_numpy(g["gt"].copy()).to(device),
         "batch_size": 2}
    d = vfe(d)
    np.testing.assert_allclose(N(d["pillar_features"]), g["pillar_features"], rtol=1e-5 * tol, atol=1e-6 * tol)
    d = scatter(d)
    np.testing.assert_allclose(N(d["spatial_features"].sum(dim=(2, 3))), g["spatial_features_sum"], rtol=1e-4, atol=1e-4)
    d = head(bev(d))
    fr = head.forward_ret_dict
    np.testing.assert_allclose(N(fr["cls_preds"]), g["cls_preds"], rtol=1e-4 * tol, atol=1e-5 * tol)
    np.testing.assert_allclose(N(fr["box_preds"]), g["box_preds"], rtol=1e-4 * tol, atol=1e-5 * tol)
    np.testing.assert_allclose(N(fr["dir_cls_preds"]), g["dir_preds"], rtol=1e-4 * tol, atol=1e-5 * tol)
    assert np.array_equal(N(fr["box_cls_labels"]), g["box_cls_labels"])
    np.testing.assert_allclose(N(fr["box_reg_targets"]), g["box_reg_targets"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(N(fr["reg_weights"]), g["reg_weights"], rtol=0, atol=0)
    loss, tb = head.get_loss()
    for mine, ref in ((loss, "loss"), (tb["rpn_loss_cls"], "loss_cls"), (tb["rpn_loss_loc"], "loss_loc"),
                      (tb["rpn_loss_dir"], "loss_dir")):
        assert abs(float(mine) - float(g[ref])) < 1e-4 * tol * max(1.0, abs(float(g[ref]))), ref
    loss.backward()
    np.testing.assert_allclose(N(voxels.grad), g["gvoxels"], rtol=1e-3 * tol, atol=1e-6 * tol)
    np.testing.assert_allclose(N(head.conv_cls.weight.grad), g["g_conv_cls"], rtol=1e-3 * tol, atol=1e-5 * tol)
    np.testing.assert_allclose(N(vfe.pfn_layers[0].linear.weight.grad), g["g_pfn"], rtol=1e-3 * tol, atol=1e-5 * tol)


def test_c1_pointpillar_chain_matches_reference():
    check_c1_pointpillar_chain("cpu")


def test_collate_batch_matches_reference():
    """DatasetTemplate.collate_batch against the reference's own (pcdet/datasets/dataset.py:161-233): concatenation order, the
    batch-index column of points / voxel_coords, zero-padded gt_boxes, dtypes."""
    from toda_amd.pcdet.datasets import DatasetTemplate

    z = np.load(os.path.join(G, "collate_batch.npz"), allow_pickle=False)
    samples = []
    for k in range(2):
        keys = [n[len(f"in{k}_"):] for n in z.files if n.startswith(f"in{k}_")]
        s = {key: z[f"in{k}_{key}"] for key in keys}
        s["frame_id"], s["use_lead_xyz"] = str(s["frame_id"]), bool(s["use_lead_xyz"])
        samples.append(s)
    out = DatasetTemplate.collate_batch(samples)
    for name in z.files:
        if not name.startswith("out_"):
            continue
        key = name[4:]
        want, got = z[name], np.asarray(out[key])
        assert got.shape == want.shape, key
        if want.dtype.kind in "fiub":
            np.testing.assert_array_equal(got, want, err_msg=key)
            if key in ("points", "voxels", "voxel_coords", "voxel_num_points", "gt_boxes"):
                assert got.dtype == want.dtype, (key, got.dtype, want.dtype)
        else:
            assert got.astype(str).tolist() == want.astype(str).tolist(), key
    assert out["points_per_sample"] == [11, 6]      # this build's extra key (sizes for the device voxeliser)


def test_data_processor_mask_and_shuffle_match_reference():
    """mask_points_and_boxes_outside_range (x / y only, inclusive ends, boxes by >= 1 corner in the 3-D range) and the seeded
    shuffle against the reference's DataProcessor (data_processor.py:78-103)."""
    from toda_amd.pcdet.config import AttrDict
    from toda_amd.pcdet.datasets.processor.data_processor import DataProcessor

    z = np.load(os.path.join(G, "data_processor.npz"))
    cfgs = [AttrDict({"NAME": "mask_points_and_boxes_outside_range", "REMOVE_OUTSIDE_BOXES": True}),
            AttrDict({"NAME": "shuffle_points", "SHUFFLE_ENABLED": {"train": True, "test": False}})]
    proc = DataProcessor(cfgs, point_cloud_range=z["range"], training=True, num_point_features=4)
    np.random.seed(int(z["seed"]))
    out = proc.forward({"points": z["in_points"].copy(), "gt_boxes": z["in_boxes"].copy()})
    np.testing.assert_array_equal(out["gt_boxes"], z["out_boxes"])
    np.testing.assert_array_equal(out["points"], z["out_points"])
    assert 0 < len(z["out_boxes"]) < len(z["in_boxes"]) and len(z["out_points"]) < len(z["in_points"])


def test_decode_bbox_from_heatmap_matches_reference():
    """Top-K decode of the CenterHead outputs (centernet_utils.py:154-216): boxes, scores, labels per sample."""
    from toda_amd.pcdet.models.model_utils import centernet_utils

    z = np.load(os.path.join(G, "decode_bbox.npz"))
    ins = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    out = centernet_utils.decode_bbox_from_heatmap(point_cloud_range=[-9.6, -12.0, -5.0, 9.6, 12.0, 3.0], voxel_size=[0.1, 0.125, 0.2],
                                                   feature_map_stride=8, K=40, circle_nms=False, score_thresh=0.2,
                                                   post_center_limit_range=torch.tensor([-9.0, -11.0, -6.0, 9.0, 11.0, 4.0]), **ins)
    assert len(out) == 2
    for i, d in enumerate(out):
        assert 0 < len(z[f"out{i}_pred_scores"]) < 40
        for key in ("pred_boxes", "pred_scores", "pred_labels"):
            np.testing.assert_array_equal(d[key].numpy(), z[f"out{i}_{key}"], err_msg=f"{i} {key}")


def test_consistency_helpers_match_reference():
    """reverse_transform (and the forward direction) and get_consistency_loss of the stage-2 step against the reference's
    functions (pcdet/models/__init__.py:127-260) on three samples: flips + rotation + scaling, an empty sample, gt_sampling in
    the augmentation list."""
    from toda_amd.pcdet import models as M

    z = np.load(os.path.join(G, "consistency.npz"))
    aug_list = [["random_world_flip", "random_world_rotation", "random_world_scaling"], ["random_world_rotation"], ["gt_sampling", "random_world_flip"]]
    aug_params = [{"random_world_flip": ["x", "y"], "random_world_rotation": 0.3, "random_world_scaling": 1.04},
                  {"random_world_rotation": -0.2}, {"gt_sampling": None, "random_world_flip": ["y"]}]
    meta = {"augmentation_list": aug_list, "augmentation_params": aug_params}
    fwd = [{"pred_boxes": torch.from_numpy(z[f"fwd{i}"].copy())} for i in range(3)]
    back = M.reverse_transform(fwd, meta)
    for i in range(3):
        np.testing.assert_array_equal(back[i]["pred_boxes"].numpy(), z[f"back{i}"], err_msg=f"sample {i}")
        np.testing.assert_allclose(z[f"back{i}"], z[f"org{i}"], atol=2e-5)        # and the round trip really is one
    adv = [{"pred_boxes": torch.from_numpy(z[f"adv{i}"].copy())} for i in range(3)]
    org = [{"pred_boxes": torch.from_numpy(z[f"org{i}"].copy())} for i in range(3)]
    closs, sloss = M.get_consistency_loss(adv, org)
    np.testing.assert_allclose(float(closs), float(z["center_loss"]), rtol=1e-6)
    np.testing.assert_allclose(float(sloss), float(z["size_loss"]), rtol=1e-6)


def test_feature_encoder_and_box_helpers_match_reference():
    from toda_amd.pcdet.config import AttrDict
    from toda_amd.pcdet.datasets.processor.point_feature_encoder import PointFeatureEncoder
    from toda_amd.pcdet.utils import box_utils, common_utils

    z = np.load(os.path.join(G, "small_utils.npz"))
    enc = PointFeatureEncoder(AttrDict({"encoding_type": "absolute_coordinates_encoding", "used_feature_list": ["x", "y", "z", "intensity"],
                                        "src_feature_list": ["x", "y", "z", "intensity", "timestamp"], "normalize_intensity": True}),
                              point_cloud_range=np.array([-54.0, -54.0, -5.0, 54.0, 54.0, 4.8], np.float32))
    out = enc.forward({"points": z["points"].copy()})
    assert enc.num_point_features == int(z["enc_num_features"]) and bool(out["use_lead_xyz"]) == bool(z["enc_use_lead_xyz"])
    np.testing.assert_array_equal(out["points"], z["enc_points"])
    boxes, limit = z["boxes"], z["limit"]
    np.testing.assert_array_equal(box_utils.boxes_to_corners_3d(boxes), z["corners"])
    np.testing.assert_array_equal(box_utils.mask_boxes_outside_range_numpy(boxes, limit, 1), z["mask_c1"])
    np.testing.assert_array_equal(box_utils.mask_boxes_outside_range_numpy(boxes, limit, 8), z["mask_c8"])
    assert 0 < z["mask_c1"].sum() < len(boxes) and z["mask_c8"].sum() <= z["mask_c1"].sum()
    np.testing.assert_array_equal(box_utils.enlarge_box3d(boxes, (0.2, 0.3, 0.1)).numpy(), z["enlarged"])
    np.testing.assert_array_equal(common_utils.limit_period(boxes[:, 6], offset=0.5, period=np.pi), z["limited_pi"])
    np.testing.assert_array_equal(common_utils.limit_period(boxes[:, 6], offset=0.5, period=2 * np.pi), z["limited_2pi"])
    np.testing.assert_array_equal(common_utils.rotate_points_along_z(z["points"][None, :50, :], np.array([0.77]))[0], z["rotated"])
