#!/usr/bin/env python
"""Capture golden vectors from the REFERENCE's own torch-only modules (build container only).

    python tests/golden/capture_reference.py        # writes tests/golden/*.npz

The reference (rasd3/TODA, /root/reference) cannot be imported as a package (SharedArray, numba,
easydict, spconv absent; pcdet/datasets/__init__.py:39-40 is a SyntaxError), so each needed file is
loaded by path under an alias package with stub parents.  Nothing of the
reference is copied: only inputs, weights (state_dict), outputs, losses and gradients of tiny
cases are stored.  The GPU box never sees /root/reference; tests read the .npz files only.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
ALIAS = "refpcdet"


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in {**(d or {}), **kw}.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        elif isinstance(v, list):
            v = [EasyDict(x) if isinstance(x, dict) else x for x in v]
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    __setattr__ = __setitem__


def _pkg(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], leaf, m)
    return m


def _load(alias_name, rel_path):
    spec = importlib.util.spec_from_file_location(alias_name, os.path.join(REF, rel_path))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[alias_name] = mod
    parent, _, leaf = alias_name.rpartition(".")
    setattr(sys.modules[parent], leaf, mod)
    spec.loader.exec_module(mod)
    return mod


def setup():
    sys.modules["SharedArray"] = types.ModuleType("SharedArray")
    sk = types.ModuleType("skimage")
    sk.transform = types.ModuleType("skimage.transform")
    sys.modules["skimage"], sys.modules["skimage.transform"] = sk, sk.transform
    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (lambda f: f)
    sys.modules["numba"] = nb
    ed = types.ModuleType("easydict")
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed
    for name in [ALIAS, f"{ALIAS}.utils", f"{ALIAS}.ops", f"{ALIAS}.ops.iou3d_nms", f"{ALIAS}.ops.roiaware_pool3d",
                 f"{ALIAS}.models", f"{ALIAS}.models.model_utils", f"{ALIAS}.models.dense_heads",
                 f"{ALIAS}.models.dense_heads.target_assigner", f"{ALIAS}.models.backbones_3d",
                 f"{ALIAS}.models.backbones_3d.vfe", f"{ALIAS}.models.backbones_2d",
                 f"{ALIAS}.models.backbones_2d.map_to_bev", f"{ALIAS}.datasets", f"{ALIAS}.datasets.processor",
                 f"{ALIAS}.datasets.augmentor", "reftools", "reftools.optimization"]:
        _pkg(name)
    for stub in (f"{ALIAS}.ops.iou3d_nms.iou3d_nms_cuda", f"{ALIAS}.ops.roiaware_pool3d.roiaware_pool3d_cuda"):
        sys.modules[stub] = types.ModuleType(stub)
        parent, _, leaf = stub.rpartition(".")
        setattr(sys.modules[parent], leaf, sys.modules[stub])
    torch.Tensor.cuda = lambda self, *a, **k: self  # the heads hard-code .cuda()

    L = {}
    L["common_utils"] = _load(f"{ALIAS}.utils.common_utils", "pcdet/utils/common_utils.py")
    _load(f"{ALIAS}.ops.roiaware_pool3d.roiaware_pool3d_utils", "pcdet/ops/roiaware_pool3d/roiaware_pool3d_utils.py")
    _load(f"{ALIAS}.ops.iou3d_nms.iou3d_nms_utils", "pcdet/ops/iou3d_nms/iou3d_nms_utils.py")
    L["box_utils"] = _load(f"{ALIAS}.utils.box_utils", "pcdet/utils/box_utils.py")
    L["loss_utils"] = _load(f"{ALIAS}.utils.loss_utils", "pcdet/utils/loss_utils.py")
    L["box_coder_utils"] = _load(f"{ALIAS}.utils.box_coder_utils", "pcdet/utils/box_coder_utils.py")
    _load(f"{ALIAS}.models.model_utils.model_nms_utils", "pcdet/models/model_utils/model_nms_utils.py")
    L["centernet_utils"] = _load(f"{ALIAS}.models.model_utils.centernet_utils", "pcdet/models/model_utils/centernet_utils.py")
    L["center_head"] = _load(f"{ALIAS}.models.dense_heads.center_head", "pcdet/models/dense_heads/center_head.py")
    ta = f"{ALIAS}.models.dense_heads.target_assigner"
    L["anchor_generator"] = _load(f"{ta}.anchor_generator", "pcdet/models/dense_heads/target_assigner/anchor_generator.py")
    _load(f"{ta}.atss_target_assigner", "pcdet/models/dense_heads/target_assigner/atss_target_assigner.py")
    _load(f"{ta}.axis_aligned_target_assigner", "pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py")
    L["anchor_head_template"] = _load(f"{ALIAS}.models.dense_heads.anchor_head_template", "pcdet/models/dense_heads/anchor_head_template.py")
    L["anchor_head_single"] = _load(f"{ALIAS}.models.dense_heads.anchor_head_single", "pcdet/models/dense_heads/anchor_head_single.py")
    vfe = f"{ALIAS}.models.backbones_3d.vfe"
    _load(f"{vfe}.vfe_template", "pcdet/models/backbones_3d/vfe/vfe_template.py")
    L["mean_vfe"] = _load(f"{vfe}.mean_vfe", "pcdet/models/backbones_3d/vfe/mean_vfe.py")
    L["pillar_vfe"] = _load(f"{vfe}.pillar_vfe", "pcdet/models/backbones_3d/vfe/pillar_vfe.py")
    L["base_bev_backbone"] = _load(f"{ALIAS}.models.backbones_2d.base_bev_backbone", "pcdet/models/backbones_2d/base_bev_backbone.py")
    L["pointpillar_scatter"] = _load(f"{ALIAS}.models.backbones_2d.map_to_bev.pointpillar_scatter",
                                     "pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py")
    L["fastai_optim"] = _load("reftools.optimization.fastai_optim", "tools/train_utils/optimization/fastai_optim.py")
    L["schedules"] = _load("reftools.optimization.learning_schedules_fastai",
                           "tools/train_utils/optimization/learning_schedules_fastai.py")
    return L


def sd_np(module, prefix="w."):
    return {prefix + k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def cap_mean_vfe(L):
    rng = np.random.default_rng(0)
    vox = rng.standard_normal((40, 5, 5)).astype(np.float32)
    num = rng.integers(0, 6, 40).astype(np.float32)
    for v in range(40):
        vox[v, int(num[v]):] = 0
    m = L["mean_vfe"].MeanVFE(EasyDict(), 5)
    x = torch.from_numpy(vox).requires_grad_(True)
    out = m({"voxels": x, "voxel_num_points": torch.from_numpy(num)})["voxel_features"]
    g = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(torch.from_numpy(g))
    np.savez_compressed(os.path.join(OUT, "mean_vfe.npz"), voxels=vox, num=num, out=out.detach().numpy(), gout=g,
             gvoxels=x.grad.numpy())


BEV_CFG = dict(LAYER_NUMS=[1, 2], LAYER_STRIDES=[1, 2], NUM_FILTERS=[8, 16], UPSAMPLE_STRIDES=[1, 2],
               NUM_UPSAMPLE_FILTERS=[16, 16])


def cap_bev(L):
    np.int = int  # reference base_bev_backbone.py:60 uses the removed alias (only in the stride<1 branch)
    torch.manual_seed(1)
    m = L["base_bev_backbone"].BaseBEVBackbone(EasyDict(BEV_CFG), 12).train()
    x = torch.randn(2, 12, 16, 16, requires_grad=True)
    y = m({"spatial_features": x})["spatial_features_2d"]
    w0 = sd_np(m)  # state BEFORE the forward would differ in running stats; store the post-forward buffers apart
    g = torch.randn_like(y)
    y.backward(g)
    grads = {"g." + n: p.grad.numpy() for n, p in m.named_parameters()}
    torch.manual_seed(1)
    m0 = L["base_bev_backbone"].BaseBEVBackbone(EasyDict(BEV_CFG), 12)
    np.savez_compressed(os.path.join(OUT, "bev_backbone.npz"), x=x.detach().numpy(), y=y.detach().numpy(), gy=g.numpy(),
             gx=x.grad.numpy(), **sd_np(m0), **{"after." + k[2:]: v for k, v in w0.items() if "running" in k}, **grads)


# Wide variants: channel counts that are multiples of 32 and an even map width, so that the GPU twins of these
# fixtures run on the hand-written kernels (Winograd 3x3 convolutions, bn2d, the fused hidden layer of the head, the narrow output
# convolutions, the fused loss) instead of the library ones the 8 / 16-channel fixtures above fall back to.
BEV_WIDE_CFG = dict(LAYER_NUMS=[1, 2], LAYER_STRIDES=[1, 2], NUM_FILTERS=[32, 64], UPSAMPLE_STRIDES=[1, 2],
                    NUM_UPSAMPLE_FILTERS=[32, 32])


RELU_MARGIN = 6e-5


def relu_margin_probe(module):
    """Forward hooks on every nn.ReLU of a reference module: after a forward, probe() = the smallest |input| any of them saw.
    Why: the gradient of a network is discontinuous where a ReLU input crosses zero, so two CORRECT fp32 implementations whose
    activations differ by rounding (Winograd F(4x4,3x3): ~5e-6 of the output scale, times the BatchNorm's 1 / std) disagree by a
    whole gradient element wherever an input lies inside that distance of zero - with ~150 k ReLU inputs per fixture one does,
    more often than not (first capture of these fixtures: ONE flipped mask bit of 12 288 moved the upstream gradients by up to 9e-2
    of their maximum while every forward value agreed to 2e-6).  The wide fixtures therefore draw seeds until every ReLU input of
    the reference's run is at least RELU_MARGIN away from zero: a well-conditioned vector, not a looser tolerance."""
    seen = []

    def hook(mod, args):
        seen.append(float(args[0].detach().abs().min()))

    handles = [m.register_forward_pre_hook(hook) for m in module.modules() if isinstance(m, torch.nn.ReLU)]
    return (lambda: min(seen) if seen else float("inf")), (lambda: [h.remove() for h in handles])


def cap_bev_wide(L):
    np.int = int
    for seed in range(11, 11 + 20000):
        torch.manual_seed(seed)
        m = L["base_bev_backbone"].BaseBEVBackbone(EasyDict(BEV_WIDE_CFG), 32).train()
        w_init = sd_np(m)
        probe, done = relu_margin_probe(m)
        x = torch.randn(2, 32, 16, 24, requires_grad=True)
        y = m({"spatial_features": x})["spatial_features_2d"]
        done()
        if probe() >= RELU_MARGIN:
            break
    else:
        raise RuntimeError("no well-conditioned seed found")
    print(f"bev_backbone_wide: seed {seed}, smallest |ReLU input| {probe():.2e}")
    after = sd_np(m)
    g = torch.randn_like(y)
    y.backward(g)
    grads = {"g." + n: p.grad.numpy() for n, p in m.named_parameters()}
    np.savez_compressed(os.path.join(OUT, "bev_backbone_wide.npz"), x=x.detach().numpy(), y=y.detach().numpy(), gy=g.numpy(),
             gx=x.grad.numpy(), seed=np.int64(seed), relu_margin=np.float64(probe()), **w_init,
             **{"after." + k[2:]: v for k, v in after.items() if "running" in k}, **grads)


def cap_center_head_wide(L):
    CenterHead = L["center_head"].CenterHead
    pc_range = np.array([-6.4, -6.4, -2, 6.4, 6.4, 4], np.float32)
    vs = [0.1, 0.1, 0.15]
    cfg = EasyDict(HEAD_CFG)
    cfg.SHARED_CONV_CHANNEL = 64
    gt = make_gt(np.random.default_rng(12), 2, 9, -7.0, 7.0)
    for seed in range(13, 13 + 100000):
        torch.manual_seed(seed)
        head = CenterHead(cfg, 32, 3, CLASSES, np.array([128, 128, 40]), pc_range, vs, predict_boxes_when_training=False).train()
        w0 = sd_np(head)
        probe, done = relu_margin_probe(head)
        x = torch.randn(2, 32, 16, 16, requires_grad=True)
        data = {"spatial_features_2d": x, "gt_boxes": torch.from_numpy(gt.copy()), "batch_size": 2}
        with torch.no_grad():
            head.shared_conv(x)
            for h in head.heads_list:
                h(head.shared_conv(x))
        done()
        if probe() >= RELU_MARGIN:
            break
    else:
        raise RuntimeError("no well-conditioned seed found")
    print(f"center_head_wide: seed {seed}, smallest |ReLU input| {probe():.2e}")
    torch.manual_seed(seed)
    head = CenterHead(cfg, 32, 3, CLASSES, np.array([128, 128, 40]), pc_range, vs, predict_boxes_when_training=False).train()
    x = torch.randn(2, 32, 16, 16, requires_grad=True)
    data = {"spatial_features_2d": x, "gt_boxes": torch.from_numpy(gt.copy()), "batch_size": 2}
    head(data)
    td = head.forward_ret_dict["target_dicts"]
    preds = {k: v.detach().numpy().copy() for k, v in head.forward_ret_dict["pred_dicts"][0].items()}
    loss, tb = head.get_loss()
    loss.backward()
    after = sd_np(head)
    np.savez_compressed(os.path.join(OUT, "center_head_wide.npz"), x=x.detach().numpy(), gt=gt, pc_range=pc_range, voxel_size=np.array(vs),
             heatmap=td["heatmaps"][0].numpy(), target_boxes=td["target_boxes"][0].numpy(), inds=td["inds"][0].numpy(),
             masks=td["masks"][0].numpy(), loss=np.float32(loss.item()), hm_loss=np.float32(tb["hm_loss_head_0"]),
             loc_loss=np.float32(tb["loc_loss_head_0"]), gx=x.grad.numpy(), seed=np.int64(seed), relu_margin=np.float64(probe()),
             **w0, **{"pred." + k: v for k, v in preds.items()},
             **{"after." + k[2:]: v for k, v in after.items() if "running" in k},
             **{"g." + n: p.grad.numpy() for n, p in head.named_parameters() if p.grad is not None})


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
CLASSES = ["Vehicle", "Pedestrian", "Cyclist"]


def make_gt(rng, batch, n_max, lo, hi, n_cls=3):
    gt = np.zeros((batch, n_max, 8), np.float32)
    for b in range(batch):
        k = n_max - 3 * b
        gt[b, :k, 0:2] = rng.uniform(lo, hi, (k, 2))
        gt[b, :k, 2] = rng.uniform(-1, 2, k)
        gt[b, :k, 3:6] = rng.uniform(0.4, 6, (k, 3))
        gt[b, :k, 6] = rng.uniform(-3.14, 3.14, k)
        gt[b, :k, 7] = rng.integers(1, n_cls + 1, k)
    return gt


def cap_center_head(L):
    CenterHead = L["center_head"].CenterHead
    rng = np.random.default_rng(2)
    # (a) small map: whole head forward + targets + loss + backward
    pc_range = np.array([-6.4, -6.4, -2, 6.4, 6.4, 4], np.float32)
    vs = [0.1, 0.1, 0.15]
    torch.manual_seed(3)
    head = CenterHead(EasyDict(HEAD_CFG), 24, 3, CLASSES, np.array([128, 128, 40]), pc_range, vs,
                      predict_boxes_when_training=False).train()
    w0 = sd_np(head)
    x = torch.randn(2, 24, 16, 16, requires_grad=True)
    gt = make_gt(rng, 2, 9, -7.0, 7.0)
    gt[0, 2, 3] = 0.0  # degenerate box
    data = {"spatial_features_2d": x, "gt_boxes": torch.from_numpy(gt.copy()), "batch_size": 2}
    head(data)
    td = head.forward_ret_dict["target_dicts"]
    preds = {k: v.detach().numpy().copy() for k, v in head.forward_ret_dict["pred_dicts"][0].items()}
    loss, tb = head.get_loss()
    loss.backward()
    np.savez_compressed(os.path.join(OUT, "center_head.npz"), x=x.detach().numpy(), gt=gt, pc_range=pc_range, voxel_size=np.array(vs),
             heatmap=td["heatmaps"][0].numpy(), target_boxes=td["target_boxes"][0].numpy(), inds=td["inds"][0].numpy(),
             masks=td["masks"][0].numpy(), loss=np.float32(loss.item()), hm_loss=np.float32(tb["hm_loss_head_0"]),
             loc_loss=np.float32(tb["loc_loss_head_0"]), gx=x.grad.numpy(), **w0, **{"pred." + k: v for k, v in preds.items()},
             **{"g." + n: p.grad.numpy() for n, p in head.named_parameters() if p.grad is not None})
    # (b) target assignment alone at the Waymo geometry (188 x 188 map, stride 8), incl. boxes near the border
    pc_range = np.array([-75.2, -75.2, -2, 75.2, 75.2, 4], np.float32)
    head = CenterHead(EasyDict(HEAD_CFG), 24, 3, CLASSES, np.array([1504, 1504, 40]), pc_range, vs,
                      predict_boxes_when_training=False)
    gt = make_gt(rng, 3, 40, -78.0, 78.0)
    td = head.assign_targets(torch.from_numpy(gt.copy()), feature_map_size=(188, 188))
    np.savez_compressed(os.path.join(OUT, "center_assign_waymo.npz"), gt=gt, pc_range=pc_range, voxel_size=np.array(vs),
             heatmap=td["heatmaps"][0].numpy(), target_boxes=td["target_boxes"][0].numpy(), inds=td["inds"][0].numpy(),
             masks=td["masks"][0].numpy())
    # (c) two head groups (class remapping + per-head compaction)
    cfg = EasyDict(HEAD_CFG)
    cfg.CLASS_NAMES_EACH_HEAD = [["Vehicle"], ["Pedestrian", "Cyclist"]]
    head = CenterHead(cfg, 24, 3, CLASSES, np.array([1504, 1504, 40]), pc_range, vs, predict_boxes_when_training=False)
    gt = make_gt(rng, 2, 25, -70.0, 70.0)
    td = head.assign_targets(torch.from_numpy(gt.copy()), feature_map_size=(188, 188))
    np.savez_compressed(os.path.join(OUT, "center_assign_two_heads.npz"), gt=gt, pc_range=pc_range, voxel_size=np.array(vs),
             **{f"heatmap{i}": td["heatmaps"][i].numpy() for i in range(2)},
             **{f"target_boxes{i}": td["target_boxes"][i].numpy() for i in range(2)},
             **{f"inds{i}": td["inds"][i].numpy() for i in range(2)}, **{f"masks{i}": td["masks"][i].numpy() for i in range(2)})


def cap_optim(L):
    from functools import partial

    import torch.nn as nn

    OptimWrapper, OneCycle = L["fastai_optim"].OptimWrapper, L["schedules"].OneCycle
    torch.manual_seed(4)
    model = nn.Sequential(nn.Linear(6, 8), nn.BatchNorm1d(8), nn.ReLU(), nn.Linear(8, 3))
    w0 = sd_np(model)
    flatten = lambda m: sum(map(flatten, m.children()), []) if len(list(m.children())) else [m]  # noqa: E731
    opt = OptimWrapper.create(partial(torch.optim.Adam, betas=(0.9, 0.99)), 3e-3, [nn.Sequential(*flatten(model))], wd=0.01,
                              true_wd=True, bn_wd=True)
    sched = OneCycle(opt, 100, 1e-3, [0.95, 0.85], 10, 0.4)
    lrs, moms = [], []
    for it in range(100):
        sched.step(it)
        lrs.append(opt.lr)
        moms.append(opt.mom)
    x = torch.randn(16, 6)
    t = torch.randn(16, 3)
    sched2 = OneCycle(opt, 10, 3e-3, [0.95, 0.85], 10, 0.4)
    snaps = {}
    for it in range(3):
        sched2.step(it)
        opt.zero_grad()
        ((model(x) - t) ** 2).mean().backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10)
        opt.step()
        snaps.update({f"s{it}.{k}": v.detach().numpy().copy() for k, v in model.state_dict().items()})
    np.savez_compressed(os.path.join(OUT, "optim_onecycle.npz"), lrs=np.array(lrs), moms=np.array(moms), x=x.numpy(), t=t.numpy(),
             groups=np.array([len(g["params"]) for g in opt.opt.param_groups]), **w0, **snaps)


def cap_losses(L):
    lu = L["loss_utils"]
    rng = np.random.default_rng(5)
    logits = torch.from_numpy(rng.standard_normal((2, 50, 3)).astype(np.float32))
    onehot = torch.zeros(2, 50, 3)
    onehot[torch.arange(2)[:, None], torch.arange(50)[None], torch.from_numpy(rng.integers(0, 3, (2, 50)))] = 1.0
    onehot[:, ::3] = 0
    w = torch.from_numpy(rng.uniform(0, 1, (2, 50)).astype(np.float32))
    focal = lu.SigmoidFocalClassificationLoss(alpha=0.25, gamma=2.0)(logits, onehot, w)
    a = torch.from_numpy(rng.standard_normal((2, 50, 7)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal((2, 50, 7)).astype(np.float32))
    sl1 = lu.WeightedSmoothL1Loss(code_weights=[1, 1, 1, 1, 1, 1, 0.5])(a, b, w)
    d = torch.from_numpy(rng.standard_normal((2, 50, 2)).astype(np.float32))
    dt = torch.zeros(2, 50, 2)
    dt[..., 0] = 1
    ce = lu.WeightedCrossEntropyLoss()(d, dt, w)
    np.savez_compressed(os.path.join(OUT, "anchor_losses.npz"), logits=logits.numpy(), onehot=onehot.numpy(), w=w.numpy(),
             focal=focal.numpy(), a=a.numpy(), b=b.numpy(), sl1=sl1.numpy(), d=d.numpy(), dt=dt.numpy(), ce=ce.numpy())


C1_RANGE = [0, -3.84, -3, 7.68, 3.84, 1]
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


def cap_c1_chain(L):
    """BASELINE config 1 in miniature: PillarVFE -> PointPillarScatter -> BaseBEVBackbone -> AnchorHeadSingle
    (anchors, AxisAlignedTargetAssigner, cls / loc / dir losses, backward), CPU, bs 2, 48 x 48 pillars."""
    rng = np.random.default_rng(7)
    grid = np.array([48, 48, 1])
    pc_range = np.array(C1_RANGE, np.float32)
    # random pillars: unique (y, x), 1..8 points each inside the pillar
    vox, coords, num = [], [], []
    for b in range(2):
        cells = rng.choice(48 * 48, size=150 + 30 * b, replace=False)
        for c in cells:
            y, x = divmod(int(c), 48)
            n = int(rng.integers(1, 9))
            pts = np.zeros((8, 4), np.float32)
            pts[:n, 0] = pc_range[0] + (x + rng.uniform(0, 1, n)) * 0.16
            pts[:n, 1] = pc_range[1] + (y + rng.uniform(0, 1, n)) * 0.16
            pts[:n, 2] = rng.uniform(-2.5, 0.5, n)
            pts[:n, 3] = rng.uniform(0, 1, n)
            vox.append(pts)
            coords.append([b, 0, y, x])
            num.append(n)
    vox, coords, num = np.stack(vox), np.array(coords, np.float32), np.array(num, np.float32)
    gt = np.zeros((2, 6, 8), np.float32)
    sizes = {1: [3.9, 1.6, 1.56], 2: [0.8, 0.6, 1.73], 3: [1.76, 0.6, 1.73]}
    for b in range(2):
        for k in range(6 - 2 * b):
            cls = 1 + (k % 3)
            gt[b, k] = [rng.uniform(0.5, 7), rng.uniform(-3.5, 3.5), rng.uniform(-1.2, -0.6),
                        *(np.array(sizes[cls]) * rng.uniform(0.9, 1.1, 3)), rng.uniform(-3.14, 3.14), cls]
    torch.manual_seed(8)
    vfe = L["pillar_vfe"].PillarVFE(EasyDict(C1_VFE), 4, C1_VOXEL, pc_range).train()
    scatter = L["pointpillar_scatter"].PointPillarScatter(EasyDict(NUM_BEV_FEATURES=32), grid)
    bev = L["base_bev_backbone"].BaseBEVBackbone(EasyDict(C1_BEV), 32).train()
    head = L["anchor_head_single"].AnchorHeadSingle(EasyDict(C1_HEAD), 48, 3, ["Car", "Pedestrian", "Cyclist"], grid, pc_range,
                                                    predict_boxes_when_training=False).train()
    w = {**sd_np(vfe, "vfe."), **sd_np(bev, "bev."), **sd_np(head, "head.")}
    voxels = torch.from_numpy(vox).requires_grad_(True)
    d = {"voxels": voxels, "voxel_num_points": torch.from_numpy(num), "voxel_coords": torch.from_numpy(coords),
         "gt_boxes": torch.from_numpy(gt.copy()), "batch_size": 2}
    d = vfe(d)
    pillar = d["pillar_features"]
    d = head(bev(scatter(d)))
    loss, tb = head.get_loss()
    loss.backward()
    fr = head.forward_ret_dict
    np.savez_compressed(
        os.path.join(OUT, "c1_pointpillar_chain.npz"), voxels=vox, coords=coords, num=num, gt=gt, pc_range=pc_range,
        pillar_features=pillar.detach().numpy(), spatial_features_sum=d["spatial_features"].detach().sum(dim=(2, 3)).numpy(),
        cls_preds=fr["cls_preds"].detach().numpy(), box_preds=fr["box_preds"].detach().numpy(),
        dir_preds=fr["dir_cls_preds"].detach().numpy(), box_cls_labels=fr["box_cls_labels"].numpy(),
        box_reg_targets=fr["box_reg_targets"].numpy(), reg_weights=fr["reg_weights"].numpy(),
        anchors=torch.cat(head.anchors, dim=-3).numpy(), loss=np.float32(loss.item()),
        loss_cls=np.float32(tb["rpn_loss_cls"]), loss_loc=np.float32(tb["rpn_loss_loc"]), loss_dir=np.float32(tb["rpn_loss_dir"]),
        gvoxels=voxels.grad.numpy(), g_conv_cls=head.conv_cls.weight.grad.numpy(),
        g_pfn=vfe.pfn_layers[0].linear.weight.grad.numpy(), **w)


def main():
    L = setup()
    caps = {"mean_vfe": cap_mean_vfe, "bev": cap_bev, "center_head": cap_center_head, "optim": cap_optim, "losses": cap_losses,
            "c1_chain": cap_c1_chain, "bev_wide": cap_bev_wide, "center_head_wide": cap_center_head_wide}
    only = sys.argv[1:]          # e.g. `capture_reference.py bev_wide center_head_wide`: just these (the others stay as committed)
    for name, fn in caps.items():
        if not only or name in only:
            fn(L)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
