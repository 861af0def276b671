"""Host-side logic that needs neither the GPU nor the reference: config system, dataset contract,
collate, processors, checkpoint layout conversion, stage-2 consistency helpers."""
import os

import numpy as np
import pytest
import torch

from toda_amd.pcdet.config import AttrDict, cfg_from_list, cfg_from_yaml_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "toda_amd/tools/cfgs/models")


def load(name):
    cfg = AttrDict()
    cfg_from_yaml_file(os.path.join(CFG, name + ".yaml"), cfg)
    return cfg


@pytest.mark.parametrize("name,grid,p,cap,c", [
    ("pointpillar_kitti", [432, 496, 1], 32, 16000, 4),
    ("second_backbone_nuscenes", [1024, 1024, 40], 10, 60000, 5),
    ("centerpoint_voxel_waymo", [1504, 1504, 40], 5, 150000, 5),
    ("toda_stage1_centerpoint_res", [1440, 1440, 49], 10, 120000, 4),
])
def test_baseline_configs_resolve_to_the_survey_geometry(name, grid, p, cap, c):
    from toda_amd.pcdet.datasets import SyntheticLidarDataset

    cfg = load(name)
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES)
    assert list(ds.grid_size) == grid
    assert ds.voxel_cfg["max_points_per_voxel"] == p and ds.voxel_cfg["max_num_voxels"] == cap
    assert ds.point_feature_encoder.num_point_features == c


def test_config_set_overrides_and_base_include():
    cfg = load("centerpoint_voxel_waymo")
    assert cfg.DATA_CONFIG.DATASET == "SyntheticLidarDataset"  # came through _BASE_CONFIG_
    cfg_from_list(["OPTIMIZATION.LR", "0.01", "MODEL.BACKBONE_2D.LAYER_NUMS", "[3,3]", "OPTIMIZATION.LR_WARMUP", "True"], cfg)
    assert cfg.OPTIMIZATION.LR == 0.01 and cfg.MODEL.BACKBONE_2D.LAYER_NUMS == [3, 3] and cfg.OPTIMIZATION.LR_WARMUP is True
    with pytest.raises(AssertionError):
        cfg_from_list(["MODEL.NOPE", "1"], cfg)
    assert cfg.MODEL.get("ROI_HEAD", None) is None


def test_parameter_counts_match_survey_a4():
    from toda_amd.pcdet.datasets import SyntheticLidarDataset
    from toda_amd.pcdet.models import build_network

    for name, expect in (("centerpoint_voxel_waymo", 5775803), ("toda_stage1_centerpoint_res", 7757225)):
        cfg = load(name)
        ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES)
        model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds)
        assert sum(p.numel() for p in model.parameters()) == expect
        keys = set(model.state_dict())
        assert {"backbone_3d.conv_input.0.weight", "backbone_3d.conv_out.0.weight", "backbone_2d.blocks.0.1.weight",
                "dense_head.shared_conv.0.weight", "dense_head.heads_list.0.hm.1.bias", "global_step"} <= keys


def test_dataset_sample_is_deterministic_and_masks_range():
    from toda_amd.pcdet.datasets import SyntheticLidarDataset

    cfg = load("second_backbone_nuscenes")
    cfg.DATA_CONFIG.SYNTHETIC.NUM_POINTS = 5000
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES)
    a, b = ds[3], ds[3]
    assert np.array_equal(a["points"], b["points"]) and np.array_equal(a["gt_boxes"], b["gt_boxes"])
    r = cfg.DATA_CONFIG.POINT_CLOUD_RANGE
    p = a["points"]
    assert (p[:, 0] >= r[0]).all() and (p[:, 0] <= r[3]).all() and (p[:, 1] >= r[1]).all() and (p[:, 1] <= r[4]).all()
    assert a["gt_boxes"].shape[1] == 8 and set(np.unique(a["gt_boxes"][:, 7])) <= {1.0, 2.0, 3.0}


def test_collate_batch_contract():
    from toda_amd.pcdet.datasets import DatasetTemplate

    s0 = {"points": np.ones((5, 4), np.float32), "voxels": np.zeros((3, 2, 4), np.float32), "voxel_coords": np.zeros((3, 3), np.int32),
          "voxel_num_points": np.ones(3, np.int32), "gt_boxes": np.ones((2, 8), np.float32), "frame_id": "a"}
    s1 = {"points": np.ones((7, 4), np.float32), "voxels": np.zeros((4, 2, 4), np.float32), "voxel_coords": np.zeros((4, 3), np.int32),
          "voxel_num_points": np.ones(4, np.int32), "gt_boxes": np.ones((5, 8), np.float32), "frame_id": "b"}
    out = DatasetTemplate.collate_batch([s0, s1])
    assert out["batch_size"] == 2 and out["points"].shape == (12, 5) and out["voxel_coords"].shape == (7, 4)
    assert out["points"][:5, 0].tolist() == [0] * 5 and out["points"][5:, 0].tolist() == [1] * 7
    assert out["voxel_coords"][3:, 0].tolist() == [1] * 4
    assert out["gt_boxes"].shape == (2, 5, 8) and (out["gt_boxes"][0, 2:] == 0).all()
    assert out["voxels"].shape == (7, 2, 4) and out["points_per_sample"] == [5, 7]


def test_spconv_v1_checkpoint_layout_is_converted_on_load(tmp_path):
    from toda_amd.pcdet.datasets import SyntheticLidarDataset
    from toda_amd.pcdet.models import build_network

    cfg = load("centerpoint_voxel_waymo")
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES)
    model = build_network(cfg.MODEL, 3, ds)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    key = "backbone_3d.conv2.0.0.weight"            # [Cout,kz,ky,kx,Cin] here; spconv 1.x stores [kz,ky,kx,Cin,Cout]
    v1 = sd[key].permute(1, 2, 3, 4, 0).contiguous()
    sd[key] = v1
    path = tmp_path / "ckpt.pth"
    torch.save({"model_state": sd, "epoch": 3, "it": 7, "optimizer_state": None}, path)
    fresh = build_network(cfg.MODEL, 3, ds)
    fresh.load_params_from_file(str(path), to_cpu=True)
    assert torch.equal(fresh.state_dict()[key], model.state_dict()[key])
    it, epoch = fresh.load_params_with_optimizer(str(path), to_cpu=True)
    assert (it, epoch) == (7, 3)


def test_consistency_helpers_round_trip():
    from toda_amd.pcdet.models import get_consistency_loss, random_world_flip, random_world_rotation, random_world_scaling

    rng = np.random.default_rng(0)
    boxes = torch.from_numpy(rng.uniform(-5, 5, (6, 7)).astype(np.float32))
    b = boxes.clone()
    b = random_world_flip(b, ["x"])
    b = random_world_rotation(b, 0.3)
    b = random_world_scaling(b, 1.04)
    b = random_world_scaling(b, 1.04, reverse=True)
    b = random_world_rotation(b, 0.3, reverse=True)
    b = random_world_flip(b, ["x"], reverse=True)
    assert torch.allclose(b, boxes, atol=1e-5)
    c, s = get_consistency_loss([{"pred_boxes": boxes.clone()}], [{"pred_boxes": boxes.clone()}])
    assert float(c) == 0.0 and float(s) == 0.0
    shifted = boxes.clone()
    shifted[:, 0] += 0.5
    c, s = get_consistency_loss([{"pred_boxes": boxes.clone()}], [{"pred_boxes": shifted}])
    assert abs(float(c) - 0.5) < 1e-5 and float(s) == 0.0


def test_stage2_two_forward_one_backward_step_on_cpu():
    from oracle.cpu_backend import oracle_backend
    from toda_amd.pcdet.datasets import SyntheticPairDataset
    from toda_amd.pcdet.models import DistModel, build_network, model_fn_decorator_cl

    cfg = load("toda_stage1_centerpoint_res")
    cfg.DATA_CONFIG.POINT_CLOUD_RANGE = [-7.2, -7.2, -5.0, 7.2, 7.2, 4.8]
    cfg.DATA_CONFIG.SYNTHETIC.NUM_POINTS = 4000
    cfg.MODEL.BACKBONE_2D.LAYER_NUMS = [1, 1]
    cfg.MODEL.DENSE_HEAD.POST_PROCESSING.POST_CENTER_LIMIT_RANGE = [-7.2, -7.2, -10, 7.2, 7.2, 10]
    ds = SyntheticPairDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES)
    torch.manual_seed(0)
    model = build_network(cfg.MODEL, 1, ds).train()
    adv, org = ds.collate_batch([ds[0], ds[1]])
    assert org["augmentation_list"][0][-1] == "random_world_scaling"
    with oracle_backend():
        ret = model_fn_decorator_cl()(DistModel(model), adv, org)
        ret.loss.backward()
    assert torch.isfinite(ret.loss) and {"loss_org", "cl_center", "cl_size"} <= set(ret.tb_dict)
    assert int(model.global_step) == 1
    assert all(p.grad is not None for n, p in model.named_parameters() if "backbone_3d.conv_input.0" in n)


def test_install_as_pcdet_aliases_every_submodule_lazily():
    """`pcdet.*` / `spconv.*` imports resolve to the SAME module objects as `toda_amd.pcdet.*` / `toda_amd.spconv.*` (run in a
    child interpreter so this session's sys.modules stay clean)."""
    import subprocess
    import sys
    code = """
import toda_amd.pcdet as tp
tp.install_as_pcdet()
from pcdet.config import cfg, cfg_from_yaml_file
from pcdet.models import build_network, model_fn_decorator
from pcdet.utils.spconv_utils import spconv
from pcdet.ops.iou3d_nms import iou3d_nms_utils
from pcdet.ops.roiaware_pool3d import roiaware_pool3d_utils
from pcdet.datasets.augmentor import data_augmentor, database_sampler
from pcdet.datasets.processor.inter_domain_point_polarmix import inter_domain_point_polarmix
from pcdet.datasets.processor.intra_domain_point_mixup import intra_domain_point_mixup_cd
from pcdet.utils import box_utils
import spconv.pytorch as sp2
import toda_amd.pcdet.utils.box_utils as real_bu, toda_amd.spconv as real_sp
assert spconv.SubMConv3d is sp2.SubMConv3d is real_sp.SubMConv3d and box_utils is real_bu
assert isinstance(spconv.SubMConv3d(4, 8, 3, indice_key='k'), spconv.conv.SparseConvolution)
print('ok')
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_train_one_epoch_moves_train_percent_like_the_reference():
    """reference tools/train_utils/train_utils.py:47-48: dataset.train_percent = (epoch * its + it) / (epochs * its)."""
    from toda_amd.pcdet.config import AttrDict
    from toda_amd.tools.train_utils.train_utils import train_one_epoch

    class DS(torch.utils.data.Dataset):
        train_percent = 0.0

        def __len__(self):
            return 4

        def __getitem__(self, i):
            return torch.zeros(1)

    seen = []
    lin = torch.nn.Linear(1, 1)
    loader = torch.utils.data.DataLoader(DS(), batch_size=1)

    def model_func(model, batch):
        seen.append(loader.dataset.train_percent)
        return model(batch).sum(), {}, {}

    class Sched:
        def step(self, it):
            pass

    opt = torch.optim.SGD(lin.parameters(), lr=0.1)
    it = 0
    for epoch in range(2):
        it = train_one_epoch(lin, opt, loader, model_func, Sched(), it, AttrDict(GRAD_NORM_CLIP=10), rank=1, tbar=None,
                             total_it_each_epoch=4, dataloader_iter=iter(loader), cur_epoch=epoch, total_epoch=2)
    assert seen == [k / 8 for k in range(8)]


def test_polarmix_sector_schedule_follows_train_percent():
    """ASC sectors widen with train_percent (reference inter_domain_point_polarmix.py:62-75)."""
    from toda_amd.pcdet.datasets.processor import point_mix

    w = []
    for pct in (0.0, 0.5, 1.0):
        sectors = point_mix.polarmix_sectors([np.pi / 4, np.pi / 2], pct, ["ASC"], np.random.RandomState(0))
        w.append(sum(hi - lo for lo, hi in sectors))
    assert w[0] < w[1] < w[2]


def test_sync_batchnorm_is_never_taken_by_the_fused_row_path():
    """--sync_bn (reference tools/train.py:117-118): a converted SparseBasicBlock must use the SyncBatchNorm modules, not the
    per-rank fused BatchNorm1d row passes."""
    from toda_amd import ops
    from toda_amd.pcdet.models.backbones_3d.spconv_backbone import SparseBasicBlock
    from functools import partial

    block = SparseBasicBlock(16, 16, norm_fn=partial(torch.nn.BatchNorm1d, eps=1e-3, momentum=0.01), indice_key="res1")
    conv = torch.nn.SyncBatchNorm.convert_sync_batchnorm(block)
    assert type(conv.bn1) is torch.nn.SyncBatchNorm

    class FakeCuda:      # the predicate must say no before it looks at the tensor
        is_cuda, dtype, shape = True, torch.float32, (1000, 16)

    assert not ops.bn_rows_supported(FakeCuda(), conv.bn1)
    assert ops.bn_rows_supported(FakeCuda(), torch.nn.BatchNorm1d(16, eps=1e-3, momentum=0.01))


def test_clip_grad_norm_matches_torch():
    """train_utils.optimization.clip_grad_norm_ against torch.nn.utils.clip_grad_norm_ (reference train_utils.py:57): same total
    norm, same clipped gradients, with and without clipping taking effect, parameters without a gradient skipped."""
    import copy

    from toda_amd.tools.train_utils.optimization import clip_grad_norm_

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    net(torch.randn(4, 7))[:, 0].sum().backward()
    net[3].weight.grad = None
    for max_norm in (0.05, 1e6):
        a, b = copy.deepcopy(net), copy.deepcopy(net)
        for (_, p), (_, q), (_, r) in zip(net.named_parameters(), a.named_parameters(), b.named_parameters()):
            q.grad = None if p.grad is None else p.grad.clone()
            r.grad = None if p.grad is None else p.grad.clone()
        ta = clip_grad_norm_(list(a.parameters()), max_norm)
        tb = torch.nn.utils.clip_grad_norm_(b.parameters(), max_norm)
        assert torch.equal(ta, tb)
        for q, r in zip(a.parameters(), b.parameters()):
            assert (q.grad is None) == (r.grad is None)
            if q.grad is not None:
                assert torch.equal(q.grad, r.grad)


def test_grad_bucket_reducer_single_rank_keeps_gradients_and_views(tmp_path):
    """GradBucketReducer at world size 1 (gloo): after backward every .grad is a slice of its bucket's flat buffer holding the
    plain gradient; a second backward without dropping the gradients accumulates in place and is reduced without a gather."""
    import os
    import torch
    import torch.distributed as dist
    from toda_amd.pcdet.utils.common_utils import GradBucketReducer

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29641"
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        red = GradBucketReducer(list(net.parameters()), bucket_bytes=64)       # several tiny buckets
        x = torch.randn(4, 7)
        net(x).sum().backward()
        g1 = [p.grad.clone() for p in net.parameters()]
        spans = [(f.data_ptr(), f.data_ptr() + f.numel() * 4) for f in red.flat]
        assert all(any(lo <= p.grad.data_ptr() < hi for lo, hi in spans) for p in net.parameters())
        plain = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        plain.load_state_dict(net.state_dict())
        plain(x).sum().backward()
        for a, b in zip(g1, [p.grad for p in plain.parameters()]):
            assert torch.equal(a, b)
        net(x).sum().backward()                  # gradients kept: accumulated in place, reduced in place
        for p, a in zip(net.parameters(), g1):
            assert torch.allclose(p.grad, 2 * a)
        with_off = [p.grad.clone() for p in net.parameters()]
        red.enabled = False                      # no_sync: hooks stand aside
        net(x).sum().backward()
        for p, a, b in zip(net.parameters(), with_off, g1):
            assert torch.allclose(p.grad, a + b)
    finally:
        dist.destroy_process_group()


def test_consistency_loss_ignores_inf_and_nan_in_unselected_candidate_rows():
    """the padded form of get_consistency_loss (reference pcdet/models/__init__.py:216-260) keeps unselected candidate rows
    behind a mask; such a row may hold inf / nan (dim.exp() of a wild regression output) and must not reach the loss."""
    import torch
    from toda_amd.pcdet.models import get_consistency_loss

    g = torch.Generator().manual_seed(0)
    a = torch.randn(12, 7, generator=g)
    o = a + 0.05 * torch.randn(12, 7, generator=g)
    va = torch.tensor([True] * 8 + [False] * 4)
    vo = torch.tensor([True] * 9 + [False] * 3)
    clean = get_consistency_loss([{"pred_boxes": a.clone(), "mask": va}], [{"pred_boxes": o.clone(), "mask": vo}])
    a_bad, o_bad = a.clone(), o.clone()
    a_bad[9] = float("inf")
    a_bad[10, 3:6] = float("nan")
    o_bad[11] = float("-inf")
    dirty = get_consistency_loss([{"pred_boxes": a_bad, "mask": va}], [{"pred_boxes": o_bad, "mask": vo}])
    for c, d in zip(clean, dirty):
        assert torch.isfinite(d) and float(c) == float(d)
    # and the unpadded form (selected rows only) gives the same numbers
    plain = get_consistency_loss([{"pred_boxes": a[va]}], [{"pred_boxes": o[vo]}])
    for c, p in zip(clean, plain):
        assert abs(float(c) - float(p)) < 1e-6
