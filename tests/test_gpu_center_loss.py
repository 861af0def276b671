"""toda_center_loss_{fwd,bwd} (the fused CenterHead.get_loss) against the operator-by-operator torch formulation of the
reference losses (pcdet/utils/loss_utils.py:264-385, center_head.py:229-262) in float64."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def torch_losses(hm, regs, heatmap, inds, mask, target, code_w, cls_w, loc_w):
    from toda_amd.pcdet.utils import loss_utils

    p = torch.clamp(hm.sigmoid(), min=1e-4, max=1 - 1e-4)
    hm_loss = loss_utils.FocalLossCenterNet()(p, heatmap) * cls_w
    reg = loss_utils.RegLossCenterNet()(torch.cat(regs, dim=1), mask, inds, target)
    loc_loss = (reg * reg.new_tensor(code_w)).sum() * loc_w
    return hm_loss, loc_loss, p


@pytest.mark.parametrize("geom", [(2, 3, 188, 188, 500, (2, 1, 3, 2)), (1, 1, 12, 20, 7, (2, 1, 3, 2, 2)), (3, 2, 16, 16, 64, (2, 1, 3, 2))])
def test_fused_center_loss_matches_torch_fp64(geom):
    from toda_amd import ops

    b, c, h, w, k, chans = geom
    g = torch.Generator().manual_seed(h + k)
    hm = (torch.randn((b, c, h, w), generator=g) * 3 - 2).cuda()
    hm[0, 0, 0, :4] = torch.tensor([-20.0, 20.0, -9.3, 9.3])                    # both clamp sides and their edges
    heatmap = torch.rand((b, c, h, w), generator=g).pow(6).cuda()
    d = sum(chans)
    inds = torch.randint(0, h * w, (b, k), generator=g)
    inds[:, 1] = inds[:, 0]                                                      # two objects in one cell
    if k > 5:
        inds[:, 5] = inds[:, 0]                                                  # ... and a third
    mask = (torch.rand((b, k), generator=g) < 0.4).long()
    mask[:, :2] = 1
    inds[mask == 0] = 0                                                          # empty slots point at cell 0, like the assigner's
    for bi in range(b):
        cells = inds[bi][mask[bi] == 1]
        heatmap[bi, torch.randint(0, c, (len(cells),), generator=g), cells // w, cells % w] = 1.0
    target = torch.randn((b, k, d), generator=g).cuda()
    regs = [torch.randn((b, ch, h, w), generator=g).cuda() for ch in chans]
    inds, mask = inds.cuda(), mask.cuda()
    code_w = [1.0 + 0.1 * i for i in range(d)]
    cls_w, loc_w = 1.0, 0.25

    hm_a = hm.clone().requires_grad_(True)
    regs_a = [r.clone().requires_grad_(True) for r in regs]
    assert ops.center_loss_supported(hm_a, regs_a, target)
    la, lb, prob = ops.center_loss(hm_a, regs_a, heatmap, inds, mask, target, code_w, cls_w, loc_w)
    (1.7 * la + 0.6 * lb).backward()

    hm_r = hm.double().requires_grad_(True)
    regs_r = [r.double().requires_grad_(True) for r in regs]
    ra, rb, p_ref = torch_losses(hm_r, regs_r, heatmap.double(), inds, mask, target.double(), code_w, cls_w, loc_w)
    (1.7 * ra + 0.6 * rb).backward()

    assert abs(float(la.detach()) - float(ra.detach())) <= 2e-5 * max(1.0, abs(float(ra.detach())))     # fp32 element arithmetic vs fp64
    assert abs(float(lb.detach()) - float(rb.detach())) <= 2e-5 * max(1.0, abs(float(rb.detach())))
    assert float((prob.double() - p_ref.detach()).abs().max()) < 1e-6
    gz, gz_ref = hm_a.grad.double(), hm_r.grad
    assert float((gz - gz_ref).abs().max()) <= 5e-5 * float(gz_ref.abs().max())
    for ga, gr in zip(regs_a, regs_r):
        assert float((ga.grad.double() - gr.grad).abs().max()) <= 1e-5 * max(1e-6, float(gr.grad.abs().max()))

    # run-to-run identical (no atomics)
    hm_b = hm.clone().requires_grad_(True)
    regs_b = [r.clone().requires_grad_(True) for r in regs]
    la2, lb2, _ = ops.center_loss(hm_b, regs_b, heatmap, inds, mask, target, code_w, cls_w, loc_w)
    (1.7 * la2 + 0.6 * lb2).backward()
    assert torch.equal(la, la2) and torch.equal(lb, lb2) and torch.equal(hm_a.grad, hm_b.grad)
    assert all(torch.equal(x.grad, y.grad) for x, y in zip(regs_a, regs_b))


def test_center_head_get_loss_fused_equals_operator_path(monkeypatch):
    """CenterHead.get_loss through the fused kernels and through the torch operators on the same head and batch: loss, tb_dict
    and the gradient of the shared feature map."""
    from tests.test_gpu_golden import load
    from tests.test_golden_reference import build_head

    g = load("center_head")
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TODA_FUSED_LOSS", mode)
        head = build_head(g).cuda()
        x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
        head({"spatial_features_2d": x, "gt_boxes": torch.from_numpy(g["gt"].copy()).cuda(), "batch_size": 2})
        loss, tb = head.get_loss()
        loss.backward()
        out[mode] = (float(loss), {k: float(v) for k, v in tb.items()}, x.grad.clone())
    assert abs(out["1"][0] - out["0"][0]) < 1e-5 * max(1.0, abs(out["0"][0]))
    for k, v in out["0"][1].items():
        assert abs(out["1"][1][k] - v) < 1e-5 * max(1.0, abs(v)), k
    assert float((out["1"][2] - out["0"][2]).abs().max()) < 1e-4 * float(out["0"][2].abs().max())


def test_multi_group_center_head_with_velocity_branch_kernels_equal_operator_path(monkeypatch):
    """A nuScenes-style CenterHead (two head groups, HEAD_ORDER with a `vel` branch, 64 shared channels, no conv bias before the
    norms) through the hand-written path (Winograd shared conv, fused 64 -> 5 x 64 hidden layer, narrow output convs, GPU target
    assignment, fused loss) and through the plain torch operators on the same weights and batch: predictions, loss, tb_dict, the
    input gradient and every parameter gradient."""
    import copy

    import numpy as np

    from toda_amd import ops
    from toda_amd.pcdet.config import AttrDict
    from toda_amd.pcdet.models.dense_heads import CenterHead

    names = ["car", "truck", "pedestrian"]
    order = ["center", "center_z", "dim", "rot", "vel"]
    cfg = dict(CLASS_AGNOSTIC=False, CLASS_NAMES_EACH_HEAD=[["car", "truck"], ["pedestrian"]], SHARED_CONV_CHANNEL=64, USE_BIAS_BEFORE_NORM=False,
               NUM_HM_CONV=2,
               SEPARATE_HEAD_CFG=dict(HEAD_ORDER=order, HEAD_DICT=dict(center=dict(out_channels=2, num_conv=2), center_z=dict(out_channels=1, num_conv=2),
                                                                       dim=dict(out_channels=3, num_conv=2), rot=dict(out_channels=2, num_conv=2),
                                                                       vel=dict(out_channels=2, num_conv=2))),
               TARGET_ASSIGNER_CONFIG=dict(FEATURE_MAP_STRIDE=8, NUM_MAX_OBJS=100, GAUSSIAN_OVERLAP=0.1, MIN_RADIUS=2),
               LOSS_CONFIG=dict(LOSS_WEIGHTS=dict(cls_weight=1.0, loc_weight=0.25, code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2])),
               POST_PROCESSING=dict(SCORE_THRESH=0.1, POST_CENTER_LIMIT_RANGE=[-60, -60, -10, 60, 60, 10], MAX_OBJ_PER_SAMPLE=100,
                                    NMS_CONFIG=dict(NMS_TYPE="nms_gpu", NMS_THRESH=0.2, NMS_PRE_MAXSIZE=1000, NMS_POST_MAXSIZE=83)))
    torch.manual_seed(7)
    head = CenterHead(AttrDict(cfg), 96, 3, names, np.array([512, 512, 40]), [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], [0.2, 0.2, 0.2],
                      predict_boxes_when_training=False).cuda().train()
    ref = copy.deepcopy(head)
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 96, 64, 64), generator=g).cuda()
    gt = torch.zeros((2, 12, 10))
    gt[:, :, 0:2] = torch.rand((2, 12, 2), generator=g) * 90 - 45
    gt[:, :, 2] = torch.rand((2, 12), generator=g) * 2 - 2
    gt[:, :, 3:6] = torch.rand((2, 12, 3), generator=g) * 3 + 0.5
    gt[:, :, 6] = torch.rand((2, 12), generator=g) * 6.28 - 3.14
    gt[:, :, 7:9] = torch.randn((2, 12, 2), generator=g)
    gt[:, :, 9] = torch.randint(1, 4, (2, 12), generator=g).float()
    gt[1, 9:] = 0                                                    # padding rows
    gt = gt.cuda()

    out = {}
    for mode, net in (("kernels", head), ("operators", ref)):
        if mode == "operators":
            monkeypatch.setattr(ops, "DENSE_CONV", "miopen")
            monkeypatch.setattr(ops, "FUSED_BN2D", False)
            monkeypatch.setenv("TODA_FUSED_LOSS", "0")
        xi = x.clone().requires_grad_(True)
        net({"spatial_features_2d": xi, "gt_boxes": gt.clone(), "batch_size": 2})
        preds = [{k: v.detach().clone() for k, v in d.items()} for d in net.forward_ret_dict["pred_dicts"]]
        loss, tb = net.get_loss()
        loss.backward()
        out[mode] = (preds, float(loss.detach()), {k: float(v) for k, v in tb.items()}, xi.grad.clone())
    for da, db in zip(out["kernels"][0], out["operators"][0]):
        for k in db:
            assert float((da[k] - db[k]).abs().max()) < 2e-4 * max(1.0, float(db[k].abs().max())), k
    assert abs(out["kernels"][1] - out["operators"][1]) < 1e-4 * max(1.0, abs(out["operators"][1]))
    for k, v in out["operators"][2].items():
        assert abs(out["kernels"][2][k] - v) < 1e-4 * max(1.0, abs(v)), k
    assert float((out["kernels"][3] - out["operators"][3]).abs().max()) < 2e-3 * float(out["operators"][3].abs().max())
    # (random-init head: a hidden activation within rounding of zero takes the other side of its ReLU in one of the two
    # implementations and moves single weight gradients by a few 1e-4 of the largest one)
    for (n, p), (_, q) in zip(head.named_parameters(), ref.named_parameters()):
        assert float((p.grad - q.grad).abs().max()) <= 4e-3 * float(q.grad.abs().max()) + 1e-6, n
    for (n, p), (_, q) in zip(head.named_buffers(), ref.named_buffers()):
        assert torch.allclose(p.float(), q.float(), rtol=1e-4, atol=1e-6), n
