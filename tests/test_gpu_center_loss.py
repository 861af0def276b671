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
    assert float((prob.double() - p_ref).abs().max()) < 1e-6
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
