"""LDS-staged halo tiles for the SubM gather-GEMM (toda_halo_plan_build / toda_spconv_gather_gemm_halo; reference
spconv_backbone.py:21-25,54-64: SubMConv3d forward and, through autograd, its data gradient).  The plan is checked as a
structure (every row in exactly one block, every neighbour resolved), the kernel bit for bit against the per-offset gather
kernel on the same operands, including blocks whose unique rows do not fit (spill path), duplicate / out-of-lattice rows and
the BatchNorm moments of the epilogue."""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = [pytest.mark.gpu, H.needs_variants]


def _level(shape, batch, npb, seed, clustered=True, c=64):
    from toda_amd import ops

    idx, feat = (H.clustered_sparse if clustered else H.random_sparse)(batch, shape, npb, c, seed=seed)
    ind = torch.from_numpy(idx).cuda()
    rb, _ = ops.build_subm_rulebook(ind, batch, shape)
    return ind, torch.from_numpy(feat).cuda(), rb


def _same(ref, got, c):
    """32 channels: one pass over the offsets, the per-offset kernel's summation order - the same bits.  64 channels: the gathered
    channels are worked off in two passes of 32 (all offsets with channels 0-31, then all offsets with 32-63), a different but
    equally valid fp32 summation order: equal to rounding (1e-5 of the output's scale; products of ~N(0,1) x 0.1 operands)."""
    if c == 32:
        assert torch.equal(ref, got)
    else:
        assert float((ref - got).abs().max()) <= 1e-5 * float(ref.abs().max())


def _plan_parts(plan, n, K, c):
    R, umax = 128, 320          # rows per block / resident rows: halo_geom in csrc/halo_plan.hip
    nb = (n + R - 1) // R
    al = lambda v: (v + 255) // 256 * 256
    raw = plan.cpu().numpy()
    order = raw[:nb * R * 4].view(np.int32).reshape(nb, R)
    o1 = al(nb * R * 4)
    urows = raw[o1:o1 + nb * umax * 4].view(np.int32).reshape(nb, umax)
    o2 = o1 + al(nb * umax * 4)
    lids = raw[o2:o2 + nb * K * R * 2].view(np.uint16).reshape(nb, K, R)
    return order, urows, lids, umax


@pytest.mark.parametrize("c", [64, 32])
@pytest.mark.parametrize("shape,npb,clustered", [((11, 96, 96), 9000, True), ((40, 24, 24), 21000, False), ((5, 33, 47), 700, False)])
def test_halo_plan_structure(shape, npb, clustered, c):
    from toda_amd import ops

    ind, feat, rb = _level(shape, 2, npb, seed=3, clustered=clustered, c=c)
    n, K = rb.nbr_fwd.shape[1], rb.nbr_fwd.shape[0]
    plan = ops.build_halo_plan(rb, ind, 2, shape, c)
    order, urows, lids, umax = _plan_parts(plan, n, K, c)
    flat = order.reshape(-1)
    assert np.array_equal(np.sort(flat[flat >= 0]), np.arange(n)) and (flat[n:] == -1).all()      # a permutation, padded behind
    nbr = rb.nbr_fwd.cpu().numpy()
    spills = 0
    for b in range(order.shape[0]):
        rows = order[b]
        live = rows >= 0
        want = np.where(live[None, :], nbr[:, np.where(live, rows, 0)], -1)           # [K, R] neighbour rows of the block
        uniq = np.unique(want[want >= 0])
        nu = min(len(uniq), umax)
        assert np.array_equal(urows[b, :nu], uniq[:nu]) and (urows[b, nu:] == -1).all()      # ascending unique rows, padded
        l = lids[b].astype(np.int64)
        assert ((l == 0xFFFF) == (want < 0)).all()
        res = (l < 0xFFFE)
        assert np.array_equal(urows[b][l[res]], want[res])                         # every resident id names its neighbour
        sp = (l == 0xFFFE)
        spills += int(sp.sum())
        assert (np.searchsorted(uniq, want[sp]) >= umax).all()                      # spilled = did not fit, nothing else
    if shape == (40, 24, 24):
        assert spills > 0          # the dense, deep lattice is there to exercise the spill path


@pytest.mark.parametrize("c", [64, 32])
@pytest.mark.parametrize("shape,npb,clustered", [((11, 96, 96), 9000, True), ((40, 24, 24), 21000, False), ((5, 33, 47), 700, False)])
def test_halo_gather_gemm_is_bit_identical_to_the_per_offset_kernel(shape, npb, clustered, c):
    from toda_amd import ops

    ind, feat, rb = _level(shape, 2, npb, seed=5, clustered=clustered, c=c)
    K = rb.nbr_fwd.shape[0]
    rng = np.random.default_rng(1)
    w = torch.from_numpy((rng.standard_normal((c, 3, 3, 3, c)) * 0.1).astype(np.float32)).cuda()
    bias = torch.from_numpy(rng.standard_normal(c).astype(np.float32)).cuda()
    plan = ops.build_halo_plan(rb, ind, 2, shape, c)
    for transpose, flip, b in ((False, False, bias), (True, True, None)):        # forward operand with bias, dgrad operand
        wp = ops.pack_weight(w, transpose, flip)
        ref = ops.gather_gemm(feat, wp, rb.nbr_fwd, c, b)
        got = ops.gather_gemm_halo(feat, wp, rb.nbr_fwd, c, plan, b)
        _same(ref, got, c)
        again = ops.gather_gemm_halo(feat, wp, rb.nbr_fwd, c, plan, b)
        assert torch.equal(got, again)
    wp = ops.pack_weight(w, False, False)
    ref, sums_ref = ops.gather_gemm_with_stats(feat, wp, rb.nbr_fwd, c, None)
    got, sums = ops.gather_gemm_halo(feat, wp, rb.nbr_fwd, c, plan, None, True)
    _same(ref, got, c)
    a, b2 = sums[:2 * c].cpu().numpy(), sums_ref[:2 * c].cpu().numpy()
    np.testing.assert_allclose(a, b2, rtol=1e-5, atol=1e-5 * np.abs(b2).max())
    direct = torch.stack([got.double().sum(0), (got.double() ** 2).sum(0)]).reshape(-1).cpu().numpy()
    np.testing.assert_allclose(a, direct, rtol=1e-5, atol=1e-5 * np.abs(direct).max())


def test_halo_with_duplicate_and_stray_rows():
    """Hand-built tensors: duplicate coordinates and rows outside the lattice.  Every row is still produced exactly once and
    equals the per-offset kernel's result."""
    from toda_amd import ops

    shape = (9, 40, 40)
    idx, feat = H.clustered_sparse(2, shape, 4000, 64, seed=9)
    idx = np.concatenate([idx, idx[:37], np.array([[0, 20, 5, 5], [3, 1, 1, 1], [1, 2, 50, 2]], np.int32)]).astype(np.int32)
    feat = np.random.default_rng(2).standard_normal((len(idx), 64)).astype(np.float32)
    ind, f = torch.from_numpy(idx).cuda(), torch.from_numpy(feat).cuda()
    rb, _ = ops.build_subm_rulebook(ind, 2, shape)
    plan = ops.build_halo_plan(rb, ind, 2, shape, 64)
    w = torch.from_numpy((np.random.default_rng(3).standard_normal((64, 3, 3, 3, 64)) * 0.1).astype(np.float32)).cuda()
    wp = ops.pack_weight(w, False, False)
    _same(ops.gather_gemm(f, wp, rb.nbr_fwd, 64), ops.gather_gemm_halo(f, wp, rb.nbr_fwd, 64, plan), 64)


def test_backbone_takes_the_halo_kernel_and_keeps_its_bits(monkeypatch):
    """VoxelBackBone8x forward + backward with the halo kernel on every eligible SubM level against the same step with it
    switched off: identical features and gradients (and the C-ABI call count shows the kernel ran)."""
    from tests.helpers import abi_calls
    from tests.test_gpu_e2e import small_cfg
    from toda_amd import ops
    from toda_amd.pcdet.datasets import SyntheticLidarDataset
    from toda_amd.pcdet.models import build_network, voxelize_on_gpu

    cfg = small_cfg("centerpoint_voxel_waymo", rng_xy=16.0, n_points=20000)
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
    torch.manual_seed(3)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds).cuda().train()
    col = ds.collate_batch([ds[0], ds[1]])
    points = torch.from_numpy(col["points"]).float().cuda()

    def run(bn_train):
        model.zero_grad(set_to_none=True)
        for m in model.modules():          # same BatchNorm state for both runs
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.reset_running_stats()
                m.train(bn_train)
        batch = {"points": points, "points_per_sample": col["points_per_sample"], "batch_size": 2}
        voxelize_on_gpu(batch, ds.voxel_cfg)
        for m in (model.vfe, model.backbone_3d, model.map_to_bev_module):
            batch = m(batch)
        out = batch["spatial_features"]
        (out * torch.linspace(-1, 1, out.numel(), device="cuda").view_as(out)).sum().backward()
        return out.detach().clone(), [p.grad.clone() for p in model.backbone_3d.parameters()]

    monkeypatch.setattr(ops, "HALO_MIN_ROWS", 1)
    monkeypatch.setattr(ops, "HALO", True)
    with abi_calls("toda_spconv_gather_gemm_halo", "toda_halo_plan_build") as n:
        a, a_train = run(False), run(True)
    assert n["toda_halo_plan_build"] >= 6 and n["toda_spconv_gather_gemm_halo"] >= 24, n      # subm2 / subm3 / subm4: 2 layers x (fwd + dgrad), two runs
    monkeypatch.setattr(ops, "HALO", False)
    with abi_calls("toda_spconv_gather_gemm_halo") as n:
        b, b_train = run(False), run(True)
    assert n["toda_spconv_gather_gemm_halo"] == 0
    # the 64-channel layers sum in another (equally valid) order.  Forward: BatchNorm on its running statistics agrees to rounding
    # (1e-4 of the tensor's scale through 12 layers).  Backward: the ReLU masks are recomputed from pre-activations that differ in
    # the last bits, so an element that sits on the threshold flips and moves single gradient entries by 1e-3 of the tensor's maximum
    # (profiles/scripts/compact_e2e.py: ANY two implementations differ like that here - the packed kernels are 2.5e-3 from the CPU oracle backend
    # on conv3.2.0.weight, the compacting ones 3e-4): gradients are compared in the L2 norm, which a flipped element barely moves.
    # In training mode the moments come out of the epilogues as per-workgroup partial sums whose grouping follows the row blocks, and a
    # train-mode BatchNorm amplifies rounding (tests/test_noise_floor.py: 7e-3 between two runs of the SAME CPU code with permuted rows): 2e-2
    err_out = float((a[0] - b[0]).abs().max()) / (float(b[0].abs().max()) + 1e-12)
    errs = [float((x - y).norm()) / (float(y.norm()) + 1e-12) for x, y in zip(a[1], b[1])]
    print("eval mode: output", f"{err_out:.1e}", "gradients (L2)", [f"{e:.1e}" for e in errs])
    assert err_out <= 1e-4, err_out
    for err in errs:
        assert err <= 2e-3, errs
    for x, y in zip([a_train[0]] + a_train[1], [b_train[0]] + b_train[1]):
        err = float((x - y).norm()) / (float(y.norm()) + 1e-12)
        assert err <= 2e-2, err
