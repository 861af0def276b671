"""Train-mode BACKWARD at the full C3 sizes, layer by layer: the same upstream gradient is fed into one
hand-written backward at a time and compared with float64 torch / the CPU oracle at 1e-4 of the tensor's scale - no whole-net
ReLU amplification between the thing measured and the reference.

  * rows_bn_bwd(+res) at [389 533, 64] and [682 284, 32]   (BatchNorm1d + ReLU / + shortcut of the sparse blocks,
                                                             reference pcdet/models/backbones_3d/spconv_backbone.py:8-66)
  * bn2d_bwd / bn2d_bwd_from at [2,128,188,188] / [2,256,94,94]   (BEV neck, reference base_bev_backbone.py:81-112)
  * the 64 -> 64 SubM convolution with the statistics epilogue on the full C3 stride-4 rulebook: forward, moments, data and
    weight gradient against the oracle
The dense-head / whole-net comparisons stay in test_gpu_e2e.py."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def rel(a, b):
    """max |a - b| over the scale of b"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _bn1d_pair(c, seed):
    rng = np.random.default_rng(seed)
    ref = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).double()
    with torch.no_grad():
        ref.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, c)))
        ref.bias.copy_(torch.from_numpy(rng.uniform(-0.5, 0.5, c)))
    mine = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)
    mine.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in ref.state_dict().items()})
    return ref.train(), mine.cuda().train()


@pytest.mark.parametrize("n,c,residual", [(389533, 64, False), (389533, 64, True), (682284, 32, False), (682284, 32, True)])
def test_full_size_rows_bn_backward_matches_float64(n, c, residual):
    from toda_amd import ops

    rng = np.random.default_rng(n + c)
    x = (rng.standard_normal((n, c)) * 1.3 + 0.1).astype(np.float32)
    g = rng.standard_normal((n, c)).astype(np.float32)
    r = rng.standard_normal((n, c)).astype(np.float32) if residual else None
    ref, mine = _bn1d_pair(c, c)
    xr = torch.from_numpy(x).double().requires_grad_(True)
    rr = torch.from_numpy(r).double().requires_grad_(True) if residual else None
    pre = ref(xr) + (rr if residual else 0.0)
    torch.relu(pre).backward(torch.from_numpy(g).double())
    xm = torch.from_numpy(x).cuda().requires_grad_(True)
    rm = torch.from_numpy(r).cuda().requires_grad_(True) if residual else None
    ym = ops.bn_rows(xm, mine, True, residual=rm)
    ym.backward(torch.from_numpy(g).cuda())
    # the ReLU mask is a discontinuity of the reference too: entries whose pre-activation sits within fp32 noise of zero are left out
    safe = (pre.detach().abs() > 2e-5).numpy()
    assert safe.mean() > 0.9999
    assert rel(ym.detach().cpu().numpy()[safe], torch.relu(pre).detach().numpy()[safe]) < 1e-5
    assert rel(xm.grad.cpu().numpy()[safe], xr.grad.numpy()[safe]) < 1e-4
    assert rel(mine.weight.grad.cpu().numpy(), ref.weight.grad.numpy()) < 1e-4
    assert rel(mine.bias.grad.cpu().numpy(), ref.bias.grad.numpy()) < 1e-4
    if residual:
        assert rel(rm.grad.cpu().numpy()[safe], rr.grad.numpy()[safe]) < 1e-6
    assert rel(mine.running_mean.cpu().numpy(), ref.running_mean.numpy()) < 1e-5
    assert rel(mine.running_var.cpu().numpy(), ref.running_var.numpy()) < 1e-5


@pytest.mark.parametrize("n,c,residual", [(389533, 64, True), (97511, 128, False), (1000, 16, False), (3, 4, True)])
def test_rows_bn_backward_column_sums_of_dx_are_the_bias_gradient(n, c, residual):
    """toda_rows_bn_bwd_res_colsum: dx and the parameter gradients are those of toda_rows_bn_bwd_res bit for bit, and dx_colsum is
    the sum of dx over the rows.  With true batch statistics that sum is rounding noise (BatchNorm's input gradient sums to zero),
    so the kernel is also called with a mean that is NOT the batch mean: then every channel has its own, large sum."""
    from toda_amd import lib as L

    lib = L.load()
    rng = np.random.default_rng(n * 7 + c)
    x = torch.from_numpy((rng.standard_normal((n, c)) * 1.3 + 0.1).astype(np.float32)).cuda()
    g = torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32)).cuda() + 0.5 * x      # correlated with x: a large d gamma
    r = torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32)).cuda() if residual else None
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)).cuda()
    beta = torch.from_numpy(rng.uniform(-0.5, 0.5, c).astype(np.float32)).cuda()
    for shift_mean in (0.0, 0.25):
        mean = x.double().mean(0) + shift_mean * torch.arange(1, c + 1, device="cuda", dtype=torch.float64) / c
        invstd = 1.0 / torch.sqrt(x.double().var(0, unbiased=False) + 1e-3)
        scale = gamma.double() * invstd
        stats = torch.stack([mean, invstd, scale, beta.double() - mean * scale]).float().contiguous()
        outs = []
        for with_colsum in (False, True):
            sums = torch.zeros((lib.toda_rows_reduce_doubles(n, c),), dtype=torch.float64, device="cuda")
            dx, dres = torch.empty_like(x), (torch.empty_like(x) if residual else None)
            if with_colsum:
                ws = torch.empty((lib.toda_rows_bn_bwd_colsum_doubles(n, c),), dtype=torch.float64, device="cuda")
                cs = torch.full((c,), float("nan"), device="cuda")
                rc = lib.toda_rows_bn_bwd_res_colsum(L.ptr(g), L.ptr(x), L.ptr(r), L.ptr(stats), L.ptr(gamma), n, c, 1, L.ptr(sums), L.ptr(dx),
                                                     L.ptr(dres), L.ptr(ws), L.ptr(cs), L.stream())
            else:
                cs = None
                rc = lib.toda_rows_bn_bwd_res(L.ptr(g), L.ptr(x), L.ptr(r), L.ptr(stats), L.ptr(gamma), n, c, 1, L.ptr(sums), L.ptr(dx), L.ptr(dres),
                                              L.stream())
            L.check(rc, "rows_bn_bwd")
            outs.append((dx, dres, sums[:2 * c].clone(), cs))
        (dx0, dres0, s0, _), (dx1, dres1, s1, cs) = outs
        assert torch.equal(dx0, dx1) and torch.equal(s0, s1) and (not residual or torch.equal(dres0, dres1))
        want = dx1.double().sum(0)
        noise = dx1.double().abs().sum(0) * 2e-7 + 1e-30
        assert bool(((cs.double() - want).abs() <= noise).all()), float(((cs.double() - want).abs() / noise).max())
        if shift_mean and n > 100:
            assert float(want.abs().min()) > 1e2 * float(noise.max())          # a discriminating case: channel sums far from zero and distinct
    # a second call gives the same bits (fixed-order fold)
    cs2 = torch.empty_like(cs)
    rc = lib.toda_rows_bn_bwd_res_colsum(L.ptr(g), L.ptr(x), L.ptr(r), L.ptr(stats), L.ptr(gamma), n, c, 1, L.ptr(sums), L.ptr(dx), L.ptr(dres),
                                         L.ptr(ws), L.ptr(cs2), L.stream())
    L.check(rc, "rows_bn_bwd")
    assert torch.equal(cs, cs2)


def test_res_block_bias_gradients_come_from_the_bn_backward():
    """SparseBasicBlock (convolutions WITH bias, reference spconv_backbone.py:37-40): the bias gradients are the column sums the
    BatchNorm backward leaves behind - same values (to the rounding noise of a sum that is zero in exact arithmetic) as autograd's
    own grad_output.sum(0), and no torch reduction over the rows is launched for them."""
    import functools
    from toda_amd import ops, spconv
    from toda_amd.pcdet.models.backbones_3d.spconv_backbone import SparseBasicBlock

    rng = np.random.default_rng(5)
    shape = (16, 80, 80)
    cells = rng.choice(shape[0] * shape[1] * shape[2], 30000, replace=False)
    idx = np.stack([np.zeros_like(cells), cells // (shape[1] * shape[2]), cells // shape[2] % shape[1], cells % shape[2]], 1).astype(np.int32)
    feats = rng.standard_normal((len(idx), 64)).astype(np.float32)
    norm = functools.partial(torch.nn.BatchNorm1d, eps=1e-3, momentum=0.01)
    torch.manual_seed(0)
    block = SparseBasicBlock(64, 64, norm_fn=norm, indice_key="res").cuda().train()
    gy = torch.from_numpy(rng.standard_normal(feats.shape).astype(np.float32)).cuda()      # SubM: the output has the input's rows
    grads = {}
    orig = ops._take_colsum
    for flag in (True, False):
        ops.BN_BWD_COLSUM = flag
        block.zero_grad(set_to_none=True)
        taken = []
        ops._take_colsum = lambda gout: taken.append(orig(gout)) or taken[-1]
        try:
            x = spconv.SparseConvTensor(torch.from_numpy(feats).cuda().requires_grad_(True), torch.from_numpy(idx).cuda(), shape, 1)
            block(x).features.backward(gy)
        finally:
            ops._take_colsum = orig
            ops.BN_BWD_COLSUM = True
        grads[flag] = {k: p.grad.clone() for k, p in block.named_parameters()}, [t is not None for t in taken]
    on, off = grads[True], grads[False]
    assert on[1] == [True, True] and off[1] == [False, False]              # conv2's and conv1's backward, in that order
    for k in on[0]:
        if k.endswith("conv1.bias") or k.endswith("conv2.bias"):
            scale = float(block.get_parameter(k.replace("bias", "weight")).grad.abs().max())
            assert float((on[0][k] - off[0][k]).abs().max()) < 1e-4 * scale      # both are the noise of a zero sum
        else:
            assert torch.equal(on[0][k], off[0][k]), k


@pytest.mark.parametrize("shape", [(2, 128, 188, 188), (2, 256, 94, 94)])
def test_full_size_bn2d_backward_matches_float64(shape):
    from toda_amd import ops

    b, c, h, w = shape
    rng = np.random.default_rng(c)
    x = (rng.standard_normal(shape) * 1.5 + 0.2).astype(np.float32)
    g = rng.standard_normal(shape).astype(np.float32)
    ref = torch.nn.BatchNorm2d(c, eps=1e-3, momentum=0.01).double().train()
    with torch.no_grad():
        ref.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, c)))
        ref.bias.copy_(torch.from_numpy(rng.uniform(-0.5, 0.5, c)))
    mine = torch.nn.BatchNorm2d(c, eps=1e-3, momentum=0.01)
    mine.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in ref.state_dict().items()})
    mine = mine.cuda().train()
    xr = torch.from_numpy(x).double().requires_grad_(True)
    pre = ref(xr)
    torch.relu(pre).backward(torch.from_numpy(g).double())
    xm = torch.from_numpy(x).cuda().requires_grad_(True)
    assert ops.bn2d_supported(xm, mine)
    ym = ops.bn2d(xm, mine, True)
    ym.backward(torch.from_numpy(g).cuda())
    safe = (pre.detach().abs() > 2e-5).numpy()
    assert safe.mean() > 0.9999
    assert rel(xm.grad.cpu().numpy()[safe], xr.grad.numpy()[safe]) < 1e-4
    assert rel(mine.weight.grad.cpu().numpy(), ref.weight.grad.numpy()) < 1e-4
    assert rel(mine.bias.grad.cpu().numpy(), ref.bias.grad.numpy()) < 1e-4


def test_full_size_bn2d_cat_backward_reads_its_gradient_slices_in_place():
    """The deblock tails of the BEV neck at the C3 size: two BatchNorm2d + ReLU writing [2, 512, 188, 188] (toda_bn2d_fwd_into) and
    their backward reading the slices of its gradient (toda_bn2d_bwd_from) against torch.cat of float64 modules."""
    from toda_amd import ops

    b, c, h, w = 2, 256, 188, 188
    rng = np.random.default_rng(5)
    xs = [(rng.standard_normal((b, c, h, w)) * (1.0 + 0.5 * i) - 0.1 * i).astype(np.float32) for i in range(2)]
    g = rng.standard_normal((b, 2 * c, h, w)).astype(np.float32)
    refs, mines = [], []
    for i in range(2):
        ref = torch.nn.BatchNorm2d(c, eps=1e-3, momentum=0.01).double().train()
        with torch.no_grad():
            ref.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, c)))
            ref.bias.copy_(torch.from_numpy(rng.uniform(-0.5, 0.5, c)))
        mine = torch.nn.BatchNorm2d(c, eps=1e-3, momentum=0.01)
        mine.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in ref.state_dict().items()})
        refs.append(ref), mines.append(mine.cuda().train())
    xr = [torch.from_numpy(x).double().requires_grad_(True) for x in xs]
    pres = [ref(x) for ref, x in zip(refs, xr)]
    torch.cat([torch.relu(p) for p in pres], 1).backward(torch.from_numpy(g).double())
    xm = [torch.from_numpy(x).cuda().requires_grad_(True) for x in xs]
    out = ops.bn2d_cat(list(zip(xm, mines)), relu=True)
    out.backward(torch.from_numpy(g).cuda())
    for i in range(2):
        safe = (pres[i].detach().abs() > 2e-5).numpy()
        assert rel(xm[i].grad.cpu().numpy()[safe], xr[i].grad.numpy()[safe]) < 1e-4
        assert rel(mines[i].weight.grad.cpu().numpy(), refs[i].weight.grad.numpy()) < 1e-4
        assert rel(mines[i].bias.grad.cpu().numpy(), refs[i].bias.grad.numpy()) < 1e-4


def test_full_size_stats_epilogue_conv_forward_and_backward_match_the_oracle():
    """64 -> 64 SubM convolution on the stride-4 rulebook of a full-size C3 batch (two 180 k-point clouds -> ~389 k rows, ~6.3 M
    pairs), the launch bench.py prices: output, epilogue moments, data gradient (same table, reversed offsets) and weight gradient
    against the CPU oracle, 1e-4 of each tensor's scale."""
    import os
    from toda_amd import ops
    from toda_amd.pcdet.config import AttrDict, cfg_from_yaml_file
    from toda_amd.pcdet.datasets import SyntheticLidarDataset

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = AttrDict()
    cfg_from_yaml_file(os.path.join(root, "toda_amd/tools/cfgs/models/centerpoint_voxel_waymo.yaml"), cfg)
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
    vc = ds.voxel_cfg
    clouds = [torch.from_numpy(ds[i]["points"]).cuda() for i in range(2)]
    gx, gy, gz = ops.grid_size_xyz(vc["point_cloud_range"], vc["voxel_size"])
    shape = [int(gz) + 1, int(gy), int(gx)]
    steps = [
        {"kind": "subm", "key": "subm1", "ksize": [3, 3, 3], "dilation": [1, 1, 1]},
        {"kind": "conv", "key": "spconv2", "ksize": [3, 3, 3], "stride": [2, 2, 2], "padding": [1, 1, 1]},
        {"kind": "conv", "key": "spconv3", "ksize": [3, 3, 3], "stride": [2, 2, 2], "padding": [1, 1, 1]},
        {"kind": "subm", "key": "subm3", "ksize": [3, 3, 3], "dilation": [1, 1, 1]},
    ]
    _, _, _, plan = ops.build_input_plan(ops._cloud_list(clouds), vc, 2, shape, steps, training=False)
    rb = plan["subm3"]["rb"]
    n = rb.n_out
    assert 300000 < n < 500000
    rng = np.random.default_rng(11)
    x = np.maximum(rng.standard_normal((n, 64)), 0).astype(np.float32)       # post-ReLU activations, as in the step
    w = (rng.standard_normal((64, 3, 3, 3, 64)) * 0.05).astype(np.float32)
    g = rng.standard_normal((n, 64)).astype(np.float32)
    nbr = rb.nbr_fwd.cpu().numpy()
    O.set_threads(min(16, os.cpu_count() or 1))
    y0 = O.spconv_fwd(x, w, nbr)
    xm = torch.from_numpy(x).cuda().requires_grad_(True)
    wm = torch.from_numpy(w).cuda().requires_grad_(True)
    y1, sums = ops.sparse_conv(xm, wm, None, rb, want_stats=True)
    assert sums is not None
    assert rel(y1.detach().cpu().numpy(), y0) < 1e-4
    # the epilogue's moments against a float64 pass over the oracle's output
    part, blocks = sums if isinstance(sums, tuple) else (sums, 0)
    if blocks:
        mom = part[2 * 64:2 * 64 + 2 * 64 * blocks].view(2 * 64, blocks).sum(1).cpu().numpy()
    else:
        mom = part[:2 * 64].cpu().numpy()
    y64 = y0.astype(np.float64)
    assert rel(mom[:64], y64.sum(0)) < 1e-4 and rel(mom[64:], (y64 * y64).sum(0)) < 1e-4
    y1.backward(torch.from_numpy(g).cuda())
    dx0 = O.spconv_dgrad(g, w, nbr, True)
    dw0 = O.spconv_wgrad(x, g, nbr, w.shape)
    assert rel(xm.grad.cpu().numpy(), dx0) < 1e-4
    assert rel(wm.grad.cpu().numpy(), dw0) < 1e-4
