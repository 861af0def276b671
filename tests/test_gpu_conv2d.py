"""Dense 3x3 convolution (Winograd F(4x4,3x3) on the fp32 matrix cores, toda_amd/csrc/conv2d.hip) against a plain PyTorch
reference of the same op: torch.nn.functional.conv2d in float64 on the CPU (the reference's layers are torch.nn.Conv2d,
base_bev_backbone.py:37-58, center_head.py:20-28,73-80).  Tolerance: 2e-5 of the output scale (north star 1e-3 fp32)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (batch, cin, cout, H, W): full tiles, partial tiles in x / y (W = 4k + 2), ragged last tile block, bias, neck / head channel pairs
SHAPES = [
    (1, 32, 32, 8, 8),
    (2, 32, 64, 12, 16),
    (2, 64, 32, 22, 26),
    (1, 128, 128, 47, 46),
    (3, 256, 128, 10, 94),
    (2, 512, 64, 20, 20),
    (1, 64, 320, 9, 30),
]


def _ref(x, w, b):
    return F.conv2d(x.double().cpu(), w.double().cpu(), None if b is None else b.double().cpu(), padding=1)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("with_bias", [False, True])
def test_conv3x3_forward_matches_fp64_reference(shape, with_bias):
    from toda_amd import ops

    b, cin, cout, h, w_ = shape
    g = torch.Generator().manual_seed(b * 1000 + cin + cout + h)
    x = torch.relu(torch.randn((b, cin, h, w_), generator=g)).cuda()
    w = (torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5).cuda()
    bias = torch.randn((cout,), generator=g).cuda() if with_bias else None
    y = ops.conv3x3(x, w, bias)
    ref = _ref(x, w, bias)
    err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    assert y.shape == ref.shape and err < 2e-5, err


@pytest.mark.parametrize("shape", SHAPES[1:5])
def test_conv3x3_backward_matches_fp64_autograd(shape):
    from toda_amd import ops

    b, cin, cout, h, w_ = shape
    g = torch.Generator().manual_seed(7)
    x = torch.randn((b, cin, h, w_), generator=g).cuda().requires_grad_(True)
    w = (torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5).cuda().requires_grad_(True)
    bias = torch.randn((cout,), generator=g).cuda().requires_grad_(True)
    gy = torch.randn((b, cout, h, w_), generator=g).cuda()
    ops.conv3x3(x, w, bias).backward(gy)
    xd, wd, bd = (t.detach().double().cpu().requires_grad_(True) for t in (x, w, bias))
    F.conv2d(xd, wd, bd, padding=1).backward(gy.double().cpu())
    for name, got, ref in (("dx", x.grad, xd.grad), ("dw", w.grad, wd.grad), ("db", bias.grad, bd.grad)):
        err = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
        assert err < 5e-5, (name, err)


def test_module_routing_keeps_unsupported_layers_on_torch():
    """run_dense_sequential: 3x3 / stride-1 layers with channels % 32 == 0 take the HIP kernel, the rest (stride 2, 1x1,
    tiny head outputs) stay on torch; results equal torch's within fp32 rounding."""
    from toda_amd import ops

    torch.manual_seed(0)
    seq = torch.nn.Sequential(torch.nn.ZeroPad2d(1), torch.nn.Conv2d(32, 64, 3, stride=1, padding=0, bias=False), torch.nn.BatchNorm2d(64),
                              torch.nn.ReLU(), torch.nn.Conv2d(64, 64, 3, padding=1), torch.nn.ReLU(),
                              torch.nn.ZeroPad2d(1), torch.nn.Conv2d(64, 32, 3, stride=2, padding=0), torch.nn.Conv2d(32, 3, 3, padding=1)).cuda()
    x = torch.randn(2, 32, 20, 24, device="cuda")
    calls = []
    orig = ops.conv3x3
    ops.conv3x3 = lambda *a: (calls.append(1), orig(*a))[1]
    try:
        y = ops.run_dense_sequential(seq, x)
    finally:
        ops.conv3x3 = orig
    assert len(calls) == 2
    ref = seq(x)
    assert float((y - ref).abs().max()) < 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize("geom", [(2, 64, 20, 24), (1, 16, 7, 188), (3, 8, 33, 4)])
def test_narrow_output_conv_group_matches_fp64(geom):
    """toda_conv3x3_narrow_{fwd,dgrad,wgrad}: the 64 -> {2, 1, 3, 2, 3} output convolutions of the five CenterHead branches
    (reference center_head.py:20-28) in one launch per direction, against torch.nn.functional.conv2d in float64."""
    from toda_amd import ops

    b, cin, h, w_ = geom
    couts = [2, 1, 3, 2, 3]
    g = torch.Generator().manual_seed(cin + h)
    convs = []
    for co in couts:
        c = torch.nn.Conv2d(cin, co, 3, padding=1, bias=True)
        c.weight.data = torch.randn(c.weight.shape, generator=g) * (2.0 / (9 * cin)) ** 0.5
        c.bias.data = torch.randn(co, generator=g)
        convs.append(c.cuda())
    xs = [torch.randn((b, cin, h, w_), generator=g).cuda().requires_grad_(True) for _ in couts]
    gys = [torch.randn((b, co, h, w_), generator=g).cuda() for co in couts]
    assert all(ops.conv3x3_narrow_supported(x, c) for x, c in zip(xs, convs))
    ys = ops.conv3x3_narrow_group(xs, convs)
    torch.autograd.backward(ys, gys)
    for x, c, y, gy in zip(xs, convs, ys, gys):
        xd = x.detach().double().cpu().requires_grad_(True)
        wd, bd = c.weight.detach().double().cpu().requires_grad_(True), c.bias.detach().double().cpu().requires_grad_(True)
        ref = F.conv2d(xd, wd, bd, padding=1)
        ref.backward(gy.double().cpu())
        for name, got, want in (("y", y, ref), ("dx", x.grad, xd.grad), ("dw", c.weight.grad, wd.grad), ("db", c.bias.grad, bd.grad)):
            err = float((got.detach().double().cpu() - want.detach()).abs().max() / want.detach().abs().max())
            assert err < 2e-5, (name, err)


def test_narrow_output_conv_group_on_channel_slices():
    """Same kernels with the branch inputs given as channel slices of one [B, n C, H, W] tensor (image stride argument): outputs
    and all gradients equal the separate-tensor call bit for bit, the input gradient comes back as one wide tensor."""
    from toda_amd import ops

    b, cin, h, w_ = 2, 32, 12, 20
    couts = [2, 1, 3]
    g = torch.Generator().manual_seed(5)
    convs_a, convs_b = [], []
    for co in couts:
        c = torch.nn.Conv2d(cin, co, 3, padding=1, bias=True)
        c.weight.data = torch.randn(c.weight.shape, generator=g) * 0.1
        convs_a.append(c.cuda())
        convs_b.append(__import__("copy").deepcopy(c).cuda())
    wide = torch.randn((b, cin * len(couts), h, w_), generator=g).cuda().requires_grad_(True)
    xs = [wide.detach()[:, i * cin:(i + 1) * cin].contiguous().requires_grad_(True) for i in range(len(couts))]
    gys = [torch.randn((b, co, h, w_), generator=g).cuda() for co in couts]
    ya = ops.conv3x3_narrow_group(xs, convs_a)
    yb = ops.conv3x3_narrow_group_fused(wide, convs_b)
    torch.autograd.backward(ya, gys)
    torch.autograd.backward(yb, gys)
    for i in range(len(couts)):
        assert torch.equal(ya[i], yb[i])
        assert torch.equal(xs[i].grad, wide.grad[:, i * cin:(i + 1) * cin])
        assert torch.equal(convs_a[i].weight.grad, convs_b[i].weight.grad)
        assert torch.equal(convs_a[i].bias.grad, convs_b[i].bias.grad)


def test_separate_head_fused_hidden_layer_matches_branchwise_modules():
    """SeparateHead.forward (one 64 -> 5 x 64 convolution + one BatchNorm over 320 channels + sliced output convolutions)
    against the plain nn.Sequential branches of the reference (center_head.py:20-41): outputs, every parameter gradient, the
    input gradient, and the BatchNorm running statistics / step counters of every branch."""
    import copy

    from toda_amd import ops
    from toda_amd.pcdet.models.dense_heads.center_head import SeparateHead

    torch.manual_seed(3)
    head_dict = {"center": dict(out_channels=2, num_conv=2), "center_z": dict(out_channels=1, num_conv=2), "dim": dict(out_channels=3, num_conv=2),
                 "rot": dict(out_channels=2, num_conv=2), "hm": dict(out_channels=3, num_conv=2)}
    head = SeparateHead(64, head_dict).cuda().train()
    ref = copy.deepcopy(head)
    x = torch.randn(2, 64, 24, 28, device="cuda")
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    with ops.bn_counter_scope():
        out = head(xa)
    want = {name: getattr(ref, name)(xb) for name in head_dict}
    gys = {name: torch.randn_like(want[name]) for name in head_dict}
    torch.autograd.backward([out[n] for n in head_dict], [gys[n] for n in head_dict])
    torch.autograd.backward([want[n] for n in head_dict], [gys[n] for n in head_dict])
    for name in head_dict:
        assert float((out[name] - want[name]).abs().max()) < 2e-4 * float(want[name].abs().max()), name
    assert float((xa.grad - xb.grad).abs().max()) < 1e-3 * float(xb.grad.abs().max())
    for (k, p), (_, q) in zip(head.named_parameters(), ref.named_parameters()):
        assert float((p.grad - q.grad).abs().max()) <= 1e-3 * float(q.grad.abs().max()) + 1e-6, k
    for (k, p), (_, q) in zip(head.named_buffers(), ref.named_buffers()):
        assert torch.allclose(p.float(), q.float(), rtol=1e-4, atol=1e-6), k
    assert set(head.state_dict()) == set(ref.state_dict())
    # a buffer replaced from outside (module.to(), load) is re-aliased without losing its value
    head.center[0][1].running_mean = head.center[0][1].running_mean.clone() + 1.0
    ref.center[0][1].running_mean += 1.0
    with ops.bn_counter_scope():
        head(x)
    for name in head_dict:
        getattr(ref, name)(x)
    for (k, p), (_, q) in zip(head.named_buffers(), ref.named_buffers()):
        assert torch.allclose(p.float(), q.float(), rtol=1e-4, atol=1e-6), k


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 128, 188, 188), (2, 256, 94, 94), (1, 64, 180, 180), (4, 32, 64, 64), (2, 320, 36, 20), (2, 7, 2, 2),
                                   (2, 64, 47, 45), (1, 5, 191, 193), (4, 3, 7, 9), (4, 16, 180, 180), (2, 12, 190, 194)])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("split", [True, False])
def test_single_pass_batchnorm2d_matches_torch(shape, relu, split, monkeypatch):
    """toda_bn2d_fwd / _bwd against nn.BatchNorm2d (+ReLU) in fp64 on the same tensors: output, running statistics, input / weight /
    bias gradients (reference base_bev_backbone.py:37-58: BatchNorm2d(eps 1e-3, momentum 0.01) + ReLU).  split: one workgroup
    per (channel, sample) plane with the partner exchange (what ops passes a workspace for); otherwise one workgroup per channel."""
    from toda_amd import ops
    if not split:
        if shape[0] == 4 and shape[2] * shape[3] > 16384:
            pytest.skip("batch 4 above 128 x 128 needs the partner exchange")
        monkeypatch.setattr(ops, "_bn2d_sync", lambda device: (None, 0))
    torch.manual_seed(3)
    b, c, h, w = shape
    bn = torch.nn.BatchNorm2d(c, eps=1e-3, momentum=0.01).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-1, 1)
        bn.running_var.uniform_(0.5, 2)
    ref = torch.nn.BatchNorm2d(c, eps=1e-3, momentum=0.01).cuda().double().train()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v.clone() for k, v in bn.state_dict().items()})
    x = (torch.randn(shape, device="cuda") * 2.0 + 0.7).requires_grad_(True)
    xr = x.detach().double().requires_grad_(True)
    assert ops.bn2d_supported(x, bn)
    y = ops.bn2d(x, bn, relu)
    yr = ref(xr)
    yr = torch.relu(yr) if relu else yr
    gy = torch.randn_like(y)
    y.backward(gy)
    yr.backward(gy.double())
    tol = 2e-5
    assert float((y.double() - yr).abs().max()) <= tol * max(1.0, float(yr.abs().max()))
    assert int(bn.num_batches_tracked) == 1
    assert torch.allclose(bn.running_mean.double(), ref.running_mean, rtol=1e-6, atol=1e-6)
    assert torch.allclose(bn.running_var.double(), ref.running_var, rtol=1e-5, atol=1e-6)
    # elements whose pre-activation is within rounding of zero may take the other side of the ReLU: compare away from it
    if relu:
        safe = (ref(xr.detach()).abs() > 1e-4)
    else:
        safe = torch.ones_like(yr, dtype=torch.bool)
    gx, gxr = x.grad.double(), xr.grad
    assert float(((gx - gxr) * safe).abs().max()) <= 2e-4 * max(1e-3, float(gxr.abs().max()))
    assert torch.allclose(bn.weight.grad.double(), ref.weight.grad, rtol=2e-4, atol=2e-3)
    assert torch.allclose(bn.bias.grad.double(), ref.bias.grad, rtol=2e-4, atol=2e-3)


# ------------------------------------------------------------------ stride-2 3x3 convolution and transposed-convolution deblocks
def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.mark.parametrize("b,cin,cout,h,w", [(2, 128, 256, 188, 188), (2, 32, 64, 16, 24), (1, 40, 24, 10, 6), (3, 128, 128, 46, 50)])
def test_conv3x3_stride2_matches_conv2d_fp64(b, cin, cout, h, w):
    """ZeroPad2d(1) + Conv2d(3, stride 2) (reference base_bev_backbone.py:32-36) through toda_conv3x3s2_*: forward, data gradient
    (four parity classes), weight gradient against torch conv2d in float64; bit-reproducible."""
    from toda_amd import ops

    torch.manual_seed(1)
    x = torch.randn(b, cin, h, w, device="cuda", requires_grad=True)
    wt = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.05).requires_grad_(True)
    gy = torch.randn(b, cout, h // 2, w // 2, device="cuda")
    y = ops.conv3x3s2(x, wt)
    y.backward(gy)
    xr, wr = x.detach().double().requires_grad_(True), wt.detach().double().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wr, stride=2, padding=1)
    yr.backward(gy.double())
    assert y.shape == yr.shape
    assert _rel(y, yr) < 2e-6 and _rel(x.grad, xr.grad) < 2e-6 and _rel(wt.grad, wr.grad) < 5e-6, (_rel(y, yr), _rel(x.grad, xr.grad), _rel(wt.grad, wr.grad))
    g1 = (x.grad.clone(), wt.grad.clone())
    x.grad = wt.grad = None
    y2 = ops.conv3x3s2(x, wt)
    y2.backward(gy)
    assert torch.equal(y, y2) and torch.equal(g1[0], x.grad) and torch.equal(g1[1], wt.grad)


@pytest.mark.parametrize("b,cin,cout,h,w,s", [(2, 128, 256, 188, 188, 1), (2, 256, 256, 94, 94, 2), (2, 64, 32, 8, 12, 2), (1, 24, 40, 7, 5, 2),
                                              (3, 40, 24, 9, 11, 1)])
def test_deconv_matches_conv_transpose2d_fp64(b, cin, cout, h, w, s):
    """ConvTranspose2d(kernel = stride = s) (reference base_bev_backbone.py:47-66) through toda_deconv_*: forward with the pixel
    shuffle in the store, data gradient with the un-shuffle in the gather, weight gradient, against torch in float64."""
    from toda_amd import ops

    torch.manual_seed(2)
    x = torch.randn(b, cin, h, w, device="cuda", requires_grad=True)
    wt = (torch.randn(cin, cout, s, s, device="cuda") * 0.05).requires_grad_(True)
    gy = torch.randn(b, cout, h * s, w * s, device="cuda")
    y = ops.deconv(x, wt, s)
    y.backward(gy)
    xr, wr = x.detach().double().requires_grad_(True), wt.detach().double().requires_grad_(True)
    yr = torch.nn.functional.conv_transpose2d(xr, wr, stride=s)
    yr.backward(gy.double())
    assert y.shape == yr.shape
    assert _rel(y, yr) < 2e-6 and _rel(x.grad, xr.grad) < 2e-6 and _rel(wt.grad, wr.grad) < 5e-6, (_rel(y, yr), _rel(x.grad, xr.grad), _rel(wt.grad, wr.grad))
    g1 = (x.grad.clone(), wt.grad.clone())
    x.grad = wt.grad = None
    ops.deconv(x, wt, s).backward(gy)
    assert torch.equal(g1[0], x.grad) and torch.equal(g1[1], wt.grad)


def test_bn2d_cat_equals_batchnorm_relu_cat():
    """ops.bn2d_cat (toda_bn2d_fwd_into / toda_bn2d_bwd_from: every BatchNorm2d + ReLU writes its channel slice of the concatenated map
    and reads the slice of its gradient in place) against torch.cat([relu(bn(x))]) in float64: forward, input / weight / bias gradients,
    running statistics; the per-channel and the per-plane (partner-exchange) kernels (C3's 2 x 256 x 188 x 188 takes the latter)."""
    from toda_amd import ops

    for shapes in ([(2, 32, 24, 20), (2, 64, 24, 20)], [(2, 256, 188, 188), (2, 256, 188, 188)], [(1, 32, 7, 5), (1, 96, 7, 5), (1, 16, 7, 5)]):
        torch.manual_seed(len(shapes))
        xs = [torch.randn(s, device="cuda") * 2 + 0.3 for s in shapes]
        bns = [torch.nn.BatchNorm2d(s[1], eps=1e-3, momentum=0.01).cuda().train() for s in shapes]
        refs = [torch.nn.BatchNorm2d(s[1], eps=1e-3, momentum=0.01).cuda().double().train() for s in shapes]
        for bn, rf in zip(bns, refs):
            with torch.no_grad():
                bn.weight.uniform_(0.5, 1.5), bn.bias.uniform_(-0.5, 0.5)
                rf.weight.copy_(bn.weight), rf.bias.copy_(bn.bias)
        if not all(ops.bn2d_supported(x, bn) for x, bn in zip(xs, bns)):
            continue
        xa = [x.clone().requires_grad_(True) for x in xs]
        xr = [x.double().requires_grad_(True) for x in xs]
        out = ops.bn2d_cat(list(zip(xa, bns)), relu=True)
        ref = torch.cat([torch.relu(rf(x)) for rf, x in zip(refs, xr)], dim=1)
        g = torch.randn_like(out)
        out.backward(g)
        ref.backward(g.double())
        assert float((out.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
        for x, r, bn, rf in zip(xa, xr, bns, refs):
            assert float((x.grad.double() - r.grad).abs().max()) <= 5e-6 * float(r.grad.abs().max())
            assert float((bn.weight.grad.double() - rf.weight.grad).abs().max()) <= 5e-6 * float(rf.weight.grad.abs().max())
            assert float((bn.bias.grad.double() - rf.bias.grad).abs().max()) <= 5e-6 * float(rf.bias.grad.abs().max())
            assert torch.allclose(bn.running_mean.double(), rf.running_mean, rtol=1e-5, atol=1e-7) and torch.allclose(bn.running_var.double(), rf.running_var, rtol=1e-5, atol=1e-7)
            assert int(bn.num_batches_tracked) == 1


_GANG_CHILD = r"""
import sys, torch
import torch.nn.functional as F
sys.path.insert(0, sys.argv[1])
from toda_amd import ops
torch.manual_seed(3)
worst = 0.0
for (B, ci, co, H, W) in [(2, 256, 256, 94, 94), (1, 128, 128, 47, 46), (2, 512, 64, 20, 20), (1, 64, 320, 9, 30), (3, 256, 128, 10, 94)]:
    x = torch.randn(B, ci, H, W, device="cuda"); w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    y = ops.conv3x3(x, w)
    ref = F.conv2d(x.double().cpu(), w.double().cpu(), padding=1)
    worst = max(worst, float((y.double().cpu() - ref).abs().max() / ref.abs().max()))
print("WORST", worst)
"""


@pytest.mark.parametrize("gang,kb", [("-1", "2560"), ("1", "600"), ("1", "1280"), ("1", "5120"), ("0", "2560")])
def test_winograd_forward_is_the_same_for_every_gang_size(gang, kb):
    """The stream-K sequence of wino_fwd_ws_kernel is (channel-block group, tile block, chunk) with gangs sized by an L2 budget (round 4): all
    block counts per gang - every block (round 3's form), 1, 2, 4, none - give the convolution (the switch is read once: child process)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TODA_WINO_GANG=gang, TODA_WINO_GANG_KB=kb)
    p = subprocess.run([sys.executable, "-c", _GANG_CHILD, root], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert float(p.stdout.strip().split("WORST")[-1]) < 2e-5
