"""Native (fp32 MFMA) against split (bf16 hi/mid/lo, 6 terms) gather-GEMM on the C3 / C5 levels, alone: time per launch and the error of
both against an fp64 evaluation of the same sums (rms and max, relative to the rms of the exact result).
    python profiles/scripts/split_bench.py [c3|c5] [quick]
"""
import os
import sys

R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R)
sys.path.insert(0, R + '/tests/golden')
import torch

import make_counts as MC
from toda_amd import ops

name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
quick = (len(sys.argv) > 2 and sys.argv[2] == 'quick') or bool(os.environ.get('SPLIT_ONLY'))
ds = MC.load_dataset(name)
vc = ds.voxel_cfg
clouds = [torch.from_numpy(ds[i]['points']).cuda() for i in range(2)]
vox, coords, num = ops.voxelize_batch(clouds, vc['point_cloud_range'], vc['voxel_size'], vc['max_points_per_voxel'], vc['max_num_voxels'])
gx, gy, gz = (int(v) for v in ds.grid_size)
shape = [gz + 1, gy, gx]
steps = [dict(kind=k, key=key, **{a: b for a, b in kw.items()}) for key, k, kw in MC.PLAN]
for st in steps:
    if st['kind'] == 'conv':
        st['padding'] = st.pop('pad')
plan = ops.build_index_plan(coords, 2, shape, steps)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n


def exact(feat, w, nbr, transpose, flip):
    """fp64 sums over the table: out[o] = sum_k feat[nbr[k, o]] @ W_k (W_k = w[:, k, :]^T forward, w[:, K-1-k or k, :] for the data gradient)."""
    K, n_out = nbr.shape
    cout, cin = w.shape[0], w.shape[-1]
    wk = w.reshape(cout, K, cin).double()
    f64 = torch.cat([feat.double(), torch.zeros(1, feat.shape[1], dtype=torch.float64, device=feat.device)], 0)
    out = torch.zeros(n_out, cin if transpose else cout, dtype=torch.float64, device=feat.device)
    for k in range(K):
        idx = nbr[k].long()
        idx = torch.where(idx >= 0, idx, torch.full_like(idx, feat.shape[0]))
        kk = K - 1 - k if flip else k
        m = wk[:, kk, :] if transpose else wk[:, kk, :].t()
        out += f64[idx] @ m
    return out


def run(tag, feat, w, nbr, c_produce, transpose=False, flip=False, order=None, classed=None):
    """classed = (rulebook, (order, cls_sorted)): the class-sorted data gradient of a strided convolution."""
    res = {}

    def call(wp):
        if classed is not None:
            rb, co = classed
            return ops.gather_gemm_classed(feat, wp, nbr, c_produce, co[0], co[1], rb.ksize, rb.geom["stride"], rb.geom["padding"])
        return ops.gather_gemm(feat, wp, nbr, c_produce, None, order=order)

    for mm in ("native", "split"):
        ops.set_matrix_path(mm)
        wp = ops.pack_weight(w, transpose, flip)
        out = call(wp)
        t = timeit(lambda: call(wp), 5 if quick else 20)
        res[mm] = (t, out)
    ref = exact(feat, w, nbr, transpose, flip)
    scale = float(ref.pow(2).mean().sqrt())
    pairs = int((nbr >= 0).sum())
    fl = 2.0 * pairs * feat.shape[1] * c_produce
    line = f"{tag}: rows {nbr.shape[1]} pairs {pairs} {feat.shape[1]}->{c_produce}"
    for mm in ("native", "split"):
        t, out = res[mm]
        d = out.double() - ref
        line += f" | {mm} {t:.4f} ms {fl / t / 1e9:.1f} TF/s rms {float(d.pow(2).mean().sqrt()) / scale:.3e} max {float(d.abs().max()) / scale:.3e}"
    same = bool((res['native'][1] == res['split'][1]).all())
    o2 = None
    ops.set_matrix_path("split")
    wp = ops.pack_weight(w, transpose, flip)
    o2 = call(wp)
    line += f" | split/native time {res['split'][0] / res['native'][0]:.3f} rerun-identical {bool((o2 == res['split'][1]).all())} bit-equal-native {same}"
    print(line, flush=True)
    ops.set_matrix_path("native")


def run_wgrad(tag, feat, g, nbr, wshape):
    """weight gradient, both paths, against fp64: dW[co][k][ci] = sum over the pairs of offset k of feat[i][ci] g[o][co]."""
    K, n_out = nbr.shape
    cout, cin = wshape[0], wshape[-1]
    f64 = torch.cat([feat.double(), torch.zeros(1, cin, dtype=torch.float64, device=feat.device)], 0)
    ref = torch.zeros(cout, K, cin, dtype=torch.float64, device=feat.device)
    for k in range(K):
        idx = nbr[k].long()
        idx = torch.where(idx >= 0, idx, torch.full_like(idx, feat.shape[0]))
        ref[:, k, :] = g.double().t() @ f64[idx]
    scale = float(ref.pow(2).mean().sqrt())
    pairs = int((nbr >= 0).sum())
    fl = 2.0 * pairs * cin * cout
    line = f"{tag}: rows {n_out} pairs {pairs} wgrad {cin}x{cout}"
    outs = {}
    for mm in ("native", "split"):
        ops.set_matrix_path(mm)
        dw = ops.wgrad(feat, g, nbr, wshape)
        t = timeit(lambda: ops.wgrad(feat, g, nbr, wshape), 5 if quick else 20)
        d = dw.reshape(cout, K, cin).double() - ref
        outs[mm] = dw
        line += f" | {mm} {t:.4f} ms {fl / t / 1e9:.1f} TF/s rms {float(d.pow(2).mean().sqrt()) / scale:.3e} max {float(d.abs().max()) / scale:.3e}"
    ops.set_matrix_path("split")
    line += f" | rerun {bool((ops.wgrad(feat, g, nbr, wshape) == outs['split']).all())}"
    print(line, flush=True)
    ops.set_matrix_path("native")


torch.manual_seed(0)
ONLY = os.environ.get("SPLIT_ONLY")          # e.g. subm3: that level's forward alone (counter passes)
WG_ONLY = os.environ.get("SPLIT_WGRAD_ONLY")  # e.g. subm3: that level's weight gradient alone
chan = {'c3': {'subm2': 32, 'subm3': 64, 'subm4': 64}, 'c5': {'subm2': 32, 'subm3': 64, 'subm4': 128}}.get(name, {})
for key, c in chan.items():
    if ONLY and key != ONLY:
        continue
    rb = plan[key]['rb']
    n = rb.n_out
    feat = torch.relu(torch.randn(n, c, device='cuda')) * (1.0 + 3.0 * torch.rand(1, c, device='cuda'))     # post-ReLU-like, channel scales differ
    w = torch.randn(c, 3, 3, 3, c, device='cuda') * 0.05
    if WG_ONLY:
        if key == WG_ONLY:
            run_wgrad(key, feat, torch.randn(n, c, device='cuda'), rb.nbr_fwd, tuple(w.shape))
        continue
    run(key + " fwd", feat, w, rb.nbr_fwd, c)
    if ONLY:
        continue
    g = torch.randn(n, c, device='cuda')
    run(key + " dgrad", g, w, rb.nbr_bwd, c, True, rb.flip_bwd)
    run_wgrad(key, feat, g, rb.nbr_fwd, tuple(w.shape))
for key, cin, cout in {'c5': (('spconv3', 32, 64), ('spconv4', 64, 128))}.get(name, (('spconv3', 32, 64), ('spconv4', 64, 64))):
    if ONLY or WG_ONLY:
        break
    rb = plan[key]['rb']
    feat = torch.relu(torch.randn(rb.n_in, cin, device='cuda'))
    w = torch.randn(cout, 3, 3, 3, cin, device='cuda') * 0.05
    run(key + " fwd", feat, w, rb.nbr_fwd, cout)
    g = torch.randn(rb.n_out, cout, device='cuda')
    run_wgrad(key, feat, g, rb.nbr_fwd, tuple(w.shape))
    run(key + " dgrad(plain)", g, w, rb.nbr_bwd, cin, True, rb.flip_bwd)
    co = rb.class_order()
    if co is not None:
        run(key + " dgrad(classed)", g, w, rb.nbr_bwd, cin, True, rb.flip_bwd, classed=(rb, co))
