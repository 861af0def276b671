"""Narrow K = 27 layers of the C3 batch: the output-stationary kernels against the compacting kernel."""
import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + '/tests/golden')
import numpy as np, torch
import make_counts as MC
from toda_amd import ops
name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
ds = MC.load_dataset(name); vc = ds.voxel_cfg
clouds = [torch.from_numpy(ds[i]['points']).cuda() for i in range(2)]
vox, coords, num = ops.voxelize_batch(clouds, vc['point_cloud_range'], vc['voxel_size'], vc['max_points_per_voxel'], vc['max_num_voxels'])
gx, gy, gz = (int(v) for v in ds.grid_size); shape = [gz + 1, gy, gx]
steps = [dict(kind=k, key=key, **{a: b for a, b in kw.items()}) for key, k, kw in MC.PLAN]
for st in steps:
    if st['kind'] == 'conv': st['padding'] = st.pop('pad')
plan = ops.build_index_plan(coords, 2, shape, steps)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n): fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n
# (key, table, cin(gathered), cout(produced), transpose, flip)
cases = [('subm1', 'fwd', 5, 16, False, False), ('subm1', 'fwd', 16, 16, False, False), ('subm1', 'bwd', 16, 16, True, True),
         ('spconv2', 'fwd', 16, 32, False, False), ('spconv2', 'bwd', 32, 16, True, False),
         ('subm2', 'fwd', 32, 32, False, False), ('subm2', 'bwd', 32, 32, True, True)]
for key, tab, cg, cp, tr, fl in cases:
    rb = plan[key]['rb']
    nbr = rb.nbr_fwd if tab == 'fwd' else rb.nbr_bwd
    K, n_out = nbr.shape
    n_in = int(nbr.max()) + 1
    feat = torch.randn(n_in, cg, device='cuda')
    w = torch.randn((cp, 3, 3, 3, cg) if not tr else (cg, 3, 3, 3, cp), device='cuda') * 0.1
    wp = ops.pack_weight(w, tr, fl)
    bias = torch.randn(cp, device='cuda') if not tr else None
    co = rb.class_order() if (tab == 'bwd' and rb.kind == 'conv') else None
    if co is not None:
        old = lambda: ops.gather_gemm_classed(feat, wp, nbr, cp, co[0], co[1], rb.ksize, rb.geom["stride"], rb.geom["padding"])
    else:
        old = lambda: ops.gather_gemm(feat, wp, nbr, cp, bias)
    new = lambda: ops.gather_gemm_compact(feat, w, nbr, cp, bias, tr, fl)
    a, b = old(), new()
    err = float((a - b).abs().max() / a.abs().max())
    ta, tb = timeit(old), timeit(new)
    pairs = int((nbr >= 0).sum())
    byts = (n_in * cg + n_out * cp) * 4 + nbr.numel() * 4
    print(f"{key} {tab} {cg}->{cp}: n_out {n_out} pairs {pairs} ({pairs/n_out:.2f}/row): old {ta*1e3:.1f} us  compact {tb*1e3:.1f} us ({byts/tb/1e6:.0f} GB/s alg)  rel err {err:.1e}")
