"""Times sparse wgrad on the C3 / C5 levels (full-size synthetic batch): SubM 32->32, 64->64 and the strided 32->64.
TODA_WG_TILE=0/1 selects the kernel (read once per process): run twice."""
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
print("keys", [(k, plan[k]['rb'].kind, plan[k]['rb'].n_out, plan[k]['rb'].k_vol) for k in plan if isinstance(plan[k], dict) and 'rb' in plan[k]])
cases = [('subm2', 32, 32), ('subm3', 64, 64), ('subm4', 64, 64), ('conv3', 32, 64), ('conv4', 64, 64)]
for key, ci, co in cases:
    if key not in plan: continue
    rb = plan[key]['rb']
    n, K = rb.n_out, rb.k_vol
    n_in = int(rb.nbr_fwd.max()) + 1
    feat = torch.randn(n_in, ci, device='cuda'); g = torch.randn(n, co, device='cuda')
    ws = (co,) + ((3, 3, 3) if K == 27 else (3, 1, 1)) + (ci,)
    t = timeit(lambda: ops.wgrad(feat, g, rb.nbr_fwd, ws))
    pairs = int((rb.nbr_fwd >= 0).sum()); fl = 2.0 * pairs * ci * co
    dw = ops.wgrad(feat, g, rb.nbr_fwd, ws)
    # float64 check on three offsets
    err = 0.0
    for k in (0, 13, 26) if K == 27 else (0,):
        m = rb.nbr_fwd[k] >= 0
        ref = g[m].double().T @ feat[rb.nbr_fwd[k][m].long()].double()
        err = max(err, float((dw.reshape(co, K, ci)[:, k].double() - ref).abs().max() / ref.abs().max()))
    print(f"{key}: n_out {n} n_in {n_in} K {K} {ci}->{co} pairs {pairs}: {t:.4f} ms  {fl/t/1e9:.1f} TF/s ({fl/t/1e9/157.3:.3f})  err {err:.1e}  TILE={os.environ.get('TODA_WG_TILE','1')}")
