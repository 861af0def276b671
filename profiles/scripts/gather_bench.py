"""Times the SubM gather-GEMMs of the C3 levels alone (full-size synthetic batch): 32 -> 32 at the stride-2 level, 64 -> 64 at the stride-4
and stride-8 levels.  TODA_GG_ABLATE=1 (read once per process) gives the gathers an empty table: same instruction stream, no row traffic."""
import sys, os
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R); sys.path.insert(0, R + '/tests/golden')
import torch
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
for key, c in (('subm2', 32), ('subm3', 64), ('subm4', 64)):
    rb = plan[key]['rb']
    n = rb.n_out
    feat = torch.randn(n, c, device='cuda')
    w = torch.randn(c, 3, 3, 3, c, device='cuda') * 0.05
    wp = ops.pack_weight(w, False, False)
    t = timeit(lambda: ops.gather_gemm(feat, wp, rb.nbr_fwd, c, None))
    pairs = int((rb.nbr_fwd >= 0).sum()); fl = 2.0 * pairs * c * c
    print(f"{key}: rows {n} pairs {pairs} {c}->{c}: {t:.4f} ms  {fl/t/1e9:.1f} TF/s ({fl/t/1e9/157.3:.3f})", flush=True)
