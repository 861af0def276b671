import sys, os, os
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R); sys.path.insert(0, R + '/tests/golden')
import numpy as np, torch
import make_counts as MC
from toda_amd import ops
ds = MC.load_dataset('c3'); vc = ds.voxel_cfg
clouds = [torch.from_numpy(ds[i]['points']).cuda() for i in range(2)]
vox, coords, num = ops.voxelize_batch(clouds, vc['point_cloud_range'], vc['voxel_size'], vc['max_points_per_voxel'], vc['max_num_voxels'])
gx, gy, gz = (int(v) for v in ds.grid_size); shape = [gz + 1, gy, gx]
steps = [dict(kind=k, key=key, **kw) for key, k, kw in MC.PLAN]
for st in steps:
    if st['kind'] == 'conv': st['padding'] = st.pop('pad')
plan = ops.build_index_plan(coords, 2, shape, steps)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, key, cin, cout, table in [("conv_input 5->16", "subm1", 5, 16, "fwd"), ("subm1 16->16", "subm1", 16, 16, "fwd"), ("spconv2 16->32", "spconv2", 16, 32, "fwd")]:
    rb = plan[key]['rb']
    nbr = rb.nbr_fwd
    feat = torch.relu(torch.randn(rb.n_in, cin, device='cuda'))
    w = torch.randn(cout, *rb.ksize, cin, device='cuda') * 0.05
    wp = ops.pack_weight(w, False, False)
    t = timeit(lambda: ops.gather_gemm(feat, wp, nbr, cout))
    pairs = int((nbr >= 0).sum())
    print(f"{name}: rows {nbr.shape[1]} pairs {pairs} ({pairs/nbr.numel():.3f} dense): {t:.1f} us")
