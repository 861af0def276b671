"""Diagnostic build only (libtoda_stamps.so = spconv.hip with -DSP_STAMPS=1): per-wave cycle sums of the split gather-GEMM's loop segments,
C3 64 -> 64 @ 389 k rows.  Read the SHARES, not the length (the stamps' fences forbid overlaps the real kernel has).
    TODA_HIP_LIB=$PWD/toda_amd/libtoda_stamps.so python profiles/scripts/split_stamps.py"""
import os, sys
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R); sys.path.insert(0, R + '/tests/golden')
import torch
import make_counts as MC
from toda_amd import ops
ds = MC.load_dataset('c3'); vc = ds.voxel_cfg
clouds = [torch.from_numpy(ds[i]['points']).cuda() for i in range(2)]
vox, coords, num = ops.voxelize_batch(clouds, vc['point_cloud_range'], vc['voxel_size'], vc['max_points_per_voxel'], vc['max_num_voxels'])
gx, gy, gz = (int(v) for v in ds.grid_size); shape = [gz + 1, gy, gx]
steps = [dict(kind=k, key=key, **{a: b for a, b in kw.items()}) for key, k, kw in MC.PLAN]
for st in steps:
    if st['kind'] == 'conv': st['padding'] = st.pop('pad')
plan = ops.build_index_plan(coords, 2, shape, steps)
ops.set_matrix_path("split")
for key, c in (('subm3', 64), ('subm2', 32)):
    rb = plan[key]['rb']; n = rb.n_out
    feat = torch.relu(torch.randn(n, c, device='cuda'))
    w = torch.randn(c, 3, 3, 3, c, device='cuda') * 0.05
    wp = ops.pack_weight(w, False, False)
    for _ in range(3):
        out, sums, blocks = ops.gather_gemm_with_stats(feat, wp, rb.nbr_fwd, c, None, partials=True)
    torch.cuda.synchronize()
    waves = (n + 31) // 32
    t = sums.view(torch.int64)[:waves * 6].view(waves, 6).double()
    names = ["split", "issue loads", "matrix phase", "wait slice (vmcnt)", "barrier", "top (flags, loop)"]
    tot = t.sum(1)
    print(f"{key}: waves {waves}, cycles per wave {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f}), per offset {tot.mean() / 27:.0f}")
    for i, nm in enumerate(names):
        print(f"   {nm:22s} {t[:, i].mean() / 27:8.0f} cycles per offset  {t[:, i].sum() / tot.sum():.3f}")
