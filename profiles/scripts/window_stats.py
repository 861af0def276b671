"""For the submanifold tables of C3 / C5: per block of B consecutive output rows and per (dz, dy) line group (three offsets dx = -1, 0, +1), the
contiguous range of input rows [min id, max id] its neighbours fall into (adding a constant offset preserves the canonical (b, z, y, x) order, so
the neighbours of consecutive rows at a fixed offset are an increasing sequence).  Range length against the block's pair count decides whether
staging the range once in LDS moves fewer bytes than gathering per offset.
    python profiles/scripts/window_stats.py [c3|c5] [block rows]"""
import os, sys
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R); sys.path.insert(0, R + '/tests/golden')
import torch
import make_counts as MC
from toda_amd import ops
name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ds = MC.load_dataset(name); vc = ds.voxel_cfg
clouds = [torch.from_numpy(ds[i]['points']).cuda() for i in range(2)]
vox, coords, num = ops.voxelize_batch(clouds, vc['point_cloud_range'], vc['voxel_size'], vc['max_points_per_voxel'], vc['max_num_voxels'])
gx, gy, gz = (int(v) for v in ds.grid_size); shape = [gz + 1, gy, gx]
steps = [dict(kind=k, key=key, **{a: b for a, b in kw.items()}) for key, k, kw in MC.PLAN]
for st in steps:
    if st['kind'] == 'conv': st['padding'] = st.pop('pad')
plan = ops.build_index_plan(coords, 2, shape, steps)
for key in ('subm1', 'subm2', 'subm3', 'subm4'):
    if key not in plan: continue
    nbr = plan[key]['rb'].nbr_fwd.long()            # [27, N]
    K, N = nbr.shape
    nb = (N + B - 1) // B
    pad = nb * B - N
    big = torch.full((K, pad), -1, dtype=torch.long, device=nbr.device)
    t = torch.cat([nbr, big], 1).view(9, 3, nb, B)   # group (dz, dy), dx, block, row
    valid = t >= 0
    lo = torch.where(valid, t, torch.full_like(t, 1 << 40)).amin(dim=(1, 3))
    hi = torch.where(valid, t, torch.full_like(t, -1)).amax(dim=(1, 3))
    pairs = valid.sum(dim=(1, 3)).float()
    ln = (hi - lo + 1).clamp(min=0).float()
    ln = torch.where(pairs > 0, ln, torch.zeros_like(ln))
    print(f"{key}: rows {N}, blocks of {B}: {nb}; per (block, line group): pairs mean {pairs.mean():.1f}, range length mean {ln.mean():.1f} "
          f"p50 {ln.flatten().quantile(0.5):.0f} p90 {ln.flatten().quantile(0.9):.0f} p99 {ln.flatten().quantile(0.99):.0f} max {ln.max():.0f}; "
          f"over 160: {(ln > 160).float().mean():.4f}, over 192: {(ln > 192).float().mean():.4f}, over 256: {(ln > 256).float().mean():.4f}; "
          f"staged rows / gathered rows = {ln.sum() / pairs.sum():.3f} (capped at 192, rest gathered: {(torch.where(ln <= 192, ln, pairs).sum() / pairs.sum()):.3f})")
