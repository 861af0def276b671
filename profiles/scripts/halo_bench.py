"""Times the halo kernel against the per-offset kernel on the C3 levels (full-size synthetic batch)."""
import sys, os, time
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R); sys.path.insert(0, R + '/tests/golden')
import numpy as np, torch
import make_counts as MC
from toda_amd import ops, lib as L
name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
ds = MC.load_dataset(name); vc = ds.voxel_cfg
clouds = [torch.from_numpy(ds[i]['points']).cuda() for i in range(2)]
vox, coords, num = ops.voxelize_batch(clouds, vc['point_cloud_range'], vc['voxel_size'], vc['max_points_per_voxel'], vc['max_num_voxels'])
gx, gy, gz = (int(v) for v in ds.grid_size); shape = [gz + 1, gy, gx]
steps = [dict(kind=k, key=key, **{a: b for a, b in kw.items()}) for key, k, kw in MC.PLAN]
for st in steps:
    if st['kind'] == 'conv': st['padding'] = st.pop('pad')
plan = ops.build_index_plan(coords, 2, shape, steps)
levels = {'subm3': 64} if os.environ.get('TODA_HALO_ABLATE', '0') not in ('0', '32') else {'subm2': 32, 'subm3': 64, 'subm4': 64}
idx, sh = coords, shape
lvl_idx = {}
for st in steps:
    if st['kind'] == 'subm': lvl_idx[st['key']] = (idx, sh)
    else:
        e = plan[st['key']]; idx, sh = e['out_indices'], e['out_shape']
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n): fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n
for key, c in levels.items():
    rb = plan[key]['rb']; ind, shp = lvl_idx[key]
    n = rb.n_out; K = rb.k_vol
    feat = torch.randn(n, c, device='cuda')
    w = torch.randn(c, 3, 3, 3, c, device='cuda') * 0.05
    wp = ops.pack_weight(w, False, False)
    t0 = time.time(); hp = ops.build_halo_plan(rb, ind, 2, shp, c); torch.cuda.synchronize(); t_plan = time.time() - t0
    rb.halo.clear()
    t_plan2 = timeit(lambda: (rb.halo.clear(), ops.build_halo_plan(rb, ind, 2, shp, c)), 5)
    a = timeit(lambda: ops.gather_gemm(feat, wp, rb.nbr_fwd, c))
    b = timeit(lambda: ops.gather_gemm_halo(feat, wp, rb.nbr_fwd, c, hp))
    pairs = int((rb.nbr_fwd >= 0).sum())
    fl = 2.0 * pairs * c * c
    R, umax = 128, 320
    nb = (n + R - 1) // R
    al = lambda v: (v + 255) // 256 * 256
    o2 = al(nb * R * 4) + al(nb * umax * 4)
    lid = hp[o2:o2 + nb * K * R * 2].cpu().numpy().view(np.uint16).reshape(nb, K, R // 16, 16)
    hit = (lid != 0xFFFF).any(3)
    print(f"   plan: executed (tile, offset) fraction {hit.mean():.3f}, spilled ids {(lid == 0xFFFE).sum()}, useful {(lid != 0xFFFF).mean() / hit.mean():.3f}")
    r_, g_ = ops.gather_gemm(feat, wp, rb.nbr_fwd, c), ops.gather_gemm_halo(feat, wp, rb.nbr_fwd, c, hp)
    if os.environ.get('TODA_HALO_ABLATE', '0') in ('0', '32'):
        err = (r_ - g_).abs().max(1).values / float(r_.abs().max())
        bad = torch.nonzero(err > 1e-5).flatten()
        if len(bad):
            order = hp[:nb * R * 4].cpu().numpy().view(np.int32)
            posof = np.empty(n, np.int64); valid = order >= 0; posof[order[valid]] = np.nonzero(valid)[0]
            bb = np.unique(posof[bad.cpu().numpy()] // R)
            spill_blocks = np.nonzero((lid == 0xFFFE).reshape(nb, -1).any(1))[0]
            ur = hp[al(nb * R * 4):al(nb * R * 4) + nb * umax * 4].cpu().numpy().view(np.int32).reshape(nb, umax)
            lid2 = lid.reshape(nb, K, R)
            for bblk in bb[:4]:
                rows_bad = [int(x) for x in bad.cpu().numpy() if posof[int(x)] // R == bblk]
                print(f"   block {bblk}: nu {(ur[bblk] >= 0).sum()} spill lids {(lid2[bblk] == 0xFFFE).sum()} spill rows {np.nonzero((lid2[bblk] == 0xFFFE).any(0))[0][:12]} bad rows at pos {[int(posof[x] % R) for x in rows_bad]}")
                x = rows_bad[0]; pos_ = int(posof[x] % R)
                ks = np.nonzero(lid2[bblk][:, pos_] == 0xFFFE)[0]
                contrib = sum((feat[int(rb.nbr_fwd[k, x])] @ w[:, k // 9, (k // 3) % 3, k % 3, :].T) for k in ks) if len(ks) else 0
                print(f"      row {x}: spilled offsets {ks}; |ref - got| {float((r_[x] - g_[x]).abs().max()):.3e}; |ref - got - spilled contribution| {float((r_[x] - g_[x] - contrib).abs().max()) if len(ks) else -1:.3e}")
            print(f"   MISMATCH: {len(bad)} rows in {len(bb)} blocks; blocks with spills {len(spill_blocks)}; bad blocks that spill {np.isin(bb, spill_blocks).sum()}; max err {float(err.max()):.2e}; first bad blocks {bb[:8]} positions in block {posof[bad.cpu().numpy()][:8] % R}")
    print(f"{key}: n {n} pairs {pairs} c {c}: per-offset {a:.4f} ms ({fl/a/1e9:.1f} TF/s, {fl/a/1e9/157.3:.3f}) | halo {b:.4f} ms ({fl/b/1e9:.1f} TF/s, {fl/b/1e9/157.3:.3f}) | plan build {t_plan2:.3f} ms")
