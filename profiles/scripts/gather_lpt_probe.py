"""Does the ORDER in which the row blocks of a SubM gather-GEMM are dispatched matter?  Workgroup = 128 rows (4 waves x 2 tiles); cost of a
workgroup ~ its (16-row tile, offset) pairs with at least one neighbour.  Through the kernel's `order` argument (a row permutation that keeps
every 128-row block intact) the blocks are dispatched heaviest first (LPT), lightest first, or in a random order; identity passed the same way
is the control.  The permutation is composed with the inverse of the kernel's blockIdx -> XCD-chunk mapping, so `position in time` is what is
sorted."""
import sys, os
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R); sys.path.insert(0, R + '/tests/golden')
import numpy as np, torch
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
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n): fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n
def xcd_pos(b, nblk, C=32):
    per = 8 * C; full = (nblk // per) * per
    if b >= full: return b
    xcd, local = b & 7, b >> 3
    return ((local // C) * 8 + xcd) * C + local % C
for key, c in (('subm3', 64), ('subm2', 32), ('subm4', 64)):
    rb = plan[key]['rb']; n = rb.n_out
    feat = torch.randn(n, c, device='cuda'); w = torch.randn(c, 3, 3, 3, c, device='cuda') * 0.05
    wp = ops.pack_weight(w, False, False)
    nbr = rb.nbr_fwd
    nb_full = n // 128; nblk = (n + 127) // 128
    hit = (nbr[:, :nb_full * 128] >= 0).view(nbr.shape[0], nb_full * 8, 16).any(2)          # [K, tiles]
    cost = hit.view(nbr.shape[0], nb_full, 8).sum((0, 2)).cpu().numpy()                       # per block
    print(f"{key}: blocks {nblk}, tile-offset pairs per block min/mean/max {cost.min()}/{cost.mean():.1f}/{cost.max()}", flush=True)
    base = timeit(lambda: ops.gather_gemm(feat, wp, nbr, c, None))
    ref = ops.gather_gemm(feat, wp, nbr, c, None)
    rng = np.random.default_rng(0)
    seqs = {'identity': np.arange(nb_full), 'heaviest first': np.argsort(-cost, kind='stable'), 'lightest first': np.argsort(cost, kind='stable'),
            'random': rng.permutation(nb_full)}
    pos_of = np.array([xcd_pos(b, nblk) for b in range(nblk)])                                 # dispatch index b -> row-block position
    for name, seq in seqs.items():
        blocks_at_pos = np.arange(nblk)
        # the b-th workgroup to be dispatched works on position pos_of[b]; give that position the b-th block of the sequence
        full_seq = np.concatenate([seq, np.arange(nb_full, nblk)])
        if name == 'identity':
            blocks_at_pos = np.arange(nblk)
        else:
            disp = [b for b in range(nblk) if pos_of[b] < nb_full]                            # dispatch slots that land on full blocks, in time order
            blocks_at_pos = np.arange(nblk)
            for slot, blk in zip(disp, seq):
                blocks_at_pos[pos_of[slot]] = blk
        order = (blocks_at_pos[:, None] * 128 + np.arange(128)[None, :]).reshape(-1)[:n]
        order[nb_full * 128:] = np.arange(nb_full * 128, n)
        assert np.array_equal(np.sort(order), np.arange(n))
        od = torch.from_numpy(order.astype(np.int32)).cuda()
        t = timeit(lambda: ops.gather_gemm(feat, wp, nbr, c, None, order=od))
        same = torch.equal(ops.gather_gemm(feat, wp, nbr, c, None, order=od), ref)
        print(f"   {name:15s} {t:.4f} ms   (no order argument: {base:.4f})  bit-identical {same}", flush=True)
