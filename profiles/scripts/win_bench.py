"""(Runs with profiles/r05_window_kernel.patch applied: the kernel lost and is not in the tree.)
Line-window gather-GEMM (gg_win_kernel) against the per-offset split kernel on the submanifold levels of C3 / C5: time, max difference,
reproducibility.    python profiles/scripts/win_bench.py [c3|c5]"""
import os, sys
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
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(n): fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / n
torch.manual_seed(0)
ops.set_matrix_path("split")
chan = {'c3': {'subm2': 32, 'subm3': 64, 'subm4': 64}, 'c5': {'subm2': 32, 'subm3': 64, 'subm4': 128}}[name]
for key, c in chan.items():
    rb = plan[key]['rb']; n = rb.n_out
    feat = torch.relu(torch.randn(n, c, device='cuda')) * (1.0 + 3.0 * torch.rand(1, c, device='cuda'))
    w = torch.randn(c, 3, 3, 3, c, device='cuda') * 0.05
    bias = torch.randn(c, device='cuda')
    wp = ops.pack_weight(w, False, False)
    t_w = timeit(lambda: ops.rulebook_windows(rb.nbr_fwd))
    win = ops.rulebook_windows(rb.nbr_fwd)
    ref = ops.gather_gemm(feat, wp, rb.nbr_fwd, c, bias)
    got = ops.gather_gemm_win(feat, wp, rb.nbr_fwd, c, win, bias)
    again = ops.gather_gemm_win(feat, wp, rb.nbr_fwd, c, win, bias)
    d = float((got - ref).abs().max()); sc = float(ref.abs().max())
    a = timeit(lambda: ops.gather_gemm(feat, wp, rb.nbr_fwd, c, bias)); b = timeit(lambda: ops.gather_gemm_win(feat, wp, rb.nbr_fwd, c, win, bias))
    o1, s1 = ops.gather_gemm_with_stats(feat, wp, rb.nbr_fwd, c, bias)
    o2, s2 = ops.gather_gemm_win(feat, wp, rb.nbr_fwd, c, win, bias, stats="fold")
    ds_ = float((s1[:2 * c] - s2[:2 * c]).abs().max() / s1[:2 * c].abs().max())
    wlen = win[:, :, 1].float()
    print(f"{key}: rows {n} {c}->{c}: per-offset {a:.4f} ms | windows {b:.4f} ms ({b / a:.3f}) | max diff {d:.2e} of {sc:.2e} | rerun identical {bool((got == again).all())} | "
          f"stats rel diff {ds_:.1e} | window table {t_w * 1e3:.1f} us, over-cap groups {float((wlen > 192).float().mean()):.4f}", flush=True)
