"""Why the train-mode-BatchNorm gradient comparison of tests/test_gpu_e2e.py uses 3e-2 and not the north star's 1e-3
(VERDICT r1): the model itself amplifies fp32 rounding.  Here the SAME CPU oracle runs the SAME detector on the SAME
voxels twice; the second time the voxel rows are permuted, which changes nothing mathematically (every operator is
row-order equivariant and the losses are sums) but changes the order of the fp32 additions in the BatchNorm statistics
and the weight gradients.  The gradient difference between the two CPU runs is the noise floor any other correct fp32
implementation (the HIP path) is measured against."""
import copy

import numpy as np
import pytest
import torch


@pytest.mark.timeout(900)
@pytest.mark.parametrize("bn_train", [True, False])
def test_gradient_noise_floor_cpu_vs_cpu(bn_train):
    from oracle import cpu_backend as CB
    from oracle.cpu_backend import oracle_backend
    from tests.test_gpu_e2e import small_cfg
    from toda_amd.pcdet.datasets import SyntheticLidarDataset
    from toda_amd.pcdet.models import build_network, model_fn_decorator

    torch.set_num_threads(4)
    cfg = small_cfg("centerpoint_voxel_waymo", 16.0)
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES)
    torch.manual_seed(0)
    model_a = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds).train()
    if not bn_train:          # BatchNorm on running statistics: the case tests/test_gpu_e2e.py compares at the tighter tolerance
        from tests.test_gpu_e2e import _freeze_bn
        _freeze_bn(model_a)
    model_b = copy.deepcopy(model_a)
    col = ds.collate_batch([ds[0], ds[1]])
    pts = torch.from_numpy(col["points"]).float()
    clouds = [pts[pts[:, 0] == b][:, 1:].contiguous() for b in range(2)]
    vc = ds.voxel_cfg
    vox, coords, num = CB.voxelize_batch(clouds, vc["point_cloud_range"], vc["voxel_size"], vc["max_points_per_voxel"],
                                         vc["max_num_voxels"])
    perm = torch.from_numpy(np.random.default_rng(0).permutation(len(vox)))
    fn = model_fn_decorator()

    # eval-mode BN: a row permutation changes no sum at all in the oracle (every reduction it has left is per row), so the
    # probe is a one-ulp perturbation of the input point features instead: mathematically a relative change of 6e-8
    ulp = (torch.from_numpy(np.random.default_rng(1).integers(0, 2, tuple(vox.shape))).float() * 2 - 1) * 6e-8

    def run(model, order):
        v = vox if bn_train or order is ident else vox * (1 + ulp)
        batch = {"voxels": v[order].clone(), "voxel_coords": coords[order].clone(), "voxel_num_points": num[order].clone(),
                 "gt_boxes": torch.from_numpy(col["gt_boxes"].copy()).float(), "batch_size": 2}
        with oracle_backend():
            ret = fn(model, batch)
            ret.loss.backward()
        return ret

    ident = torch.arange(len(vox))
    ra = run(model_a, ident)
    rb = run(model_b, perm if bn_train else ident.clone())
    assert abs(float(ra.loss) - float(rb.loss)) <= 1e-4 * max(1.0, abs(float(ra.loss)))     # the loss itself is stable
    grads = [(n, p.grad, q.grad) for (n, p), q in zip(model_a.named_parameters(), model_b.parameters()) if p.grad is not None]
    g_all = torch.cat([p.flatten() for _, p, _ in grads])
    d_all = torch.cat([(q - p).flatten() for _, p, q in grads])
    global_err = float(d_all.norm() / g_all.norm())
    floor = 1e-3 * float(g_all.norm())
    per_param = sorted(((float((q - p).norm() / (p.norm() + floor)), n) for n, p, q in grads), reverse=True)
    worst, worst_name = per_param[0]
    print(f"CPU-vs-CPU noise floor (bn_train={bn_train}): global {global_err:.2e}, worst parameter {worst:.2e} ({worst_name}); next: {per_param[1:4]}")
    if bn_train:
        # the floor sits far above fp32 epsilon and above the north star's 1e-3 for single parameters ...
        assert worst > 1e-4, "rounding noise unexpectedly small: tighten the GPU tolerance"
        # ... and below the tolerance the GPU comparison uses, so that tolerance still detects real errors
        assert worst < 3e-2 and global_err < 3e-2
    else:
        assert worst < 1e-2 and global_err < 2e-3
