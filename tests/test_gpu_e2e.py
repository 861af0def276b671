"""End-to-end parity on the GPU: the same detector (same weights, same batch) through the HIP path
and through the CPU oracle backend.  Loss and gradients must agree within 1e-3 (north star)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = "toda_amd/tools/cfgs/models/{}.yaml"


def small_cfg(name, rng_xy=16.0, n_points=12000):
    import os
    from toda_amd.pcdet.config import AttrDict, cfg_from_yaml_file

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = AttrDict()
    cfg_from_yaml_file(os.path.join(root, CFG.format(name)), cfg)
    r = cfg.DATA_CONFIG.POINT_CLOUD_RANGE
    cfg.DATA_CONFIG.POINT_CLOUD_RANGE = [-rng_xy, -rng_xy, r[2], rng_xy, rng_xy, r[5]]
    cfg.DATA_CONFIG.SYNTHETIC.NUM_POINTS = n_points
    cfg.MODEL.DENSE_HEAD.POST_PROCESSING.POST_CENTER_LIMIT_RANGE = [-rng_xy, -rng_xy, -10, rng_xy, rng_xy, 10]
    return cfg


def _freeze_bn(model):
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()


@pytest.mark.parametrize("bn_train", [True, False])
@pytest.mark.parametrize("name,rng_xy", [("centerpoint_voxel_waymo", 16.0), ("toda_stage1_centerpoint_res", 14.4)])
def test_detector_loss_and_grads_match_oracle_backend(name, rng_xy, bn_train):
    """bn_train=False (BN uses running statistics): gradients must agree to 2e-3 globally and 1e-2 per parameter.  (Per parameter
    the comparison is at the mercy of single ReLU decisions: with the untrained mean 0 / var 1 statistics the activations shrink
    layer by layer to ~1e-12 at the BEV neck, where a rounding difference of 1e-16 - the library's stride-2 convolution is not
    run-to-run reproducible at that level - flips a mask bit; 2 flipped elements of 819 200 were measured to move a neck gradient by
    1e-2 of its maximum and a bias gradient by 3e-3 of its floored norm between two runs of the SAME code.  With the statistics
    calibrated to the batch the activations are O(1) and the whole comparison sits at the train-mode floor, 5e-3..8e-3 globally.)
    bn_train=True: the loss must agree to 1e-3, but gradients are compared loosely (3e-2): a
    train-mode BatchNorm behind the nearly constant heat-map gradient at initialisation cancels
    ~99.99 % of dy (dx = dy - mean(dy) - xhat*mean(dy*xhat)), so fp32 rounding is amplified ~1e4x.
    Measured noise floor: CPU oracle vs the SAME CPU oracle with the voxel rows permuted differs by
    6e-3..8e-3 per parameter (and 100 % on the mathematically-zero biases in front of a BN)."""
    from oracle.cpu_backend import oracle_backend
    from toda_amd.pcdet.datasets import SyntheticLidarDataset
    from toda_amd.pcdet.models import build_network, model_fn_decorator

    cfg = small_cfg(name, rng_xy)
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES)
    torch.manual_seed(0)
    cpu_model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds).train()
    if not bn_train:
        _freeze_bn(cpu_model)
    gpu_model = copy.deepcopy(cpu_model).cuda()
    batch = ds.collate_batch([ds[0], ds[1]])
    fn = model_fn_decorator()

    with oracle_backend():
        ref = fn(cpu_model, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()})
        ref.loss.backward()
    out = fn(gpu_model, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()})
    out.loss.backward()

    assert abs(float(out.loss) - float(ref.loss)) <= 1e-3 * max(1.0, abs(float(ref.loss)))
    for key in ref.tb_dict:
        assert abs(float(out.tb_dict[key]) - float(ref.tb_dict[key])) <= 1e-3 * max(1.0, abs(float(ref.tb_dict[key])))
    grads = [(n, p.grad, q.grad.cpu()) for (n, p), q in zip(cpu_model.named_parameters(), gpu_model.parameters())
             if p.grad is not None]
    assert len(grads) > 50 and all(q is not None for _, _, q in grads)
    g_all = torch.cat([p.flatten() for _, p, _ in grads])
    d_all = torch.cat([(q - p).flatten() for _, p, q in grads])
    tol = 3e-2 if bn_train else 2e-3
    global_err = float(d_all.norm() / g_all.norm())
    assert global_err < tol, f"global relative grad error {global_err:.2e}"
    if not bn_train:
        tol = 1e-2          # per parameter: see the docstring
    floor = 1e-3 * float(g_all.norm())  # parameters with (near) zero gradient are judged on the global scale
    worst = 0.0
    for n, p, q in grads:
        err = float((q - p).norm() / (p.norm() + floor))
        worst = max(worst, err)
        assert err < tol, f"{n}: relative L2 grad error {err:.2e}"
    # BN running statistics went through the same batches
    for (n, b), c in zip(cpu_model.named_buffers(), gpu_model.buffers()):
        if b.dtype.is_floating_point and bn_train:
            assert torch.allclose(c.cpu(), b, rtol=1e-3, atol=1e-4), n
    print(f"bn_train={bn_train}: global grad error {global_err:.2e}, worst per-parameter {worst:.2e}")


def test_sparse_backbone_step_is_bitwise_reproducible():
    """No float atomics anywhere in the hand-written path: voxelise -> MeanVFE -> VoxelBackBone8x -> dense BEV map, forward
    and backward, twice on the same input gives the same bits (features, weight gradients, input gradients)."""
    from toda_amd.pcdet.datasets import SyntheticLidarDataset
    from toda_amd.pcdet.models import build_network, voxelize_on_gpu

    cfg = small_cfg("centerpoint_voxel_waymo", rng_xy=16.0, n_points=20000)
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
    torch.manual_seed(3)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds).cuda().train()
    col = ds.collate_batch([ds[0], ds[1]])
    points = torch.from_numpy(col["points"]).float().cuda()

    def run():
        model.zero_grad(set_to_none=True)
        batch = {"points": points, "points_per_sample": col["points_per_sample"], "batch_size": 2}
        voxelize_on_gpu(batch, ds.voxel_cfg)
        batch["voxels"].requires_grad_(True)
        vox = batch["voxels"]
        for m in (model.vfe, model.backbone_3d, model.map_to_bev_module):
            batch = m(batch)
        out = batch["spatial_features"]
        (out * torch.linspace(-1, 1, out.numel(), device="cuda").view_as(out)).sum().backward()
        grads = [p.grad.clone() for p in model.backbone_3d.parameters()]
        return out.detach().clone(), vox.grad.clone(), grads

    a, b = run(), run()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert all(torch.equal(x, y) for x, y in zip(a[2], b[2])) and len(a[2]) > 20
    assert float(a[0].abs().sum()) > 0 and float(a[1].abs().sum()) > 0


@pytest.mark.parametrize("n_second", [7, 0])
def test_batch_with_a_tiny_or_empty_sample_trains(n_second):
    """Ragged batches: a sample with a handful of points, or none at all (every level of that sample is empty), goes through
    voxelisation, both backbones' index plans, the heads and the backward pass."""
    from toda_amd.pcdet.datasets import SyntheticLidarDataset
    from toda_amd.pcdet.models import build_network, voxelize_on_gpu

    cfg = small_cfg("toda_stage1_centerpoint_res", rng_xy=14.4, n_points=8000)
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
    torch.manual_seed(0)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds).cuda().train()
    a = ds[0]
    b = dict(a)
    b["points"] = a["points"][:n_second].copy()
    col = ds.collate_batch([a, b])
    batch = {"points": torch.from_numpy(col["points"]).float().cuda(), "points_per_sample": col["points_per_sample"],
             "gt_boxes": torch.from_numpy(col["gt_boxes"]).float().cuda(), "batch_size": 2}
    voxelize_on_gpu(batch, ds.voxel_cfg)
    per = torch.bincount(batch["voxel_coords"][:, 0].long(), minlength=2).tolist()
    assert per[0] > 1000 and per[1] <= n_second
    ret, _, _ = model(batch)
    ret["loss"].backward()
    assert torch.isfinite(ret["loss"]) and all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("name", ["centerpoint_voxel_waymo", "toda_stage1_centerpoint_res"])
def test_full_size_detector_matches_oracle_backend(name):
    """BASELINE configs 3 and 5 at their FULL size (180k-pt Waymo-shape clouds / mixed 180k + 35k, bs 2, the real grids
    and voxel caps): same weights through the HIP path and through the CPU oracle backend, BatchNorm on its running
    statistics.  Loss and every tb_dict entry <= 1e-3, the stage-4 sparse features (as the dense BEV map, which is
    independent of the row order of generated index sets) <= 1e-3 relative, gradients <= 2e-3 (north star: 1e-3 fp32 for
    features / losses)."""
    import os
    from oracle.cpu_backend import oracle_backend
    from toda_amd.pcdet.config import AttrDict, cfg_from_yaml_file
    from toda_amd.pcdet.datasets import SyntheticLidarDataset
    from toda_amd.pcdet.models import build_network, model_fn_decorator

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = AttrDict()
    cfg_from_yaml_file(os.path.join(root, CFG.format(name)), cfg)
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES)
    torch.manual_seed(0)
    cpu_model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds).train()
    # running statistics that are not the identity, so the eval-mode BN really scales and shifts
    g = torch.Generator().manual_seed(1)
    for m in cpu_model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.copy_(0.05 * torch.randn(m.running_mean.shape, generator=g))
            m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
    _freeze_bn(cpu_model)
    gpu_model = copy.deepcopy(cpu_model).cuda()
    batch = ds.collate_batch([ds[0], ds[1]])
    assert batch["points"].shape[0] > 150_000
    fn = model_fn_decorator()
    feats = {}

    def grab(tag):
        def hook(mod, args, out):
            feats[tag] = out["spatial_features"].detach().cpu()
        return hook

    cpu_model.map_to_bev_module.register_forward_hook(grab("cpu"))
    gpu_model.map_to_bev_module.register_forward_hook(grab("gpu"))
    with oracle_backend():
        ref = fn(cpu_model, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()})
        ref.loss.backward()
    out = fn(gpu_model, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()})
    out.loss.backward()
    assert abs(float(out.loss) - float(ref.loss)) <= 1e-3 * max(1.0, abs(float(ref.loss)))
    for key in ref.tb_dict:
        assert abs(float(out.tb_dict[key]) - float(ref.tb_dict[key])) <= 1e-3 * max(1.0, abs(float(ref.tb_dict[key]))), key
    a, b = feats["cpu"], feats["gpu"]
    assert a.shape == b.shape
    rel = float((a - b).norm() / a.norm())
    worst = float((a - b).abs().max() / a.abs().max())
    assert rel < 1e-3 and worst < 1e-3, (rel, worst)
    grads = [(n, p.grad, q.grad.cpu()) for (n, p), q in zip(cpu_model.named_parameters(), gpu_model.parameters()) if p.grad is not None]
    g_all = torch.cat([p.flatten() for _, p, _ in grads])
    d_all = torch.cat([(q - p).flatten() for _, p, q in grads])
    global_err = float(d_all.norm() / g_all.norm())
    assert global_err < 2e-3, global_err
    print(f"{name}: loss {float(out.loss):.5f} vs {float(ref.loss):.5f}, BEV features rel {rel:.2e} / max {worst:.2e}, grads {global_err:.2e}")


def _full_cfg(name):
    import os
    from toda_amd.pcdet.config import AttrDict, cfg_from_yaml_file

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = AttrDict()
    cfg_from_yaml_file(os.path.join(root, CFG.format(name)), cfg)
    return cfg


def _bn_buffers(model):
    return {n: b.detach().cpu().clone() for n, b in model.named_buffers() if n.endswith(("running_mean", "running_var"))}


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("name", ["centerpoint_voxel_waymo", "toda_stage1_centerpoint_res"])
def test_full_size_train_mode_forward_matches_oracle_backend(name):
    """The path bench.py times - TRAIN-mode BatchNorm: statistics from the gather-GEMM epilogue, bn2d, the fused hidden layer of
    the head - at the FULL size of BASELINE configs 3 and 5 (reference spconv_backbone.py:21-25,128-180,
    base_bev_backbone.py:81-112, center_head.py:221-272).  One training forward of the same weights through the HIP path and
    through the CPU oracle backend: loss and every tb_dict entry <= 1e-3, the stage-4 sparse features as the dense BEV map
    <= 1e-3, and every BatchNorm's running_mean / running_var after the step <= 1e-3 of the buffer's scale (those ARE the batch
    statistics: momentum x batch moment + (1 - momentum) x init)."""
    from oracle.cpu_backend import oracle_backend
    from toda_amd.pcdet.datasets import SyntheticLidarDataset
    from toda_amd.pcdet.models import build_network, model_fn_decorator

    cfg = _full_cfg(name)
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES)
    torch.manual_seed(0)
    cpu_model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds).train()
    gpu_model = copy.deepcopy(cpu_model).cuda()
    before = _bn_buffers(cpu_model)
    batch = ds.collate_batch([ds[0], ds[1]])
    assert batch["points"].shape[0] > 150_000
    fn = model_fn_decorator()
    feats = {}

    def grab(tag):
        def hook(mod, args, out):
            feats[tag] = out["spatial_features"].detach().cpu()
        return hook

    cpu_model.map_to_bev_module.register_forward_hook(grab("cpu"))
    gpu_model.map_to_bev_module.register_forward_hook(grab("gpu"))
    with torch.no_grad():
        with oracle_backend():
            ref = fn(cpu_model, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()})
        out = fn(gpu_model, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()})
    assert abs(float(out.loss) - float(ref.loss)) <= 1e-3 * max(1.0, abs(float(ref.loss))), (float(out.loss), float(ref.loss))
    for key in ref.tb_dict:
        assert abs(float(out.tb_dict[key]) - float(ref.tb_dict[key])) <= 1e-3 * max(1.0, abs(float(ref.tb_dict[key]))), key
    a, b = feats["cpu"], feats["gpu"]
    rel = float((a - b).norm() / a.norm())
    worst = float((a - b).abs().max() / a.abs().max())
    assert rel < 1e-3 and worst < 1e-3, (rel, worst)
    cpu_after, gpu_after = _bn_buffers(cpu_model), _bn_buffers(gpu_model)
    moved, worst_bn = 0, 0.0
    for n, v in cpu_after.items():
        moved += int(not torch.equal(v, before[n]))
        scale = float(v.abs().max()) + 1e-12
        err = float((gpu_after[n] - v).abs().max()) / scale
        worst_bn = max(worst_bn, err)
        assert err <= 1e-3, (n, err)
    assert moved == len(cpu_after) and moved >= 40          # every norm of the detector saw the batch
    print(f"{name} train-mode: loss {float(out.loss):.5f} vs {float(ref.loss):.5f}, BEV rel {rel:.2e} / max {worst:.2e}, "
          f"{moved} BN buffers, worst {worst_bn:.2e}")


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("bn_train", [True, False])
def test_full_size_c2_backbone_forward_matches_oracle_backend(bn_train):
    """BASELINE config 2 at its full size (reference spconv_backbone.py:128-180): VoxelBackBone8x forward on
    four 60k-pt nuScenes-shape clouds, MeanVFE -> 12 sparse convolutions -> dense BEV map, HIP path against the CPU oracle
    backend, <= 1e-3 (north star).  bn_train=True is what `bench.py --workload c2` runs (a fresh model in train mode under
    no_grad); bn_train=False uses non-trivial running statistics."""
    from oracle.cpu_backend import oracle_backend
    from toda_amd.pcdet.datasets import SyntheticLidarDataset
    from toda_amd.pcdet.models import build_network, load_data_to_gpu, voxelize_on_gpu

    cfg = _full_cfg("second_backbone_nuscenes")
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES)
    torch.manual_seed(0)
    cpu_model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds).train()
    if not bn_train:
        g = torch.Generator().manual_seed(1)
        for m in cpu_model.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.copy_(0.05 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
        _freeze_bn(cpu_model)
    gpu_model = copy.deepcopy(cpu_model).cuda()
    batch = ds.collate_batch([ds[i] for i in range(4)])
    assert batch["points"].shape[0] > 200_000 and batch["batch_size"] == 4

    def forward(model):
        b = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()}
        load_data_to_gpu(b)
        voxelize_on_gpu(b, ds.voxel_cfg)
        for m in (model.vfe, model.backbone_3d, model.map_to_bev_module):
            b = m(b)
        return b["spatial_features"].detach().cpu(), int(b["voxel_coords"].shape[0])

    with torch.no_grad():
        with oracle_backend():
            a, n_cpu = forward(cpu_model)
        b, n_gpu = forward(gpu_model)
    assert n_cpu == n_gpu and a.shape == b.shape and a.shape[0] == 4
    rel = float((a - b).norm() / a.norm())
    worst = float((a - b).abs().max() / a.abs().max())
    assert rel < 1e-3 and worst < 1e-3, (rel, worst)
    print(f"c2 bn_train={bn_train}: {n_gpu} voxels, BEV {tuple(a.shape)}, rel {rel:.2e} / max {worst:.2e}")


@pytest.mark.timeout(1500)
def test_full_size_stage2_consistency_step_matches_oracle_backend():
    """BASELINE config 5, stage 2, at full size (reference pcdet/models/__init__.py:88-260 model_fn_decorator_cl behind
    tools/stage2_mixup_train_cl.py): the (adversarial, original) pair of two 180 k-point frames, 2 forwards + 1 backward through
    DistModel, same weights through the HIP path and the CPU oracle backend (BatchNorm on non-trivial running statistics).  Total loss
    and every tb_dict entry <= 1e-3, gradients <= 2e-3 globally - the toy-size pair of tests/test_gpu_eval.py was the only stage-2
    parity so far ("stage-2 c5cl only at toy size in tests")."""
    from oracle.cpu_backend import oracle_backend
    from toda_amd.pcdet.datasets import SyntheticPairDataset
    from toda_amd.pcdet.models import DistModel, build_network, model_fn_decorator_cl

    cfg = _full_cfg("toda_stage1_centerpoint_res")
    if "KINDS" not in cfg.DATA_CONFIG.SYNTHETIC and "KIND" not in cfg.DATA_CONFIG.SYNTHETIC:
        cfg.DATA_CONFIG.SYNTHETIC.KINDS = [cfg.DATA_CONFIG.SYNTHETIC.get("SOURCE_KIND", "waymo_toda")]
    ds = SyntheticPairDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
    torch.manual_seed(0)
    cpu_model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds).train()
    g = torch.Generator().manual_seed(1)
    for m in cpu_model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.copy_(0.05 * torch.randn(m.running_mean.shape, generator=g))
            m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
    _freeze_bn(cpu_model)
    gpu_model = copy.deepcopy(cpu_model).cuda()
    adv, org = ds.collate_batch([ds[0], ds[1]])
    assert adv["points"].shape[0] > 150_000 and org["points"].shape[0] > 150_000
    fn = model_fn_decorator_cl()

    def clone(b):
        return {k: (v.copy() if isinstance(v, np.ndarray) else copy.deepcopy(v)) for k, v in b.items()}

    with oracle_backend():
        ref = fn(DistModel(cpu_model), clone(adv), clone(org), False)
        ref.loss.backward()
    out = fn(DistModel(gpu_model), clone(adv), clone(org), False)
    out.loss.backward()
    assert abs(float(out.loss.detach()) - float(ref.loss.detach())) <= 1e-3 * max(1.0, abs(float(ref.loss.detach())))
    for key in ref.tb_dict:
        # cl_center / cl_size sit behind a top-500 selection over a near-flat random-init heatmap (sigmoid(sigmoid(logit)), see
        # filter_boxes_centerpoint): a 1e-7 score difference swaps which cell is the 500th, so those two carry a selection tolerance;
        # they are detached (no gradient) and enter the loss with weight 0.1.  Everything differentiable is held to 1e-3.
        tol = 2e-2 if key in ("cl_center", "cl_size") else 1e-3
        a, b = float(out.tb_dict[key]), float(ref.tb_dict[key])
        assert abs(a - b) <= tol * max(1.0, abs(b)), (key, a, b)
    grads = [(n, p.grad, q.grad.cpu()) for (n, p), q in zip(cpu_model.named_parameters(), gpu_model.parameters()) if p.grad is not None]
    g_all = torch.cat([p.flatten() for _, p, _ in grads])
    d_all = torch.cat([(q - p).flatten() for _, p, q in grads])
    err = float(d_all.norm() / g_all.norm())
    assert err < 2e-3, err
    print(f"stage-2 full size: loss {float(out.loss.detach()):.5f} vs {float(ref.loss.detach()):.5f}, grads {err:.2e}, "
          + ", ".join(f"{k} {float(out.tb_dict[k]):.5f}/{float(ref.tb_dict[k]):.5f}" for k in sorted(ref.tb_dict)))


def test_input_prefetcher_worker_thread_hands_out_the_same_batches(monkeypatch):
    """pcdet.models.InputPrefetcher with the preparation on its worker thread (the default) and on the caller's thread
    (TODA_PREFETCH_THREAD=0): the same voxels, coordinates and rulebook tables in the same order, StopIteration behind the last
    batch, and an exception of the batch source surfaces in next()."""
    from toda_amd.pcdet.datasets import SyntheticLidarDataset
    from toda_amd.pcdet.models import InputPrefetcher, build_network

    cfg = small_cfg("centerpoint_voxel_waymo", rng_xy=16.0, n_points=20000)
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
    net = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds).cuda().train()
    cols = [ds.collate_batch([ds[2 * i], ds[2 * i + 1]]) for i in range(3)]

    def source(fail_at=None):
        for i, c in enumerate(cols):
            if i == fail_at:
                raise RuntimeError("source failed")
            yield {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in c.items()}

    def drain(threaded):
        monkeypatch.setenv("TODA_PREFETCH_THREAD", "1" if threaded else "0")
        pre = InputPrefetcher(source(), net, torch.device("cuda", 0))
        assert pre.threaded == threaded
        out = []
        while True:
            try:
                b = pre.next()
            except StopIteration:
                break
            pre.kick()
            torch.cuda.current_stream().synchronize()
            plan = b["sparse_index_plan"][0] if "sparse_index_plan" in b else None
            out.append((b["voxels"].clone(), b["voxel_coords"].clone(), b["voxel_num_points"].clone(), plan))
        return out

    a, b = drain(True), drain(False)
    assert len(a) == len(b) == 3
    for (va, ca, na, pa), (vb, cb, nb, pb) in zip(a, b):
        assert torch.equal(va, vb) and torch.equal(ca, cb) and torch.equal(na, nb)
        assert pa is not None and sorted(pa) == sorted(pb) and len(pa) >= 8          # the whole backbone's rulebooks came with the batch
        for key in pa:
            ea, eb = pa[key], pb[key]
            ra = ea["rb"] if isinstance(ea, dict) else getattr(ea, "rb", ea)
            rbb = eb["rb"] if isinstance(eb, dict) else getattr(eb, "rb", eb)
            assert torch.equal(ra.nbr_fwd, rbb.nbr_fwd), key
    monkeypatch.setenv("TODA_PREFETCH_THREAD", "1")
    pre = InputPrefetcher(source(fail_at=1), net, torch.device("cuda", 0))
    pre.next()
    pre.kick()
    with pytest.raises(RuntimeError, match="source failed"):
        pre.next()
