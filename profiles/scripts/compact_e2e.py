import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import torch
from tests.test_gpu_e2e import small_cfg
from toda_amd import ops
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
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.reset_running_stats(); m.train(False)
    batch = {"points": points, "points_per_sample": col["points_per_sample"], "batch_size": 2}
    voxelize_on_gpu(batch, ds.voxel_cfg)
    for m in (model.vfe, model.backbone_3d, model.map_to_bev_module):
        batch = m(batch)
    out = batch["spatial_features"]
    (out * torch.linspace(-1, 1, out.numel(), device="cuda").view_as(out)).sum().backward()
    return out.detach().clone(), [(n, p.grad.clone()) for n, p in model.backbone_3d.named_parameters()]
def cmp(tag, A, B):
    errs = [(n, float((x - y).abs().max()) / (float(y.abs().max()) + 1e-12)) for (n, x), (_, y) in zip(A[1], B[1])]
    print(tag, "out", float((A[0] - B[0]).abs().max() / B[0].abs().max()), "worst", sorted(errs, key=lambda t: -t[1])[:6])
ops.COMPACT = True
a1, a2 = run(), run()
cmp("compact twice", a1, a2)
ops.COMPACT = False
b1 = run()
cmp("compact vs packed", a1, b1)
ops.COMPACT = True; ops.HALO = True; ops.HALO_MIN_ROWS = 1
h1 = run()
cmp("compact: halo vs plain", h1, a1)
ops.COMPACT = False
h2 = run()
cmp("packed: halo vs plain", h2, b1)
# which variant is closer to the CPU oracle backend?
import copy
from oracle.cpu_backend import oracle_backend
cpu_model = copy.deepcopy(model).cpu()
def run_cpu():
    cpu_model.zero_grad(set_to_none=True)
    for m in cpu_model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.reset_running_stats(); m.train(False)
    with oracle_backend():
        batch = {"points": points.cpu(), "points_per_sample": col["points_per_sample"], "batch_size": 2}
        voxelize_on_gpu(batch, ds.voxel_cfg)
        for m in (cpu_model.vfe, cpu_model.backbone_3d, cpu_model.map_to_bev_module):
            batch = m(batch)
        out = batch["spatial_features"]
        (out * torch.linspace(-1, 1, out.numel()).view_as(out)).sum().backward()
    return out.detach().clone(), [(n, p.grad.clone()) for n, p in cpu_model.backbone_3d.named_parameters()]
o = run_cpu()
oc = (o[0].cuda(), [(n, g.cuda()) for n, g in o[1]])
cmp("compact vs oracle", a1, oc)
cmp("packed vs oracle", b1, oc)
