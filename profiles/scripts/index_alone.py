"""The input pipeline of one C3 batch alone on the GPU (no training beside it): voxelise + collate + every rulebook of VoxelBackBone8x through
ops.build_input_plan into rotating arena slots, 30 times.  Run under `rocprofv3 --kernel-trace --stats` for the uncontended duration of
every index kernel (profiles/r04_index_alone_kernel_stats.csv); prints the wall time per plan."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from toda_amd import arena, ops  # noqa: E402
from toda_amd.pcdet.config import AttrDict, cfg_from_yaml_file  # noqa: E402
from toda_amd.pcdet.datasets import SyntheticLidarDataset  # noqa: E402
from toda_amd.pcdet.models import build_network  # noqa: E402
from toda_amd.spconv.plan import ordered_steps  # noqa: E402

cfg = AttrDict()
cfg_from_yaml_file(os.path.join(ROOT, "toda_amd/tools/cfgs/models/centerpoint_voxel_waymo.yaml"), cfg)
ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
net = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds).cuda().train()
steps = ordered_steps(net.backbone_3d)
clouds = [[torch.from_numpy(ds[2 * b + i]["points"]).cuda() for i in range(2)] for b in range(2)]
ar = arena.IndexArena("cuda", slots=3)
stream = torch.cuda.current_stream()
times = []
for it in range(30):
    slot = ar.acquire(stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with arena.use_slot(slot), torch.no_grad():
        ops.build_input_plan(ops._cloud_list(clouds[it % 2]), ds.voxel_cfg, 2, net.backbone_3d.sparse_shape, steps, training=True)
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
    ar.release(slot, stream)
print("wall ms per plan (host + GPU, one sync inside):", " ".join(f"{t * 1e3:.2f}" for t in times[5:]))
