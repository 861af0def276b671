"""N>1 path on the CPU: two `gloo` ranks, each with its own shard of scenes, through the same
detector + DistributedDataParallel path the trainers and bench.py use.  The averaged gradients must
equal the mean of the two shards' gradients computed in one process (BatchNorm in eval mode so
that per-rank statistics do not differ)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _tiny_cfg():
    from toda_amd.pcdet.config import AttrDict, cfg_from_yaml_file

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = AttrDict()
    cfg_from_yaml_file(os.path.join(root, "toda_amd/tools/cfgs/models/centerpoint_voxel_waymo.yaml"), cfg)
    cfg.DATA_CONFIG.POINT_CLOUD_RANGE = [-6.4, -6.4, -2, 6.4, 6.4, 4]
    cfg.DATA_CONFIG.SYNTHETIC.NUM_POINTS = 6000
    cfg.MODEL.BACKBONE_2D.LAYER_NUMS = [1, 1]
    cfg.MODEL.DENSE_HEAD.POST_PROCESSING.POST_CENTER_LIMIT_RANGE = [-6.4, -6.4, -2, 6.4, 6.4, 4]
    return cfg


def _freeze_bn(model):
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()


def _loss_and_grads(model, ds, indices):
    from oracle.cpu_backend import oracle_backend
    from toda_amd.pcdet.models import model_fn_decorator

    batch = ds.collate_batch([ds[i] for i in indices])
    with oracle_backend():
        ret = model_fn_decorator()(model, batch)
        ret.loss.backward()
    return ret.loss.detach()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from toda_amd.pcdet.datasets import SyntheticLidarDataset
    from toda_amd.pcdet.models import build_network

    torch.set_num_threads(2)
    cfg = _tiny_cfg()
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES)
    torch.manual_seed(0)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds).train()
    _freeze_bn(model)
    from toda_amd.pcdet.utils.common_utils import wrap_ddp
    ddp = wrap_ddp(model)          # as tools/train.py: gradient buckets + the coalesced buffer broadcast
    # rank 1 starts from different BatchNorm buffers: the forward's pre-hook must replace them by rank 0's
    if rank == 1:
        with torch.no_grad():
            for b in model.buffers():
                b.add_(3)
    loss = _loss_and_grads(ddp, ds, [2 * rank, 2 * rank + 1])  # DistributedSampler-style shard
    ddp.toda_buffer_broadcaster.sync()
    digest = torch.stack([b.double().sum() for b in model.buffers()]).sum().reshape(1)
    both = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(both, digest)
    if rank == 0:
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        torch.save({"loss": loss, "grads": grads, "buffer_digests": [float(t) for t in both]}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("reducer", ["own", "torch"])      # common_utils.DataParallel (flat-bucket reducer) / torch's DistributedDataParallel
def test_two_rank_gloo_ddp_equals_single_process(tmp_path, reducer, monkeypatch):
    out = str(tmp_path / "rank0.pt")
    monkeypatch.setenv("TODA_DDP", reducer)                 # inherited by the spawned ranks (wrap_ddp reads it)
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["buffer_digests"][0] == got["buffer_digests"][1]        # rank 1 holds rank 0's BatchNorm buffers (it started 3 off)

    from toda_amd.pcdet.datasets import SyntheticLidarDataset
    from toda_amd.pcdet.models import build_network

    cfg = _tiny_cfg()
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES)
    torch.manual_seed(0)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds).train()
    _freeze_bn(model)
    l0 = _loss_and_grads(model, ds, [0, 1])
    g0 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad()
    _loss_and_grads(model, ds, [2, 3])
    assert abs(float(got["loss"]) - float(l0)) < 1e-5 * max(1.0, abs(float(l0)))  # rank 0 saw shard 0
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        mean_grad = 0.5 * (g0[n] + p.grad)
        scale = float(mean_grad.abs().max()) + 1e-8
        assert float((got["grads"][n] - mean_grad).abs().max()) <= 1e-4 * scale + 1e-7, n
