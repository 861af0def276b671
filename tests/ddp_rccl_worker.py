"""One rank of tests/test_gpu_ddp_rccl.py (started by torch.distributed.run, one process per GPU): the tiny CenterPoint
model through the HIP path under DistributedDataParallel over RCCL; rank 0 stores loss + averaged gradients."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out = sys.argv[1]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    backend = os.environ.get("TODA_TEST_DDP_BACKEND", "nccl")
    if backend == "gloo":          # one-GPU rehearsal: both ranks on cuda:0, collectives over gloo (staged through the host)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "gloo":
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from test_ddp_gloo import _freeze_bn, _tiny_cfg
    from toda_amd.pcdet.datasets import SyntheticLidarDataset
    from toda_amd.pcdet.models import build_network, model_fn_decorator

    cfg = _tiny_cfg()
    ds = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES)
    torch.manual_seed(0)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds).to(dev).train()
    _freeze_bn(model)
    from toda_amd.pcdet.utils.common_utils import wrap_ddp
    ddp = wrap_ddp(model, device_ids=[local])      # as tools/train.py: 8 MB bucket views + the coalesced buffer broadcast
    batch = ds.collate_batch([ds[2 * rank], ds[2 * rank + 1]])      # DistributedSampler-style shard
    ret = model_fn_decorator()(ddp, batch)
    ret.loss.backward()
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    grads = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}
    # the optimizer step of the training loop on the synchronised gradients (bucket views): two launches through toda_clip_adam_step;
    # every rank must end up with the same parameters, and they must equal the torch path's on a copy of the model
    from helpers import abi_calls
    from toda_amd.tools.train_utils.optimization import OneCycleAdam, clip_and_step
    twin = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds).to(dev).train()      # (the stepped model holds its forward's activations: no deepcopy)
    twin.load_state_dict(model.state_dict())
    for (_, p), q in zip(model.named_parameters(), twin.parameters()):
        q.grad = None if p.grad is None else p.grad.detach().clone()
    opt, opt_t = OneCycleAdam(model, wd=0.01), OneCycleAdam(twin, wd=0.01)
    opt_t._hip_step = False
    opt.lr = opt_t.lr = 1e-3
    with abi_calls("toda_clip_adam_step") as calls:
        norm = clip_and_step(opt, list(ddp.parameters()), 10.0)
    clip_and_step(opt_t, list(twin.parameters()), 10.0)
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    flat_t = torch.cat([p.detach().flatten() for p in twin.parameters()])
    lo, hi = flat.clone(), flat.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if rank == 0:
        torch.save({"loss": ret.loss.detach().cpu(), "grads": grads, "ranks": int(ones.item()), "fused_calls": calls["toda_clip_adam_step"],
                    "rank_spread": float((hi - lo).abs().max()), "vs_torch": float((flat - flat_t).abs().max() / flat_t.abs().max()),
                    "grad_norm": float(norm)}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
