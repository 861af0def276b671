"""Small host utilities shared by the trainers (reference pcdet/utils/common_utils.py)."""
import logging
import os
import random

import numpy as np
import torch
import torch.distributed as dist


def check_numpy_to_torch(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).float(), True
    return x, False


def limit_period(val, offset=0.5, period=np.pi):
    val, is_numpy = check_numpy_to_torch(val)
    ans = val - torch.floor(val / period + offset) * period
    return ans.numpy() if is_numpy else ans


def mask_points_by_range(points, limit_range):
    """x/y only, both ends inclusive (reference common_utils.py:60-63; z is NOT tested)."""
    return (points[:, 0] >= limit_range[0]) & (points[:, 0] <= limit_range[3]) \
        & (points[:, 1] >= limit_range[1]) & (points[:, 1] <= limit_range[4])


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False


def create_logger(log_file=None, rank=0, log_level=logging.INFO):
    logger = logging.getLogger(__name__)
    logger.setLevel(log_level if rank == 0 else "ERROR")
    logger.handlers.clear()
    fmt = logging.Formatter("%(asctime)s  %(levelname)5s  %(message)s")
    console = logging.StreamHandler()
    console.setLevel(log_level if rank == 0 else "ERROR")
    console.setFormatter(fmt)
    logger.addHandler(console)
    if log_file is not None:
        fh = logging.FileHandler(filename=log_file)
        fh.setLevel(log_level if rank == 0 else "ERROR")
        fh.setFormatter(fmt)
        logger.addHandler(fh)
    logger.propagate = False
    return logger


def get_dist_info(return_gpu_per_machine=False):
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    if return_gpu_per_machine:
        return rank, world, max(torch.cuda.device_count(), 1)
    return rank, world


def init_dist_pytorch(tcp_port=None, local_rank=None, backend="nccl"):
    """torchrun / torch.distributed.launch rendezvous (reference common_utils.py:161-176).
    backend 'nccl' is RCCL on ROCm.  Rank / world size come from the environment."""
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if backend == "nccl":
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if tcp_port is not None:
            os.environ.setdefault("MASTER_PORT", str(tcp_port))
        dist.init_process_group(backend=backend)
    return dist.get_world_size(), dist.get_rank()


def init_dist_slurm(tcp_port, local_rank=None, backend="nccl"):
    """SLURM launch (reference common_utils.py:134-158): one task per GPU; rank / world size from SLURM_PROCID /
    SLURM_NTASKS, rendezvous on the first host of SLURM_NODELIST."""
    import subprocess

    rank, world = int(os.environ["SLURM_PROCID"]), int(os.environ["SLURM_NTASKS"])
    if backend == "nccl":
        torch.cuda.set_device(rank % max(torch.cuda.device_count(), 1))
    first = subprocess.run(["scontrol", "show", "hostname", os.environ["SLURM_NODELIST"]], capture_output=True, text=True)
    os.environ["MASTER_ADDR"] = (first.stdout.split() or ["127.0.0.1"])[0]
    os.environ["MASTER_PORT"] = str(tcp_port)
    os.environ["WORLD_SIZE"], os.environ["RANK"] = str(world), str(rank)
    dist.init_process_group(backend=backend)
    return dist.get_world_size(), dist.get_rank()


class AverageMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def rotate_points_along_z(points, angle):
    """points [B, N, 3+C], angle [B] (radians, counter-clockwise seen from +z): fp32 batched product with the
    row-vector rotation matrix [[c, s, 0], [-s, c, 0], [0, 0, 1]] (reference common_utils.py:34-57)."""
    points, is_numpy = check_numpy_to_torch(points)
    angle, _ = check_numpy_to_torch(angle)
    c, s = torch.cos(angle), torch.sin(angle)
    o, l = torch.zeros_like(angle), torch.ones_like(angle)
    rot = torch.stack((c, s, o, -s, c, o, o, o, l), dim=1).view(-1, 3, 3).float()
    xyz = torch.matmul(points[:, :, 0:3], rot)
    out = torch.cat((xyz, points[:, :, 3:]), dim=-1)
    return out.numpy() if is_numpy else out


def merge_results_dist(result_part, size, tmpdir=None):
    """Gather per-rank result lists in dataset order (reference common_utils.py:205-238 does this through pickles in a
    shared tmpdir; here one all_gather_object over the process group).  The evaluation sampler deals indices round
    robin, so parts are interleaved and the padded tail is cut at `size`."""
    rank, world = get_dist_info()
    if world == 1:
        return result_part[:size]
    parts = [None] * world
    dist.all_gather_object(parts, result_part)
    if rank != 0:
        return None
    ordered = []
    for group in zip(*parts):
        ordered.extend(group)
    return ordered[:size]


class BufferBroadcaster:
    """What DistributedDataParallel(broadcast_buffers=True) does before every forward - every rank takes rank 0's buffers (the
    BatchNorm running statistics and step counters; reference tools/train.py:143 uses that default) - as ONE flat tensor per
    dtype: torch.cat of the buffers, one broadcast, one foreach copy back on the other ranks.  DDP's own buffer sync walks the
    ~150 small tensors of a detector in buckets and costs 0.85 ms of a 19 ms step on one MI355X (measured at world size 1);
    this form costs a handful of launches.  Same values in every buffer afterwards."""

    def __init__(self, module, process_group=None, src=0):
        import torch.distributed as dist
        self.dist = dist
        self.group = process_group
        self.src = src
        self.module = module
        self._groups = None

    def _collect(self):
        groups = {}
        seen = set()
        for b in self.module.buffers():
            if b is None or b.numel() == 0 or id(b) in seen:
                continue
            seen.add(id(b))
            groups.setdefault((b.dtype, b.device), []).append(b)
        return list(groups.values())

    def sync(self):
        dist = self.dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        groups = self._collect()          # every call: modules may re-point their buffers (aliased running statistics of the fused head)
        is_src = dist.get_rank(self.group) == self.src
        for bufs in groups:
            flat = torch.cat([b.detach().reshape(-1) for b in bufs])
            dist.broadcast(flat, self.src, group=self.group)
            if not is_src:
                with torch.no_grad():
                    dst, src = [], []
                    for b, piece in zip(bufs, flat.split([b.numel() for b in bufs])):
                        if b.is_contiguous():
                            dst.append(b.detach().view(-1))
                            src.append(piece)
                        else:
                            b.detach().copy_(piece.view(b.shape))
                    if dst:
                        torch._foreach_copy_(dst, src)


def wrap_ddp(model, device_ids=None, **kw):
    """DistributedDataParallel as the trainers use it (8 MB gradient buckets as bucket views) with the buffer broadcast of
    torch's default broadcast_buffers=True done by BufferBroadcaster before each forward."""
    kw.setdefault("gradient_as_bucket_view", True)
    kw.setdefault("bucket_cap_mb", 8)
    own = kw.pop("coalesced_buffer_broadcast", True)
    if own:
        kw["broadcast_buffers"] = False
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=device_ids, **kw)
    if own:
        bb = BufferBroadcaster(model)
        ddp.register_forward_pre_hook(lambda m, args: bb.sync())
        ddp.toda_buffer_broadcaster = bb
    return ddp
