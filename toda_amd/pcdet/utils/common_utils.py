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


class GradBucketReducer:
    """The gradient all-reduce of data-parallel training (reference tools/train.py:143: DistributedDataParallel) over a few
    flat buckets, without a kernel per parameter.  Why: optimizer.zero_grad() drops the gradients every step, so DDP's reducer
    finds fresh gradient tensors in every backward and copies each of them into its bucket with its own scaled-copy kernel -
    109 launches and 0.5 ms of a 19 ms CenterPoint step on one MI355X even at world size 1.  Here the parameters are bucketed
    in reverse registration order (the order backward produces them); when the last gradient of a bucket has been accumulated
    (post-accumulate hook) ONE torch.cat gathers the bucket's gradients into its flat buffer and one asynchronous all-reduce
    starts, overlapping the rest of backward; a callback at the end of the backward pass waits for the buckets, scales them by
    1 / world and re-points every .grad at its slice of the flat buffer (no copy back).  Same averaged gradients as DDP
    (tests/test_ddp_gloo.py).  no_sync(): gradients stay local, as DDP.no_sync()."""

    def __init__(self, params, bucket_bytes=8 << 20, process_group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.enabled = True
        params = [p for p in params if p.requires_grad]
        self.buckets = []
        cur, cur_bytes = [], 0
        for p in reversed(params):
            key = (p.dtype, p.device)
            if cur and (cur_bytes >= bucket_bytes or key != (cur[0].dtype, cur[0].device)):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
        if cur:
            self.buckets.append(cur)
        self.flat = [torch.empty((sum(p.numel() for p in b),), dtype=b[0].dtype, device=b[0].device) for b in self.buckets]
        self.where = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self.where[p] = bi
                p.register_post_accumulate_grad_hook(self._ready)
        self._reset()

    def _reset(self):
        self.pending = [len(b) for b in self.buckets]
        self.seen = [set() for _ in self.buckets]
        self.work = [None] * len(self.buckets)
        self.next_launch = 0          # collectives are issued strictly in bucket-index order (see _ready)
        self.armed = False

    def _launch(self, bi):
        b, flat = self.buckets[bi], self.flat[bi]
        # a trainer that keeps its gradients (zero_grad(set_to_none=False)) accumulates straight into the slices handed out by
        # finish(): nothing to gather then (and cat must not read what it writes)
        base, size, off, in_place = flat.data_ptr(), flat.element_size(), 0, True
        for p in b:
            if p.grad is None or p.grad.data_ptr() != base + off * size or not p.grad.is_contiguous():
                in_place = False
                break
            off += p.numel()
        if not in_place:
            grads = [p.grad.reshape(-1) if p.grad is not None else torch.zeros((p.numel(),), dtype=p.dtype, device=p.device) for p in b]
            if any(g.data_ptr() >= base and g.data_ptr() < base + flat.numel() * size for g in grads):
                grads = [g.clone() for g in grads]          # some still alias the buffer, some do not: gather from copies
            torch.cat(grads, out=flat)
        self.work[bi] = self.dist.all_reduce(flat, group=self.group, async_op=True)

    def _ready(self, p):
        if not self.enabled:
            return
        if not self.armed:          # first gradient of this backward pass: finish() runs when the pass ends
            self.armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.finish)
        bi = self.where[p]
        if id(p) in self.seen[bi]:
            return
        self.seen[bi].add(id(p))
        self.pending[bi] -= 1
        # Every rank must issue its collectives in the SAME order: a bucket goes out only once all buckets before it have gone
        # (as DDP's reducer does).  Launching in completion order would depend on each rank's autograd order and on which
        # parameters got a gradient in this pass - data dependent (an empty head, a loss term that is off) - and two ranks
        # issuing all-reduces in different orders hang or add up the wrong buckets.
        while self.next_launch < len(self.buckets) and self.pending[self.next_launch] == 0:
            self._launch(self.next_launch)
            self.next_launch += 1

    def finish(self):
        if not self.enabled:
            self._reset()
            return
        # the rest in index order too: buckets holding a parameter without a gradient in this pass (zeros reduced for it) and
        # whatever was queued behind them
        while self.next_launch < len(self.buckets):
            self._launch(self.next_launch)
            self.next_launch += 1
        scale = 1.0 / self.world
        for bi, b in enumerate(self.buckets):
            self.work[bi].wait()
            flat = self.flat[bi]
            if self.world > 1:
                flat.mul_(scale)
            off = 0
            for p in b:
                n = p.numel()
                p.grad = flat[off:off + n].view(p.shape)
                off += n
        self._reset()


class DataParallel(torch.nn.Module):
    """One replica per process: forward = the wrapped module's (after the buffer broadcast), gradients averaged over the ranks by
    GradBucketReducer during backward, parameters of all ranks set to rank 0's at construction - the surface of
    torch.nn.parallel.DistributedDataParallel that the trainers use (.module, no_sync(), state_dict of the wrapped module under
    the same `module.` prefix)."""

    def __init__(self, module, bucket_cap_mb=8, process_group=None):
        super().__init__()
        import torch.distributed as dist
        self.module = module
        self.broadcaster = BufferBroadcaster(module, process_group)
        self.toda_buffer_broadcaster = self.broadcaster
        with torch.no_grad():        # every rank starts from rank 0's parameters (DDP does this in its constructor)
            groups = {}
            for p in module.parameters():
                groups.setdefault((p.dtype, p.device), []).append(p)
            for ps in groups.values():
                flat = torch.cat([p.detach().reshape(-1) for p in ps])
                dist.broadcast(flat, 0, group=process_group)
                if dist.get_rank(process_group) != 0:
                    torch._foreach_copy_([p.detach().view(-1) if p.is_contiguous() else p.detach() for p in ps],
                                         [piece if p.is_contiguous() else piece.view(p.shape) for p, piece in zip(ps, flat.split([p.numel() for p in ps]))])
        self.reducer = GradBucketReducer(list(module.parameters()), int(bucket_cap_mb) << 20, process_group)

    def forward(self, *args, **kwargs):
        self.broadcaster.sync()
        return self.module(*args, **kwargs)

    def no_sync(self):
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old = self.reducer.enabled
            self.reducer.enabled = False
            try:
                yield
            finally:
                self.reducer.enabled = old
        return ctx()


def wrap_ddp(model, device_ids=None, **kw):
    """Data-parallel wrapper of the trainers and bench.py.  Default: torch's DistributedDataParallel with 8 MB bucket views, its
    buffer sync replaced by BufferBroadcaster unless coalesced_buffer_broadcast=False.  TODA_DDP=own (or reducer="own"):
    DataParallel above (flat-bucket gradient reducer, 0.5 ms / 109 launches per step cheaper on one MI355X).  The own reducer
    stays opt-in until it has run over RCCL with more than one GPU: it is covered by 2-rank gloo tests only (no multi-GPU box
    was available to any round), and a first N = 8 run is no place for a collective path that has never seen RCCL."""
    reducer = kw.pop("reducer", os.environ.get("TODA_DDP", "torch"))
    own_bcast = kw.pop("coalesced_buffer_broadcast", True)
    for m in model.modules():       # fused heads re-point their BatchNorm buffers: do it before a wrapper captures module.buffers()
        if hasattr(m, "alias_fused_buffers") and next(m.parameters(), torch.empty(0)).is_cuda:
            m.alias_fused_buffers()
    if reducer != "torch" and not kw.get("static_graph") and own_bcast:
        return DataParallel(model, bucket_cap_mb=kw.get("bucket_cap_mb", 8))
    kw.setdefault("gradient_as_bucket_view", True)
    kw.setdefault("bucket_cap_mb", 8)
    if own_bcast:
        kw["broadcast_buffers"] = False
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=device_ids, **kw)
    if own_bcast:
        bb = BufferBroadcaster(model)
        ddp.register_forward_pre_hook(lambda m, args: bb.sync())
        ddp.toda_buffer_broadcaster = bb
    return ddp
