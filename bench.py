#!/usr/bin/env python
"""bench.py — training samples/s of the TODA LiDAR-detection hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N>1: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the env: this process IS a rank), or
started plain, in which case this process is only a launcher: before anything touches the GPU it starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child (one fresh process per GPU, RCCL over xGMI),
lets rank 0's JSON line through and exits with the child's status (reference: tools/scripts/dist_train.sh:18,
tools/train.py:65-74,143, pcdet/utils/common_utils.py:161-176).

A step = one pass of the hot path over one batch of synthetic clouds.  --input host (default): the collated batch sits in PINNED
HOST memory and every step uploads it (points + boxes, 7.2 MB at C3) on the input stream inside the timed region, as the reference's
load_data_to_gpu does every iteration (pcdet/models/__init__.py:23-34); --input resident keeps the raw clouds in HBM (the line says
which one ran).  Then: GPU voxelisation + MeanVFE -> VoxelBackBone8x (rulebooks + sparse convs) -> HeightCompression
-> BaseBEVBackbone -> CenterHead (GPU target assignment, losses) -> backward -> grad-norm clip ->
Adam one-cycle step.  Workload at every N: BASELINE.json configs[2]/[3] (CenterPoint-Voxel on
180k-point Waymo-shape clouds, 2 samples per GPU, fp32) — the configuration the "training
samples/s" metric is quoted on; scenes are sharded across ranks (weak scaling), the only exchange
is the gradient all-reduce.  One JSON line on rank 0 (contract in the task prompt) carrying
`roofline` (dominant hand-written kernel, HIP events inside the timed region) and `cpu_baseline`
(the CPU oracle port + torch-CPU dense part, timed on the host cores, rank 0, N=1 only).
"""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from toda_amd import ops  # noqa: E402
from toda_amd.pcdet.config import AttrDict, cfg_from_yaml_file  # noqa: E402
from toda_amd.pcdet.datasets import SyntheticLidarDataset  # noqa: E402
from toda_amd.pcdet.models import build_network, voxelize_on_gpu  # noqa: E402
from toda_amd.tools.train_utils.optimization import build_optimizer, build_scheduler, clip_and_step, clip_grad_norm_  # noqa: E402

WORKLOADS = {
    # name: (yaml, samples per GPU, description)
    "c3": ("toda_amd/tools/cfgs/models/centerpoint_voxel_waymo.yaml", 2,
           "CenterPoint-Voxel fwd+bwd+optimizer, synthetic 180k-pt Waymo-shape clouds, bs 2 per GPU"),
    "c2": ("toda_amd/tools/cfgs/models/second_backbone_nuscenes.yaml", 4,
           "SECOND-style VoxelBackBone8x forward only (voxelize+MeanVFE+backbone+dense BEV), 60k-pt nuScenes-shape clouds, bs 4"),
    "c5": ("toda_amd/tools/cfgs/models/toda_stage1_centerpoint_res.yaml", 2,
           "TODA stage-1 CenterPoint (VoxelResBackBone8x), mixed 180k/35k-pt clouds, bs 2 per GPU"),
    "c5mix": ("toda_amd/tools/cfgs/models/toda_stage1_polarmix.yaml", 2,
              "TODA stage-1 step with the inter-domain PolarMix in the loop: mix (180k-pt source x 35k-pt target, on the device) "
              "+ range mask + shuffle + voxelize + CenterPoint (VoxelResBackBone8x) fwd+bwd+optimizer, bs 2 per GPU"),
    "c5cl": ("toda_amd/tools/cfgs/models/toda_stage1_centerpoint_res.yaml", 2,
             "TODA stage-2 consistency step (2 fwd + 1 bwd, VoxelResBackBone8x), mixed 180k/35k-pt clouds, bs 2 per GPU"),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: fp32 MFMA peak (dense)
# matrix path "split" (toda_amd/csrc/spconv_split.cuh): six bf16 matrix instructions stand for one fp32 product, so a kernel on that
# path is priced in fp32-EQUIVALENT FLOPs against the dense BF16 peak / 6 (the BF16 rate is 16 x the fp32 one: MI355X_MICROARCH.md,
# Matrix cores) - never against 157.3
MFMA_BF16_PEAK_TF = 16 * MFMA_F32_PEAK_TF
MFMA_SPLIT_PEAK_TF = round(MFMA_BF16_PEAK_TF / 6.0, 1)


def load_cfg(path):
    cfg = AttrDict()
    cfg_from_yaml_file(os.path.join(ROOT, path), cfg)
    return cfg


def _place(a, device, host):
    """fp32 tensor of a collated array: pinned host memory (uploaded per step by load_data_to_gpu) or resident on the device."""
    t = torch.from_numpy(np.ascontiguousarray(a)).float()
    return t.pin_memory() if host else t.to(device)


def make_device_batches(dataset, per_gpu, n_batches, rank, device, host=False):
    """Collated batches with `points` / `gt_boxes` in pinned host memory (host=True) or already on the device."""
    batches = []
    for b in range(n_batches):
        base = (rank * n_batches + b) * per_gpu
        samples = [dataset[(base + i) % len(dataset)] for i in range(per_gpu)]
        col = dataset.collate_batch(samples)
        out = {"batch_size": col["batch_size"], "points_per_sample": col["points_per_sample"]}
        out["points"] = _place(col["points"], device, host)
        out["gt_boxes"] = _place(col["gt_boxes"], device, host)
        batches.append(out)
    return batches


def make_device_pair_batches(dataset, per_gpu, n_batches, rank, device, host=False):
    """(adv, org) batches of the stage-2 consistency step, tensors in pinned host memory or resident on the device."""
    out = []
    for b in range(n_batches):
        base = (rank * n_batches + b) * per_gpu
        pair = dataset.collate_batch([dataset[(base + i) % len(dataset)] for i in range(per_gpu)])
        dev = []
        for col in pair:
            d = {k: v for k, v in col.items() if k in ("batch_size", "points_per_sample", "augmentation_list", "augmentation_params")}
            d["points"] = _place(col["points"], device, host)
            d["gt_boxes"] = _place(col["gt_boxes"], device, host)
            dev.append(d)
        out.append(tuple(dev))
    return out


class KernelTimer:
    """Per-launch durations of the gather-GEMM kernels inside the timed region.  The library stamps a start / stop
    event pair on each kernel dispatch (toda_timing_begin / toda_timing_end, hipExtLaunchKernelGGL), so a duration is the
    kernel's own - the number rocprofv3 --kernel-trace reports - and no host sync happens while the clock runs.
    ops.gather_gemm is wrapped only to remember each launch's shape in launch order."""

    CAPACITY = 512          # launches timed and labelled (the pair counters of their tables stay referenced until the summary)

    def __init__(self):
        self.records = []          # (pair counters [K] of the table, n_out, K, n_src, c_gather, c_produce) in launch order
        self.ms = []
        self._enabled = False
        self._orig = ops.gather_gemm

        def labelled(feat, wp, nbr, c_produce, bias=None, order=None):
            if self._enabled and len(self.records) < self.CAPACITY:
                self.records.append(self._record_of(nbr, feat, c_produce))
            return self._orig(feat, wp, nbr, c_produce, bias, order)

        ops.gather_gemm = labelled
        self._orig_stats = ops.gather_gemm_with_stats      # forward convs whose epilogue also takes the BatchNorm moments

        def labelled_stats(feat, wp, nbr, c_produce, bias=None, **kw):
            if self._enabled and len(self.records) < self.CAPACITY:
                self.records.append(self._record_of(nbr, feat, c_produce))
            return self._orig_stats(feat, wp, nbr, c_produce, bias, **kw)

        ops.gather_gemm_with_stats = labelled_stats
        self._orig_classed = ops.gather_gemm_classed        # data gradient of the strided convs (same dispatch-stamped launches)

        def labelled_classed(feat, wp, nbr, c_produce, *rest):
            if self._enabled and len(self.records) < self.CAPACITY:
                self.records.append(self._record_of(nbr, feat, c_produce))
            return self._orig_classed(feat, wp, nbr, c_produce, *rest)

        ops.gather_gemm_classed = labelled_classed

    @property
    def enabled(self):
        return self._enabled

    @enabled.setter
    def enabled(self, on):
        from toda_amd import lib as L
        lib = L.load()
        if on and not self._enabled:
            L.check(lib.toda_timing_begin(self.CAPACITY), "toda_timing_begin")
        elif not on and self._enabled:
            import ctypes
            buf = (ctypes.c_float * self.CAPACITY)()
            seen = ctypes.c_int(0)
            L.check(lib.toda_timing_end(ctypes.cast(buf, ctypes.c_void_p), self.CAPACITY, ctypes.cast(ctypes.pointer(seen), ctypes.c_void_p)),
                    "toda_timing_end")
            # every dispatch the library stamped must have been labelled here, in order: a gather-GEMM entry point that is not
            # wrapped above would shift all following durations onto the wrong shapes
            assert min(seen.value, self.CAPACITY) == len(self.records), \
                f"KernelTimer: {seen.value} timed launches but {len(self.records)} labelled ones"
            self.ms = list(buf[:min(seen.value, self.CAPACITY, len(self.records))])
        self._enabled = bool(on)

    def summary(self):
        """Per launch shape: mean ms, algorithmic bytes and flops (valid pairs counted exactly, from the counters the rulebook
        kernels left - never from the tables: with the arena those are views into slots later batches have overwritten)."""
        groups = {}
        cnts = torch.stack([c.to(torch.int64).sum() for c, *_ in self.records]).tolist() if self.records else []
        for (cnt, n_out, K, n_src, cg, cp), ms, pairs in zip(self.records, self.ms, cnts):
            pairs = int(pairs)
            key = (n_out, K, pairs, n_src, cg, cp)      # one group per rulebook (its row AND pair count) and channel pair, over all steps
            g = groups.setdefault(key, {"ms": [], "pairs": pairs, "n_out": n_out, "K": K, "n_src": n_src, "cg": cg, "cp": cp})
            g["ms"].append(ms)
        return groups

    @staticmethod
    def _record_of(nbr, feat, c_produce):
        cnt = getattr(nbr, "_toda_pair_cnt", None)
        if cnt is None:          # a table that did not come out of ops.Rulebook: count now (one reduction on the current stream)
            cnt = (nbr >= 0).sum().reshape(1)
        return (cnt, nbr.shape[1], nbr.shape[0], feat.shape[0], feat.shape[1], c_produce)


def run_gpu(args, rank, world, device):
    yaml_path, per_gpu, desc = WORKLOADS[args.workload]
    if getattr(args, "per_gpu", None):          # exploration only: the BASELINE configs fix the per-GPU batch
        per_gpu = int(args.per_gpu)
        desc += f" [per-GPU batch overridden to {per_gpu}]"
    cfg = load_cfg(yaml_path)
    pair = args.workload.endswith("cl")
    mixed = args.workload.endswith("mix")
    host_input = getattr(args, "input", "host") == "host" and os.environ.get("TODA_PREFETCH", "1") == "1"
    if pair:
        from toda_amd.pcdet.datasets import SyntheticPairDataset
        from toda_amd.pcdet.models import DistModel, model_fn_decorator_cl
        dataset = SyntheticPairDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
    elif mixed:
        from toda_amd.pcdet.datasets import SyntheticMixDataset
        # the raw clouds of both domains are generated before the clock starts and kept in pinned host memory (every access uploads the
        # frame: the mix runs on the device) or, --input resident, in HBM
        cfg.DATA_CONFIG.CACHE_FRAMES = "host" if host_input else True
        for step_cfg in cfg.DATA_CONFIG.DATA_PROCESSOR:
            if step_cfg.NAME == "shuffle_points":
                step_cfg.SHUFFLE_ON_DEVICE = True   # permutation drawn by torch on the device instead of numpy on the host
        cfg.DATA_CONFIG.SYNTHETIC.NUM_SOURCE = cfg.DATA_CONFIG.SYNTHETIC.NUM_TARGET = 2 * per_gpu * world
        dataset = SyntheticMixDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
        np.random.seed(4321 + rank)
    else:
        dataset = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
    torch.manual_seed(1234)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), dataset).to(device)
    model.train()
    optimizer = build_optimizer(model, cfg.OPTIMIZATION)
    total_steps = max(args.steps + args.warmup, 10) + 32      # + the untimed steps of --layers / the comm report
    scheduler, _ = build_scheduler(optimizer, total_steps, 1, -1, cfg.OPTIMIZATION)
    net = model
    if pair:
        model = DistModel(model)
    if world > 1 or os.environ.get("TODA_FORCE_DDP") == "1":  # the env knob rehearses the DDP path on one GPU
        # 23-31 MB of fp32 gradients: 8 MB buckets let the all-reduce of the dense part's gradients (ready first) run over
        # xGMI while the sparse backbone is still in backward; the default 25 MB would make it one bucket at the very end
        ddp_kw = dict(gradient_as_bucket_view=True, bucket_cap_mb=8)
        # buffers: rank 0's BN running statistics reach every rank before each forward, as with the reference's
        # DistributedDataParallel default (tools/train.py:143) - done by wrap_ddp as one flat broadcast per dtype (DDP's own
        # per-tensor buffer sync costs 0.85 ms of a 19 ms step; TODA_DDP_BCAST_BUFFERS=torch selects it, =0 none at all)
        bcast = os.environ.get("TODA_DDP_BCAST_BUFFERS", "1")
        if bcast == "0":
            ddp_kw.update(broadcast_buffers=False, coalesced_buffer_broadcast=False)
        elif bcast == "torch":
            ddp_kw.update(broadcast_buffers=True, coalesced_buffer_broadcast=False)
        if os.environ.get("TODA_DDP_STATIC", "0") == "1":
            ddp_kw["static_graph"] = True
        if os.environ.get("TODA_DDP_BUCKET_MB"):
            ddp_kw["bucket_cap_mb"] = int(os.environ["TODA_DDP_BUCKET_MB"])
        from toda_amd.pcdet.utils.common_utils import wrap_ddp
        model = wrap_ddp(model, device_ids=[device.index], **ddp_kw)
    if pair:
        batches = make_device_pair_batches(dataset, per_gpu, args.batches, rank, device, host_input)
        cl_fn = model_fn_decorator_cl()
    elif mixed:
        batches = None
        for i in range(len(dataset)):            # generate + upload every raw frame before the clock starts
            dataset._frame(dataset.source_kind, i % dataset.num_source)
            dataset._frame(dataset.target_kind, 100_000 + i % dataset.num_target)
    else:
        batches = make_device_batches(dataset, per_gpu, args.batches, rank, device, host_input)
    params = [p for p in net.parameters() if p.requires_grad]
    clip = cfg.OPTIMIZATION.GRAD_NORM_CLIP
    timer = KernelTimer()

    import collections
    phases = collections.defaultdict(float) if os.environ.get("TODA_BENCH_PHASES") else None
    fwd_only = args.workload == "c2"
    # device-side input pipeline as in tools/train_utils/train_utils.py (the reference voxelises in DataLoader workers, concurrently
    # with training): step t + 1's voxelisation and rulebooks run on a side stream while step t's backward runs.  Every timed step
    # still contains one voxelisation + index build (the one for the next step).  C3 18.1-18.2 ms/step with it, 18.8-19.0 without
    # (TODA_PREFETCH=0); C5 25.7-26.0 against 26.4-26.7.
    prefetch = None

    def mixed_batch(it):
        # the whole input path of a TODA stage-1 step: mix -> range mask -> shuffle -> collate, all on the device
        base = ((it * world + rank) * per_gpu) % len(dataset)
        batch = dataset.collate_batch([dataset[(base + i) % len(dataset)] for i in range(per_gpu)])
        batch["gt_boxes"] = torch.from_numpy(batch["gt_boxes"]).float().to(device)
        return {k: batch[k] for k in ("points", "points_per_sample", "gt_boxes", "batch_size")}

    # c5mix: the mix -> mask -> shuffle -> collate chain (small kernels with host decisions and syncs between them) rides on the side
    # stream as well: 75.5-75.9 samples/s against 69.9-70.2 on the training stream.
    if os.environ.get("TODA_PREFETCH", "1") == "1":
        from toda_amd.pcdet.models import InputPrefetcher

        def batch_stream():
            it = 0
            while True:
                if pair:
                    adv, org = batches[it % len(batches)]
                    yield dict(adv), dict(org)
                elif mixed:
                    yield mixed_batch(it)
                else:
                    yield dict(batches[it % len(batches)])
                it += 1

        prefetch = InputPrefetcher(batch_stream(), net, device)

    # experiments (not a workload): TODA_BENCH_REUSE_BATCH=1 prepares ONE batch and trains on it every step (no input pipeline at all);
    # TODA_BENCH_THROTTLE=k holds the host at most k steps ahead of the GPU (host-side wait for the end of step t - k)
    step_sizes = []          # (points, voxels) of every timed step
    reuse = None
    throttle = int(os.environ.get("TODA_BENCH_THROTTLE", "0"))
    step_done = []
    if os.environ.get("TODA_BENCH_REUSE_BATCH") == "1" and not pair and not mixed and not fwd_only:
        from toda_amd.pcdet.models import prepare_batch_on_gpu
        if prefetch is not None:
            prefetch.close()
            prefetch = None
        with torch.no_grad():
            reuse = prepare_batch_on_gpu(dict(batches[0]), net, dataset.voxel_cfg)
        torch.cuda.synchronize()

    def step(it):
        if throttle:
            if len(step_done) >= throttle:
                step_done[-throttle].synchronize()
        if reuse is not None:
            scheduler.step(it)
            optimizer.zero_grad()
            ret, tb, _ = model(dict(reuse))
            loss = ret["loss"].mean()
            loss.backward()
            clip_and_step(optimizer, params, clip)
            net.update_global_step()
            if throttle:
                ev = torch.cuda.Event()
                ev.record()
                step_done.append(ev)
                del step_done[:-8]
            return loss
        if fwd_only:  # BASELINE config 2: inference through the sparse backbone only
            with torch.no_grad():
                if prefetch is not None:
                    batch = prefetch.next()
                    if prefetch.threaded:
                        prefetch.kick()      # the worker thread prepares the next batch while this thread enqueues the forward
                else:
                    batch = dict(batches[it % len(batches)])
                    voxelize_on_gpu(batch, dataset.voxel_cfg)
                for m in (net.vfe, net.backbone_3d, net.map_to_bev_module):
                    batch = m(batch)
                out = batch["spatial_features"].sum()
                if prefetch is not None:
                    prefetch.kick()          # (same thread: the next batch's voxelisation + rulebooks on the side stream, under this forward)
            return out
        ph = phases if timer.enabled else None          # host-side phase clock of the timed steps (TODA_BENCH_PHASES)
        t_ph = time.perf_counter()
        scheduler.step(it)
        optimizer.zero_grad()
        if pair:
            adv, org = prefetch.next() if prefetch is not None else batches[it % len(batches)]
            loss = cl_fn(model, dict(adv), dict(org), world > 1).loss
        else:
            if prefetch is not None:
                batch = prefetch.next()
            elif mixed:
                batch = mixed_batch(it)
            else:
                batch = dict(batches[it % len(batches)])
            if prefetch is None:
                voxelize_on_gpu(batch, dataset.voxel_cfg)
            if timer.enabled:       # shapes only (no read-back): the step's input size beside its time
                step_sizes.append((int(batch["points"].shape[0]), int(batch["voxel_coords"].shape[0])))
            if ph is not None:
                ph["next"] += time.perf_counter() - t_ph
                t_ph = time.perf_counter()
            ret, tb, _ = model(batch)
            loss = ret["loss"].mean()
        if ph is not None:
            ph["forward"] += time.perf_counter() - t_ph
            t_ph = time.perf_counter()
        loss.backward()
        if ph is not None:
            ph["backward"] += time.perf_counter() - t_ph
            t_ph = time.perf_counter()
        clip_and_step(optimizer, params, clip)
        if ph is not None:
            ph["clip+optimizer"] += time.perf_counter() - t_ph
            t_ph = time.perf_counter()
        if prefetch is not None:
            # next batch's index work goes to the side stream once this step's backward + optimizer are ENQUEUED: the host then
            # sits in the side stream's two syncs while the GPU still has the whole backward to run (kicking before backward() -
            # the first version - parked the host there with nothing queued behind the forward)
            prefetch.kick()
        if ph is not None:
            ph["kick (next batch: voxelise + plan, 2 syncs)"] += time.perf_counter() - t_ph
        if not pair:
            net.update_global_step()
        return loss

    # The cyclic garbage collector is paused over the warm-up + timed steps (one collection up front): a generation-2 pass over
    # the interpreter's heap (torch, sympy, the model tree) lands as a 10+ ms stall inside a random step and is the kind of
    # host-side outlier a single driver run cannot average away.  Reference counting still frees every tensor at once.
    import gc
    gc.collect()
    gc.disable()
    loss_log = [] if os.environ.get("TODA_BENCH_LOSSES") else None       # every step's loss (device tensors, read after the clock has stopped)
    for it in range(args.warmup):
        loss = step(it)
        if loss_log is not None:
            loss_log.append(loss.detach())
    # everything the timed region needs is set up BEFORE the synchronisation (1024 + K events, the library's dispatch stamps): the GPU
    # should idle for microseconds, not milliseconds, between the last warm-up kernel and the first timed one (an idle GPU drops its
    # clocks; the first timed step of a run measured 1-1.6 ms longer than the steps after it)
    timer.enabled = True
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cpu0 = time.thread_time()
    marks[0].record()
    # next to each step's GPU time (event to event): the host time to enqueue it and the allocator's counters, so that a slow step
    # can be told apart as host-side (long enqueue, GPU starved), allocator-side (a device allocation or a retry inside the step)
    # or GPU-side (neither) - VERDICT r2 item 5
    host_step_s, alloc_marks = [], []

    def alloc_counters():
        st = torch.cuda.memory_stats(device)
        return (int(st.get("num_device_alloc", 0)), int(st.get("num_alloc_retries", 0)))

    alloc_marks.append(alloc_counters())
    for k, it in enumerate(range(args.warmup, args.warmup + args.steps)):
        th = time.perf_counter()
        loss = step(it)
        if loss_log is not None:
            loss_log.append(loss.detach())
        marks[k + 1].record()        # no sync: the median step time is read after the clock has stopped
        host_step_s.append(time.perf_counter() - th)
        alloc_marks.append(alloc_counters())
    t_issued = time.perf_counter() - t0      # host wall time to ENQUEUE the K steps (includes waiting at the two host syncs per step)
    cpu_busy = time.thread_time() - cpu0     # CPU time of this thread over the same span: what the host really WORKS per step
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    timer.enabled = False
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if prefetch is not None and getattr(prefetch, "arena", None) is not None and rank == 0 and os.environ.get("TODA_BENCH_STEP_MS"):
        print(f"[arena] slots {len(prefetch.arena.slots)}, waits for a free slot {prefetch.arena.waits}, buffers allocated {prefetch.arena.grown}", file=sys.stderr)
    if prefetch is not None and getattr(prefetch, "_trace", None) and rank == 0:
        tr = prefetch._trace[-args.steps:]
        print("[prefetch] preparation wall ms (of which waiting for the counts): " + " ".join(f"{a * 1e3:.1f}({b * 1e3:.1f})" for a, b in tr), file=sys.stderr)
    if phases is not None and rank == 0:
        print("[host phases, ms per step] " + ", ".join(f"{k} {v / args.steps * 1e3:.2f}" for k, v in phases.items()), file=sys.stderr)
    final_loss = float(loss.item())
    if loss_log is not None and rank == 0:
        print("[losses] " + " ".join(f"{float(v):.7g}" for v in torch.stack([v.float().reshape(()) for v in loss_log]).tolist()), file=sys.stderr)
    assert np.isfinite(final_loss), "training diverged"
    step_ms = [marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps)]
    op_rows = None
    if args.layers and rank == 0:
        from toda_amd.tools.op_table import OpTable
        table = OpTable()
        table.enabled = True
        extra = 5
        gc.disable()          # a collection inside an event bracket reads as milliseconds of "kernel" time in that row
        for it in range(args.warmup + args.steps, args.warmup + args.steps + extra):
            step(it)
        gc.enable()
        table.enabled = False
        op_rows = table.rows(extra)
        table.restore()
    if os.environ.get("TODA_CPROFILE") and rank == 0:
        # host-side cost per Python function over 10 extra steps (the step is close to host-bound: CPU ms / step is in the line)
        import cProfile
        import pstats
        prof = cProfile.Profile()
        prof.enable()
        for it in range(args.warmup + args.steps + 5, args.warmup + args.steps + 15):
            step(it)
        torch.cuda.synchronize()
        prof.disable()
        with open(os.environ["TODA_CPROFILE"], "w") as f:
            pstats.Stats(prof, stream=f).sort_stats("tottime").print_stats(60)
    if os.environ.get("TODA_TORCH_PROFILE") and rank == 0:
        # which Python line launches which small kernel: two extra steps under torch.profiler, grouped by call stack
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            for it in range(args.warmup + args.steps + 3, args.warmup + args.steps + 5):
                step(it)
            torch.cuda.synchronize()
        with open(os.environ["TODA_TORCH_PROFILE"], "w") as f:
            f.write(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=150, max_name_column_width=60,
                                                                 max_src_column_width=110))
    comm = None
    if world > 1 and not fwd_only:
        comm = comm_report(model, net, step, args, world, device, elapsed / args.steps * 1e3,
                           backend="gloo, ranks SHARING one GPU (TODA_BENCH_SHARE_GPU rehearsal)" if os.environ.get("TODA_BENCH_SHARE_GPU") == "1" else "rccl")
    return {"host_input": host_input, "step_sizes": step_sizes, "elapsed": elapsed, "per_gpu": per_gpu, "desc": desc, "loss": final_loss, "timer": timer, "cfg": cfg,
            "dataset": dataset, "model": net, "step_ms": step_ms, "comm": comm, "op_rows": op_rows, "issue_s": t_issued, "cpu_s": cpu_busy,
            "host_step_ms": [t * 1e3 for t in host_step_s],
            "device_allocs": [alloc_marks[k + 1][0] - alloc_marks[k][0] for k in range(args.steps)],
            "alloc_retries": [alloc_marks[k + 1][1] - alloc_marks[k][1] for k in range(args.steps)]}


# (gathered channels, produced channels, K) of a dominant launch shape -> (PMC summary under profiles/, kernel instantiation,
# threads per output row of that instantiation, slack of the grid in threads)
PMC_FILES = {("native", 64, 64, 27): ("r05_pmc_gather_gemm_64x64.json", "gather_gemm_lds_kernel<4, 4, 2", 2, 1024),       # 64 lanes per 32-row tile
             ("native", 128, 128, 27): ("r05_pmc_gather_gemm_128x128.json", "gather_gemm_lds_kernel<8, 8, 1", 4, 2048),  # 512 threads per 128 rows
             ("split", 64, 64, 27): ("r05_pmc_split_64x64.json", "gg_split_kernel<2, 1, 4, 2", 2, 1024),
             ("split", 128, 128, 27): ("r05_pmc_split_128x128.json", "gg_split_kernel<4, 1, 8, 2", 2, 1024),
             ("native", 32, 32, 27): ("r05_pmc_gather_gemm_32x32.json", "gather_gemm_lds_kernel<2, 2, 2", 2, 1024),
             ("split", 32, 32, 27): ("r05_pmc_split_32x32.json", "gg_split_kernel<1, 1, 2, 2", 2, 1024)}
PMC_SOURCES = ("toda_amd/csrc/spconv.hip", "toda_amd/csrc/spconv_split.cuh", "toda_amd/csrc/split_common.cuh")


def _sha256(path):
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def pmc_traffic(d):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary (separate FETCH_SIZE / WRITE_SIZE
    passes of this same command; PMC counters cannot be read from inside bench.py).  The summary records the SHA-256 of the
    kernel source it was collected on: when toda_amd/csrc/spconv.hip has changed since, or no profiled launch shape matches,
    the traffic is reported as null instead of a stale number.  Returns (bytes or None, where it came from)."""
    ent = PMC_FILES.get((d["path"], d["c_gather"], d["c_produce"], d["K"]))
    if ent is None:
        return None, "dominant launch shape is not one of the profiled kernels (32->32, 64->64, 128->128 at K=27)"
    name, inst, per_row, slack = ent
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, f"no PMC summary profiles/{name} committed"
    pmc = json.load(open(path))
    for rel in PMC_SOURCES:
        if pmc.get("source_sha256", {}).get(rel) != _sha256(os.path.join(ROOT, rel)):
            return None, f"profiles/{name} was collected on another version of {rel} (stale)"
    for shape in pmc["launch_shapes"]:
        if inst not in shape.get("kernel", "") or "hbm_bytes" not in shape:
            continue
        if 0 <= shape["grid_threads"] // per_row - d["n_out"] < slack // per_row:
            return shape["hbm_bytes"], (f"profiles/{name}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (2*FETCH + WRITE), "
                                        f"{shape['kernel']}, grid {shape['grid_threads']}, {shape['dispatches']} dispatches")
    return None, "no profiled launch shape matches"


def roofline_from_timer(timer):
    """Dominant gather-GEMM launch shape (largest total time inside the timed region)."""
    from toda_amd import lib as L
    groups = timer.summary()
    if not groups:
        return None, []
    split_on = ops.matrix_path() == "split"
    rows = []
    for g in groups.values():
        ms = float(np.mean(g["ms"]))
        # B = 4*(N_in*Cin + N_out*Cout + K*Cin*Cout) + 8*Pairs ; FLOPs = 2*Pairs*Cin*Cout
        bytes_alg = 4.0 * (g["n_src"] * g["cg"] + g["n_out"] * g["cp"] + g["K"] * g["cg"] * g["cp"]) + 8.0 * g["pairs"]
        flops = 2.0 * g["pairs"] * g["cg"] * g["cp"]
        # which matrix instructions this launch ran on: the split path covers the pairs toda_spconv_split_supported names (and only
        # the launches that go through the packed operand: the narrow layers' compaction kernel never does)
        path = "split" if (split_on and L.load().toda_spconv_split_supported(int(g["cg"]), int(g["cp"]))) else "native"
        mfma_peak = MFMA_SPLIT_PEAK_TF if path == "split" else MFMA_F32_PEAK_TF
        t_hbm, t_mfma = bytes_alg / (HBM_PEAK_GBS * 1e9), flops / (mfma_peak * 1e12)
        rows.append({"n_out": g["n_out"], "K": g["K"], "c_gather": g["cg"], "c_produce": g["cp"], "pairs": g["pairs"],
                     "launches": len(g["ms"]), "ms": ms, "total_ms": float(np.sum(g["ms"])), "bytes": bytes_alg,
                     "flops": flops, "bound": "hbm" if t_hbm >= t_mfma else "mfma", "path": path, "mfma_peak": mfma_peak,
                     "frac": max(t_hbm, t_mfma) / (ms * 1e-3),
                     "frac_of_fp32_mfma_peak": flops / (MFMA_F32_PEAK_TF * 1e12) / (ms * 1e-3)})
    rows.sort(key=lambda r: -r["total_ms"])
    d = rows[0]
    if d["bound"] == "hbm":
        achieved, peak, unit = d["bytes"] / (d["ms"] * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
    else:
        achieved, peak, unit = d["flops"] / (d["ms"] * 1e-3) / 1e12, d["mfma_peak"], "TFLOP/s"
    traffic, traffic_source = pmc_traffic(d)
    roof = {"bound": d["bound"], "achieved": round(achieved, 3), "peak": peak, "unit": unit,
            "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
            "kernel": f"{'gg_split_kernel' if d['path'] == 'split' else 'gather_gemm_kernel'} rows={d['n_out']} K={d['K']} "
                      f"{d['c_gather']}->{d['c_produce']} pairs={d['pairs']}",
            "avg_launch_ms": round(d["ms"], 4), "algorithmic_bytes": int(d["bytes"])}
    if d["path"] == "split":
        roof["peak_note"] = (f"fp32-equivalent FLOPs (2 x pairs x Cin x Cout) against the dense BF16 MFMA peak {MFMA_BF16_PEAK_TF:.0f} / 6: each fp32 product "
                             f"is six bf16 matrix instructions (hi/mid/lo split, fp32 accumulate); the same launch is "
                             f"{d['flops'] / (d['ms'] * 1e-3) / 1e12 / MFMA_F32_PEAK_TF:.3f} of the fp32 MFMA peak {MFMA_F32_PEAK_TF}")
    return roof, rows


def cpu_baseline(cfg, workload, n_scenes=5):
    """The SAME workload on the host cores: CPU oracle for the sparse part (oracle/, "port"), torch CPU for the dense
    part, on a BOUNDED sample of it (a few full-size scenes at bs 1, one step each; ~10-30 s of CPU work): a full train
    step for c3 / c5 / c5mix, the forward-only backbone pass for c2, the 2 fwd + 1 bwd consistency step for c5cl.
    Point counts, ranges and model are the workload's own config."""
    from oracle.cpu_backend import oracle_backend
    from toda_amd.pcdet.models import model_fn_decorator

    def say(msg):
        print(f"[cpu_baseline] {msg}", file=sys.stderr, flush=True)

    mixed, pair, fwd_only = workload.endswith("mix"), workload.endswith("cl"), workload == "c2"
    cfg = copy.deepcopy(cfg)
    if "KINDS" not in cfg.DATA_CONFIG.SYNTHETIC and "KIND" not in cfg.DATA_CONFIG.SYNTHETIC:
        cfg.DATA_CONFIG.SYNTHETIC.KINDS = [cfg.DATA_CONFIG.SYNTHETIC.get("SOURCE_KIND", "waymo_toda")]
    if pair:
        from toda_amd.pcdet.datasets import SyntheticPairDataset
        dataset = SyntheticPairDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
    else:
        dataset = SyntheticLidarDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
    torch.manual_seed(1234)
    from oracle import oracle as O

    # the GPU box reports every host core (256) but one GPU's share is 16: never oversubscribe
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get("TODA_CPU_BASELINE_THREADS", "16"))))
    torch.set_num_threads(cores)
    cores = O.set_threads(cores)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), dataset)
    model.train()
    optimizer = build_optimizer(model, cfg.OPTIMIZATION)
    fn = model_fn_decorator()
    mix_s = 0.0
    if mixed:   # the CPU path of the mix: numpy restatement of the reference's PolarMix (oracle/mix.py) on a 180k x 35k pair
        from oracle import mix as OM
        from toda_amd.pcdet.datasets.synthetic import synth_cloud
        dc = cfg.DATA_CONFIG
        frames = []
        for kind, seed in ((dc.SYNTHETIC.SOURCE_KIND, 5000), (dc.SYNTHETIC.TARGET_KIND, 105000)):
            pts, bx, _ = synth_cloud(kind, seed, class_count=len(cfg.CLASS_NAMES))
            frames.append({"points": pts[:, :4].copy(), "gt_boxes": np.concatenate([bx, np.ones((len(bx), 1), np.float32)], 1)})
        np.random.seed(4321)
        t0 = time.perf_counter()
        mixed_scene = OM.polarmix(frames[0], frames[1], int(dc.POLARMIX_RC_NUM), dc.POLARMIX_DEGREE, 0.0, list(dc.POLARMIX_UPDATE_METHOD),
                                  dc.MIX_INC_METHOD)
        mix_s = time.perf_counter() - t0
        data = dataset.data_processor.forward({"points": mixed_scene["points"], "gt_boxes": mixed_scene["gt_boxes"], "use_lead_xyz": True})
        batches = [dataset.collate_batch([data])]
        say(f"PolarMix of one scene pair on the host: {mix_s:.2f} s")
    else:
        batches = [dataset.collate_batch([dataset[i]]) for i in range(n_scenes)]
    npts = [int(len((b[0] if pair else b)["points"])) for b in batches]
    say(f"{len(batches)} scene(s) of {npts} points on {cores} host threads ...")
    if pair:
        from toda_amd.pcdet.models import DistModel, model_fn_decorator_cl
        cl_fn, cl_model = model_fn_decorator_cl(), DistModel(model)
    with oracle_backend():
        t0 = time.perf_counter()
        for i, batch in enumerate([batches[0]] + batches):        # one step per scene (bs 1); the first one is an UNTIMED warm-up
            if i == 1:                                             # (first-touch of the worker threads' buffers, torch's thread pool: 2.1 s against 1.7 s)
                t0 = time.perf_counter()
            if fwd_only:
                with torch.no_grad():
                    b = dict(batch)
                    from toda_amd.pcdet.models import load_data_to_gpu
                    load_data_to_gpu(b)
                    voxelize_on_gpu(b, dataset.voxel_cfg)      # ops.voxelize_batch is the oracle's inside oracle_backend()
                    for m in (model.vfe, model.backbone_3d, model.map_to_bev_module):
                        b = m(b)
            else:
                optimizer.zero_grad()
                if pair:
                    loss = cl_fn(cl_model, dict(batch[0]), dict(batch[1]), False).loss
                else:
                    loss = fn(model, dict(batch)).loss
                loss.backward()
                clip_and_step(optimizer, list(model.parameters()), cfg.OPTIMIZATION.GRAD_NORM_CLIP)
            say(f"warm-up step done after {time.perf_counter() - t0:.1f} s" if i == 0 else f"step {i}/{len(batches)} done after {time.perf_counter() - t0:.1f} s")
        dt = time.perf_counter() - t0 + mix_s
    what = {"c2": "forward pass voxel features -> VoxelBackBone8x -> dense BEV (no gradients)",
            "c5cl": "consistency step (2 forwards + 1 backward + optimizer)"}.get(workload, "full train step (fwd + bwd + clip + Adam)")
    return {"value": round(len(batches) / dt, 4), "unit": "samples/s", "cores": cores, "kind": "port",
            "cores_note": f"{avail} CPUs in this rank's affinity mask; threads capped at the GPU box's share for one GPU (16) unless "
                          "TODA_CPU_BASELINE_THREADS says otherwise",
            "sample": f"one untimed warm-up step, then {len(batches)} scene(s) of {sorted(set(npts))} points from this workload's own generator (bs 1), one {what} each = "
                      f"{dt:.1f} s in total; sparse part = oracle/ C port (OpenMP, AVX2), dense part = torch CPU"}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _allowed_cpus():
    try:
        return sorted(os.sched_getaffinity(0))
    except AttributeError:      # pragma: no cover
        return list(range(os.cpu_count() or 1))


def place_rank(local_rank, world):
    """Pin this rank to its share of the CPUs the job may use (contiguous block per LOCAL_RANK) BEFORE anything touches the GPU:
    the HIP runtime's threads and the prefetch worker inherit the mask, and eight ranks stop migrating over each other's cores.
    TODA_BENCH_AFFINITY=0 leaves the mask alone.  Returns the CPUs of this rank."""
    cpus = _allowed_cpus()
    if os.environ.get("TODA_BENCH_AFFINITY", "1") != "1" or world <= 1 or not hasattr(os, "sched_setaffinity"):
        return cpus
    per = max(1, len(cpus) // world)
    mine = cpus[(local_rank * per) % len(cpus):][:per] or cpus
    try:
        os.sched_setaffinity(0, mine)
    except OSError:             # a cgroup that refuses: keep the inherited mask
        return cpus
    torch.set_num_threads(max(1, min(per, int(os.environ.get("OMP_NUM_THREADS", per)))))
    return mine


def launch_ranks(args, argv):
    """--gpus N > 1 without a rank environment: this process never touches the GPU.  It starts one fresh process per GPU
    through torch.distributed.run (the same command line the driver uses), lets their output through (rank 0 prints the
    JSON line) and returns the launcher's exit status (non-zero when any rank failed)."""
    import subprocess
    dry = os.environ.get("TODA_BENCH_DRYRUN") == "1"
    if not dry and os.environ.get("TODA_BENCH_SHARE_GPU") != "1":
        have = torch.cuda.device_count()      # counting devices does not initialise HIP
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible on this node", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver (RCCL needs it)
    # host budget per rank: the CPUs this process may run on, dealt evenly (a GPU box exposes every core of the host, its share per
    # GPU is 16; 8 ranks of a node therefore live on 2 cores each: main thread + prefetch worker)
    env.setdefault("OMP_NUM_THREADS", str(max(1, len(_allowed_cpus()) // max(args.gpus, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, rank, world):
    """TODA_BENCH_DRYRUN=1: the N > 1 path of this file without a GPU, over gloo.  Plain: rendezvous + one all-reduce (exercises
    the launcher).  With TODA_BENCH_DRYRUN_REHEARSAL=module:function the ranks also run what the driver's N = 8 run executes
    around the kernels: the factory (test infrastructure, tests/bench_rehearsal.py - a tiny CenterPoint on the CPU) returns a
    model wrapped by common_utils.wrap_ddp and its train step; warm-up, barrier, timed steps, barrier, max over ranks and
    comm_report (no_sync steps, stand-alone all-reduce, exposed / hidden split) then run exactly as in run_gpu / main."""
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ones = torch.ones(1)
    dist.all_reduce(ones)
    dist.barrier()
    cpus = [None] * world
    dist.all_gather_object(cpus, sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else [])
    line = {"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "comm": {"ranks": int(ones.item()), "backend": "gloo"}, "rank_cpus": cpus}
    spec = os.environ.get("TODA_BENCH_DRYRUN_REHEARSAL")
    if spec:
        import importlib
        mod, fn = spec.split(":")
        made = getattr(importlib.import_module(mod), fn)(rank, world)
        model, net, step = made["model"], made["net"], made["step"]
        for it in range(args.warmup):
            loss = step(it)
        dist.barrier()
        t0 = time.perf_counter()
        for it in range(args.warmup, args.warmup + args.steps):
            loss = step(it)
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        assert np.isfinite(float(loss)), "rehearsal diverged"
        line.update(ms_per_step=round(elapsed / args.steps * 1e3, 3), value=round(args.steps * made["per_gpu"] * world / elapsed, 3),
                    reducer=type(model).__name__,
                    comm=comm_report(model, net, step, args, world, torch.device("cpu"), elapsed / args.steps * 1e3,
                                     sync=lambda: None, backend="gloo"))
    if rank == 0:
        print(json.dumps(line))
    dist.destroy_process_group()
    if os.environ.get("TODA_BENCH_DRYRUN_FAIL_RANK") == str(rank):
        raise SystemExit(3)


def comm_report(model, net, step, args, world, device, ms_per_step, sync=None, backend="rccl"):
    """Outside the timed region: how much of the gradient all-reduce the backward hides.  (i) the same steps with
    the all-reduce switched off (DDP.no_sync) -> ms per step without communication; (ii) one flat all-reduce of the
    gradient payload on an otherwise idle GPU.  exposed = ms_per_step - (i); hidden = (ii) - exposed.
    `sync` / `backend`: the CPU rehearsal (dry_run) passes a no-op and "gloo"."""
    sync = sync or torch.cuda.synchronize
    ones = torch.ones(1, device=device)
    dist.all_reduce(ones)
    n_par = sum(p.numel() for p in net.parameters() if p.requires_grad)
    flat = torch.zeros(n_par, device=device)
    for _ in range(3):
        dist.all_reduce(flat)
    sync()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(10):
        dist.all_reduce(flat)
    sync()
    ar_ms = (time.perf_counter() - t0) / 10 * 1e3
    k = min(args.steps, 10)
    it0 = args.warmup + args.steps
    with model.no_sync():
        step(it0)
        sync()
        dist.barrier()
        t0 = time.perf_counter()
        for it in range(it0 + 1, it0 + 1 + k):
            step(it)
        sync()
        dist.barrier()
        nosync_ms = (time.perf_counter() - t0) / k * 1e3
    t = torch.tensor([ar_ms, nosync_ms], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ar_ms, nosync_ms = (float(v) for v in t.tolist())
    exposed = max(ms_per_step - nosync_ms, 0.0)
    return {"backend": backend, "ranks": int(ones.item()), "grad_bytes": 4 * n_par, "reducer": type(model).__name__,
            "allreduce_ms_standalone": round(ar_ms, 3), "ms_per_step_without_allreduce": round(nosync_ms, 3),
            "allreduce_exposed_ms": round(exposed, 3), "allreduce_hidden_ms": round(max(ar_ms - exposed, 0.0), 3)}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--batches", type=int, default=2, help="distinct pre-generated batches cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--input", default="host", choices=("host", "resident"),
                    help="host: collated batches in pinned host memory, uploaded every step inside the timed region; resident: in HBM")
    ap.add_argument("--per-gpu", type=int, default=None, help="override the workload's samples per GPU (exploration)")
    ap.add_argument("--layers", action="store_true",
                    help="after the timed region run 3 more steps with every hand-written kernel bracketed by HIP events and print the "
                         "per-kernel roofline table (gather-GEMM, wgrad, Winograd conv, voxelise, rulebooks, BN rows, .dense()) to stderr")
    ap.add_argument("--layers-out", default=None, help="also write that table as JSON to this path")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # plain `python bench.py --gpus N`: become the launcher
        raise SystemExit(launch_ranks(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    rank_cpus = place_rank(local_rank, world)      # before any GPU call
    if os.environ.get("TODA_BENCH_DRYRUN") == "1":
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # TODA_BENCH_SHARE_GPU=1 (REHEARSAL, not a measurement of scaling): every rank uses cuda:0 and the collectives go over gloo on the
    # device tensors (RCCL refuses two ranks on one device).  What it shows on a 1-GPU box: the N > 1 code path of this file end to end
    # through the HIP kernels - DDP wrapper, bucketed all-reduce behind the backward, barrier, max over ranks, comm block - and the host
    # side of a rank (host_issue_ms_per_step, host_cpu_ms_per_step) while a second process competes for the same launch queue.
    share = os.environ.get("TODA_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1 or os.environ.get("TODA_FORCE_DDP") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    if os.environ.get("TODA_MIOPEN_FIND", "0") == "1":      # experiment knob: MIOpen find mode for the dense convs
        torch.backends.cudnn.benchmark = True
    res = run_gpu(args, rank, world, device)
    if rank == 0:
        per_gpu = res["per_gpu"]
        total_samples = args.steps * per_gpu * world
        roof, rows = roofline_from_timer(res["timer"])
        if args.layers:
            for r in rows:
                print(json.dumps(dict(r, source="dispatch-stamped gather-GEMM launches inside the timed region")), file=sys.stderr)
            for r in res["op_rows"] or []:
                print(json.dumps(r), file=sys.stderr)
            if args.layers_out:
                json.dump({"workload": args.workload, "gather_gemm_timed_region": rows, "kernels": res["op_rows"]}, open(args.layers_out, "w"),
                          indent=1)
        step_ms = res["step_ms"]
        if os.environ.get("TODA_BENCH_STEP_MS"):      # every timed step's GPU time (event to event), to look at outliers
            print("[step_ms] " + " ".join(f"{t:.2f}" for t in step_ms), file=sys.stderr)
            print("[host_enqueue_ms] " + " ".join(f"{t:.2f}" for t in res["host_step_ms"]), file=sys.stderr)
            print("[device_allocs] " + " ".join(str(v) for v in res["device_allocs"]) + "  [alloc_retries] " +
                  " ".join(str(v) for v in res["alloc_retries"]), file=sys.stderr)
            if len(res["step_sizes"]) == len(step_ms):
                print("[step table] step points voxels gpu_ms host_enqueue_ms device_allocs", file=sys.stderr)
                for k, ((npt, nvx), t, h, a) in enumerate(zip(res["step_sizes"], step_ms, res["host_step_ms"], res["device_allocs"])):
                    print(f"[step table] {k} {npt} {nvx} {t:.2f} {h:.2f} {a}", file=sys.stderr)
        line = {
            "metric": "LiDAR training samples/sec" if args.workload != "c2" else "LiDAR backbone forward samples/sec", "value": round(total_samples / res["elapsed"], 3),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(res["elapsed"] / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": res["desc"], "global_batch": per_gpu * world,
                       "points_per_cloud": {"c3": 180000, "c2": 60000}.get(args.workload, "180000/35000 alternating"),
                       "parallelism": f"dp{world}" + (" (REHEARSAL: all ranks share cuda:0, gloo collectives - not a scaling measurement)"
                                                        if os.environ.get("TODA_BENCH_SHARE_GPU") == "1" else ""),
                       "final_loss": round(res["loss"], 4),
                       "input": ("pinned host memory, uploaded every step on the input stream inside the timed region (--input host)"
                                 if res["host_input"] else "raw clouds resident in HBM before the clock starts (--input resident)"),
                       "matrix_path": ("bf16 hi/mid/lo split, 6 terms, fp32 accumulate (sparse gather-GEMMs and weight gradients of the 32/64/128-channel "
                                       "pairs, pixel-GEMMs of the BEV neck; exact operand split, TODA_MM=split)" if ops.matrix_path() == "split"
                                       else "native fp32 MFMA (TODA_MM=native)"),
                       "peak_hbm_gib": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
                       "alloc_retries": int(torch.cuda.memory_stats(device).get("num_alloc_retries", 0)),
                       "reserved_gib": round(torch.cuda.memory_reserved(device) / 2 ** 30, 2),
                       "device_allocs_in_timed_steps": int(sum(res["device_allocs"])),
                       "rank0_cpus": len(rank_cpus),
                       "slowest_step_ms": round(max(step_ms), 3) if step_ms else None,
                       "slowest_step_host_enqueue_ms": round(res["host_step_ms"][int(np.argmax(step_ms))], 3) if step_ms else None},
            # per-step GPU time between events recorded at the step boundaries of rank 0 (no sync inside the region)
            "host_issue_ms_per_step": round(res["issue_s"] / args.steps * 1e3, 3),
            "host_cpu_ms_per_step": round(res["cpu_s"] / args.steps * 1e3, 3),
            "ms_per_step_median": round(float(np.median(step_ms)), 3) if step_ms else None,
            "ms_per_step_p10_p90": [round(float(np.percentile(step_ms, q)), 3) for q in (10, 90)] if step_ms else None,
            "roofline": roof,
        }
        if len(res["step_sizes"]) == len(step_ms) and len(set(res["step_sizes"])) > 2:
            # input size varies from step to step (the mix): how much of the step-time spread is the voxel count
            vx = np.array([v for _, v in res["step_sizes"]], np.float64)
            tm = np.array(step_ms, np.float64)
            slope, icpt = np.polyfit(vx, tm, 1)
            resid = tm - (slope * vx + icpt)
            line["step_time_vs_voxels"] = {"corr": round(float(np.corrcoef(vx, tm)[0, 1]), 3), "ms_per_100k_voxels": round(float(slope * 1e5), 3),
                                           "intercept_ms": round(float(icpt), 3), "residual_p10_p90_ms": [round(float(np.percentile(resid, q)), 3) for q in (10, 90)],
                                           "voxels_min_max": [int(vx.min()), int(vx.max())]}
        if res.get("comm"):
            line["comm"] = res["comm"]
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(res["cfg"], args.workload)
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
