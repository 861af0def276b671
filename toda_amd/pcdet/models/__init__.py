"""build_network / load_data_to_gpu / model_fn_decorator (reference pcdet/models/__init__.py:16-69).

One addition for the MI355X path: when a batch carries raw `points` but no `voxels`, voxelisation
runs on the GPU here (ops.voxelize_batch) instead of in DataLoader workers on the CPU."""
from collections import namedtuple

import numpy as np
import torch

from .detectors import build_detector


def build_network(model_cfg, num_class, dataset):
    return build_detector(model_cfg=model_cfg, num_class=num_class, dataset=dataset)


def load_data_to_gpu(batch_dict):
    """Every ndarray -> fp32 CUDA tensor, integers included (reference :23-34).  A host TENSOR (a collate function that pins its
    output) is uploaded on the current stream without blocking the host."""
    for key, val in batch_dict.items():
        if torch.is_tensor(val) and not val.is_cuda and key not in ("frame_id", "metadata", "calib"):
            batch_dict[key] = val.cuda(non_blocking=True)
            continue
        if not isinstance(val, np.ndarray):
            continue
        if key in ("frame_id", "metadata", "calib"):
            continue
        if key == "image_shape":
            batch_dict[key] = torch.from_numpy(val).int().cuda()
        else:
            batch_dict[key] = torch.from_numpy(val).float().cuda(non_blocking=True)


def voxelize_on_gpu(batch_dict, voxel_cfg):
    """points [sum N, 1 + C] (batch index in column 0) -> voxels / voxel_coords / voxel_num_points.
    voxel_cfg: dict(point_cloud_range, voxel_size, max_points_per_voxel, max_num_voxels)."""
    from ... import ops

    pts = batch_dict["points"]
    bs = int(batch_dict["batch_size"])
    counts = batch_dict.get("points_per_sample")
    if counts is None:
        counts = torch.bincount(pts[:, 0].long(), minlength=bs).tolist()
    clouds, start = [], 0
    for n in counts:
        clouds.append(pts[start:start + int(n), 1:].contiguous())
        start += int(n)
    vox, coords, num = ops.voxelize_batch(clouds, voxel_cfg["point_cloud_range"], voxel_cfg["voxel_size"],
                                          voxel_cfg["max_points_per_voxel"], voxel_cfg["max_num_voxels"])
    batch_dict["voxels"], batch_dict["voxel_coords"], batch_dict["voxel_num_points"] = vox, coords, num
    return batch_dict


def prepare_batch_on_gpu(batch_dict, net, voxel_cfg=None):
    """Everything of a training step that depends only on the INPUT: H2D, voxelisation, the sparse backbone's rulebooks.
    The reference does this part (voxelisation) in DataLoader workers, concurrently with the previous training step
    (pcdet/datasets/processor/data_processor.py:115-143); InputPrefetcher below does it on a side stream."""
    load_data_to_gpu(batch_dict)
    backbone = getattr(net, "backbone_3d", None)
    if "voxels" not in batch_dict and "points" in batch_dict:
        cfg = voxel_cfg if voxel_cfg is not None else net.dataset.voxel_cfg
        # voxels AND rulebooks behind one host sync when the backbone can plan from raw points
        if backbone is not None and hasattr(backbone, "plan_input") and backbone.plan_input(batch_dict, cfg):
            return batch_dict
        voxelize_on_gpu(batch_dict, cfg)
    if backbone is not None and hasattr(backbone, "plan") and "voxel_coords" in batch_dict and batch_dict["voxel_coords"].is_cuda:
        backbone.plan(batch_dict)
    return batch_dict


def _record_stream(obj, stream, seen=None):
    """Tell the caching allocator that every CUDA tensor reachable from obj is used on `stream` too."""
    seen = set() if seen is None else seen
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream, seen)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream, seen)
    elif hasattr(obj, "__dict__") and not isinstance(obj, torch.nn.Module):
        _record_stream(vars(obj), stream, seen)


class InputPrefetcher:
    """Device-side input pipeline: batch t + 1 is voxelised and its rulebooks are built on a side HIP stream while the main
    stream still runs step t, so the two host syncs of the index building (voxel counts, output-set sizes) and the ~1 ms of
    small index kernels leave the critical path.  next() hands out a prepared batch (the main stream waits for its event);
    kick() starts the preparation of the following one - call it once the backward pass and the optimizer step of the current
    iteration are ENQUEUED: the host then waits in the side stream's syncs while the GPU has the whole backward to run (kicked
    right behind the forward, the host waits with nothing queued behind it and the GPU runs dry: measured no gain)."""

    def __init__(self, batches, net, device, eager=True, voxel_cfg=None):
        self.it = iter(batches)
        self.net = net
        self.voxel_cfg = voxel_cfg          # the voxeliser settings of the dataset being READ (evaluation of a train-built model: the eval set's cap / range)
        # HIGH priority: the side stream's kernels are small (index building) and the host WAITS for them twice per batch; at
        # normal priority they queue behind the backward pass's chip-filling kernels, the host comes back from kick() only when
        # that pass is done and the GPU then idles while the next forward is being enqueued (r02 trace: 0.6-0.9 ms gaps at the
        # step boundary).  TODA_PREFETCH_PRIORITY=0 restores the default priority.
        import os
        prio = int(os.environ.get("TODA_PREFETCH_PRIORITY", "-1"))
        self.side = torch.cuda.Stream(device=device, priority=prio)
        # device tensors that exist already (resident clouds, an object database, the model's weights) were produced on the
        # caller's stream: the side stream waits for that stream ONCE, here.  Not in kick(): kick() runs right behind an enqueued
        # backward pass and must not wait for it - what a later batch reads is either uploaded on the side stream itself or
        # older than this point.
        self.side.wait_stream(torch.cuda.current_stream(device))
        self.pending = None
        # Device memory of the prepared batches: arena slots (toda_amd.arena) instead of the caching allocator - what the side stream
        # produces is consumed on the caller's stream a step later, the worst case for a stream-aware allocator (round 3: hipMalloc
        # calls inside steady-state steps).  A slot is re-used when its last consumer has finished (host-side event query by the worker;
        # neither stream waits for the other on the GPU).  TODA_PREFETCH_ARENA=0 goes back to the allocator.
        from ... import arena as _arena
        self._arena_mod = _arena
        self.arena = None
        if os.environ.get("TODA_PREFETCH_ARENA", "1") == "1":
            n_slots = int(os.environ.get("TODA_PREFETCH_SLOTS", "2"))
            self.arena = _arena.IndexArena(device, n_slots, int(os.environ.get("TODA_PREFETCH_MAX_SLOTS", n_slots)),
                                           host_wait=os.environ.get("TODA_PREFETCH_STREAM_WAIT", "0") != "1")
        self._in_use = None          # slot of the batch the caller is consuming
        self._poison = os.environ.get("TODA_PREFETCH_DEBUG_POISON", "0") == "1"
        # (Round 4 measured WHERE in the step the index kernels run - start of the forward, behind the sparse forward, at the start of the
        # dense or sparse backward - and whether the side stream waits for the training stream on the GPU or the worker waits on the host:
        # 16.87-16.99 ms per step everywhere.  A batch's ~0.5 ms of index kernels costs ~0.4 ms wherever it lands: the training stream
        # leaves no idle capacity to hide it in - without any input pipeline the step is 16.4 ms.  DESIGN.md section 5.)
        self._host_wait = os.environ.get("TODA_PREFETCH_HOST_WAIT", "1") == "1" and os.environ.get("TODA_PREFETCH_THREAD", "1") == "1"
        # Phase of the index kernels inside the consumer's step: with two slots a preparation starts when a step ends, i.e. it lands
        # under the next step's sparse forward - on the C3 step right on the dominant 64 -> 64 gather-GEMMs (0.605 instead of 0.58 ms per
        # launch).  The worker therefore also waits (on the host) until the training stream has passed the sparse backbone's forward of
        # the step in flight: an event recorded by a forward hook on backbone_3d.  The index kernels then run beside the dense neck.
        self._fwd_done = None
        self._phase_hook = None
        bb = getattr(net, "backbone_3d", None)
        if os.environ.get("TODA_PREFETCH_PHASE", "1") == "1" and self.arena is not None and self._host_wait and bb is not None:
            def _mark(_m, _a, _out):
                if torch.is_grad_enabled():      # training steps only (the forward-only loops have no slack to wait in)
                    ev = torch.cuda.Event(blocking=self._arena_mod.BLOCKING_EVENTS)
                    ev.record()
                    self._fwd_done = ev
            self._phase_hook = bb.register_forward_hook(_mark)
        self._trace = [] if os.environ.get("TODA_PREFETCH_TRACE") else None      # (seconds per preparation, of which waiting for the counts)
        # The preparation runs on a worker thread (TODA_PREFETCH_THREAD=0: on the caller's), so its two host syncs and its ~150
        # launches overlap the caller's own enqueueing instead of following it: the forward-only workload is host-bound otherwise
        # (69 launches of the backbone + 155 of the next batch's index plan per 3.2 ms of GPU work: 907 -> 1144 samples/s); the
        # training step, GPU-bound, measures the same either way.  Streams, current device and grad mode are per thread.
        self.device = device
        self.pool = None
        if os.environ.get("TODA_PREFETCH_THREAD", "1") == "1":
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="toda-prefetch")
        if eager:
            self.kick()

    def _prepare(self):
        import time as _t
        t0 = _t.perf_counter()
        try:
            return self._prepare_inner()
        finally:
            if self._trace is not None:
                from ... import ops as _ops
                self._trace.append((_t.perf_counter() - t0, getattr(_ops, "_LAST_WAIT_S", 0.0)))

    def _prepare_inner(self):
        if self.pool is not None:
            torch.cuda.set_device(self.device)       # the worker thread's own current device
        with torch.no_grad(), torch.cuda.stream(self.side):
            try:
                batch = next(self.it)        # inside the side-stream context: a source that mixes / collates on the device runs there too
            except StopIteration:
                return None
            slot = self.arena.acquire(self.side) if self.arena is not None else None      # (waits on the host for a free slot)
            gate = self._fwd_done
            if gate is not None and slot is not None:
                self._arena_mod._host_wait(gate)
            try:
                with self._arena_mod.use_slot(slot):
                    if isinstance(batch, (tuple, list)):      # the (adversarial, original) pair of the stage-2 consistency step
                        batch = tuple(prepare_batch_on_gpu(b, self.net, self.voxel_cfg) for b in batch)
                    else:
                        batch = prepare_batch_on_gpu(batch, self.net, self.voxel_cfg)
            except BaseException:
                # the slot must not stay "in use" for good (with two slots the NEXT acquire would find none and hide this error);
                # the caller sees the original exception from next()
                if slot is not None:
                    self.arena.abandon(slot)
                raise
            if self.arena is not None:
                self.arena.prewarm()         # once: the other slots get the first one's layout (no device allocation after the warm-up steps)
            ev = torch.cuda.Event(blocking=self._arena_mod.BLOCKING_EVENTS)
            ev.record(self.side)
            if self._host_wait:
                # hand the batch over only when its preparation has FINISHED on the GPU (this is the worker thread; it sleeps in the runtime
                # with the interpreter lock released): next() then needs no stream-side wait, and the training stream's queue carries no
                # cross-queue dependency at all
                self._arena_mod._host_wait(ev)
        return batch, ev, slot

    def kick(self):
        if self.pending is not None:
            return
        self.pending = self.pool.submit(self._prepare) if self.pool is not None else self._prepare()

    def next(self):
        """The next prepared batch.  LIFETIME: with the arena (the default) every device tensor the pipeline built for the PREVIOUS batch -
        voxels, voxel_coords, voxel_num_points, every neighbour table and the out_indices inside sparse tensors derived from it - lives in
        a slot that this call hands back: the worker overwrites it as soon as the work enqueued so far on the caller's stream has
        finished.  A caller that holds a batch across next() (gradient accumulation over two batches, an evaluation loop that collects
        multi_scale_3d_features, a debugging dump) must take InputPrefetcher.keep(batch) first.  TODA_PREFETCH_DEBUG_POISON=1 fills a
        released slot with 0xFF so that a stale reader fails loudly instead of reading the next batch's tables."""
        if self.pending is None:
            self.kick()
        pend, self.pending = self.pending, None      # cleared first: a preparation that raised must not be handed out again by the next call
        got = pend.result() if self.pool is not None else pend
        if got is None:
            raise StopIteration
        batch, ev, slot = got
        main = torch.cuda.current_stream()
        if self._in_use is not None:
            # everything that reads the PREVIOUS batch (forward, backward, optimizer) has been enqueued on this stream by now
            if self._poison:
                for ch in self._in_use.chunks:
                    ch.fill_(0xFF)           # on the caller's stream, behind the batch's last reader and before the release event
            self.arena.release(self._in_use, main)
        self._in_use = slot
        if not ev.query():
            main.wait_event(ev)
        _record_stream(batch, main)
        return batch

    @staticmethod
    def keep(obj, _seen=None):
        """A copy of a batch (dict / list / tuple / object tree) whose device tensors are owned by the caching allocator instead of an arena
        slot: safe to hold across next().  Tensors are cloned on the current stream; everything else is shared."""
        seen = {} if _seen is None else _seen
        if id(obj) in seen:
            return seen[id(obj)]
        if torch.is_tensor(obj):
            out = obj.clone() if obj.is_cuda else obj
        elif isinstance(obj, dict):
            out = {}
            seen[id(obj)] = out
            for k, v in obj.items():
                out[k] = InputPrefetcher.keep(v, seen)
            return out
        elif isinstance(obj, (list, tuple)):
            out = type(obj)(InputPrefetcher.keep(v, seen) for v in obj)
        elif hasattr(obj, "__dict__") and not isinstance(obj, torch.nn.Module):
            import copy as _copy
            out = _copy.copy(obj)
            seen[id(obj)] = out
            for k, v in vars(obj).items():
                setattr(out, k, InputPrefetcher.keep(v, seen))
            return out
        else:
            out = obj
        seen[id(obj)] = out
        return out

    def close(self):
        """Stop the worker: a pending preparation is waited for and dropped (its side-stream work and pinned buffers would otherwise
        stay alive until interpreter exit - one prefetcher per epoch and per evaluation).  Idempotent."""
        pend, self.pending = self.pending, None
        if self._phase_hook is not None:
            self._phase_hook.remove()
            self._phase_hook = None
        if self.pool is not None:
            if pend is not None:
                try:
                    pend.result()
                except Exception:       # the loop that owned this prefetcher is already unwinding
                    pass
            self.pool.shutdown(wait=True, cancel_futures=True)
            self.pool = None
        self.it = iter(())

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    @property
    def threaded(self):
        """True when kick() only hands the preparation to the worker thread: a caller may then kick right behind next() (the worker
        prepares the following batch while the caller enqueues this one) instead of behind its own last launch."""
        return self.pool is not None


ModelReturn = namedtuple("ModelReturn", ["loss", "tb_dict", "disp_dict"])


def model_fn_decorator():
    def model_func(model, batch_dict):
        load_data_to_gpu(batch_dict)
        if "voxels" not in batch_dict and "points" in batch_dict:
            net = model.module if hasattr(model, "module") and not hasattr(model, "dataset") else model
            voxelize_on_gpu(batch_dict, net.dataset.voxel_cfg)
        ret_dict, tb_dict, disp_dict = model(batch_dict)
        loss = ret_dict["loss"].mean()
        (model.module if hasattr(model, "module") and not hasattr(model, "update_global_step") else model).update_global_step()
        return ModelReturn(loss, tb_dict, disp_dict)

    return model_func


# --------------------------------------------------------------------------------------------
# TODA stage 2: "2 forward + 1 backward" consistency step (reference :88-260).
# ---------------------------------------------------------------------------------------------
def common_utils_rot(a):
    """(r00, r01, r10, r11, angle) of the row-vector rotation [[c, s], [-s, c]] in fp32, as Python floats."""
    ang = torch.tensor([a], dtype=torch.float32)
    c, s_ = float(torch.cos(ang)), float(torch.sin(ang))
    return c, s_, -s_, c, float(ang)


def random_world_flip(box_preds, params, reverse=False):
    order = ("y", "x") if reverse else ("x", "y")
    for axis in order:
        if axis not in params:
            continue
        if axis == "x":
            box_preds[:, 1] = -box_preds[:, 1]
            box_preds[:, 6] = -box_preds[:, 6]
        else:
            box_preds[:, 0] = -box_preds[:, 0]
            box_preds[:, 6] = -(box_preds[:, 6] + np.pi)
    return box_preds


def random_world_rotation(box_preds, params, reverse=False):
    if box_preds.is_cuda:
        # the same rotation with the matrix built on the host and handed over as kernel scalars: torch.tensor(..., device=cuda) is a
        # synchronous copy and float(angle) a read-back - each drains the stream (27 of the 54 ms of a stage-2 step sat in them)
        a = float(-params if reverse else params)
        rot = common_utils_rot(a)
        x, y = box_preds[:, 0].clone(), box_preds[:, 1].clone()
        box_preds[:, 0] = x * rot[0] + y * rot[2]
        box_preds[:, 1] = x * rot[1] + y * rot[3]
        box_preds[:, 6] += rot[4]
        return box_preds
    angle = torch.tensor([-params if reverse else params], dtype=torch.float32, device=box_preds.device)
    c, s_, z, o = torch.cos(angle), torch.sin(angle), angle.new_zeros(1), angle.new_ones(1)
    rot = torch.stack((c, s_, z, -s_, c, z, z, z, o), dim=1).reshape(3, 3)
    box_preds[:, :3] = box_preds[:, :3] @ rot
    box_preds[:, 6] += float(angle)
    return box_preds


def random_world_scaling(box_preds, params, reverse=False):
    box_preds[:, :6] *= (1.0 / params) if reverse else params
    return box_preds


_AUGS = {"random_world_flip": random_world_flip, "random_world_rotation": random_world_rotation,
         "random_world_scaling": random_world_scaling}


@torch.no_grad()
def reverse_transform(boxes, batch_dict):
    """Undo the recorded world augmentations (last applied first) on the decoded boxes of each sample."""
    lists, params = batch_dict.get("augmentation_list"), batch_dict.get("augmentation_params")
    if lists is None:
        return boxes
    for b, box in enumerate(boxes):
        preds = box["pred_boxes"]
        for key in list(lists[b])[::-1]:
            if key == "gt_sampling":
                continue
            preds = _AUGS[key](preds, params[b][key], reverse=True)
        box["pred_boxes"] = preds
    return boxes


def filter_boxes_centerpoint(batch_dict, model):
    """Decoded (top-K, score-thresholded, un-NMSed) boxes of a CenterHead forward (reference :310-360)."""
    net = model.module if hasattr(model, "module") and not hasattr(model, "dense_head") else model
    net = getattr(net, "onepass", net)
    head = net.dense_head
    from .model_utils import centernet_utils

    post = head.model_cfg.POST_PROCESSING
    ref = batch_dict["pred_dicts"][0]["hm"]
    limit = head._device_const("limit", ref.device, lambda: torch.tensor(post.POST_CENTER_LIMIT_RANGE, dtype=torch.float32))     # uploaded once
    padded = ref.is_cuda       # on the GPU the K candidates stay in place behind a mask (no row count read back per sample)
    out = [{"pred_boxes": [], "pred_scores": [], "mask": []} for _ in range(batch_dict["batch_size"])]
    for pred in batch_dict["pred_dicts"]:
        # As the reference (:328): .sigmoid() of whatever pred_dict holds.  In the training step get_loss has already replaced
        # pred["hm"] by its clamped sigmoid in place (center_head.py:233), so the scores here are sigmoid(sigmoid(logit)) in
        # 0.5 .. 0.73 and every top-K box passes SCORE_THRESH - reproduced as is (no host-side probing of the values).
        hm = pred["hm"].sigmoid()
        decoded = centernet_utils.decode_bbox_from_heatmap(
            heatmap=hm, rot_cos=pred["rot"][:, 0:1], rot_sin=pred["rot"][:, 1:2], center=pred["center"],
            center_z=pred["center_z"], dim=pred["dim"].exp(), vel=None, point_cloud_range=head.point_cloud_range,
            voxel_size=head.voxel_size, feature_map_stride=head.feature_map_stride, K=post.MAX_OBJ_PER_SAMPLE,
            circle_nms=False, score_thresh=post.SCORE_THRESH, post_center_limit_range=limit, padded=padded)
        for k, d in enumerate(decoded):
            out[k]["pred_boxes"].append(d["pred_boxes"])
            out[k]["pred_scores"].append(d["pred_scores"])
            if padded:
                out[k]["mask"].append(d["mask"])
    for d in out:
        d["pred_boxes"] = torch.cat(d["pred_boxes"], dim=0)
        d["pred_scores"] = torch.cat(d["pred_scores"], dim=0)
        if padded:
            d["mask"] = torch.cat(d["mask"], dim=0)
        else:
            del d["mask"]
    return out


def get_consistency_loss(adv_boxes, org_boxes):
    """Nearest-centre matching (< 1 m^2) between the two passes; L1 on centres, MSE on sizes.  The
    reference detaches both box sets (:229-230), so this term carries no gradient; kept as is."""
    import torch.nn.functional as F

    centre_terms, size_terms, norm = [], [], 0
    for adv, org in zip(adv_boxes, org_boxes):
        a, o = adv["pred_boxes"].detach(), org["pred_boxes"].detach()
        norm += 1
        if "mask" in adv and "mask" in org:
            # the same terms over padded candidate lists: a pair counts when both boxes are selected (invalid rows / columns sit at
            # infinite distance), sums run over the selected rows only, n = number of selected boxes of both passes
            va, vo = adv["mask"], org["mask"]
            d2 = ((a[:, None, :3] - o[None, :, :3]) ** 2).sum(-1)
            d2 = d2.masked_fill(~(va[:, None] & vo[None, :]), float("inf"))
            d_a, idx_o_of_a = d2.min(1)
            d_o, idx_a_of_o = d2.min(0)
            m_o = ((d_a < 1) & va).unsqueeze(-1)
            m_a = ((d_o < 1) & vo).unsqueeze(-1)
            n = (va.sum() + vo.sum()).clamp(min=1).float()
            # unselected candidate rows are SELECTED away, not multiplied by zero: such a row may hold inf / nan (dim.exp() of a wild
            # regression output) and 0 * inf is nan; the unpadded path drops these rows by boolean indexing
            zero = a.new_zeros(())
            centre_terms.append((torch.where(m_o, (a[:, :3] - o[idx_o_of_a, :3]).abs(), zero).sum()
                                 + torch.where(m_a, (o[:, :3] - a[idx_a_of_o, :3]).abs(), zero).sum()) / n)
            size_terms.append((torch.where(m_o, F.mse_loss(o[idx_o_of_a, 3:6], a[:, 3:6], reduction="none"), zero).sum()
                               + torch.where(m_a, F.mse_loss(a[idx_a_of_o, 3:6], o[:, 3:6], reduction="none"), zero).sum()) / n)
            continue
        if a.shape[0] == 0 or o.shape[0] == 0:
            continue
        d2 = ((a[:, None, :3] - o[None, :, :3]) ** 2).sum(-1)
        d_a, idx_o_of_a = d2.min(1)
        d_o, idx_a_of_o = d2.min(0)
        m_o = (d_a < 1).float().unsqueeze(-1)   # adv boxes that found an org partner
        m_a = (d_o < 1).float().unsqueeze(-1)
        n = a.shape[0] + o.shape[0]
        centre_terms.append((((a[:, :3] - o[idx_o_of_a, :3]) * m_o).abs().sum()
                             + ((o[:, :3] - a[idx_a_of_o, :3]) * m_a).abs().sum()) / n)
        size_terms.append(((F.mse_loss(o[idx_o_of_a, 3:6], a[:, 3:6], reduction="none") * m_o).sum()
                           + (F.mse_loss(a[idx_a_of_o, 3:6], o[:, 3:6], reduction="none") * m_a).sum()) / n)
    zero = 0.0
    return (sum(centre_terms) if centre_terms else zero) / max(norm, 1), (sum(size_terms) if size_terms else zero) / max(norm, 1)


class DistModel(torch.nn.Module):
    """Both passes inside ONE module forward so that DDP sees a single backward and issues a single
    gradient all-reduce (reference tools/stage2_mixup_train_cl.py:61-74)."""

    def __init__(self, model):
        super().__init__()
        self.onepass = model

    def forward(self, batch_adv, batch_org):
        return self.onepass(batch_adv), self.onepass(batch_org)

    def update_global_step(self):
        self.onepass.update_global_step()


def model_fn_decorator_cl():
    """loss = loss_adv + loss_org + 0.1 * (centre consistency + size consistency)."""

    def model_func(model, adv_batch_dict, org_batch_dict, dist=False):
        net = model.module if hasattr(model, "module") and not hasattr(model, "dataset") else model
        net = getattr(net, "onepass", net)
        for b in (adv_batch_dict, org_batch_dict):
            load_data_to_gpu(b)
            if "voxels" not in b and "points" in b:
                voxelize_on_gpu(b, net.dataset.voxel_cfg)
        net.return_batch_dict = True
        try:
            if isinstance(model, DistModel) or isinstance(getattr(model, "module", None), DistModel):
                (adv_b, adv_ret, adv_tb, adv_disp), (org_b, org_ret, org_tb, org_disp) = model(adv_batch_dict, org_batch_dict)
            else:
                adv_b, adv_ret, adv_tb, adv_disp = model(adv_batch_dict)
                org_b, org_ret, org_tb, org_disp = model(org_batch_dict)
        finally:
            net.return_batch_dict = False
        loss_adv, loss_org = adv_ret["loss"].mean(), org_ret["loss"].mean()
        adv_boxes = filter_boxes_centerpoint(adv_b, net)
        org_boxes = reverse_transform(filter_boxes_centerpoint(org_b, net), org_b)
        centre_loss, size_loss = get_consistency_loss(adv_boxes, org_boxes)
        loss = loss_adv + loss_org + 0.1 * (centre_loss + size_loss)
        net.update_global_step()
        adv_tb = dict(adv_tb, loss_org=loss_org.detach(), cl_center=centre_loss, cl_size=size_loss)
        return ModelReturn(loss, adv_tb, adv_disp)

    return model_func
