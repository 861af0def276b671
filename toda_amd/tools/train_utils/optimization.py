"""Optimizer + LR schedule of the reference trainers (tools/train_utils/optimization/__init__.py:11-63,
fastai_optim.py:104-236, learning_schedules_fastai.py:12-79), restated for the GPU:

  adam_onecycle = Adam(betas=(mom, 0.99), eps 1e-8) with DECOUPLED weight decay applied before the
  step (p *= 1 - wd*lr, BN parameters included), two parameter groups (non-BN leaves / BN leaves),
  lr and beta1 driven per iteration by a one-cycle cosine schedule.

The reference walks every parameter in Python (one tiny kernel each); here the decay is one
torch._foreach_mul_ per group and Adam runs in its multi-tensor form."""
import math

import torch
import torch.nn as nn

BN_TYPES = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm)


def split_bn_params(model):
    """(non-BN leaf params, BN leaf params) in module traversal order, trainable only."""
    plain, bn = [], []

    def visit(m):
        kids = list(m.children())
        if kids:
            for k in kids:
                visit(k)
            return
        dst = bn if isinstance(m, BN_TYPES) else plain
        dst.extend(p for p in m.parameters(recurse=False) if p.requires_grad)

    visit(model)
    # parameters registered directly on non-leaf modules (none in the reference's models)
    seen = {id(p) for p in plain + bn}
    plain.extend(p for p in model.parameters() if p.requires_grad and id(p) not in seen)
    return plain, bn


class OneCycleAdam:
    """The reference's OptimWrapper(Adam, true_wd=True, bn_wd=True) with its `lr` / `mom` properties."""

    def __init__(self, model, wd, betas=(0.9, 0.99), lr=3e-3, fused=None):
        plain, bn = split_bn_params(model)
        groups = [{"params": plain, "lr": 0.0}, {"params": bn, "lr": 0.0}]
        kw = {}
        if fused is None:
            fused = all(p.is_cuda for p in plain + bn) and len(plain + bn) > 0
        if fused:
            kw["fused"] = True
        else:
            kw["foreach"] = True
        self.opt = torch.optim.Adam(groups, betas=betas, weight_decay=0.0, **kw)
        self.wd = wd
        self.true_wd, self.bn_wd = True, True
        self._beta2 = betas[1]
        self.lr, self.mom = lr, betas[0]
        import os

        self._hip_step = bool(fused) and os.environ.get("TODA_HIP_OPTIMIZER", "1") == "1"
        self._hip_cache = {}

    # hyper-parameters as properties, like the reference wrapper
    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, val):
        self._lr = float(val)
        for g in self.opt.param_groups:
            g["lr"] = self._lr

    @property
    def mom(self):
        return self._mom

    @mom.setter
    def mom(self, val):
        self._mom = float(val)
        for g in self.opt.param_groups:
            g["betas"] = (self._mom, self._beta2)

    @property
    def param_groups(self):
        return self.opt.param_groups

    def zero_grad(self, set_to_none=True):
        self.opt.zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self):
        self._sync_steps()
        self._hip_cache = {}      # the torch path advances the counters itself
        factor = 1.0 - self.wd * self._lr
        for g in self.opt.param_groups:
            ps = [p for p in g["params"] if p.requires_grad]
            if ps:
                torch._foreach_mul_(ps, factor)
        self.opt.step()

    # ------------------------------------------------------------------ clip + decay + Adam in two launches (toda_clip_adam_step)
    def _fused_ready(self, params):
        return (self._hip_step and len(params) > 0 and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                                                           and p.grad.dtype == torch.float32 and p.grad.device == p.device for p in params)
                and len({p.device for p in params}) == 1)

    @torch.no_grad()
    def clip_and_step(self, max_norm):
        """clip_grad_norm_(parameters, max_norm) followed by step(), as the reference's loop does them
        (tools/train_utils/train_utils.py:57-58); returns the total gradient norm (a 0-d tensor).  On one GPU with fp32 contiguous
        tensors both run inside toda_clip_adam_step (the state stays torch.optim.Adam's: exp_avg, exp_avg_sq, step - checkpoints
        are interchangeable with the torch path, which is what runs otherwise)."""
        every = [p for g in self.opt.param_groups for p in g["params"] if p.requires_grad]
        params = [p for p in every if p.grad is not None]
        if len(params) != len(every):
            # a parameter without a gradient this step: the reference's wrapper (OptimWrapper.step) still DECAYS it and Adam leaves its
            # moments and step counter alone - per-parameter behaviour the two-launch kernel (one step count, one table) does not
            # model.  The torch path does exactly that.
            self._sync_steps()
            self._hip_cache = {}
            total = clip_grad_norm_(params, max_norm) if max_norm is not None and max_norm > 0 and params else None
            self.step()
            return total
        key = tuple(map(id, params))
        c = self._hip_cache
        # (the per-tensor checks and the table of parameter / moment addresses are made once per parameter set: at ~150 tensors they
        # cost 2 ms of host time per step when repeated; per step only the gradient addresses are read)
        if c.get("key") != key and not self._fused_ready(params):
            self._sync_steps()
            total = clip_grad_norm_(params, max_norm) if max_norm is not None and max_norm > 0 else None
            self.step()
            return total
        from toda_amd import lib as L

        lib = L.load()
        dev = params[0].device
        st = self.opt.state
        if c.get("key") != key:
            self._sync_steps()
            for p in params:
                if len(st[p]) == 0:       # torch.optim.Adam._init_group for fused=True: device step counter, zero moments
                    st[p]["step"] = torch.zeros((), dtype=torch.float32, device=dev)
                    st[p]["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st[p]["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        if c.get("key") != key:
            chunk = lib.toda_clip_adam_chunk()
            numel = [p.numel() for p in params]
            ct, co = [], []
            for i, n in enumerate(numel):
                for off in range(0, max(n, 1), chunk):
                    ct.append(i)
                    co.append(off)
            c.clear()
            c.update(key=key, n=len(params), n_chunks=len(ct),
                     numel=torch.tensor(numel, dtype=torch.int64, device=dev),
                     chunk_tensor=torch.tensor(ct, dtype=torch.int32, device=dev),
                     chunk_off=torch.tensor(co, dtype=torch.int64, device=dev),
                     partial=torch.empty((len(ct),), dtype=torch.float64, device=dev),
                     # pinned staging copies of the pointer table: the upload is stream-ordered and may run several steps after
                     # the host wrote it, so a buffer is rewritten only after its own upload has executed (event)
                     host=[torch.empty((4, len(params)), dtype=torch.int64).pin_memory() for _ in range(4)],
                     host_done=[None] * 4, turn=0,
                     table=torch.empty((4, len(params)), dtype=torch.int64, device=dev),
                     moments=[(st[p]["exp_avg"], st[p]["exp_avg_sq"]) for p in params],
                     steps=[st[p]["step"] for p in params],
                     count=None)
            if any(not (m.is_contiguous() and v.is_contiguous() and m.dtype == torch.float32) for m, v in c["moments"]):
                c.clear()
                self._hip_step = False
                return self.clip_and_step(max_norm)
            fixed = [[p.data_ptr() for p in params], [m.data_ptr() for m, _ in c["moments"]], [v.data_ptr() for _, v in c["moments"]]]
            for h in c["host"]:       # parameters and moments do not move: rows 0, 2, 3 of every staging copy are written once
                t = h.numpy()
                t[0, :], t[2, :], t[3, :] = fixed
            # first step / resumed from a checkpoint: ONE read of the device counters.  The kernel applies one bias correction to every
            # tensor, so the counters must agree (they differ when a parameter sat out earlier steps on the torch path): otherwise
            # this parameter set stays on the torch path for good.
            counts = torch.stack([t.reshape(()).float() for t in c["steps"]])
            lo, hi = (float(v) for v in torch.stack([counts.min(), counts.max()]).tolist())
            if lo != hi:
                c.clear()
                self._hip_step = False
                return self.clip_and_step(max_norm)
            c["count"] = int(lo)
        # a staging copy whose upload has executed (the host may be several steps ahead of the GPU: never wait here, grow the pool)
        turn = next((i for i, ev in enumerate(c["host_done"]) if ev is None or ev.query()), None)
        if turn is None:
            h = torch.empty_like(c["host"][0]).pin_memory()
            h.copy_(c["host"][0])
            c["host"].append(h)
            c["host_done"].append(None)
            turn = len(c["host"]) - 1
        host = c["host"][turn]
        # gradient storage is reallocated by zero_grad(set_to_none=True): row 1 is rewritten every step (pinned buffer -> device, stream-ordered)
        host.numpy()[1, :] = [p.grad.data_ptr() for p in params]
        c["table"].copy_(host, non_blocking=True)
        ev = c["host_done"][turn] or torch.cuda.Event()
        ev.record()
        c["host_done"][turn] = ev
        c["count"] += 1       # the state's device step counters follow lazily (_sync_steps: state_dict, a change of path)
        c["steps_dirty"] = True
        norm = torch.empty((1,), dtype=torch.float32, device=dev)
        t = c["table"]
        rc = lib.toda_clip_adam_step(L.ptr(t[0]), L.ptr(t[1]), L.ptr(t[2]), L.ptr(t[3]), L.ptr(c["numel"]), L.ptr(c["chunk_tensor"]),
                                     L.ptr(c["chunk_off"]), c["n_chunks"], L.ptr(c["partial"]), L.ptr(norm),
                                     float(max_norm) if max_norm is not None else 0.0, self._lr, self._mom, self._beta2,
                                     float(self.opt.param_groups[0]["eps"]), float(self.wd), c["count"], L.stream())
        L.check(rc, "toda_clip_adam_step")
        return norm[0]

    def _sync_steps(self):
        """Write the host-side step count into the state's device counters (torch.optim.Adam's `step` tensors), which the two-launch
        path does not touch per step."""
        c = self._hip_cache
        if c.get("steps_dirty"):
            torch._foreach_mul_(c["steps"], 0.0)
            torch._foreach_add_(c["steps"], float(c["count"]))
            c["steps_dirty"] = False

    def state_dict(self):
        self._sync_steps()
        return self.opt.state_dict()

    def load_state_dict(self, sd):
        self.opt.load_state_dict(sd)
        self._hip_cache = {}      # new moment tensors, new step count


def annealing_cos(start, end, pct):
    return end + (start - end) / 2 * (math.cos(math.pi * pct) + 1)


class OneCycle:
    """lr: lr_max/div -> lr_max over the first pct_start of the steps, then -> lr_max/div/1e4;
    momentum mirrors it moms[0] -> moms[1] -> moms[0].  step(it) sets both on the optimizer."""

    def __init__(self, optimizer, total_step, lr_max, moms, div_factor, pct_start):
        self.optimizer = optimizer
        self.total_step = int(total_step)
        self.lr_max, self.moms, self.div_factor, self.pct_start = lr_max, list(moms), div_factor, pct_start
        low = lr_max / div_factor
        split = int(pct_start * self.total_step)
        # (first step, last step, lr from, lr to, mom from, mom to)
        self.phases = [(0, split, low, lr_max, self.moms[0], self.moms[1]),
                       (split, self.total_step, lr_max, low / 1e4, self.moms[1], self.moms[0])]
        optimizer.lr, optimizer.mom = low, self.moms[0]

    def values(self, step):
        lr, mom = self.optimizer.lr, self.optimizer.mom
        for start, end, lr0, lr1, m0, m1 in self.phases:
            if step >= start and end > start:  # an empty warm-up phase (tiny runs) is skipped instead of dividing by zero
                pct = (step - start) / (end - start)
                lr, mom = annealing_cos(lr0, lr1, pct), annealing_cos(m0, m1, pct)
        return lr, mom

    def step(self, step):
        self.optimizer.lr, self.optimizer.mom = self.values(step)


def build_optimizer(model, optim_cfg):
    if optim_cfg.OPTIMIZER == "adam":
        return torch.optim.Adam(model.parameters(), lr=optim_cfg.LR, weight_decay=optim_cfg.WEIGHT_DECAY)
    if optim_cfg.OPTIMIZER == "sgd":
        return torch.optim.SGD(model.parameters(), lr=optim_cfg.LR, weight_decay=optim_cfg.WEIGHT_DECAY,
                               momentum=optim_cfg.MOMENTUM)
    if optim_cfg.OPTIMIZER == "adam_onecycle":
        return OneCycleAdam(model, wd=optim_cfg.WEIGHT_DECAY, betas=(0.9, 0.99), lr=3e-3)
    raise NotImplementedError(optim_cfg.OPTIMIZER)


def build_scheduler(optimizer, total_iters_each_epoch, total_epochs, last_epoch, optim_cfg):
    total_steps = total_iters_each_epoch * total_epochs
    if optim_cfg.OPTIMIZER == "adam_onecycle":
        sched = OneCycle(optimizer, total_steps, optim_cfg.LR, list(optim_cfg.MOMS), optim_cfg.DIV_FACTOR,
                         optim_cfg.PCT_START)
        return sched, None
    decay_steps = [x * total_iters_each_epoch for x in optim_cfg.DECAY_STEP_LIST]

    def factor(cur):
        f = 1.0
        for s in decay_steps:
            if cur >= s:
                f *= optim_cfg.LR_DECAY
        return max(f, optim_cfg.LR_CLIP / optim_cfg.LR)

    return torch.optim.lr_scheduler.LambdaLR(optimizer, factor, last_epoch=last_epoch), None


def clip_grad_norm_(parameters, max_norm, norm_type=2.0):
    """torch.nn.utils.clip_grad_norm_ (reference tools/train_utils/train_utils.py:57) for gradients that live on ONE device, in
    four launches and no per-parameter Python work: torch's version calls `.to(device)` on each of the ~110 per-tensor norms
    (2 ms of host time per step here).  Same operations in the same order: foreach norms -> norm of the stacked norms ->
    coefficient max_norm / (total + 1e-6) clamped to 1 -> foreach multiply.  Returns the total norm (a 0-d tensor)."""
    parameters = [parameters] if torch.is_tensor(parameters) else list(parameters)
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.tensor(0.0)
    if any(g.device != grads[0].device or g.dtype != grads[0].dtype for g in grads):
        return torch.nn.utils.clip_grad_norm_([p for p in parameters if p.grad is not None], max_norm, norm_type)
    norms = torch._foreach_norm(grads, norm_type)
    total = torch.linalg.vector_norm(torch.stack(norms), norm_type)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    torch._foreach_mul_(grads, coef)
    return total


def clip_and_step(optimizer, parameters, max_norm):
    """The two lines of the reference's loop (tools/train_utils/train_utils.py:57-58) - clip_grad_norm_(parameters, max_norm);
    optimizer.step() - in the optimizer's own fused form where it has one (OneCycleAdam.clip_and_step)."""
    if hasattr(optimizer, "clip_and_step"):
        return optimizer.clip_and_step(max_norm)
    total = clip_grad_norm_(parameters, max_norm)
    optimizer.step()
    return total
