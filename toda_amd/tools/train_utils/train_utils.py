"""Epoch loop and checkpoints of the reference trainers (tools/train_utils/train_utils.py:11-176):
same call signatures (train_one_epoch / train_model / checkpoint_state / save_checkpoint) and the
same step order: lr_scheduler.step(it) -> zero_grad -> model_func -> backward -> clip_grad_norm_
-> optimizer.step.

MI355X notes: the reference averages three wall-clock meters through SIX pickle all-gathers and
three host syncs per iteration (train_utils.py:63-65 -> commu_utils.py:50-111); here the meters
are reduced with one 3-float all-reduce every `log_interval` iterations, and scalars in tb_dict
stay on the device until they are logged."""
import glob
import os
import time

import torch
import torch.distributed as dist
from .optimization import clip_and_step, clip_grad_norm_  # noqa: F401

from ...pcdet.utils import common_utils


def _to_float(v):
    return float(v.item()) if torch.is_tensor(v) else float(v)


def train_one_epoch(model, optimizer, train_loader, model_func, lr_scheduler, accumulated_iter, optim_cfg, rank, tbar,
                    total_it_each_epoch, dataloader_iter, tb_log=None, leave_pbar=False, cur_epoch=0, total_epoch=0,
                    log_interval=10, logger=None):
    if total_it_each_epoch == len(train_loader):
        dataloader_iter = iter(train_loader)
    data_time, forward_time, batch_time = (common_utils.AverageMeter() for _ in range(3))
    disp_dict = {}

    def _next_host_batch():
        nonlocal dataloader_iter
        try:
            return next(dataloader_iter)
        except StopIteration:
            dataloader_iter = iter(train_loader)
            return next(dataloader_iter)

    # Device-side input pipeline (pcdet.models.InputPrefetcher): the batch of iteration t + 1 is uploaded, voxelised and indexed
    # on a side stream once iteration t's backward + optimizer step are enqueued, so its two host syncs and small index kernels
    # run under that backward (the reference does its voxelisation in DataLoader workers, concurrently with training).
    # CenterPoint-Voxel on one MI355X: 18.1 instead of 18.9 ms per step.  TODA_PREFETCH=0: everything on the training stream.
    prefetch = None
    first = next(model.parameters(), None)
    if os.environ.get("TODA_PREFETCH", "1") == "1" and first is not None and first.is_cuda:
        from ...pcdet.models import InputPrefetcher

        def _stream():
            while True:
                yield _next_host_batch()
        net = model.module if hasattr(model, "module") and hasattr(model, "no_sync") else model
        net = getattr(net, "onepass", net)
        if hasattr(net, "dataset"):
            prefetch = InputPrefetcher(_stream(), net, first.device, eager=False)
    def _progress(it):     # reference :47-48: progress in [0, 1) drives the PolarMix ASC / DESC / SIGMOID sector schedules
        if total_epoch:
            train_loader.dataset.train_percent = (cur_epoch * total_it_each_epoch + it) / (total_epoch * total_it_each_epoch)

    try:
        return _train_iterations(model, optimizer, train_loader, model_func, lr_scheduler, accumulated_iter, optim_cfg, rank, tbar, total_it_each_epoch,
                                 tb_log, logger, log_interval, prefetch, _next_host_batch, _progress, data_time, forward_time, batch_time, disp_dict)
    finally:
        if prefetch is not None:
            prefetch.close()       # the worker thread, its side-stream work and its arena slots end with the epoch


def _train_iterations(model, optimizer, train_loader, model_func, lr_scheduler, accumulated_iter, optim_cfg, rank, tbar, total_it_each_epoch,
                      tb_log, logger, log_interval, prefetch, _next_host_batch, _progress, data_time, forward_time, batch_time, disp_dict):
    for cur_it in range(total_it_each_epoch):
        end = time.time()
        if cur_it == 0 or prefetch is None:
            _progress(cur_it)      # before the batch is BUILT (a device-side mix source reads it while mixing)
        batch = prefetch.next() if prefetch is not None else _next_host_batch()
        t_data = time.time() - end
        lr_scheduler.step(accumulated_iter)
        cur_lr = getattr(optimizer, "lr", None)
        if cur_lr is None:
            cur_lr = optimizer.param_groups[0]["lr"]
        model.train()
        optimizer.zero_grad()
        loss, tb_dict, disp = model_func(model, batch)
        t_fwd = time.time() - end
        loss.backward()
        clip_and_step(optimizer, model.parameters(), optim_cfg.GRAD_NORM_CLIP)
        if prefetch is not None and cur_it + 1 < total_it_each_epoch:
            _progress(cur_it + 1)    # the batch built now is the one iteration cur_it + 1 trains on
            prefetch.kick()          # not behind the epoch's last iteration: nothing is fetched that this call does not train on
        accumulated_iter += 1
        data_time.update(t_data)
        forward_time.update(t_fwd)
        batch_time.update(time.time() - end)
        if accumulated_iter % log_interval == 0 or cur_it == total_it_each_epoch - 1:
            meters = torch.tensor([data_time.avg, forward_time.avg, batch_time.avg], dtype=torch.float64)
            if dist.is_available() and dist.is_initialized():
                meters = meters.to(loss.device)
                dist.all_reduce(meters)
                meters = meters.cpu() / dist.get_world_size()
            if rank == 0:
                disp_dict.update({"loss": _to_float(loss), "lr": cur_lr, "d_time": f"{meters[0]:.3f}",
                                  "f_time": f"{meters[1]:.3f}", "b_time": f"{meters[2]:.3f}"})
                if logger is not None:
                    logger.info(f"it {accumulated_iter}: " + ", ".join(f"{k}={v}" for k, v in disp_dict.items()))
                if tb_log is not None:
                    tb_log.add_scalar("train/loss", _to_float(loss), accumulated_iter)
                    tb_log.add_scalar("meta_data/learning_rate", cur_lr, accumulated_iter)
                    for key, val in tb_dict.items():
                        tb_log.add_scalar("train/" + key, _to_float(val), accumulated_iter)
        if rank == 0 and tbar is not None:
            tbar.update()
    return accumulated_iter


def train_model(model, optimizer, train_loader, model_func, lr_scheduler, optim_cfg, start_epoch, total_epochs,
                start_iter, rank, tb_log, ckpt_save_dir, train_sampler=None, lr_warmup_scheduler=None,
                ckpt_save_interval=1, max_ckpt_save_num=50, merge_all_iters_to_one_epoch=False, logger=None):
    accumulated_iter = start_iter
    total_it_each_epoch = len(train_loader)
    if merge_all_iters_to_one_epoch:
        assert hasattr(train_loader.dataset, "merge_all_iters_to_one_epoch")
        train_loader.dataset.merge_all_iters_to_one_epoch(merge=True, epochs=total_epochs)
        total_it_each_epoch = len(train_loader) // max(total_epochs, 1)
    dataloader_iter = iter(train_loader)
    for cur_epoch in range(start_epoch, total_epochs):
        if train_sampler is not None:
            train_sampler.set_epoch(cur_epoch)
        sched = lr_warmup_scheduler if (lr_warmup_scheduler is not None and cur_epoch < optim_cfg.WARMUP_EPOCH) else lr_scheduler
        accumulated_iter = train_one_epoch(model, optimizer, train_loader, model_func, lr_scheduler=sched,
                                           accumulated_iter=accumulated_iter, optim_cfg=optim_cfg, rank=rank, tbar=None,
                                           tb_log=tb_log, total_it_each_epoch=total_it_each_epoch,
                                           dataloader_iter=dataloader_iter, cur_epoch=cur_epoch, total_epoch=total_epochs,
                                           logger=logger)
        trained_epoch = cur_epoch + 1
        if trained_epoch % ckpt_save_interval == 0 and rank == 0 and ckpt_save_dir is not None:
            existing = sorted(glob.glob(os.path.join(str(ckpt_save_dir), "checkpoint_epoch_*.pth")), key=os.path.getmtime)
            for old in existing[:max(0, len(existing) - max_ckpt_save_num + 1)]:
                os.remove(old)
            save_checkpoint(checkpoint_state(model, optimizer, trained_epoch, accumulated_iter),
                            filename=os.path.join(str(ckpt_save_dir), f"checkpoint_epoch_{trained_epoch}"))
    return accumulated_iter


def model_state_to_cpu(model_state):
    return type(model_state)((k, v.cpu()) for k, v in model_state.items())


def checkpoint_state(model=None, optimizer=None, epoch=None, it=None):
    """{'epoch','it','model_state','optimizer_state','version'} exactly like the reference (:149-169)."""
    optim_state = optimizer.state_dict() if optimizer is not None else None
    model_state = None
    if model is not None:
        net = model.module if hasattr(model, "module") and hasattr(model, "no_sync") else model
        model_state = model_state_to_cpu(net.state_dict())
    from ... import pcdet

    return {"epoch": epoch, "it": it, "model_state": model_state, "optimizer_state": optim_state,
            "version": "pcdet+" + pcdet.__version__}


def save_checkpoint(state, filename="checkpoint"):
    torch.save(state, f"{filename}.pth")
