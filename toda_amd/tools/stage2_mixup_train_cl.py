#!/usr/bin/env python
"""TODA stage 2 with the consistency loss: 2 forwards + 1 backward per step (reference
tools/stage2_mixup_train_cl.py; its train_utils_cl module is missing from the reference tree, so
the loop below is written from model_fn_decorator_cl, pcdet/models/__init__.py:88-125).

    python -m toda_amd.tools.stage2_mixup_train_cl --cfg_file toda_amd/tools/cfgs/models/toda_stage1_centerpoint_res.yaml

Both passes run inside one DistModel.forward so DDP issues one gradient all-reduce per step."""
import os
import time

import torch
import torch.nn as nn
from .train_utils.optimization import clip_and_step, clip_grad_norm_  # noqa: F401

from ..pcdet.config import cfg
from ..pcdet import datasets as dataset_registry
from ..pcdet.datasets import SyntheticPairDataset
from ..pcdet.models import DistModel, build_network, model_fn_decorator_cl
from ..pcdet.utils import common_utils
from .train import parse_config
from .train_utils.optimization import build_optimizer, build_scheduler


def train_one_epoch_cl(model, optimizer, loader, model_func, lr_scheduler, accumulated_iter, optim_cfg, rank, dist_train,
                       logger=None, log_interval=10, max_iters=None):
    # device-side input pipeline as in tools/train_utils/train_utils.py: the next (adv, org) pair is uploaded, voxelised and
    # indexed on a side stream while this pair's backward runs (TODA_PREFETCH=0: on the training stream)
    n_iters = len(loader) if max_iters is None else min(len(loader), max_iters)
    source = iter(loader)
    prefetch = None
    first = next(model.parameters(), None)
    if os.environ.get("TODA_PREFETCH", "1") == "1" and first is not None and first.is_cuda:
        from ..pcdet.models import InputPrefetcher
        net = model.module if hasattr(model, "module") and hasattr(model, "no_sync") else model
        net = getattr(net, "onepass", net)
        if hasattr(net, "dataset"):
            prefetch = InputPrefetcher(source, net, first.device, eager=False)
    try:
        return _stage2_iterations(model, optimizer, model_func, lr_scheduler, accumulated_iter, optim_cfg, rank, logger, log_interval, dist_train,
                                  n_iters, source, prefetch)
    finally:
        if prefetch is not None:
            prefetch.close()


def _stage2_iterations(model, optimizer, model_func, lr_scheduler, accumulated_iter, optim_cfg, rank, logger, log_interval, dist_train, n_iters,
                       source, prefetch):
    for it in range(n_iters):
        adv, org = prefetch.next() if prefetch is not None else next(source)
        lr_scheduler.step(accumulated_iter)
        model.train()
        optimizer.zero_grad()
        loss, tb_dict, _ = model_func(model, adv, org, dist_train)
        loss.backward()
        clip_and_step(optimizer, model.parameters(), optim_cfg.GRAD_NORM_CLIP)
        if prefetch is not None and it + 1 < n_iters:
            prefetch.kick()
        accumulated_iter += 1
        if rank == 0 and logger is not None and accumulated_iter % log_interval == 0:
            logger.info(f"it {accumulated_iter}: loss={float(loss):.4f} " +
                        " ".join(f"{k}={float(v):.4f}" for k, v in tb_dict.items()))
    return accumulated_iter


def main(argv=None):
    args, _ = parse_config(argv)
    if args.launcher == "none":
        dist_train, total_gpus = False, 1
    else:
        total_gpus, cfg.LOCAL_RANK = common_utils.init_dist_pytorch(args.tcp_port, args.local_rank, backend=args.backend)
        dist_train = True
    bs = cfg.OPTIMIZATION.BATCH_SIZE_PER_GPU if args.batch_size is None else args.batch_size // total_gpus
    epochs = cfg.OPTIMIZATION.NUM_EPOCHS if args.epochs is None else args.epochs
    logger = common_utils.create_logger(None, rank=cfg.LOCAL_RANK)
    # DATA_CONFIG.DATASET names a pair dataset (SyntheticMixupPairDataset: MixUp + adversarial frames, the TODA recipe);
    # configs written for the single-frame datasets fall back to the plain (adv, org) pair of one frame
    cls = dataset_registry.__all__.get(cfg.DATA_CONFIG.DATASET, SyntheticPairDataset)
    if not issubclass(cls, SyntheticPairDataset):
        cls = SyntheticPairDataset
    dataset = cls(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True)
    sampler = torch.utils.data.distributed.DistributedSampler(dataset) if dist_train else None
    loader = torch.utils.data.DataLoader(dataset, batch_size=bs, shuffle=sampler is None, sampler=sampler,
                                         num_workers=args.workers, collate_fn=dataset.collate_batch)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), dataset).cuda()
    optimizer = build_optimizer(model, cfg.OPTIMIZATION)
    wrapped = DistModel(model)
    if dist_train:
        wrapped = common_utils.wrap_ddp(wrapped, device_ids=[cfg.LOCAL_RANK % torch.cuda.device_count()])   # buffers broadcast from rank 0 each forward, as the reference's default
    scheduler, _ = build_scheduler(optimizer, len(loader), epochs, -1, cfg.OPTIMIZATION)
    fn = model_fn_decorator_cl()
    it = 0
    if args.pretrained_model is not None:
        model.load_params_from_file(filename=args.pretrained_model, to_cpu=dist_train, logger=logger)
    for epoch in range(epochs):
        if sampler is not None:
            sampler.set_epoch(epoch)
        t0 = time.time()
        it = train_one_epoch_cl(wrapped, optimizer, loader, fn, scheduler, it, cfg.OPTIMIZATION, cfg.LOCAL_RANK, dist_train,
                                logger=logger)
        logger.info(f"epoch {epoch} done in {time.time() - t0:.1f} s")
    return it


if __name__ == "__main__":
    main()
