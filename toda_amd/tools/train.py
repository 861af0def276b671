#!/usr/bin/env python
"""tools/train.py of the reference (:21-198) on the MI355X-native stack:
    python -m toda_amd.tools.train --cfg_file toda_amd/tools/cfgs/models/centerpoint_voxel_waymo.yaml
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m toda_amd.tools.train \
        --launcher pytorch --cfg_file ...
Same flags (--cfg_file --batch_size --epochs --workers --extra_tag --ckpt --pretrained_model
--launcher {none,pytorch} --sync_bn --fix_random_seed --ckpt_save_interval --max_ckpt_save_num
--set KEY VALUE ...).  One process per GPU; backend 'nccl' is RCCL over xGMI on ROCm."""
import argparse
import datetime
import glob
import os
from pathlib import Path

import torch
import torch.nn as nn

from ..pcdet.config import cfg, cfg_from_list, cfg_from_yaml_file, log_config_to_file
from ..pcdet.datasets import build_dataloader
from ..pcdet.models import build_network, model_fn_decorator
from ..pcdet.utils import common_utils
from .train_utils.optimization import build_optimizer, build_scheduler
from .train_utils.train_utils import train_model


def parse_config(argv=None):
    p = argparse.ArgumentParser(description="train a detector")
    p.add_argument("--cfg_file", type=str, required=True)
    p.add_argument("--batch_size", type=int, default=None, help="total batch size over all GPUs")
    p.add_argument("--epochs", type=int, default=None)
    p.add_argument("--workers", type=int, default=0)
    p.add_argument("--extra_tag", type=str, default="default")
    p.add_argument("--ckpt", type=str, default=None)
    p.add_argument("--pretrained_model", type=str, default=None)
    p.add_argument("--launcher", choices=["none", "pytorch", "slurm"], default="none")
    p.add_argument("--tcp_port", type=int, default=18888)
    p.add_argument("--sync_bn", action="store_true", default=False)
    p.add_argument("--fix_random_seed", action="store_true", default=False)
    p.add_argument("--ckpt_save_interval", type=int, default=1)
    p.add_argument("--local_rank", type=int, default=None)
    p.add_argument("--max_ckpt_save_num", type=int, default=30)
    p.add_argument("--merge_all_iters_to_one_epoch", action="store_true", default=False)
    p.add_argument("--output_dir", type=str, default=None)
    p.add_argument("--backend", type=str, default="nccl")
    p.add_argument("--set", dest="set_cfgs", default=None, nargs=argparse.REMAINDER)
    args = p.parse_args(argv)
    cfg_from_yaml_file(args.cfg_file, cfg)
    cfg.TAG = Path(args.cfg_file).stem
    cfg.EXP_GROUP_PATH = "/".join(args.cfg_file.split("/")[1:-1])
    if args.set_cfgs is not None:
        cfg_from_list(args.set_cfgs, cfg)
    return args, cfg


def main(argv=None):
    args, cfg = parse_config(argv)
    if args.launcher == "none":
        dist_train, total_gpus = False, 1
    else:
        total_gpus, cfg.LOCAL_RANK = getattr(common_utils, f"init_dist_{args.launcher}")(args.tcp_port, args.local_rank, backend=args.backend)
        dist_train = True
    if args.batch_size is None:
        args.batch_size = cfg.OPTIMIZATION.BATCH_SIZE_PER_GPU
    else:
        assert args.batch_size % total_gpus == 0, "batch size must be divisible by the number of GPUs"
        args.batch_size //= total_gpus
    args.epochs = cfg.OPTIMIZATION.NUM_EPOCHS if args.epochs is None else args.epochs
    if args.fix_random_seed:
        common_utils.set_random_seed(666)

    root = Path(args.output_dir) if args.output_dir else Path(cfg.ROOT_DIR) / "output"
    output_dir = root / cfg.EXP_GROUP_PATH / cfg.TAG / args.extra_tag
    ckpt_dir = output_dir / "ckpt"
    ckpt_dir.mkdir(parents=True, exist_ok=True)
    log_file = output_dir / ("log_train_%s.txt" % datetime.datetime.now().strftime("%Y%m%d-%H%M%S"))
    logger = common_utils.create_logger(log_file, rank=cfg.LOCAL_RANK)
    logger.info("**********************Start logging**********************")
    for key, val in vars(args).items():
        logger.info("{:16} {}".format(key, val))
    log_config_to_file(cfg, logger=logger)

    train_set, train_loader, train_sampler = build_dataloader(
        dataset_cfg=cfg.DATA_CONFIG, class_names=cfg.CLASS_NAMES, batch_size=args.batch_size, dist=dist_train,
        workers=args.workers, logger=logger, training=True,
        merge_all_iters_to_one_epoch=args.merge_all_iters_to_one_epoch, total_epochs=args.epochs)
    model = build_network(model_cfg=cfg.MODEL, num_class=len(cfg.CLASS_NAMES), dataset=train_set)
    if args.sync_bn:
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
    model.cuda()
    optimizer = build_optimizer(model, cfg.OPTIMIZATION)

    start_epoch = it = 0
    last_epoch = -1
    if args.pretrained_model is not None:
        model.load_params_from_file(filename=args.pretrained_model, to_cpu=dist_train, logger=logger)
    if args.ckpt is not None:
        it, start_epoch = model.load_params_with_optimizer(args.ckpt, to_cpu=dist_train, optimizer=optimizer, logger=logger)
        last_epoch = start_epoch + 1
    else:
        ckpts = sorted(glob.glob(str(ckpt_dir / "*checkpoint_epoch_*.pth")), key=os.path.getmtime)
        if ckpts:
            it, start_epoch = model.load_params_with_optimizer(ckpts[-1], to_cpu=dist_train, optimizer=optimizer,
                                                               logger=logger)
            last_epoch = start_epoch + 1
    model.train()
    if dist_train:
        # reference tools/train.py:143: DistributedDataParallel with broadcast_buffers at its default (True) - rank 0's BN running
        # statistics reach every rank before each forward; wrap_ddp does that broadcast as one flat tensor per dtype
        model = common_utils.wrap_ddp(model, device_ids=[cfg.LOCAL_RANK % torch.cuda.device_count()])
    logger.info(model)
    lr_scheduler, lr_warmup_scheduler = build_scheduler(optimizer, total_iters_each_epoch=len(train_loader),
                                                        total_epochs=args.epochs, last_epoch=last_epoch,
                                                        optim_cfg=cfg.OPTIMIZATION)
    logger.info("**********************Start training %s/%s(%s)**********************"
                % (cfg.EXP_GROUP_PATH, cfg.TAG, args.extra_tag))
    train_model(model, optimizer, train_loader, model_func=model_fn_decorator(), lr_scheduler=lr_scheduler,
                optim_cfg=cfg.OPTIMIZATION, start_epoch=start_epoch, total_epochs=args.epochs, start_iter=it,
                rank=cfg.LOCAL_RANK, tb_log=None, ckpt_save_dir=ckpt_dir, train_sampler=train_sampler,
                lr_warmup_scheduler=lr_warmup_scheduler, ckpt_save_interval=args.ckpt_save_interval,
                max_ckpt_save_num=args.max_ckpt_save_num,
                merge_all_iters_to_one_epoch=args.merge_all_iters_to_one_epoch, logger=logger)
    logger.info("**********************End training**********************")


if __name__ == "__main__":
    main()
