#!/usr/bin/env python
"""tools/test.py of the reference (:21-204), single-checkpoint path:
    python -m toda_amd.tools.test --cfg_file toda_amd/tools/cfgs/models/centerpoint_voxel_waymo.yaml --ckpt out/ckpt/checkpoint_epoch_1.pth
Flags: --cfg_file --batch_size --workers --extra_tag --ckpt --launcher {none,pytorch} --eval_tag --save_to_file --set ...
(--eval_all / --ckpt_dir polling of a training run is not built)."""
import argparse
import datetime
import re
from pathlib import Path

import torch

from ..pcdet.config import cfg, cfg_from_list, cfg_from_yaml_file, log_config_to_file
from ..pcdet.datasets import build_dataloader
from ..pcdet.models import build_network
from ..pcdet.utils import common_utils
from .eval_utils import eval_utils


def parse_config(argv=None):
    p = argparse.ArgumentParser(description="evaluate a checkpoint")
    p.add_argument("--cfg_file", type=str, required=True)
    p.add_argument("--batch_size", type=int, default=None)
    p.add_argument("--workers", type=int, default=0)
    p.add_argument("--extra_tag", type=str, default="default")
    p.add_argument("--ckpt", type=str, default=None)
    p.add_argument("--launcher", choices=["none", "pytorch", "slurm"], default="none")
    p.add_argument("--tcp_port", type=int, default=18888)
    p.add_argument("--local_rank", type=int, default=None)
    p.add_argument("--eval_tag", type=str, default="default")
    p.add_argument("--save_to_file", action="store_true", default=False)
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


def eval_single_ckpt(model, test_loader, args, eval_output_dir, logger, epoch_id, dist_test=False):
    if args.ckpt is not None:
        model.load_params_from_file(filename=args.ckpt, logger=logger, to_cpu=dist_test)
    model.cuda()
    return eval_utils.eval_one_epoch(cfg, model, test_loader, epoch_id, logger, dist_test=dist_test, result_dir=eval_output_dir,
                                     save_to_file=args.save_to_file)


def main(argv=None):
    args, cfg_ = parse_config(argv)
    if args.launcher == "none":
        dist_test, total_gpus = False, 1
    else:
        total_gpus, cfg_.LOCAL_RANK = getattr(common_utils, f"init_dist_{args.launcher}")(args.tcp_port, args.local_rank, backend=args.backend)
        dist_test = True
    if args.batch_size is None:
        args.batch_size = cfg_.OPTIMIZATION.BATCH_SIZE_PER_GPU
    else:
        assert args.batch_size % total_gpus == 0
        args.batch_size //= total_gpus
    root = Path(args.output_dir) if args.output_dir else Path(cfg_.ROOT_DIR) / "output"
    output_dir = root / cfg_.EXP_GROUP_PATH / cfg_.TAG / args.extra_tag
    num = re.findall(r"\d+", Path(args.ckpt).stem) if args.ckpt else []
    epoch_id = num[-1] if num else "no_number"
    eval_output_dir = output_dir / "eval" / f"epoch_{epoch_id}" / cfg_.DATA_CONFIG.get("DATA_SPLIT", {}).get("test", "val") / args.eval_tag
    eval_output_dir.mkdir(parents=True, exist_ok=True)
    logger = common_utils.create_logger(eval_output_dir / ("log_eval_%s.txt" % datetime.datetime.now().strftime("%Y%m%d-%H%M%S")),
                                        rank=cfg_.LOCAL_RANK)
    logger.info("**********************Start logging**********************")
    log_config_to_file(cfg_, logger=logger)
    test_set, test_loader, _ = build_dataloader(dataset_cfg=cfg_.DATA_CONFIG, class_names=cfg_.CLASS_NAMES, batch_size=args.batch_size,
                                                dist=dist_test, workers=args.workers, logger=logger, training=False)
    model = build_network(model_cfg=cfg_.MODEL, num_class=len(cfg_.CLASS_NAMES), dataset=test_set)
    with torch.no_grad():
        return eval_single_ckpt(model, test_loader, args, eval_output_dir, logger, epoch_id, dist_test=dist_test)


if __name__ == "__main__":
    main()
