#!/usr/bin/env python
"""tools/generate_pseudo_labels.py of the reference (:20-142): load a stage-1 checkpoint, run it over the unlabeled
target frames and write the thresholded detections into a pseudo-label infos pickle for stage 2.
    python -m toda_amd.tools.generate_pseudo_labels --cfg_file <yaml> --ckpt <pth> --pseudo_thresh 0.3 [--unlabel_infos <pkl>]
Without --unlabel_infos the dataset's own (label-free view of its) infos are dumped first and used."""
import argparse
from pathlib import Path

import torch

from ..pcdet.config import cfg, cfg_from_list, cfg_from_yaml_file
from ..pcdet.datasets import build_dataloader
from ..pcdet.models import build_network
from ..pcdet.utils import common_utils
from .eval_utils.generate_pseudo_labels import inference_and_generate_pseudo_labes


def parse_config(argv=None):
    p = argparse.ArgumentParser(description="generate pseudo labels with a trained detector")
    p.add_argument("--cfg_file", type=str, required=True)
    p.add_argument("--batch_size", type=int, default=None)
    p.add_argument("--workers", type=int, default=0)
    p.add_argument("--extra_tag", type=str, default="default")
    p.add_argument("--ckpt", type=str, default=None)
    p.add_argument("--launcher", choices=["none", "pytorch", "slurm"], default="none")
    p.add_argument("--tcp_port", type=int, default=18888)
    p.add_argument("--local_rank", type=int, default=None)
    p.add_argument("--save_to_file", action="store_true", default=False)
    p.add_argument("--pseudo_thresh", type=float, required=True)
    p.add_argument("--unlabel_infos", type=str, default=None)
    p.add_argument("--perturb", action="store_true", default=False,
                   help="also store d loss / d voxels per frame (reference tools/generate_pseudo_labels_perturb.py)")
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
    args, cfg_ = parse_config(argv)
    if args.launcher == "none":
        dist_test, total_gpus = False, 1
    else:
        total_gpus, cfg_.LOCAL_RANK = getattr(common_utils, f"init_dist_{args.launcher}")(args.tcp_port, args.local_rank, backend=args.backend)
        dist_test = True
    batch_size = (args.batch_size // total_gpus) if args.batch_size else cfg_.OPTIMIZATION.BATCH_SIZE_PER_GPU
    root = Path(args.output_dir) if args.output_dir else Path(cfg_.ROOT_DIR) / "output"
    result_dir = root / cfg_.EXP_GROUP_PATH / cfg_.TAG / args.extra_tag / "pseudo_labels"
    result_dir.mkdir(parents=True, exist_ok=True)
    logger = common_utils.create_logger(result_dir / "log_pseudo.txt", rank=cfg_.LOCAL_RANK)
    dataset, loader, _ = build_dataloader(dataset_cfg=cfg_.DATA_CONFIG, class_names=cfg_.CLASS_NAMES, batch_size=batch_size,
                                          dist=dist_test, workers=args.workers, logger=logger, training=False)
    infos_path = args.unlabel_infos
    if infos_path is None:
        infos_path = result_dir / "unlabel_infos.pkl"
        if cfg_.LOCAL_RANK == 0:
            dataset.dump_infos(infos_path)
    model = build_network(model_cfg=cfg_.MODEL, num_class=len(cfg_.CLASS_NAMES), dataset=dataset)
    if args.ckpt is not None:
        model.load_params_from_file(filename=args.ckpt, logger=logger, to_cpu=dist_test)
    model.cuda()
    if args.perturb:
        from .eval_utils.generate_pseudo_labels_perturb import inference_and_generate_pseudo_labes as with_perturb
        return with_perturb(cfg_, args, model, loader, logger, dist_test=dist_test, save_to_file=args.save_to_file,
                            result_dir=result_dir, unlabel_infos_path=infos_path)
    with torch.no_grad():
        return inference_and_generate_pseudo_labes(cfg_, args, model, loader, logger, dist_test=dist_test, save_to_file=args.save_to_file,
                                                   result_dir=result_dir, unlabel_infos_path=infos_path)


if __name__ == "__main__":
    main()
