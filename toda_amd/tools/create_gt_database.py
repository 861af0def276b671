#!/usr/bin/env python
"""Build the GT-sampling object database of a dataset (the reference does this inside each dataset's
`create_groundtruth_database`, e.g. pcdet/datasets/nuscenes/nuscenes_dataset.py:370-412):
    python -m toda_amd.tools.create_gt_database --cfg_file <yaml> --out <dir> [--classes car ...]
writes <dir>/gt_database/*.bin, <dir>/dbinfos.pkl and <dir>/gt_database_global.npy (packed, for USE_SHARED_MEMORY)."""
import argparse

from ..pcdet.config import cfg, cfg_from_yaml_file
from ..pcdet.datasets import __all__ as registry
from ..pcdet.datasets.augmentor.database_sampler import create_groundtruth_database


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--cfg_file", required=True)
    p.add_argument("--out", required=True)
    p.add_argument("--classes", nargs="*", default=None)
    args = p.parse_args(argv)
    cfg_from_yaml_file(args.cfg_file, cfg)
    dataset = registry[cfg.DATA_CONFIG.DATASET](dataset_cfg=cfg.DATA_CONFIG, class_names=cfg.CLASS_NAMES, training=False)
    infos = create_groundtruth_database(dataset, args.out, used_classes=args.classes)
    for name, items in infos.items():
        print(f"Database {name}: {len(items)}")


if __name__ == "__main__":
    main()
