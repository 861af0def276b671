#!/usr/bin/env python
"""TODA stage 1 (reference tools/stage1_cutmix_train.py): train.py on a dataloader that yields inter-domain mixed
scenes.  With DATA_CONFIG.DATASET = SyntheticMixDataset (cfgs/models/toda_stage1_polarmix.yaml) the mix - PolarMix /
CutMix / LaserMix, toda_amd/pcdet/datasets/processor/point_mix.py - runs on the MI355X inside the training process
(use --workers 0); the dataset's `train_percent` (PolarMix ASC / DESC sector schedules) follows the iteration count.
With the plain synthetic TODA dataset the loop sees pre-mixed shapes only.  Either way: 1 forward + 1 backward."""
from .train import main

if __name__ == "__main__":
    main()
