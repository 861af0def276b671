#!/usr/bin/env python
"""TODA stage 1 (reference tools/stage1_cutmix_train.py): identical to train.py except that the
dataloader yields inter-domain mixed scenes.  The mix processors themselves (PolarMix / CutMix /
LaserMix, host-side numpy in DataLoader workers) are out of scope; the synthetic
TODA dataset alternates Waymo-like and nuScenes-like clouds in the stage-1 range / voxel geometry,
so the GPU step sees the same shapes: 1 forward + 1 backward."""
from .train import main

if __name__ == "__main__":
    main()
