#!/usr/bin/env python
"""TODA stage 2 without the consistency term (reference tools/stage2_mixup_train.py = train.py +
--pseudo_info_path and a single forward).  Pseudo-label files belong to real nuScenes data and are
out of scope; the entry point is kept for the drop-in surface."""
from .train import main

if __name__ == "__main__":
    main()
