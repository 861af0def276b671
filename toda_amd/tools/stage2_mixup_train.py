#!/usr/bin/env python
"""TODA stage 2 without the consistency term (reference tools/stage2_mixup_train.py = train.py on the pseudo-labelled
target set, a single forward per step): point DATA_CONFIG.PSEUDO_INFO_PATH at the infos file written by
tools/generate_pseudo_labels.py, e.g.
    python -m toda_amd.tools.stage2_mixup_train --cfg_file toda_amd/tools/cfgs/models/toda_stage1_centerpoint_res.yaml \
        --pretrained_model <stage-1 ckpt> --set DATA_CONFIG.PSEUDO_INFO_PATH <score_..._infos.pkl>
The MixUp + adversarial + consistency recipe is stage2_mixup_train_cl.py."""
from .train import main

if __name__ == "__main__":
    main()
