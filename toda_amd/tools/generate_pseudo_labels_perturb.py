#!/usr/bin/env python
"""tools/generate_pseudo_labels_perturb.py of the reference: generate_pseudo_labels with the adversarial direction stored.
    python -m toda_amd.tools.generate_pseudo_labels_perturb --cfg_file <yaml> --ckpt <pth> --pseudo_thresh 0.3"""
import sys

from .generate_pseudo_labels import main as _main


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    return _main(["--perturb"] + argv)


if __name__ == "__main__":
    main()
