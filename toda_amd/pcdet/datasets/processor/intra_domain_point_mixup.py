"""Drop-in for the reference's pcdet/datasets/processor/intra_domain_point_mixup.py; the point work runs on the MI355X
(see point_mix.py)."""
import copy

from . import point_mix


def _as_dict(first, mixed):
    out = {k: copy.deepcopy(v) for k, v in first.items() if k not in ("points", "gt_boxes")}
    out.update(mixed)
    return out


def intra_domain_point_mixup(data_dict_1, data_dict_2, alpha=None):
    return _as_dict(data_dict_1, point_mix.mixup(data_dict_1, data_dict_2, alpha, collision=False))


def intra_domain_point_mixup_cd(data_dict_1, data_dict_2, alpha=None):
    return _as_dict(data_dict_1, point_mix.mixup(data_dict_1, data_dict_2, alpha, collision=True))
