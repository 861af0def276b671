"""Drop-in for the reference's pcdet/datasets/processor/inter_domain_point_cutmix.py (same function, same arguments);
the point work runs on the MI355X (see point_mix.py)."""
import copy

from . import point_mix


def inter_domain_point_cutmix(data_source, data_target, pc_range, inc_method):
    """CutMix of a source (Waymo) and a target (nuScenes) scene -> a copy of the target dict with mixed `points` and
    `gt_boxes` (reference :10-90; `inc_method` is accepted and unused there too)."""
    mixed = point_mix.cutmix(data_source, data_target, pc_range)
    out = {k: copy.deepcopy(v) for k, v in data_target.items() if k not in ("points", "gt_boxes")}
    out.update(mixed)
    return out
