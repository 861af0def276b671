"""Drop-in for the reference's pcdet/datasets/processor/inter_domain_point_polarmix.py (public entry point and argument
order of :247-300); the point work runs on the MI355X (see point_mix.py)."""
import copy

from . import point_mix
from .point_mix import polar_swap_with_range as swap_with_range  # noqa: F401  (reference :101-151, same arguments)


def inter_domain_point_polarmix(data_dict_source, data_dict_target, polarmix_rot_copy_num, polarmix_degree, train_percent,
                                update_methods, pc_range, polar_dis, inc_method, use_pitch):
    """polar_dis "FULL" (every shipped config) or "RAND"; with RAND the reference's own call of swap_with_range raises a
    TypeError (a stray keyword, :215-220) - here the call is made without it, which is what the function's signature says."""
    mixed = point_mix.polarmix(data_dict_source, data_dict_target, polarmix_rot_copy_num, polarmix_degree, train_percent,
                               update_methods, inc_method, polar_dis=polar_dis, use_pitch=bool(use_pitch), pc_range=pc_range)
    out = {k: copy.deepcopy(v) for k, v in data_dict_target.items() if k not in ("points", "gt_boxes")}
    out.update(mixed)
    return out
