"""Drop-in for the reference's pcdet/datasets/processor/inter_domain_point_polarmix.py (public entry point and argument
order of :247-300); the point work runs on the MI355X (see point_mix.py)."""
import copy

from . import point_mix


def inter_domain_point_polarmix(data_dict_source, data_dict_target, polarmix_rot_copy_num, polarmix_degree, train_percent,
                                update_methods, pc_range, polar_dis, inc_method, use_pitch):
    if polar_dis != "FULL":
        # the reference's RAND branch calls swap_with_range with an argument that function does not accept (:215-220)
        raise NotImplementedError("POLARMIX_DIS must be FULL")
    if use_pitch:
        raise NotImplementedError("POLARMIX_USE_PITCH is not supported")
    mixed = point_mix.polarmix(data_dict_source, data_dict_target, polarmix_rot_copy_num, polarmix_degree, train_percent,
                               update_methods, inc_method)
    out = {k: copy.deepcopy(v) for k, v in data_dict_target.items() if k not in ("points", "gt_boxes")}
    out.update(mixed)
    return out
