"""Drop-in for the reference's pcdet/datasets/processor/inter_domain_point_lasermix.py (entry point :175-195); the
cylindrical variant (LASERMIX_NUM_ANGLES set, every shipped config) runs on the MI355X (see point_mix.py)."""
import copy

from . import point_mix


def inter_domain_point_lasermix(data_dict_source, data_dict_target, pitch_angle, num_areas, num_angles, pc_range, inc_method):
    if num_angles is None:
        # the reference's spherical branch passes inc_method where laser_mix_transform_sph expects `order` (:186-192)
        raise NotImplementedError("LASERMIX_NUM_ANGLES must be set (cylindrical LaserMix)")
    mixed = point_mix.lasermix_cyc(data_dict_source, data_dict_target, num_areas, num_angles, pc_range, inc_method)
    out = {k: copy.deepcopy(v) for k, v in data_dict_target.items() if k not in ("points", "gt_boxes")}
    out.update(mixed)
    return out
