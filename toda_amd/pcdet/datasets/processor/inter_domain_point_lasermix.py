"""Drop-in for the reference's pcdet/datasets/processor/inter_domain_point_lasermix.py (entry point :175-195); both variants -
cylindrical (LASERMIX_NUM_ANGLES set, every shipped config) and spherical - run on the MI355X (see point_mix.py)."""
import copy

from . import point_mix


def laser_mix_transform_sph(input_dict, mix_results, pitch_angles, num_areas, order=0):
    """Reference :22-85, same arguments: bands with i % 2 == order from input_dict, the others from mix_results."""
    out = {k: copy.deepcopy(v) for k, v in mix_results.items() if k not in ("points", "gt_boxes")}
    out.update(point_mix.lasermix_sph(input_dict, mix_results, pitch_angles, num_areas, order))
    return out


def inter_domain_point_lasermix(data_dict_source, data_dict_target, pitch_angle, num_areas, num_angles, pc_range, inc_method):
    if num_angles is None:
        # as the reference (:186-192): inc_method lands in `order`, so every elevation band is taken from the target scene
        return laser_mix_transform_sph(data_dict_source, data_dict_target, pitch_angle, num_areas, inc_method)
    mixed = point_mix.lasermix_cyc(data_dict_source, data_dict_target, num_areas, num_angles, pc_range, inc_method)
    out = {k: copy.deepcopy(v) for k, v in data_dict_target.items() if k not in ("points", "gt_boxes")}
    out.update(mixed)
    return out
