"""Drop-in for the reference's pcdet/datasets/processor/inter_domain_point_pseudomix.py (:19-68: MIX_TYPE pseudobbox /
pseudobackground of tools/cfgs/stage1_pseudomix); the point work runs on the MI355X (see point_mix.py).  Like the reference,
both functions write the mixed points and boxes into data_target and return it."""
from . import point_mix


def inter_domain_point_pseudobbox(data_source, data_target):
    data_target.update(point_mix.pseudobbox(data_source, data_target))
    return data_target


def inter_domain_point_pseudobackground(data_source, data_target):
    data_target.update(point_mix.pseudobackground(data_source, data_target))
    return data_target
