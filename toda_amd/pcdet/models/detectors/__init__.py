from .centerpoint import CenterPoint
from .detector3d_template import Detector3DTemplate
from .second_net import PointPillar, SECONDNet

__all__ = {
    "Detector3DTemplate": Detector3DTemplate,
    "SECONDNet": SECONDNet,
    "PointPillar": PointPillar,
    "CenterPoint": CenterPoint,
}


def build_detector(model_cfg, num_class, dataset):
    return __all__[model_cfg.NAME](model_cfg=model_cfg, num_class=num_class, dataset=dataset)
