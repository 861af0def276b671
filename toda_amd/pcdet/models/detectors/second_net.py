"""SECONDNet (reference pcdet/models/detectors/second_net.py:4-34)."""
from .detector3d_template import Detector3DTemplate


class SECONDNet(Detector3DTemplate):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()

    def forward(self, batch_dict):
        for module in self.module_list:
            batch_dict = module(batch_dict)
        if self.training:
            loss_rpn, tb_dict = self.dense_head.get_loss()
            return {"loss": loss_rpn}, {"loss_rpn": loss_rpn.detach(), **tb_dict}, {}
        return self.post_processing(batch_dict)


class PointPillar(SECONDNet):
    """PointPillar (reference pcdet/models/detectors/pointpillar.py:4-34): same control flow."""
