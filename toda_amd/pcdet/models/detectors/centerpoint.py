"""CenterPoint (reference pcdet/models/detectors/centerpoint.py:5-63)."""
from .detector3d_template import Detector3DTemplate


class CenterPoint(Detector3DTemplate):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()

    def forward(self, batch_dict):
        for module in self.module_list:
            batch_dict = module(batch_dict)
        if self.training:
            loss, tb_dict, disp_dict = self.get_training_loss()
            if getattr(self, "return_batch_dict", False):
                # the stage-2 consistency step also needs the raw head outputs (reference models/__init__.py:96-102)
                batch_dict["pred_dicts"] = self.dense_head.forward_ret_dict["pred_dicts"]
                return batch_dict, {"loss": loss}, tb_dict, disp_dict
            return {"loss": loss}, tb_dict, disp_dict
        return self.post_processing(batch_dict)

    def get_training_loss(self):
        loss_rpn, tb_dict = self.dense_head.get_loss()
        return loss_rpn, {"loss_rpn": loss_rpn.detach(), **tb_dict}, {}

    def post_processing(self, batch_dict):
        """The head already decoded + NMS-ed (final_box_dicts); add the recall record (reference centerpoint.py:49-63)."""
        final = batch_dict["final_box_dicts"]
        recall_dict = {}
        thresholds = self.model_cfg.POST_PROCESSING.RECALL_THRESH_LIST
        for index in range(batch_dict["batch_size"]):
            recall_dict = self.generate_recall_record(final[index]["pred_boxes"], recall_dict, index, batch_dict, thresholds)
        return final, recall_dict
