"""PointPillar detector (reference: pcdet/models/detectors/pointpillar.py:4-40): module chain + Detector3DTemplate.post_processing;
train mode returns ({'loss': ...}, tb_dict, disp_dict) with loss = dense_head.get_loss() (:20-33)."""
from .centerpoint import hip_loss
from .detector3d_template import Detector3DTemplate


class PointPillar(Detector3DTemplate):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()

    def forward(self, batch_dict):
        # the score mask of post_processing is applied inside the decode kernel: tell the head before it runs
        batch_dict['_pcp_score_thresh'] = self.model_cfg.POST_PROCESSING.SCORE_THRESH
        for cur_module in self.module_list:
            batch_dict = cur_module(batch_dict)
        if self.training:
            loss, tb_dict, disp_dict = self.get_training_loss()
            return {'loss': hip_loss(loss, batch_dict.get('_pcp_tape', []))}, tb_dict, disp_dict
        return self.post_processing(batch_dict)

    def get_training_loss(self):
        disp_dict = {}
        loss_rpn, tb_dict = self.dense_head.get_loss()
        tb_dict = {'loss_rpn': loss_rpn.item(), **tb_dict}
        return loss_rpn, tb_dict, disp_dict
