"""PointPillar detector (reference: pcdet/models/detectors/pointpillar.py:4-40): module chain + Detector3DTemplate.post_processing."""
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
            raise NotImplementedError('PointPillar / AnchorHeadSingle training is not built')
        return self.post_processing(batch_dict)
