"""V2XLateFusion: box-level fusion of the agents' exchanged detections (reference: pcdet/models/detectors/v2x_late_fusion.py:7-68).
No network: metadata[b]['exchange_boxes'] = {agent_id: (n, 9) [box7, score, label]} -> class-agnostic rotated NMS
(thr 0.3, pre 4096, post 500 in v2x_late_fusion.yaml) on the device NMS kernel."""
import numpy as np
import torch

from ..model_utils import model_nms_utils
from .detector3d_template import Detector3DTemplate


class V2XLateFusion(Detector3DTemplate):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()
        self.post_process_cfg = model_cfg.POST_PROCESSING

    def forward(self, batch_dict):
        assert not self.training, 'there is nothing to train'
        final_box_dicts = []
        for meta in batch_dict['metadata']:
            exchange = meta['exchange_boxes']
            if self.model_cfg.BOX_FUSION_METHOD == 'nms':
                parts = [b for _, b in exchange.items() if b.shape[0] > 0]
                allb = np.concatenate(parts) if parts else np.zeros((0, 9), dtype=np.float32)
                boxes = torch.from_numpy(np.ascontiguousarray(allb, dtype=np.float32)).cuda()
                selected, selected_scores = model_nms_utils.class_agnostic_nms(
                    box_scores=boxes[:, -2].contiguous(), box_preds=boxes[:, :7].contiguous(),
                    nms_config=self.post_process_cfg.NMS_CONFIG, score_thresh=self.post_process_cfg.SCORE_THRESH)
                final_box_dicts.append({'pred_boxes': boxes[selected, :7], 'pred_scores': selected_scores,
                                        'pred_labels': boxes[selected, -1].long()})
            elif self.model_cfg.BOX_FUSION_METHOD == 'ego_only':
                boxes = torch.from_numpy(np.asarray(exchange[1], dtype=np.float32)).cuda()
                final_box_dicts.append({'pred_boxes': boxes[:, :7], 'pred_scores': boxes[:, -2], 'pred_labels': boxes[:, -1].long()})
            else:
                raise NotImplementedError('BOX_FUSION_METHOD: %s is not implemented' % self.model_cfg.BOX_FUSION_METHOD)
        batch_dict['final_box_dicts'] = final_box_dicts
        recall_dict = {}
        for index in range(batch_dict['batch_size']):
            recall_dict = self.generate_recall_record(
                box_preds=final_box_dicts[index]['pred_boxes'], recall_dict=recall_dict, batch_index=index, data_dict=batch_dict,
                thresh_list=self.post_process_cfg.RECALL_THRESH_LIST)
        return final_box_dicts, recall_dict
