"""class_agnostic_nms with the reference's signature (pcdet/models/model_utils/model_nms_utils.py:6-25)."""
import torch

from ...ops.iou3d_nms import iou3d_nms_utils


def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    src_box_scores = box_scores
    if score_thresh is not None:
        scores_mask = box_scores >= score_thresh
        box_scores = box_scores[scores_mask]
        box_preds = box_preds[scores_mask]
    selected = []
    if box_scores.shape[0] > 0:
        # the device op sorts by score itself; pre-max is applied after the sort exactly like topk(k) + sort
        keep_idx, _ = getattr(iou3d_nms_utils, nms_config.NMS_TYPE)(
            box_preds[:, 0:7], box_scores, nms_config.NMS_THRESH, pre_maxsize=nms_config.NMS_PRE_MAXSIZE)
        selected = keep_idx[:nms_config.NMS_POST_MAXSIZE]
    if not torch.is_tensor(selected):
        selected = torch.zeros((0,), dtype=torch.long, device=src_box_scores.device)
    if score_thresh is not None:
        original_idxs = scores_mask.nonzero().view(-1)
        selected = original_idxs[selected]
    return selected, src_box_scores[selected]
