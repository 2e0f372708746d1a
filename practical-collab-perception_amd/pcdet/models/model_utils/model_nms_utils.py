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


def multi_classes_nms(cls_scores, box_preds, nms_config, score_thresh=None):
    """per-class rotated NMS with the reference's signature and return order (pcdet/models/model_utils/model_nms_utils.py:28-66):
    cls_scores (N, num_class) activated scores, box_preds (N, 7 + C) -> (pred_scores, pred_labels, pred_boxes), classes concatenated in
    ascending order, labels 0-based (the caller maps them).  Each class: score mask -> device NMS (sort, pre-max, greedy sweep on
    pcp_nms_rotated / pcp_nms_normal) -> post-max."""
    pred_scores, pred_labels, pred_boxes = [], [], []
    for k in range(cls_scores.shape[1]):
        if score_thresh is not None:
            scores_mask = cls_scores[:, k] >= score_thresh
            box_scores = cls_scores[scores_mask, k]
            cur_box_preds = box_preds[scores_mask]
        else:
            box_scores = cls_scores[:, k]
            cur_box_preds = box_preds
        selected = torch.zeros((0,), dtype=torch.long, device=cls_scores.device)
        if box_scores.shape[0] > 0:
            keep_idx, _ = getattr(iou3d_nms_utils, nms_config.NMS_TYPE)(
                cur_box_preds[:, 0:7].contiguous(), box_scores.contiguous(), nms_config.NMS_THRESH, pre_maxsize=nms_config.NMS_PRE_MAXSIZE)
            selected = keep_idx[:nms_config.NMS_POST_MAXSIZE]
        pred_scores.append(box_scores[selected])
        pred_labels.append(box_scores.new_ones(len(selected)).long() * k)
        pred_boxes.append(cur_box_preds[selected])
    return torch.cat(pred_scores, dim=0), torch.cat(pred_labels, dim=0), torch.cat(pred_boxes, dim=0)
