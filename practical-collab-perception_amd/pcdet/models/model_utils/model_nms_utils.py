"""The two NMS front doors of the reference's plugin surface (pcdet/models/model_utils/model_nms_utils.py:6-66), same names, arguments and
return values, over the device NMS of this build (pcp_nms_rotated / pcp_nms_normal behind pcdet.ops.iou3d_nms.iou3d_nms_utils: score sort,
NMS_PRE_MAXSIZE cut and greedy sweep all happen inside one call, no host round trip)."""
import torch

from ...ops.iou3d_nms import iou3d_nms_utils


def _survivors(boxes, scores, nms_config, score_thresh):
    """rows of `boxes` / `scores` (original numbering) that survive: score >= score_thresh (when given), then the configured device NMS,
    best first, at most NMS_POST_MAXSIZE.  -> int64 index tensor"""
    dev = scores.device
    candidates = torch.arange(scores.shape[0], device=dev) if score_thresh is None else torch.nonzero(scores >= score_thresh).reshape(-1)
    if candidates.numel() == 0:
        return candidates.new_zeros((0,))
    nms = getattr(iou3d_nms_utils, nms_config.NMS_TYPE)
    kept, _ = nms(boxes[candidates, 0:7].contiguous(), scores[candidates].contiguous(), nms_config.NMS_THRESH, pre_maxsize=nms_config.NMS_PRE_MAXSIZE)
    return candidates[kept[:nms_config.NMS_POST_MAXSIZE]]


def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    """-> (indices into the caller's rows, their scores), reference :6-25"""
    winners = _survivors(box_preds, box_scores, nms_config, score_thresh)
    return winners, box_scores[winners]


def multi_classes_nms(cls_scores, box_preds, nms_config, score_thresh=None):
    """cls_scores (N, num_class) activated scores, box_preds (N, 7 + C) -> (pred_scores, pred_labels, pred_boxes): every class filtered on its
    own column, classes concatenated in ascending order, labels 0-based (the caller maps them) -- reference :28-66"""
    per_class = [_survivors(box_preds, cls_scores[:, k], nms_config, score_thresh) for k in range(cls_scores.shape[1])]
    rows = torch.cat(per_class, dim=0)
    labels = torch.cat([torch.full((w.numel(),), k, dtype=torch.long, device=cls_scores.device) for k, w in enumerate(per_class)], dim=0)
    return cls_scores[rows, labels], labels, box_preds[rows]
