"""CenterNet decode helpers with the reference's entry point (pcdet/models/model_utils/centernet_utils.py:152-214):
decode_bbox_from_heatmap, here one device kernel (sigmoid is applied by the caller, as in the reference)."""
import torch

from pcp_amd import ops


def decode_bbox_from_heatmap(heatmap, rot_cos, rot_sin, center, center_z, dim, point_cloud_range=None, voxel_size=None,
                             feature_map_stride=None, vel=None, K=100, circle_nms=False, score_thresh=None,
                             post_center_limit_range=None):
    """Inputs are (B, C, H, W) tensors: heatmap = sigmoid scores, dim = exp'ed sizes (reference call site
    center_head.py:312-333).  Returns a list of dicts(pred_boxes, pred_scores, pred_labels) per frame.
    vel (B, 2, H, W): two more box columns gathered at the selected cells (reference :174-176 -> (n, 9) boxes).  circle_nms: the reference
    itself stops with `assert False, 'not checked yet'` (:158-160), so does this adapter."""
    assert not circle_nms, 'not checked yet'
    B, C, H, W = heatmap.shape
    # the device kernel reads the caller's ACTIVATED maps as they are (descriptor flag `activated`): no inverse sigmoid / log round trip
    head = torch.zeros((B, H, W, (8 + C + 3) // 4 * 4), dtype=torch.float32, device=heatmap.device)
    head[..., 0:2] = center.permute(0, 2, 3, 1)
    head[..., 2:3] = center_z.permute(0, 2, 3, 1)
    head[..., 3:6] = dim.permute(0, 2, 3, 1)
    head[..., 6:7] = rot_cos.permute(0, 2, 3, 1)
    head[..., 7:8] = rot_sin.permute(0, 2, 3, 1)
    head[..., 8:8 + C] = heatmap.permute(0, 2, 3, 1)
    lim = [float(v) for v in post_center_limit_range]
    kw = dict(k=K, num_class=C, ch_center=0, ch_z=2, ch_dim=3, ch_rot=6, ch_hm=8, stride=feature_map_stride,
              voxel_x=float(voxel_size[0]), voxel_y=float(voxel_size[1]), min_x=float(point_cloud_range[0]),
              min_y=float(point_cloud_range[1]), limit=lim, score_thresh=score_thresh, activated=True)
    boxes, scores, labels, cell, count = ops.centerhead_decode(head, kw)
    if vel is not None:
        # the kernel returns the map cell of every candidate: the velocity columns are a gather at those cells (data movement only)
        flat = vel.permute(0, 2, 3, 1).reshape(B, H * W, vel.shape[1])
        idx = cell.long().clamp_(0, H * W - 1).unsqueeze(-1).expand(-1, -1, vel.shape[1])
        boxes = torch.cat([boxes, flat.gather(1, idx).to(boxes.dtype)], dim=-1)
    counts = count.cpu().tolist()
    return [dict(pred_boxes=boxes[b, :counts[b]], pred_scores=scores[b, :counts[b]], pred_labels=labels[b, :counts[b]])
            for b in range(B)]
