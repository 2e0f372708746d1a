"""Python surface of the rotated-IoU / NMS op, call-compatible with the reference's
pcdet/ops/iou3d_nms/iou3d_nms_utils.py (:10-99, :102-189), on top of libpcp_hip.so instead of iou3d_nms_cuda.

Unlike the reference's nms_gpu (mask on the GPU, D2H copy, greedy loop on the CPU, `keep` returned through a CPU tensor --
iou3d_nms.cpp:103-135) everything stays on the device; results are CUDA tensors.
"""
import torch

from pcp_amd import ops


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """reference :12-29 (-> iou3d_cpu.cpp boxes_iou_bev_cpu): CPU tensors or numpy arrays in, the same kind out.  The arithmetic of the
    reference's CPU file is, operation for operation, that of its CUDA kernel (oracle/nms_oracle.c pins both), so the pairs are evaluated by
    pcp_boxes_bev_pairwise on the device and copied back; without the device this raises like every other entry of the product."""
    import numpy as np
    is_numpy = isinstance(boxes_a, np.ndarray)
    a = torch.from_numpy(boxes_a).float() if isinstance(boxes_a, np.ndarray) else boxes_a
    b = torch.from_numpy(boxes_b).float() if isinstance(boxes_b, np.ndarray) else boxes_b
    assert not (a.is_cuda or b.is_cuda), 'Only support CPU tensors'
    assert a.shape[1] == 7 and b.shape[1] == 7
    if not torch.cuda.is_available():
        from pcp_amd import lib as _lib
        raise _lib.PcpError('boxes_bev_iou_cpu evaluates the pairs on the HIP device (no CPU fallback exists) and found none')
    out = ops.boxes_bev_pairwise(a.float().cuda(), b.float().cuda(), 1).cpu()
    return out.numpy() if is_numpy else out


def boxes_iou_bev(boxes_a, boxes_b):
    """(N,7) x (M,7) [x, y, z, dx, dy, dz, heading] -> (N, M) rotated BEV IoU"""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    return ops.boxes_bev_pairwise(boxes_a.float(), boxes_b.float(), 1)


def boxes_overlap_bev(boxes_a, boxes_b):
    return ops.boxes_bev_pairwise(boxes_a.float(), boxes_b.float(), 0)


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """3-D IoU = BEV overlap x height overlap / union volume (reference :48-81)."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a_max = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    a_min = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_max = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1)
    b_min = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    overlaps_bev = boxes_overlap_bev(boxes_a, boxes_b)
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """boxes (N,7), scores (N,) -> (indices of kept boxes into the input, None); sort + mask + greedy on the device."""
    assert boxes.shape[1] == 7
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.long, device=boxes.device), None
    pre = int(pre_maxsize) if pre_maxsize is not None else n
    keep, cnt = ops.nms_rotated(boxes.float().contiguous(), scores.float().contiguous(), thresh, pre, n)
    return keep[:int(cnt.item())].long(), None


def nms_normal_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """reference :102-117 (-> nms_normal_gpu, iou3d_nms.cpp:139-188): score sort + axis-aligned IoU mask + greedy, all on the device.
    pre_maxsize: the reference's callers cut to topk(NMS_PRE_MAXSIZE) BEFORE either NMS type (model_nms_utils.py:15,50); the callers here
    hand the whole candidate list over and the op applies the cut behind its own score sort (same candidates, same order)."""
    assert boxes.shape[1] == 7
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.long, device=boxes.device), None
    pre = int(pre_maxsize) if pre_maxsize is not None else n
    keep, cnt = ops.nms_normal(boxes.float().contiguous(), scores.float().contiguous(), thresh, pre, n)
    return keep[:int(cnt.item())].long(), None
