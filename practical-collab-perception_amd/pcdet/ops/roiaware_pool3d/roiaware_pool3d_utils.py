"""points_in_boxes_gpu with the reference's signature (pcdet/ops/roiaware_pool3d/roiaware_pool3d_utils.py:34-49), on pcp_points_in_boxes."""
import torch

from pcp_amd import ops


def points_in_boxes_gpu(points, boxes):
    """points: (B, M, 3) CUDA float32, boxes: (B, T, 7+) [x, y, z, dx, dy, dz, heading, ...] (centre = box centre).
    Returns (B, M) int32: index of the first box containing the point, -1 = background."""
    assert boxes.shape[0] == points.shape[0] and boxes.shape[2] >= 7 and points.shape[2] == 3
    return ops.points_in_boxes(points, boxes)
