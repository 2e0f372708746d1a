"""Host-side parameter preparation for the DiscoNet warp: the 2x3 affine `theta` exactly as the reference builds it
(pcdet/models/bev_layers/v2x_fusion_disco.py:32-35), in float32 torch-CPU arithmetic.  Six floats per (agent, frame);
the per-pixel work happens in pcp_warp_nearest.
"""
import numpy as np
import torch


def ego_se3_agent(se3_from_ego_agent):
    """float32(inv(float64 se3_from_ego[agent]))  (v2x_fusion_disco.py:93)"""
    return torch.from_numpy(np.linalg.inv(np.asarray(se3_from_ego_agent, dtype=np.float64))).float()


def warp_theta(dst_se3_src, h, w, pc_range_min, pix_size, return_ambiguous=False):
    """dst_se3_src: (4, 4) float32 torch tensor.  Returns 6 python floats (row-major 2x3).
    NB the reference normalises BOTH translation components by H (bev_in_src.shape[1])."""
    T = dst_se3_src.detach().cpu().float()
    rot = T[:2, :2]
    t = T[:2, [-1]]
    t_pix_norm = 2.0 * ((t - pc_range_min) / pix_size) / h - 1.0
    theta = torch.cat([rot.T, -torch.matmul(rot.T, t_pix_norm)], dim=1)      # (2, 3) float32
    vals = [float(v) for v in theta.reshape(-1)]
    if not return_ambiguous:
        return vals
    # pixels whose source coordinate lies within 1e-3 of a .5 tie (rounding there depends on the last ulp of the
    # affine evaluation order): used only by tests to exclude them from exact comparison
    th = theta.double()
    xs = (2.0 * torch.arange(w, dtype=torch.float64) + 1.0) / w - 1.0
    ys = (2.0 * torch.arange(h, dtype=torch.float64) + 1.0) / h - 1.0
    yy, xx = torch.meshgrid(ys, xs, indexing='ij')
    gx = xx * th[0, 0] + yy * th[0, 1] + th[0, 2]
    gy = xx * th[1, 0] + yy * th[1, 1] + th[1, 2]
    fx = ((gx + 1.0) * w - 1.0) / 2.0
    fy = ((gy + 1.0) * h - 1.0) / 2.0
    amb = ((fx - torch.floor(fx) - 0.5).abs() < 1e-3) | ((fy - torch.floor(fy) - 0.5).abs() < 1e-3)
    return vals, amb
