"""PointPillarScatter: the dense canvas is produced by the VFE kernel itself (one 256-B NHWC row per pillar), so this module
only publishes it under the reference's key and shape (pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py:14-37):
`spatial_features` (B, C, ny, nx), here an NCHW-shaped view of channels-last storage -- no .item() sync (quirk Q4)."""
import torch
import torch.nn as nn

from pcp_amd import ops


class PointPillarScatter(nn.Module):
    def __init__(self, model_cfg, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = self.model_cfg.NUM_BEV_FEATURES
        self.nx, self.ny, self.nz = [int(v) for v in grid_size]
        assert self.nz == 1

    def forward(self, batch_dict, **kwargs):
        stash = batch_dict.get('_pcp_vfe', None)
        if stash is not None:
            canvas = stash['canvas']
            if canvas is None:            # pipeline mode, sparse first layer: the backbone consumes the pillar list (no dense canvas)
                batch_dict['spatial_features'] = None
                return batch_dict
        else:
            # foreign producer: pillar_features (P, C) + voxel_coords (P, 4) -> indexed row write into an NHWC canvas
            pf, vc = batch_dict['pillar_features'], batch_dict['voxel_coords'].long()
            if not pf.is_cuda:
                raise RuntimeError('PointPillarScatter needs CUDA/ROCm tensors; there is no CPU fallback')
            bs = batch_dict.get('batch_size', None) or int(vc[:, 0].max().item()) + 1
            canvas = torch.zeros((bs, self.ny, self.nx, pf.shape[1]), dtype=pf.dtype, device=pf.device)
            canvas[vc[:, 0], vc[:, 2], vc[:, 3]] = pf
        batch_dict['spatial_features'] = ops.nchw_view(canvas)
        return batch_dict
