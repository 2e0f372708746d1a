"""DynPillarVFE on gfx950: one pcp_pillarise_rows (dense count + scan, no sort; rows left in pillar order) and one fused pcp_pfn_rows
launch replace torch.unique + 3 torch_scatter calls + 2 Linear/BN/ReLU stacks + the canvas scatter of the reference
(pcdet/models/backbones_3d/vfe/dynamic_pillar_vfe.py:49-147).  Parameter names and shapes are the reference's
(vfe.pfn_layers.{i}.linear.weight, vfe.pfn_layers.{i}.norm.*), so published checkpoints load unchanged.
"""
import numpy as np
import torch
import torch.nn as nn

from pcp_amd import ops, pack

from .vfe_template import VFETemplate
from ...packed import train_tape


class PFNLayerV2(nn.Module):
    """Parameter container only (Linear without bias + BatchNorm1d eps 1e-3); the arithmetic runs inside k_pfn."""

    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        self.last_vfe = last_layer
        self.use_norm = use_norm
        if not last_layer:
            out_channels = out_channels // 2
        self.linear = nn.Linear(in_channels, out_channels, bias=not use_norm)
        if use_norm:
            self.norm = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)
        self.relu = nn.ReLU()

    def folded(self):
        w, b = self.linear.weight.detach(), self.linear.bias
        if self.use_norm:
            n = self.norm
            return pack.fold_bn(w, n.weight.detach(), n.bias.detach(), n.running_mean, n.running_var, n.eps)
        return w.float(), b.detach().float()


SPARSE_MAX_FILL = 0.35        # points per cell below which the first backbone layer runs from the pillar list (pipeline mode)


class DynamicPillarVFE(VFETemplate):
    def __init__(self, model_cfg, num_point_features, voxel_size, grid_size, point_cloud_range, **kwargs):
        super().__init__(model_cfg=model_cfg)
        if self.model_cfg.get('NUM_RAW_POINT_FEATURES', None) is not None:
            num_point_features = self.model_cfg.NUM_RAW_POINT_FEATURES
        self.num_raw_point_features = num_point_features
        self.use_norm = self.model_cfg.USE_NORM
        self.with_distance = self.model_cfg.WITH_DISTANCE
        self.use_absolute_xyz = self.model_cfg.USE_ABSLOTE_XYZ
        self.num_filters = list(self.model_cfg.NUM_FILTERS)
        assert len(self.num_filters) > 0
        in_dim = num_point_features + (6 if self.use_absolute_xyz else 3) + (1 if self.with_distance else 0)
        # the one-launch PFN (pcp_pfn_rows) is built for the composition all five configs use; any other runs layer by layer
        # (_forward_layers: pcp_pfn_features + pcp_pointwise + pcp_segment_max per PFNLayerV2)
        self.fused = (self.use_absolute_xyz and not self.with_distance and self.num_filters == [64, 64]
                      and num_point_features in (3, 4, 5, 11))
        if num_point_features < 3:
            raise NotImplementedError('a point row holds at least x, y, z')
        dims = [in_dim] + self.num_filters
        self.pfn_layers = nn.ModuleList([
            PFNLayerV2(dims[i], dims[i + 1], self.use_norm, last_layer=(i >= len(dims) - 2)) for i in range(len(dims) - 1)])
        self.voxel_x, self.voxel_y, self.voxel_z = voxel_size
        self.point_cloud_range = np.asarray(point_cloud_range, dtype=np.float32)
        self.voxel_size = [float(v) for v in voxel_size]
        self.grid_size = [int(v) for v in grid_size]
        self.scale_xy = self.grid_size[0] * self.grid_size[1]
        self.scale_y = self.grid_size[1]
        # knobs of the MI355X pipeline (not in the reference)
        self.materialize_pillars = True     # expose exact-shape pillar_features / voxel_coords (costs one host sync)
        self.reuse_buffers = False          # keep canvas + workspace across frames
        # pipeline mode only: hand the pillar list (pillar rows + the pillariser's cell -> rank table) to the backbone, whose first layer
        # then runs from it (pcp_sparse_conv3x3_s2) and NO dense canvas is written; `spatial_features` is None in that case.  Used when
        # the cloud is sparse enough (points <= SPARSE_MAX_FILL x cells) -- a crowded canvas is faster through the dense kernel.
        self.sparse_first_layer = False
        self.keep_bucket_order = False      # also leave the row indices grouped by pillar (HunterJr's point head visits the points in that order)
        self._pf_buf = None
        self._canvas = None
        self._workspace = None

    def _forward_train(self, batch_dict):
        if not self.fused:
            raise NotImplementedError('the HIP training path covers the PillarFeatureNet of the five configs (USE_ABSLOTE_XYZ, no '
                                      'WITH_DISTANCE, NUM_FILTERS [64, 64]); this variant has inference kernels only')
        from ...train_path import VFETrain
        if getattr(self, '_pcp_train', None) is None:
            self._pcp_train = VFETrain(self)
        self.invalidate_packed()
        batch_dict = self._pcp_train.forward(batch_dict)
        train_tape(batch_dict).append(('vfe', lambda g: self._pcp_train.backward(g.t)))
        return batch_dict

    def get_output_feature_dim(self):
        return self.num_filters[-1]

    def _build_packed(self):
        if not self.fused:
            layers = []
            cin_pad = (self.pfn_layers[0].linear.in_features + 15) // 16 * 16
            for layer in self.pfn_layers:
                w, b = layer.folded()
                cout = w.shape[0]
                wp = torch.zeros((cout, cin_pad), dtype=torch.float32, device=w.device)
                wp[:, :w.shape[1]] = w
                packed, bias, cout_pad = pack.pack_plain(wp, b)
                layers.append(dict(w=packed.contiguous(), b=bias.contiguous(), cin=cin_pad, cout=cout, cout_pad=cout_pad))
                cin_pad = (2 * cout + 15) // 16 * 16
            return dict(layers=layers)
        w0, b0 = self.pfn_layers[0].folded()
        w1, b1 = self.pfn_layers[1].folded()
        return dict(w0=w0.contiguous(), b0=b0.contiguous(), w1=w1.contiguous(), b1=b1.contiguous())

    def _pillarise(self, points, grid, batch_dict):
        """pcp_pillarise_rows on `points`: the rows in pillar order + the wave tiles of pcp_pfn_rows, in this VFE's workspace"""
        nr = self.num_raw_point_features
        self._vox_borrowed = False
        kw = dict(want_coords=self.materialize_pillars, bucket_order=self.keep_bucket_order)
        want_inv = self.materialize_pillars       # drop-in mode also leaves unq_inv (bit exact against the reference's digest in the tests)
        ready = batch_dict.get('_pcp_vox_ready', None)
        if ready is not None:
            return ops.pillarise_rows(points, grid, nr, workspace=ready['workspace'], cells_ready=True, **kw)
        share = batch_dict.get('_pcp_vox_share', None)
        if share is None or not self.reuse_buffers:
            return ops.pillarise_rows(points, grid, nr, workspace=self._workspace if self.reuse_buffers else None, want_inverse=want_inv, **kw)
        # Two VFEs of one forward that pillarise the SAME cloud on the SAME grid (DiscoNet: the early-fusion BEV maker and the ego branch
        # both see all points, bev_maker.py:212-230 / SURVEY F4) share one pillar list: the first to arrive builds it, the other waits for
        # its event and only runs its own PFN.  The producer rotates through THREE workspaces: in the pipelined mode
        # (pcdet/models/pipelined.py) the producer of forward i+1 may already run while the consumer of forward i still reads the list.
        key = (points.data_ptr(), int(points.shape[0]), int(points.shape[1]), nr, grid.nx, grid.ny, grid.batch_size, grid.min_x, grid.min_y, grid.min_z,
               grid.voxel_x, grid.voxel_y, grid.voxel_z)
        cur = torch.cuda.current_stream()
        ent = share.get(key)
        if ent is not None:
            vox, ev = ent
            cur.wait_event(ev)
            for t in (vox.workspace, vox.counters):
                t.record_stream(cur)
            if vox.voxel_coords is not None:
                vox.voxel_coords.record_stream(cur)
            self._vox_borrowed = True       # the producer VFE owns this workspace (one of its three ring slots): never adopt it
            return vox
        ring = getattr(self, '_ws_ring', None)
        if ring is None:
            ring = self._ws_ring = [None, None, None]
            self._ws_idx = -1
        self._ws_idx = (self._ws_idx + 1) % 3
        vox = ops.pillarise_rows(points, grid, nr, workspace=ring[self._ws_idx], **kw)
        ring[self._ws_idx] = vox.workspace
        share[key] = (vox, cur.record_event())
        return vox

    def _forward_layers(self, batch_dict, points, grid):
        """any PillarFeatureNet composition, one PFNLayerV2 at a time (reference :110-147); publishes pillar_features / voxel_coords like
        the reference, PointPillarScatter writes the canvas from them"""
        from pcp_amd import lib as _lib, train_ops as tops
        nr = self.num_raw_point_features
        vox = ops.voxelize(points, grid, want_inverse=False, want_counts=False)
        num_pillars, rows = (int(v) for v in vox.counters[:2].tolist())
        if rows == 0:
            batch_dict['voxel_features'] = batch_dict['pillar_features'] = points.new_zeros((0, self.num_filters[-1]))
            batch_dict['voxel_coords'] = vox.voxel_coords[:0]
            return batch_dict
        x, slot_pillar, _ = ops.pfn_features(points, vox, nr, self.use_absolute_xyz, self.with_distance)
        layers = self.packed()['layers']
        for i, lw in enumerate(layers):
            y = ops.pointwise(x[:rows], lw['w'], lw['b'], _lib.PW_PLAIN, lw['cin'], lw['cout'], lw['cout_pad'], relu=True)
            y_max, _ = tops.segment_max(y, slot_pillar, num_pillars, lw['cout'], rows=rows)
            if i == len(layers) - 1:
                break
            x = ops.pfn_cat_pillar_max(y, y_max, slot_pillar, rows, lw['cout'])
        batch_dict['voxel_features'] = batch_dict['pillar_features'] = y_max
        batch_dict['voxel_coords'] = vox.voxel_coords[:num_pillars]
        return batch_dict

    def forward(self, batch_dict, **kwargs):
        if self.training:
            return self._forward_train(batch_dict)
        points = batch_dict['points']
        if points.dtype != torch.float32 or not points.is_contiguous():
            points = points.float().contiguous()
        batch_size = batch_dict.get('batch_size', None)
        if batch_size is None:                      # BEVMaker-style sub dicts carry only 'points' (bev_maker.py:196)
            batch_size = int(points[:, 0].max().item()) + 1 if points.shape[0] else 1
        grid = ops.make_grid(self.point_cloud_range, self.voxel_size, self.grid_size, batch_size)
        if not self.fused:
            return self._forward_layers(batch_dict, points, grid)
        pk = self.packed()
        dev = points.device
        nx, ny = self.grid_size[0], self.grid_size[1]
        n_eff = batch_dict.get('_pcp_valid_points_hint', points.shape[0])
        sparse = (self.sparse_first_layer and not self.materialize_pillars and self.num_filters[-1] == 64
                  and n_eff <= SPARSE_MAX_FILL * batch_size * nx * ny)
        vox = self._pillarise(points, grid, batch_dict)
        if self.reuse_buffers and not self._vox_borrowed:
            self._workspace = vox.workspace
        rows = max(points.shape[0], 1)
        if sparse:
            # pillar rows only: the backbone's first layer gathers them through the cell -> rank table in the workspace
            if self._pf_buf is None or self._pf_buf.shape[0] < rows or self._pf_buf.device != dev or not self.reuse_buffers:
                self._pf_buf = torch.empty((rows, 64), dtype=torch.float32, device=dev)
            ops.pfn_rows(vox, pk['w0'], pk['b0'], pk['w1'], pk['b1'], canvas=None, pillar_features=self._pf_buf)
            batch_dict['_pcp_vfe'] = dict(canvas=None, vox=vox, pillar_rows=self._pf_buf)
            return batch_dict
        # dense canvas: pcp_pfn_rows writes every row of it (pillar rows and zero rows), so it is neither zero-filled nor cleared
        if self.reuse_buffers:
            if self._canvas is None or self._canvas.shape[0] != batch_size or self._canvas.device != dev:
                self._canvas = torch.empty((batch_size, ny, nx, 64), dtype=torch.float32, device=dev)
            canvas = self._canvas
        else:
            canvas = torch.empty((batch_size, ny, nx, 64), dtype=torch.float32, device=dev)
        pf = torch.empty((rows, 64), dtype=torch.float32, device=dev) if self.materialize_pillars else None
        ops.pfn_rows(vox, pk['w0'], pk['b0'], pk['w1'], pk['b1'], canvas=canvas, pillar_features=pf)
        if self.materialize_pillars:
            num_pillars = int(vox.counters[0].item())            # the one host sync of the drop-in mode
            batch_dict['voxel_features'] = batch_dict['pillar_features'] = pf[:num_pillars]
            batch_dict['voxel_coords'] = vox.voxel_coords[:num_pillars]
        batch_dict['_pcp_vfe'] = dict(canvas=canvas, vox=vox)
        return batch_dict
