"""AnchorHeadSingle on gfx950 (reference: pcdet/models/dense_heads/anchor_head_single.py:7-75, anchor_head_template.py:11-60,225-272,
target_assigner/anchor_generator.py:4-60, pcdet/utils/box_coder_utils.py:5-78) -- SURVEY 8(f) row 3, the "anchor head" of north_star.

Same parameter tree (conv_cls, conv_box, conv_dir_cls: 1x1 convs) so checkpoints load unchanged.  Inference forward:
  the three 1x1 convs            -> ONE MFMA pointwise launch (weights concatenated along cout) on the NHWC map
  ResidualCoder decode + direction classifier + sigmoid / max / score mask  -> pcp_anchor_decode (one elementwise pass)
  torch.topk(NMS_PRE_MAXSIZE) + gather                                       -> pcp_topk_boxes (radix select, one workgroup / frame)
  rotated NMS                                                                -> pcp_nms_rotated (candidates arrive sorted)
The anchors are generated once on the host with the reference's own arithmetic (torch.arange / meshgrid in float32).
Training (AxisAlignedTargetAssigner + focal / smooth-L1 / direction losses) is not built: no V2X-Sim config uses this head.
"""
import numpy as np
import torch
import torch.nn as nn

from pcp_amd import lib, ops, pack

from ..packed import PackedModule, require_eval_hip


class ResidualCoder:
    """box_coder_utils.py:5-12: only the code size is needed on the host; decode_torch runs inside pcp_anchor_decode"""

    def __init__(self, code_size=7, encode_angle_by_sincos=False, **kwargs):
        if encode_angle_by_sincos:
            raise NotImplementedError('encode_angle_by_sincos is not used by any config of the reference')
        self.code_size = code_size


def generate_anchors(anchor_generator_cfg, grid_size, point_cloud_range):
    """AnchorGenerator.generate_anchors (anchor_generator.py:17-60) on the CPU: list of (1, ny, nx, n_size, n_rot, 7) tensors."""
    all_anchors, per_loc = [], []
    rng = np.asarray(point_cloud_range, dtype=np.float32)     # float32 numpy scalars, like the reference's anchor_range
    for cfg in anchor_generator_cfg:
        fm = np.asarray(grid_size[:2], dtype=np.int64) // int(cfg['feature_map_stride'])      # numpy int64: float32 / int64 -> float64 strides
        sizes, rots, heights = cfg['anchor_sizes'], cfg['anchor_rotations'], cfg['anchor_bottom_heights']
        per_loc.append(len(rots) * len(sizes) * len(heights))
        if cfg.get('align_center', False):
            xs, ys = (rng[3] - rng[0]) / fm[0], (rng[4] - rng[1]) / fm[1]
            xo, yo = xs / 2, ys / 2
        else:
            xs, ys = (rng[3] - rng[0]) / (fm[0] - 1), (rng[4] - rng[1]) / (fm[1] - 1)
            xo, yo = 0, 0
        x_shifts = torch.arange(rng[0] + xo, rng[3] + 1e-5, step=xs, dtype=torch.float32)
        y_shifts = torch.arange(rng[1] + yo, rng[4] + 1e-5, step=ys, dtype=torch.float32)
        z_shifts = x_shifts.new_tensor(heights)
        n_size, n_rot = len(sizes), len(rots)
        rot_t, size_t = x_shifts.new_tensor(rots), x_shifts.new_tensor(sizes)
        xg, yg, zg = torch.meshgrid([x_shifts, y_shifts, z_shifts], indexing='ij')
        anchors = torch.stack((xg, yg, zg), dim=-1)
        anchors = anchors[:, :, :, None, :].repeat(1, 1, 1, size_t.shape[0], 1)
        size_r = size_t.view(1, 1, 1, -1, 3).repeat([*anchors.shape[0:3], 1, 1])
        anchors = torch.cat((anchors, size_r), dim=-1)
        anchors = anchors[:, :, :, :, None, :].repeat(1, 1, 1, 1, n_rot, 1)
        rot_r = rot_t.view(1, 1, 1, 1, -1, 1).repeat([*anchors.shape[0:3], n_size, 1, 1])
        anchors = torch.cat((anchors, rot_r), dim=-1)
        anchors = anchors.permute(2, 1, 0, 3, 4, 5).contiguous()
        anchors[..., 2] += anchors[..., 5] / 2
        all_anchors.append(anchors)
    return all_anchors, per_loc


class AnchorHeadSingle(PackedModule):
    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range, predict_boxes_when_training=True,
                 **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.class_names = class_names
        self.predict_boxes_when_training = predict_boxes_when_training
        if self.model_cfg.get('USE_MULTIHEAD', False):
            raise NotImplementedError('USE_MULTIHEAD belongs to AnchorHeadMulti (outside the PointPillars path)')
        tcfg = self.model_cfg.TARGET_ASSIGNER_CONFIG
        if tcfg.BOX_CODER != 'ResidualCoder':
            raise NotImplementedError('box coder %s' % tcfg.BOX_CODER)
        self.box_coder = ResidualCoder(**dict(tcfg.get('BOX_CODER_CONFIG', {})))
        anchors, per_loc = generate_anchors(self.model_cfg.ANCHOR_GENERATOR_CONFIG, grid_size, point_cloud_range)
        self.anchors = anchors                                         # list, like the reference attribute (CPU until first forward)
        self.num_anchors_per_location = sum(per_loc)
        a = self.num_anchors_per_location
        self.conv_cls = nn.Conv2d(input_channels, a * self.num_class, kernel_size=1)
        self.conv_box = nn.Conv2d(input_channels, a * self.box_coder.code_size, kernel_size=1)
        if self.model_cfg.get('USE_DIRECTION_CLASSIFIER', None) is not None:
            self.conv_dir_cls = nn.Conv2d(input_channels, a * self.model_cfg.NUM_DIR_BINS, kernel_size=1)
        else:
            self.conv_dir_cls = None
        nn.init.constant_(self.conv_cls.bias, -np.log((1 - 0.01) / 0.01))
        nn.init.normal_(self.conv_box.weight, mean=0, std=0.001)
        self.forward_ret_dict = {}
        self._anchors_flat = None

    def _build_packed(self):
        convs = [self.conv_cls, self.conv_box] + ([self.conv_dir_cls] if self.conv_dir_cls is not None else [])
        w = torch.cat([c.weight.detach().float().reshape(c.weight.shape[0], -1) for c in convs], 0)
        b = torch.cat([c.bias.detach().float() for c in convs], 0)
        packed, bias, cout_pad = pack.pack_plain(w, b)
        return dict(w=packed, b=bias, cout=w.shape[0], cout_pad=cout_pad, cin=w.shape[1])

    def flat_anchors(self, device):
        if self._anchors_flat is None or self._anchors_flat.device != device:
            self._anchors_flat = torch.cat(self.anchors, dim=-3).reshape(-1, 7).contiguous().to(device)
        return self._anchors_flat

    def forward(self, data_dict):
        require_eval_hip(self, 'AnchorHeadSingle')
        pk = self.packed()
        x = ops.as_nhwc(data_dict['spatial_features_2d'])
        B, H, W, _ = x.shape
        a, ncls = self.num_anchors_per_location, self.num_class
        nbins = int(self.model_cfg.NUM_DIR_BINS) if self.conv_dir_cls is not None else 0
        ld = (pk['cout'] + 3) // 4 * 4
        head = torch.empty((B, H, W, ld), dtype=torch.float32, device=x.device)
        ops.pointwise(x, pk['w'], pk['b'], lib.PW_PLAIN, pk['cin'], pk['cout'], pk['cout_pad'], relu=False, out=head)
        d = lib.Anchor()
        d.batch, d.h, d.w, d.ld = B, H, W, ld
        d.anchors_per_loc, d.num_class, d.num_dir_bins = a, ncls, nbins
        d.ch_cls, d.ch_box, d.ch_dir = 0, a * ncls, a * ncls + a * 7
        if nbins:
            d.dir_offset, d.dir_limit_offset = float(self.model_cfg.DIR_OFFSET), float(self.model_cfg.DIR_LIMIT_OFFSET)
            d.dir_period = float(2 * np.pi / nbins)
        st = data_dict.get('_pcp_score_thresh', None)
        d.use_score_thresh = 0 if st is None else 1
        d.score_thresh = 0.0 if st is None else float(st)
        boxes, cls, keys, labels = ops.anchor_decode(head, self.flat_anchors(x.device), d)
        view = ops.nchw_view(head)
        self.forward_ret_dict['cls_preds'] = view[:, :a * ncls].permute(0, 2, 3, 1)
        self.forward_ret_dict['box_preds'] = view[:, a * ncls:a * ncls + a * 7].permute(0, 2, 3, 1)
        if nbins:
            self.forward_ret_dict['dir_cls_preds'] = view[:, a * ncls + a * 7:a * ncls + a * 7 + a * nbins].permute(0, 2, 3, 1)
        data_dict['batch_cls_preds'] = cls
        data_dict['batch_box_preds'] = boxes
        data_dict['cls_preds_normalized'] = False
        data_dict['_pcp_anchor'] = dict(keys=keys, labels=labels)
        return data_dict

    def get_loss(self):
        raise NotImplementedError('AnchorHeadSingle training is not built (no V2X-Sim config uses this head)')
