"""AnchorHeadSingle on gfx950 (reference: pcdet/models/dense_heads/anchor_head_single.py:7-75, anchor_head_template.py:11-60,225-272,
target_assigner/anchor_generator.py:4-60, pcdet/utils/box_coder_utils.py:5-78) -- SURVEY 8(f) row 3, the "anchor head" of north_star.

Same parameter tree (conv_cls, conv_box, conv_dir_cls: 1x1 convs) so checkpoints load unchanged.  Inference forward:
  the three 1x1 convs            -> ONE MFMA pointwise launch (weights concatenated along cout) on the NHWC map
  ResidualCoder decode + direction classifier + sigmoid / max / score mask  -> pcp_anchor_decode (one elementwise pass)
  torch.topk(NMS_PRE_MAXSIZE) + gather                                       -> pcp_topk_boxes (radix select, one workgroup / frame)
  rotated NMS                                                                -> pcp_nms_rotated (candidates arrive sorted)
The anchors are generated once on the host with the reference's own arithmetic (torch.arange / meshgrid in float32).
Training forward (reference: anchor_head_template.py:89-216, target_assigner/axis_aligned_target_assigner.py:37-210):
  the three 1x1 convs            -> the same pointwise launch with the raw weights (train_path.AnchorHeadTrain keeps what backward needs)
  AxisAlignedTargetAssigner      -> pcp_anchor_assign_targets (no host loop over frames x anchor classes, no IoU matrix, no .nonzero())
  focal + smooth-L1(sin diff) + direction losses AND dL/d(head maps) -> pcp_anchor_loss (one pass)
"""
import numpy as np
import torch
import torch.nn as nn

from pcp_amd import lib, ops, pack

from ..packed import PackedModule, require_eval_hip, train_tape


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
        if self.training:
            return self._forward_train(data_dict)
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

    # ---- training (reference anchor_head_single.py:39-75, anchor_head_template.py:89-216) ---------------------------------------
    def _assign_desc(self, B, H, W):
        gcfg = self.model_cfg.ANCHOR_GENERATOR_CONFIG
        tcfg = self.model_cfg.TARGET_ASSIGNER_CONFIG
        if tcfg.NAME != 'AxisAlignedTargetAssigner':
            raise NotImplementedError('target assigner %s (the anchor YAMLs of the reference use AxisAlignedTargetAssigner)' % tcfg.NAME)
        if tcfg.POS_FRACTION >= 0 or tcfg.get('MATCH_HEIGHT', False) or tcfg.get('NORM_BY_NUM_EXAMPLES', False):
            raise NotImplementedError('POS_FRACTION >= 0 (random sampling), MATCH_HEIGHT and NORM_BY_NUM_EXAMPLES are not used by any '
                                      'anchor YAML of the reference and have no kernel')
        d = lib.AnchorAssign()
        d.batch, d.h, d.w = B, H, W
        d.anchors_per_loc, d.num_class, d.num_groups = self.num_anchors_per_location, len(self.class_names), len(gcfg)
        if d.anchors_per_loc > 32 or d.num_groups > 8:
            raise NotImplementedError('more than 32 anchors per location / 8 anchor classes')
        slot = 0
        names = list(self.class_names)
        for g, cfg in enumerate(gcfg):
            per = len(cfg['anchor_rotations']) * len(cfg['anchor_sizes']) * len(cfg['anchor_bottom_heights'])
            for _ in range(per):
                d.slot_group[slot] = g
                slot += 1
            d.group_class[g] = names.index(cfg['class_name']) if cfg['class_name'] in names else -1
            d.matched[g], d.unmatched[g] = float(cfg['matched_threshold']), float(cfg['unmatched_threshold'])
        assert slot == d.anchors_per_loc
        return d

    def _forward_train(self, data_dict):
        from pcp_amd import train_ops as tops
        from pcp_amd.train_layers import Act
        from ..train_path import AnchorHeadTrain
        if self.predict_boxes_when_training:
            raise NotImplementedError('predict_boxes_when_training needs a RoI head (not on the PointPillars path)')
        if getattr(self, '_pcp_train', None) is None:
            self._pcp_train = AnchorHeadTrain(self)
        self.invalidate_packed()
        x = ops.as_nhwc(data_dict['spatial_features_2d'])
        buf = self._pcp_train.forward(Act(x))
        B, H, W, ld = buf.shape
        a, ncls = self.num_anchors_per_location, self.num_class
        nbins = int(self.model_cfg.NUM_DIR_BINS) if self.conv_dir_cls is not None else 0
        view = ops.nchw_view(buf)
        self.forward_ret_dict['cls_preds'] = view[:, :a * ncls].permute(0, 2, 3, 1)
        self.forward_ret_dict['box_preds'] = view[:, a * ncls:a * ncls + a * 7].permute(0, 2, 3, 1)
        if nbins:
            self.forward_ret_dict['dir_cls_preds'] = view[:, a * ncls + a * 7:a * ncls + a * 7 + a * nbins].permute(0, 2, 3, 1)
        gt = data_dict['gt_boxes']
        if gt.dtype != torch.float32 or not gt.is_contiguous():
            gt = gt.float().contiguous()
        if gt.shape[-1] != 8:
            raise NotImplementedError('gt_boxes with velocity columns are not used by the V2X-Sim configs')
        anchors = self.flat_anchors(x.device)
        labels, reg_t, reg_w = tops.anchor_assign_targets(anchors, gt, self._assign_desc(B, H, W))
        self.forward_ret_dict.update(box_cls_labels=labels, box_reg_targets=reg_t, reg_weights=reg_w)
        self._train_state = dict(buf=buf, labels=labels, reg_t=reg_t, anchors=anchors, nbins=nbins)
        train_tape(data_dict).append(('dense_head', self._backward_from_loss))
        return data_dict

    def get_loss(self):
        """the three loss terms AND dL/d(head maps) in one pass (the gradient is consumed by loss.backward())"""
        from pcp_amd import train_ops as tops
        st = self._train_state
        buf = st['buf']
        B, H, W, ld = buf.shape
        a, ncls, nbins = self.num_anchors_per_location, self.num_class, st['nbins']
        lcfg = self.model_cfg.LOSS_CONFIG
        if lcfg.get('REG_LOSS_TYPE', None) not in (None, 'WeightedSmoothL1Loss'):
            raise NotImplementedError('REG_LOSS_TYPE %s' % lcfg.REG_LOSS_TYPE)
        lw = lcfg.LOSS_WEIGHTS
        d = lib.AnchorLoss()
        d.batch, d.h, d.w, d.ld, d.ld_d = B, H, W, ld, ld
        d.anchors_per_loc, d.num_class, d.num_dir_bins = a, ncls, nbins
        d.ch_cls, d.ch_box, d.ch_dir = 0, a * ncls, a * ncls + a * 7
        if nbins:
            d.dir_offset, d.dir_period = float(self.model_cfg.DIR_OFFSET), float(2 * np.pi / nbins)
            d.dir_weight = float(lw['dir_weight'])
        d.cls_weight, d.loc_weight = float(lw['cls_weight']), float(lw['loc_weight'])
        for j in range(7):
            d.code_weights[j] = float(lw['code_weights'][j])
        dhead = torch.empty_like(buf)
        losses = tops.anchor_loss(buf, st['anchors'], st['labels'], st['reg_t'], d, dhead=dhead)
        st['dhead'] = dhead
        vals = losses.tolist()
        tb_dict = {'rpn_loss_cls': vals[0], 'rpn_loss_loc': vals[1]}
        if nbins:
            tb_dict['rpn_loss_dir'] = vals[2]
        tb_dict['rpn_loss'] = vals[3]
        return losses[3], tb_dict

    def _backward_from_loss(self, _unused):
        return self._pcp_train.backward(self._train_state['dhead'])
