"""HunterJr "aligner" corrector on gfx950 (reference: pcdet/models/bev_layers/hunter_jr.py:115-495 and hunter_toolbox.py).  Parameter tree
identical to the reference, including the training-only object_head (it exists in published checkpoints because the reference constructs
it whenever the module is built in training mode).  Training (object head, targets, the seven loss terms and the whole backward) runs in
pcdet/models/hunter_train_path.py on the kernels of csrc/hunter_train.hip; this file holds the module and the inference branch.

Forward: conv_input (MFMA 3x3, written into the first half of a 768-channel NHWC buffer) -> bilinear point sampling ->
point MLP (two fused Linear+BN+ReLU launches, residual add fused into the second) -> the three point heads as ONE
384->8 linear -> dynamic-foreground test + in-place xyz correction (reference quirk Q8: batch_dict['points'] is mutated) ->
re-sample only the corrected rows -> deterministic scatter-mean into the second half of the 768 buffer -> weightor convs
(768->768, 768->2) -> 2-way softmax blend.  No concat copy, no torch.unique, no host sync.
"""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from pcp_amd import lib, ops, pack

from ..convnet import PackedConv, pack_conv_module
from ..packed import PackedModule, require_eval_hip, train_tape


def conv_bn_relu(in_channels, out_channels, kernel_size=3, stride=1, padding=0, norm_layer=nn.BatchNorm2d):
    return nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False),
                         norm_layer(out_channels), nn.ReLU(inplace=True))


def nn_make_mlp(c_in, c_out, hidden_channels=None, is_head=True, use_drop_out=False):
    """Linear stacks with the reference's layer indices (hunter_toolbox.py:130-158)."""
    channels = [c_in] + list(hidden_channels or []) + [c_out]
    layers = []
    for i in range(1, len(channels)):
        if use_drop_out:
            layers.append(nn.Dropout(p=0.5))
        last = i == len(channels) - 1
        if last and is_head:
            layers.append(nn.Linear(channels[i - 1], channels[i], bias=True))
        else:
            layers += [nn.Linear(channels[i - 1], channels[i], bias=False), nn.BatchNorm1d(channels[i], eps=1e-3, momentum=0.01),
                       nn.ReLU(True)]
    return nn.Sequential(*layers)


class HunterObjectHead(nn.Module):
    """Training-only branch: parameters are kept for checkpoint compatibility (reference :22-76)."""

    def __init__(self, num_point_features, mlp_hidden_channels=None, use_drop_out=False):
        super().__init__()
        mk = partial(nn_make_mlp, hidden_channels=mlp_hidden_channels, use_drop_out=use_drop_out)
        self.num_local_feat = num_point_features
        self.points_shape_encoder = mk(3, num_point_features, is_head=False)
        self.local_feat_encoder = mk(2 * self.num_local_feat + 3 + 3, self.num_local_feat, is_head=False)
        self.local_tf_decoder = mk(self.num_local_feat, 7, hidden_channels=[])


class HunterPointHead(nn.Module):
    def __init__(self, num_point_features, mlp_hidden_channels=None, use_drop_out=False):
        super().__init__()
        mk = partial(nn_make_mlp, use_drop_out=use_drop_out)
        self.local_feat_predictor = mk(num_point_features, num_point_features, hidden_channels=mlp_hidden_channels, is_head=False)
        self.seg = mk(num_point_features, 3)
        self.reg_flow3d = mk(num_point_features, 3)
        self.instance_embedding = mk(num_point_features, 2)


def _pack_linear(linear, bn, relu):
    w = linear.weight.detach().float()
    if bn is not None:
        w, b = pack.fold_bn(w, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps,
                            conv_bias=linear.bias.detach().float() if linear.bias is not None else None)
    else:
        b = linear.bias.detach().float() if linear.bias is not None else torch.zeros(w.shape[0], device=w.device)
    pc = PackedConv()
    pc.kind, pc.relu, pc.stride, pc.wino = 'plain', relu, 1, None
    pc.cin, pc.cout = w.shape[1], w.shape[0]
    pc.w, pc.b, pc.cout_pad = pack.pack_plain(w, b)
    return pc


class HunterJr(PackedModule):
    def __init__(self, model_cfg, num_bev_features, voxel_size, point_cloud_range):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_sweeps = model_cfg.get('NUM_SWEEPS')
        self.bev_image_stride = model_cfg.get('BEV_IMAGE_STRIDE')
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = np.asarray(point_cloud_range, dtype=np.float32)
        self.num_points_feat = num_bev_features
        norm = partial(nn.BatchNorm2d, eps=1e-3, momentum=0.01)
        self.conv_input = conv_bn_relu(num_bev_features, num_bev_features, padding=1, norm_layer=norm)
        self.point_head = HunterPointHead(num_bev_features, list(model_cfg.get('POINT_HEAD_HIDDEN_CHANNELS')), use_drop_out=False)
        # built unconditionally: an nn.Module is in training mode while it is being constructed, so the reference
        # always creates it too (hunter_jr.py:139-142) and checkpoints contain its weights
        self.object_head = HunterObjectHead(num_bev_features, list(model_cfg.get('OBJ_HEAD_HIDDEN_CHANNELS')), use_drop_out=False)
        self.thresh_point_cls_prob = model_cfg.get('THRESHOLD_POINT_CLS_PROB', 0.3)
        self.meta_sweep_col = model_cfg.get('META_POINTS_FEAT_LOCATION_SWEEP_IDX', -2)
        self.meta_inst_col = model_cfg.get('META_POINTS_FEAT_LOCATION_INSTANCE_IDX', -1)
        self.sorted_gather = True          # MI355X knob: point-head gathers in spatially sorted order (output unchanged)
        self.conv_weightor = nn.Sequential(
            conv_bn_relu(2 * num_bev_features, 2 * num_bev_features, padding=1, norm_layer=norm),
            nn.Conv2d(2 * num_bev_features, 2, kernel_size=3, padding=1))
        self.forward_return_dict = dict()

    def _build_packed(self):
        ph = self.point_head
        lf = list(ph.local_feat_predictor)
        mlp = [_pack_linear(lf[i], lf[i + 1], relu=True) for i in range(0, len(lf), 3)]
        heads_w = torch.cat([ph.seg[0].weight, ph.reg_flow3d[0].weight, ph.instance_embedding[0].weight], 0).detach().float()
        heads_b = torch.cat([ph.seg[0].bias, ph.reg_flow3d[0].bias, ph.instance_embedding[0].bias], 0).detach().float()
        heads = PackedConv()
        heads.kind, heads.relu, heads.stride, heads.wino = 'plain', False, 1, None
        heads.cin, heads.cout = heads_w.shape[1], heads_w.shape[0]
        heads.w, heads.b, heads.cout_pad = pack.pack_plain(heads_w, heads_b)
        fused = None
        if len(lf) == 6 and heads_w.shape == (8, 384) and lf[0].weight.shape == (32, 384):
            # plain row-major folded weights for the fused point-head kernel
            f1 = pack.fold_bn(lf[0].weight.detach().float(), lf[1].weight.detach(), lf[1].bias.detach(), lf[1].running_mean,
                              lf[1].running_var, lf[1].eps)
            f2 = pack.fold_bn(lf[3].weight.detach().float(), lf[4].weight.detach(), lf[4].bias.detach(), lf[4].running_mean,
                              lf[4].running_var, lf[4].eps)
            fused = tuple(t.contiguous() for t in (f1[0], f1[1], f2[0], f2[1], heads_w, heads_b))
        # 768 -> 2 weight conv: N = 2 would waste 15/16 of a 32-wide MFMA tile -> small-N VALU kernel
        wlast = self.conv_weightor[1]
        small = None
        if wlast.weight.shape[0] <= 4 and wlast.weight.shape[1] % 64 == 0:
            small = (wlast.weight.detach().float().permute(0, 2, 3, 1).reshape(wlast.weight.shape[0], 9, -1).contiguous(),
                     wlast.bias.detach().float().contiguous(), [0, int(wlast.weight.shape[0])])
        return dict(fused=fused, w1_small=small, conv_input=pack_conv_module(self.conv_input[0], self.conv_input[1], relu=True), mlp=mlp, heads=heads,
                    w0=pack_conv_module(self.conv_weightor[0][0], self.conv_weightor[0][1], relu=True),
                    w1=pack_conv_module(self.conv_weightor[1], None, relu=False))

    def _forward_train(self, batch_dict):
        from ..hunter_train_path import HunterTrain
        if getattr(self, '_pcp_train', None) is None:
            self._pcp_train = HunterTrain(self)
        self.invalidate_packed()
        fused = self._pcp_train.forward(batch_dict)
        batch_dict.pop('spatial_features_2d')
        batch_dict['spatial_features_2d'] = ops.nchw_view(fused)
        train_tape(batch_dict).append(('corrector', self._pcp_train.backward))
        return batch_dict

    def forward(self, batch_dict):
        if self.training:
            return self._forward_train(batch_dict)
        require_eval_hip(self, 'HunterJr')
        pk = self.packed()
        points = batch_dict['points']
        x = ops.as_nhwc(batch_dict['spatial_features_2d'])
        B, H, W, C = x.shape
        dev = x.device
        cat = torch.empty((B, H, W, 2 * C), dtype=torch.float32, device=dev)      # [bev | corrected]
        pk['conv_input'].run(x, out=cat, out_ch_off=0)
        min_xy = self.point_cloud_range[:2]
        pix = [np.float32(self.voxel_size[0]) * self.bev_image_stride, np.float32(self.voxel_size[1]) * self.bev_image_stride]
        if pk['fused'] is not None and C == 384:
            # visit the points in the pillariser's bucket order (spatially sorted): the 4 x 384-float gathers of neighbouring
            # points then hit L2; results are written at the original rows, so the output does not change
            stash = batch_dict.get('_pcp_vfe', None)
            order = None
            if self.sorted_gather and stash is not None and stash['vox'].n == points.shape[0] and getattr(stash['vox'], 'has_bucket_order', True):
                order = ops.voxelize_row_order(stash['vox'])
            # one launch: sample -> MLP -> heads -> dynamic-foreground correction (in place) -> re-sampling of the corrected rows
            pf, head8, dyn = ops.hunter_point_head(cat, points, min_xy, pix, *pk['fused'], channels=C, order=order,
                                                   flow_thresh=self.thresh_point_cls_prob)
        else:
            pf = ops.bev_sample_bilinear(cat, points, min_xy, pix, channels=C)
            h = pf
            for i, layer in enumerate(pk['mlp']):
                last = i == len(pk['mlp']) - 1
                h = ops.pointwise(h, layer.w, layer.b, lib.PW_PLAIN, layer.cin, layer.cout, layer.cout_pad, relu=True,
                                  residual=pf if last else None)                      # final = pf + mlp(pf)
            head8 = pk['heads'].run(h)                                                # (N, 8) = cls(3) | flow(3) | embed(2)
            dyn = ops.hunter_apply_flow(points, head8, self.thresh_point_cls_prob)    # mutates points[:, 1:4] in place
            ops.bev_sample_bilinear(cat, points, min_xy, pix, out=pf, row_mask=dyn, channels=C)
        ops.bev_scatter_mean(points, pf, B, H, W, min_xy, pix, out=cat, out_ch_off=C)
        hid = pk['w0'].run(cat)
        if pk['w1_small'] is not None:
            wsm, bsm, offs = pk['w1_small']
            logits = torch.empty((B, H, W, offs[-1]), dtype=torch.float32, device=dev)
            ops.conv3x3_grouped_small(hid, wsm, bsm, offs, logits)
        else:
            logits = pk['w1'].run(hid)                                                # (B, H, W, 2)
        lbuf = logits                                                                 # ld 2: softmax_fuse reads columns 0..1
        fused = torch.empty((B, H, W, C), dtype=torch.float32, device=dev)
        # two "maps" that are the two channel halves of the same 768-wide buffer (pixel stride 2C)
        ops.softmax_fuse_raw([cat.data_ptr(), cat.data_ptr() + 4 * C], lbuf, C, 2 * C, fused)
        batch_dict.pop('spatial_features_2d')
        batch_dict['spatial_features_2d'] = ops.nchw_view(fused)
        if getattr(self, 'keep_point_heads', False):
            batch_dict['hunter_point_heads'] = head8                                  # (N, 8): consumers that stay on the device
        if self.model_cfg.get('GENERATING_EXCHANGE_DATA', False) or self.model_cfg.get('RETURN_SCENE_FLOW', False):
            self._emit_foreground(batch_dict, points, head8)
        return batch_dict

    def _emit_foreground(self, batch_dict, points, head8):
        """rows sent to the other agents (reference :377-397): points whose background probability is < 0.3, as
        [point features without the frame index (xyz already flow-corrected in place), sigmoid(cls)(3), flow(3)]; one compaction
        kernel instead of boolean-mask copies.  GENERATING_EXCHANGE_DATA writes one file per frame, RETURN_SCENE_FLOW keeps the
        reference's behaviour of leaving the LAST non-empty frame in batch_dict['scene_flow']."""
        rows, row_batch = ops.hunter_foreground_rows(points, head8, 0.3)
        batch_dict['hunter_point_heads'] = head8
        if rows.shape[0] == 0:
            return
        for b_idx, metadata in enumerate(batch_dict['metadata']):
            sample_points = rows[row_batch == b_idx]
            if sample_points.shape[0] == 0:
                continue
            if self.model_cfg.get('GENERATING_EXCHANGE_DATA', False):
                torch.save(sample_points, '%s/%s_id%s_foreground.pth' % (self.model_cfg.DATABASE_EXCHANGE_DATA, metadata['sample_token'],
                                                                         metadata['lidar_id']))
            else:
                batch_dict['scene_flow'] = sample_points

    def get_training_loss(self, tb_dict=None):
        """the seven loss terms of hunter_jr.py:401-495 AND their gradients w.r.t. the head outputs (consumed by loss.backward())"""
        tb_dict = {} if tb_dict is None else tb_dict
        out = self._pcp_train.losses()
        vals = out['losses'].tolist()
        for k, v in zip(('l_points_cls', 'l_points_embed', 'l_fg_offset', 'l_locals_transl', 'l_locals_rot', 'l_recon', 'l_dtl_locals_feat'), vals):
            tb_dict[k] = v
        st = self._pcp_train.s
        self.forward_return_dict.update(points_cls_target=out['labels'], meta=st['meta'], prediction={
            'points_cls_logit': st['head'][:, 0:3], 'points_flow3d': st['head'][:, 3:6], 'points_embedding': st['head'][:, 6:8],
            'locals_tf': st['locals_tf'][:, :7] if 'locals_tf' in st else None})
        return out['losses'][7], tb_dict
