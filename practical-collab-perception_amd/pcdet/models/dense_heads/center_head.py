"""CenterHead on gfx950 (reference: pcdet/models/dense_heads/center_head.py:13-429).

Parameter tree identical to the reference (shared_conv.{0,1}, heads_list.{h}.{name}.{i}.{0,1} / .{last}), so checkpoints load
unchanged.  Inference forward:
  shared 3x3 conv+BN+ReLU            -> one MFMA launch
  all first convs of all branches    -> ONE launch (weights concatenated along cout: 5 x 64 = 320 channels)
  all final convs                    -> ONE launch (block-diagonal weights, 9 real output channels)
  sigmoid/exp/atan2 + top-K + decode -> one kernel per head (pcp_centerhead_decode), no host sync
  rotated NMS                        -> sort-free (already ordered) mask + single-wavefront greedy, per frame
  one device->host copy of the per-frame keep counts at the very end to size the returned tensors.
"""
import copy

import numpy as np
import torch
import torch.nn as nn
from torch.nn.init import kaiming_normal_

from pcp_amd import ops

from ..convnet import pack_conv_module, pack_conv_raw, _fold
from ..packed import PackedModule, train_tape


class SeparateHead(nn.Module):
    """Parameter container for the per-quantity branches (reference :13-47)."""

    def __init__(self, input_channels, sep_head_dict, init_bias=-2.19, use_bias=False):
        super().__init__()
        self.sep_head_dict = sep_head_dict
        for cur_name in self.sep_head_dict:
            output_channels = self.sep_head_dict[cur_name]['out_channels']
            num_conv = self.sep_head_dict[cur_name]['num_conv']
            fc_list = []
            for _ in range(num_conv - 1):
                fc_list.append(nn.Sequential(
                    nn.Conv2d(input_channels, input_channels, kernel_size=3, stride=1, padding=1, bias=use_bias),
                    nn.BatchNorm2d(input_channels), nn.ReLU()))
            fc_list.append(nn.Conv2d(input_channels, output_channels, kernel_size=3, stride=1, padding=1, bias=True))
            fc = nn.Sequential(*fc_list)
            if 'hm' in cur_name:
                fc[-1].bias.data.fill_(init_bias)
            else:
                for m in fc.modules():
                    if isinstance(m, nn.Conv2d):
                        kaiming_normal_(m.weight.data)
                        if m.bias is not None:
                            nn.init.constant_(m.bias, 0)
            self.__setattr__(cur_name, fc)


class CenterHead(PackedModule):
    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range, voxel_size,
                 predict_boxes_when_training=True):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.grid_size = grid_size
        self.point_cloud_range = np.asarray(point_cloud_range, dtype=np.float32)
        self.voxel_size = voxel_size
        self.feature_map_stride = self.model_cfg.TARGET_ASSIGNER_CONFIG.get('FEATURE_MAP_STRIDE', None)
        self.class_names = class_names
        self.class_names_each_head = []
        self.class_id_mapping_each_head = []
        for cur_class_names in self.model_cfg.CLASS_NAMES_EACH_HEAD:
            self.class_names_each_head.append([x for x in cur_class_names if x in class_names])
            self.class_id_mapping_each_head.append(
                torch.tensor([self.class_names.index(x) for x in cur_class_names if x in class_names], dtype=torch.long))
        total_classes = sum(len(x) for x in self.class_names_each_head)
        assert total_classes == len(self.class_names), 'class_names_each_head=%s' % self.class_names_each_head
        use_bias = self.model_cfg.get('USE_BIAS_BEFORE_NORM', False)
        self.shared_conv = nn.Sequential(
            nn.Conv2d(input_channels, self.model_cfg.SHARED_CONV_CHANNEL, 3, stride=1, padding=1, bias=use_bias),
            nn.BatchNorm2d(self.model_cfg.SHARED_CONV_CHANNEL), nn.ReLU())
        self.heads_list = nn.ModuleList()
        self.separate_head_cfg = self.model_cfg.SEPARATE_HEAD_CFG
        self.head_names = []
        for cur_class_names in self.class_names_each_head:
            cur_head_dict = copy.deepcopy(dict(self.separate_head_cfg.HEAD_DICT))
            cur_head_dict = {k: dict(v) for k, v in cur_head_dict.items()}
            cur_head_dict['hm'] = dict(out_channels=len(cur_class_names), num_conv=self.model_cfg.NUM_HM_CONV)
            self.head_names.append(list(cur_head_dict.keys()))
            self.heads_list.append(SeparateHead(self.model_cfg.SHARED_CONV_CHANNEL, cur_head_dict, init_bias=-2.19, use_bias=use_bias))
        self.predict_boxes_when_training = predict_boxes_when_training
        self.forward_ret_dict = {}

    # ---- weight preparation -------------------------------------------------------------------------------------------
    def _build_packed(self):
        pk = dict(shared=pack_conv_module(self.shared_conv[0], self.shared_conv[1], relu=True), heads=[])
        c = self.model_cfg.SHARED_CONV_CHANNEL
        for head, names in zip(self.heads_list, self.head_names):
            seqs = [getattr(head, n) for n in names]
            outs = [head.sep_head_dict[n]['out_channels'] for n in names]
            offs = np.concatenate([[0], np.cumsum(outs)]).astype(int)
            entry = dict(names=names, outs=outs, offs=offs, total=int(offs[-1]))
            if all(len(s) == 2 for s in seqs):
                # stage 1: every branch's conv+BN+ReLU reads the same shared map -> concatenate along cout
                ws, bs = zip(*[_fold(s[0][0], s[0][1], out_axis=0) for s in seqs])
                entry['stage1'] = pack_conv_raw(torch.cat(ws, 0), torch.cat(bs, 0), relu=True)
                # stage 2: branch i maps its own 64 channels to outs[i] channels -> block-diagonal dense weight
                w2 = ws[0].new_zeros((int(offs[-1]), c * len(seqs), 3, 3))
                b2 = ws[0].new_zeros((int(offs[-1]),))
                for i, s in enumerate(seqs):
                    w2[offs[i]:offs[i + 1], c * i:c * (i + 1)] = s[1].weight.detach().float()
                    b2[offs[i]:offs[i + 1]] = s[1].bias.detach().float()
                entry['stage2'] = pack_conv_raw(w2, b2, relu=False)
                if c == 64 and max(outs) <= 4 and len(seqs) <= 8:
                    # grouped small-N VALU kernel: (n_out, 9, 64) weights, no zero blocks, no 32-wide MFMA padding
                    wg = torch.cat([s[1].weight.detach().float() for s in seqs], 0)          # (n_out, 64, 3, 3)
                    entry['stage2_grouped'] = (wg.permute(0, 2, 3, 1).reshape(wg.shape[0], 9, 64).contiguous(), b2.contiguous(),
                                               [int(v) for v in offs])
            else:
                entry['branches'] = []
                for s in seqs:
                    mods = list(s)
                    chain = [pack_conv_module(m[0], m[1], relu=True) for m in mods[:-1]]
                    chain.append(pack_conv_module(mods[-1], None, relu=False))
                    entry['branches'].append(chain)
            pk['heads'].append(entry)
        return pk

    # ---- forward ----------------------------------------------------------------------------------------------------------
    def _run_head_convs(self, x, entry):
        B, H, W, _ = x.shape
        ld = max(16, (entry['total'] + 3) // 4 * 4)
        buf = torch.zeros((B, H, W, ld), dtype=torch.float32, device=x.device)
        if 'stage1' in entry:
            mid = entry['stage1'].run(x)
            if mid.dtype != torch.float32:          # opt-in bf16 arithmetic (PCP_CONV_ALGO=bf16): the grouped final convs read fp32
                mid = mid.float()
            if 'stage2_grouped' in entry:
                wg, bg, goffs = entry['stage2_grouped']
                ops.conv3x3_grouped_small(mid, wg, bg, goffs, buf)
            else:
                entry['stage2'].run(mid, out=buf)
        else:
            for chain, off in zip(entry['branches'], entry['offs'][:-1]):
                y = x
                for conv in chain[:-1]:
                    y = conv.run(y)
                chain[-1].run(y, out=buf, out_ch_off=int(off))
        return buf

    def _decode_kwargs(self, entry):
        pp = self.model_cfg.POST_PROCESSING
        off = {n: int(o) for n, o in zip(entry['names'], entry['offs'][:-1])}
        for need in ('center', 'center_z', 'dim', 'rot', 'hm'):
            assert need in off, 'CenterHead decode kernel needs the %s branch' % need
        assert 'vel' not in off and 'iou' not in off, 'vel / iou branches are not used by the five configs'
        return dict(k=pp.MAX_OBJ_PER_SAMPLE, num_class=entry['outs'][entry['names'].index('hm')], ch_center=off['center'],
                    ch_z=off['center_z'], ch_dim=off['dim'], ch_rot=off['rot'], ch_hm=off['hm'],
                    stride=float(self.feature_map_stride), voxel_x=float(np.float32(self.voxel_size[0])),
                    voxel_y=float(np.float32(self.voxel_size[1])), min_x=float(self.point_cloud_range[0]),
                    min_y=float(self.point_cloud_range[1]), limit=list(pp.POST_CENTER_LIMIT_RANGE),
                    score_thresh=pp.SCORE_THRESH)

    def device_postprocess(self, head_bufs, pk):
        """decode + NMS for every head, everything left on the device (hipGraph-capturable: no host sync)."""
        nms_cfg = self.model_cfg.POST_PROCESSING.NMS_CONFIG
        assert nms_cfg.NMS_TYPE == 'nms_gpu', 'only the rotated nms_gpu of the five configs is built'
        per_head = []
        for idx, (buf, entry) in enumerate(zip(head_bufs, pk['heads'])):
            boxes, scores, labels, _cell, count = ops.centerhead_decode(buf, self._decode_kwargs(entry))
            k = boxes.shape[1]
            # candidates are already in descending score order: scores=None skips the device sort; all frames in one call
            keep, kcnt = ops.nms_rotated(boxes, None, nms_cfg.NMS_THRESH, min(nms_cfg.NMS_PRE_MAXSIZE, k), nms_cfg.NMS_POST_MAXSIZE,
                                         n_dev=count)
            per_head.append((boxes, scores, labels, keep, kcnt, idx))
        return per_head

    def finalize(self, per_head, batch_size):
        """one gather launch for all frames and heads (boxes[keep], scores[keep], class_id_mapping[labels[keep]] + 1, concatenated over
        heads: reference :335-357), then the one host sync of the path: how many boxes survive per frame -> exact-shape views."""
        ob, os_, ol, cnt = self.gather_pending(per_head, batch_size)
        counts = cnt.cpu().numpy()
        return [dict(pred_boxes=ob[b, :int(counts[b])], pred_scores=os_[b, :int(counts[b])], pred_labels=ol[b, :int(counts[b])])
                for b in range(batch_size)]

    def gather_pending(self, per_head, batch_size):
        """the gather launch of finalize() WITHOUT its host read: padded (B, M, 7) boxes, (B, M) scores, (B, M) int64 1-based labels and the
        (B,) int32 counts, all on the device (consumers that stay on the device: pcdet/models/lately_chain.py)"""
        heads = []
        for boxes, scores, labels, keep, kcnt, idx in per_head:
            cmap = self._class_maps.get(idx) if hasattr(self, '_class_maps') else None
            if cmap is None or cmap.device != boxes.device:
                if not hasattr(self, '_class_maps'):
                    self._class_maps = {}
                cmap = self.class_id_mapping_each_head[idx].to(device=boxes.device, dtype=torch.int32).contiguous()
                self._class_maps[idx] = cmap
            heads.append(dict(boxes=boxes, scores=scores, labels=labels, keep=keep, keep_count=kcnt, class_map=cmap))
        return ops.gather_detections(heads, batch_size)

    def generate_predicted_boxes(self, batch_size, head_bufs, pk):
        return self.finalize(self.device_postprocess(head_bufs, pk), batch_size)

    def forward(self, data_dict):
        if self.training:
            return self._forward_train(data_dict)
        pk = self.packed()
        x = ops.as_nhwc(data_dict['spatial_features_2d'])
        x = pk['shared'].run(x)
        head_bufs, pred_dicts = [], []
        for entry in pk['heads']:
            buf = self._run_head_convs(x, entry)
            head_bufs.append(buf)
            view = ops.nchw_view(buf)
            pred_dicts.append({n: view[:, int(entry['offs'][i]):int(entry['offs'][i + 1])] for i, n in enumerate(entry['names'])})
        self.forward_ret_dict['pred_dicts'] = pred_dicts
        if getattr(self, 'defer_finalize', False):                   # graph capture: keep the host sync outside
            data_dict['_pcp_pending_head'] = self.device_postprocess(head_bufs, pk)
            return data_dict
        final = self.generate_predicted_boxes(data_dict['batch_size'], head_bufs, pk)
        data_dict['final_box_dicts'] = final
        if self.model_cfg.get('GENERATING_EXCHANGE_DATA', False) or self.model_cfg.get('RETURN_MODAR_POINTS', False):
            # MoDAR "points": the wire / on-disk format that feeds config 3 (reference :409-427)
            for batch_idx, pred in enumerate(final):
                if pred['pred_boxes'].shape[0] == 0:
                    continue
                mo_pts = torch.cat([pred['pred_boxes'], pred['pred_scores'].reshape(-1, 1), pred['pred_labels'].reshape(-1, 1).float()], dim=1)
                if self.model_cfg.get('GENERATING_EXCHANGE_DATA', False):
                    metadata = data_dict['metadata'][batch_idx]
                    torch.save(mo_pts, '%s/%s_id%s_modar.pth' % (self.model_cfg.DATABASE_EXCHANGE_DATA, metadata['sample_token'],
                                                                 metadata['lidar_id']))
                else:
                    data_dict['mo_pts'] = mo_pts
        return data_dict

    # ---- training (reference :104-300, 377-392) -------------------------------------------------------------------------
    def _head_channels(self):
        names = self.head_names[0]
        outs = [self.heads_list[0].sep_head_dict[n]['out_channels'] for n in names]
        offs = np.concatenate([[0], np.cumsum(outs)]).astype(int)
        off = {n: int(o) for n, o in zip(names, offs[:-1])}
        order = list(self.separate_head_cfg.HEAD_ORDER)
        reg = []
        for n in order:
            reg += [off[n] + j for j in range(outs[names.index(n)])]
        if len(reg) != 8 or 'hm' not in off:
            raise NotImplementedError('loss kernel covers the 8 regression codes center/center_z/dim/rot + hm (all five configs)')
        return off, outs, reg, names

    def _forward_train(self, data_dict):
        from pcp_amd import lib
        from pcp_amd import train_ops as tops
        from pcp_amd.train_layers import Act
        from ..train_path import HeadTrain
        if self.predict_boxes_when_training:
            raise NotImplementedError('predict_boxes_when_training needs a RoI head (not on the PointPillars path)')
        if getattr(self, '_pcp_train', None) is None:
            self._pcp_train = HeadTrain(self)
        self.invalidate_packed()
        x = ops.as_nhwc(data_dict['spatial_features_2d'])
        buf = self._pcp_train.forward(Act(x))
        B, H, W, ld = buf.shape
        off, outs, reg, names = self._head_channels()
        view = ops.nchw_view(buf)
        offs = np.concatenate([[0], np.cumsum(outs)]).astype(int)
        self.forward_ret_dict['pred_dicts'] = [{n: view[:, int(offs[i]):int(offs[i + 1])] for i, n in enumerate(names)}]
        ta = self.model_cfg.TARGET_ASSIGNER_CONFIG
        gt = data_dict['gt_boxes']
        if gt.dtype != torch.float32 or not gt.is_contiguous():
            gt = gt.float().contiguous()
        if gt.shape[-1] != 8:
            raise NotImplementedError('gt_boxes with velocity columns are not used by the V2X-Sim configs')
        ncls = outs[names.index('hm')]
        tdesc = lib.Target(B, H, W, ncls, int(ta.NUM_MAX_OBJS), float(ta.FEATURE_MAP_STRIDE), float(np.float32(self.voxel_size[0])),
                           float(np.float32(self.voxel_size[1])), float(self.point_cloud_range[0]), float(self.point_cloud_range[1]),
                           float(ta.GAUSSIAN_OVERLAP), int(ta.MIN_RADIUS))
        heat, tb, inds, mask = tops.centerhead_targets(gt, tdesc)
        self.forward_ret_dict['target_dicts'] = {'heatmaps': [ops.nchw_view(heat)], 'target_boxes': [tb], 'inds': [inds.long()],
                                                 'masks': [mask.long()]}
        self._train_state = dict(buf=buf, heat=heat, tb=tb, inds=inds, mask=mask, off=off, reg=reg, ncls=ncls)
        train_tape(data_dict).append(('dense_head', self._backward_from_loss))
        return data_dict

    def get_loss(self, read_back=True):
        """focal + L1 losses AND dL/d(head maps) in one pass (the gradient is consumed by loss.backward()).
        read_back=False: the tb_dict values stay device scalars (CenterPoint.eager_backward reads them after queuing the backward)."""
        from pcp_amd import lib
        from pcp_amd import train_ops as tops
        st = self._train_state
        buf = st['buf']
        B, H, W, ld = buf.shape
        lw = self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        d = lib.HeadLoss()
        d.batch, d.h, d.w, d.ld, d.ld_d = B, H, W, ld, ld
        d.num_class, d.ch_hm = st['ncls'], st['off']['hm']
        for j in range(8):
            d.reg_ch[j] = st['reg'][j]
            d.code_weights[j] = float(lw['code_weights'][j])
        d.k = int(self.model_cfg.TARGET_ASSIGNER_CONFIG.NUM_MAX_OBJS)
        d.cls_weight, d.loc_weight = float(lw['cls_weight']), float(lw['loc_weight'])
        dhead = torch.empty_like(buf)
        losses = tops.centerhead_loss(buf, d, st['heat'], st['tb'], st['inds'], st['mask'], dhead=dhead)
        st['dhead'] = dhead
        vals = losses.tolist() if read_back else losses
        tb_dict = {'hm_loss_head_0': vals[0], 'loc_loss_head_0': vals[1], 'rpn_loss': vals[2]}
        return losses[2], tb_dict

    def _backward_from_loss(self, _unused):
        return self._pcp_train.backward(self._train_state['dhead'])
