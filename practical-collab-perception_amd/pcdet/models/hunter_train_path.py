"""Train-mode execution of HunterJr (configs 1 / 2), forward AND backward, on the kernels of include/pcp_hip_train.h section a17.

Reference: pcdet/models/bev_layers/hunter_jr.py:289-375 (forward, training branch), :22-113 (object / point heads), :165-287 (_build_meta,
assign_target, correct_bev_image), :401-495 (get_training_loss).  No torch.autograd inside: `forward` keeps what `backward` needs,
`losses()` computes the seven loss terms together with their gradients w.r.t. the head outputs, `backward(dfused)` walks the graph in reverse

    CenterHead <- fused = bev * w0 + corrected * w1 <- weightor convs <- [bev | corrected]
    corrected <- per-cell mean of the point features <- (rows the flow head moved: re-sampled at the new position, which carries a gradient
                 into the flow head through the bilinear weights; other rows: the first sampling)
    point heads, point MLP, object head (segment max routing) <- first sampling <- bev <- conv_input <- backbone

and deposits parameter gradients in param.grad.  One host read (three counts) after the locals / instances are built.
"""
import numpy as np
import torch

from pcp_amd import lib, ops, pack
from pcp_amd import train_layers as tl
from pcp_amd import train_ops as tops
from pcp_amd.train_layers import Act, ConvBNAct, LinearBNAct, MultiLinear, ensure_grad


def _mlp(seq, name):
    mods = list(seq)
    return [LinearBNAct(mods[i], mods[i + 1], relu=True, name='%s.%d' % (name, i)) for i in range(0, len(mods), 3)]


class HunterTrain:
    def __init__(self, m):
        self.m = m
        ci = m.conv_input
        self.conv_input = ConvBNAct(ci[0], ci[1], relu=True, name='corrector.conv_input')
        ph, oh = m.point_head, m.object_head
        self.mlp = _mlp(ph.local_feat_predictor, 'point_head.local_feat_predictor')
        self.heads = MultiLinear([ph.seg[0], ph.reg_flow3d[0], ph.instance_embedding[0]], name='point_head.heads')
        self.shape_enc = _mlp(oh.points_shape_encoder, 'object_head.points_shape_encoder')
        self.local_enc = _mlp(oh.local_feat_encoder, 'object_head.local_feat_encoder')
        self.tf_dec = MultiLinear([oh.local_tf_decoder[0]], name='object_head.local_tf_decoder')
        w = m.conv_weightor
        self.w0 = ConvBNAct(w[0][0], w[0][1], relu=True, name='corrector.conv_weightor.0')
        self.w1 = w[1]
        if self.w1.weight.shape[0] != 2 or self.w1.weight.shape[1] % 64 != 0:
            raise NotImplementedError('weightor head: Conv2d(2 * C, 2, 3) with 2 * C a multiple of 64')
        self._step = -1
        self.sc_ws = None
        self.s = None

    def _w1_forms(self):
        if self._step == tl.StepClock.step:
            return self._f
        w = self.w1.weight.detach().float()                        # (2, 2C, 3, 3)
        dev = w.device
        c2 = w.shape[1]
        small = (w.permute(0, 2, 3, 1).reshape(2, 9, c2).contiguous(), self.w1.bias.detach().float().contiguous())
        wt = torch.zeros((c2, 16, 3, 3), dtype=torch.float32, device=dev)   # data gradient: conv from the 16-wide logit gradient to 2C channels
        wt[:, :2] = w.flip(2, 3).transpose(0, 1)
        self._f = dict(small=small, bw=pack.pack_conv3x3(wt, torch.zeros(c2, dtype=torch.float32, device=dev)))
        self._step = tl.StepClock.step
        return self._f

    # ---- forward ---------------------------------------------------------------------------------------------------------------------
    def forward(self, batch_dict):
        m = self.m
        points = batch_dict['points']
        x = ops.as_nhwc(batch_dict['spatial_features_2d'])
        B, H, W, C = x.shape
        dev = x.device
        N = points.shape[0]
        gt = batch_dict['gt_boxes']
        if gt.dtype != torch.float32 or not gt.is_contiguous():
            gt = gt.float().contiguous()
        itf = batch_dict['instances_tf']
        if itf.dtype != torch.float32 or not itf.is_contiguous():
            itf = itf.float().contiguous()
        M, S = gt.shape[1], int(m.num_sweeps)
        if tuple(itf.shape) != (B, M, S, 3, 4):
            raise ValueError('instances_tf must be (batch, gt_boxes.shape[1], NUM_SWEEPS, 3, 4), got %s' % (tuple(itf.shape),))
        min_xy = m.point_cloud_range[:2]
        pix = [np.float32(m.voxel_size[0]) * m.bev_image_stride, np.float32(m.voxel_size[1]) * m.bev_image_stride]
        cat = torch.empty((B, H, W, 2 * C), dtype=torch.float32, device=dev)         # [bev | corrected]
        self.conv_input.forward(Act(x), out=Act(cat, 0, C))
        pts0 = points.clone()                                                          # rows before the in-place correction
        pf = ops.bev_sample_bilinear(cat, points, min_xy, pix, channels=C)
        h = pf
        for layer in self.mlp:
            h = layer.forward(h)
        local_feat = h
        final = pf.clone()
        tops.accumulate(final, local_feat, C)
        head = self.heads.forward(final)                                               # (N, 16) = cls(3) | flow(3) | embedding(2) | 0
        # ---- locals / instances, object head
        meta = tops.hunter_meta(pts0, B, M, S, sweep_col=m.meta_sweep_col, inst_col=m.meta_inst_col)
        if meta.bad_rows:
            # pcp_hunter_meta already treats such rows as background (they join no local group and no instance); a long run is not
            # aborted for one mislabelled row -- the count is kept and reported once
            self.out_of_table_rows = getattr(self, 'out_of_table_rows', 0) + meta.bad_rows
            if not getattr(self, '_warned_out_of_table', False):
                import warnings
                warnings.warn('HunterJr training: %d foreground rows carry a frame / instance / sweep index outside (batch %d, gt rows %d, '
                              'NUM_SWEEPS %d); they are treated as background (running count: .out_of_table_rows)' % (meta.bad_rows, B, M, S))
                self._warned_out_of_table = True
        s = dict(B=B, H=H, W=W, C=C, N=N, cat=cat, pts0=pts0, pf=pf, local_feat=local_feat, head=head, meta=meta, gt=gt, itf=itf,
                 min_xy=min_xy, pix=pix, M=M, S=S)
        if meta.n_fg > 0:
            centroid, centered = tops.hunter_local_centroids(pts0, meta, 16)
            e = centered
            for layer in self.shape_enc:
                e = layer.forward(e)
            shape_max, s['arg_shape'] = tops.segment_max(e, meta.fg_local, meta.n_local, C, rows=meta.n_fg)
            lf0, s['arg_feat'] = tops.segment_max(pf, meta.fg_local, meta.n_local, C, row_index=meta.fg_idx, rows=meta.n_fg)
            tops.accumulate(lf0, shape_max, C)
            gf, s['arg_g'] = tops.segment_max(lf0, meta.local_inst, meta.n_inst, C, rows=meta.n_local)
            g = tops.hunter_object_cat(lf0, gf, centroid, meta, C, pack.round_up(2 * C + 6, 16))
            for layer in self.local_enc:
                g = layer.forward(g)
            s['locals_feat'] = g
            s['locals_tf'] = self.tf_dec.forward(g)                                    # (n_local, 16) = t(3) | quaternion(4) | 0
        # ---- correct_bev_image
        dyn = ops.hunter_apply_flow(points, head, m.thresh_point_cls_prob)             # mutates batch_dict['points'] like the reference
        pf2 = pf.clone()
        ops.bev_sample_bilinear(cat, points, min_xy, pix, out=pf2, row_mask=dyn, channels=C)
        need = lib.load().pcp_bev_scatter_mean_workspace_bytes(B, H, W, N)
        if self.sc_ws is None or self.sc_ws.numel() < need or self.sc_ws.device != dev:
            self.sc_ws = torch.empty(need, dtype=torch.uint8, device=dev)               # kept: its cell tables drive the backward
        ops.bev_scatter_mean(points, pf2, B, H, W, min_xy, pix, out=cat, out_ch_off=C, workspace=self.sc_ws)
        hid = self.w0.forward(Act(cat))
        f = self._w1_forms()
        logits = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev)
        ops.conv3x3_grouped_small(hid.t, f['small'][0], f['small'][1], [0, 2], logits)
        fused = torch.empty((B, H, W, C), dtype=torch.float32, device=dev)
        ops.softmax_fuse_raw([cat.data_ptr(), cat.data_ptr() + 4 * C], logits, C, 2 * C, fused)
        s.update(dyn=dyn, hid=hid, logits=logits, points=points)
        if 'teacher_spatial_features_2d' in batch_dict:
            # hunter_jr.py:352-365: smooth-L1 between the corrected map and a teacher's, over the pixels the teacher covers.  The reference
            # stores the value and never adds it to the training loss (:490-494), so it is a reported quantity without a gradient
            teacher = ops.as_nhwc(batch_dict['teacher_spatial_features_2d'].float())
            m.forward_return_dict['loss_dtl_bev_img'] = tops.masked_smooth_l1_rows(fused, teacher, C)[0]
        self.s = s
        batch_dict['gt_boxes'] = tops.filter_gt_boxes(gt, m.point_cloud_range)
        return fused

    # ---- losses (hunter_jr.py:401-495) + gradients w.r.t. the head outputs ---------------------------------------------------------------
    def losses(self, grad_scale=1.0):
        s, m = self.s, self.m
        meta = s['meta']
        dev = s['head'].device
        N, C = s['N'], s['C']
        d = lib.HunterLoss()
        d.n, d.stride, d.n_fg, d.n_local, d.n_inst, d.c = N, s['pts0'].shape[1], meta.n_fg, meta.n_local, meta.n_inst, C
        d.batch, d.max_inst, d.num_sweeps = s['B'], s['M'], s['S']
        p = lambda t: t.data_ptr() if t is not None else None
        d.points, d.gt_boxes, d.instances_tf = p(s['pts0']), p(s['gt']), p(s['itf'])
        d.fg_idx, d.fg_local, d.local_key, d.local_inst, d.inst_key = p(meta.fg_idx), p(meta.fg_local), p(meta.local_key), p(meta.local_inst), \
            p(meta.inst_key)
        d.head, d.ld_head = p(s['head']), s['head'].shape[1]
        d.local_feat, d.ld_local_feat = p(s['local_feat']), s['local_feat'].shape[1]
        out = dict(dhead=torch.empty_like(s['head']), losses=torch.empty(8, dtype=torch.float32, device=dev),
                   labels=torch.empty(N, dtype=torch.int32, device=dev))
        if meta.n_fg > 0:
            d.locals_feat, d.ld_locals_feat = p(s['locals_feat']), s['locals_feat'].shape[1]
            d.locals_tf, d.ld_locals_tf = p(s['locals_tf']), s['locals_tf'].shape[1]
            out.update(dlocal_feat_fg=torch.empty((meta.n_fg, C), dtype=torch.float32, device=dev),
                       dlocals_feat=torch.empty((meta.n_local, C), dtype=torch.float32, device=dev),
                       dlocals_tf=torch.empty_like(s['locals_tf']),
                       tgt_embedding=torch.empty((meta.n_fg, 2), dtype=torch.float32, device=dev),
                       tgt_offset=torch.empty((meta.n_fg, 3), dtype=torch.float32, device=dev))
            d.dlocal_feat_fg, d.dlocals_feat = p(out['dlocal_feat_fg']), p(out['dlocals_feat'])
            d.dlocals_tf, d.ld_dlocals_tf = p(out['dlocals_tf']), out['dlocals_tf'].shape[1]
            d.tgt_embedding, d.tgt_offset = p(out['tgt_embedding']), p(out['tgt_offset'])
        d.coef_fg = float(m.model_cfg.get('LOSS_HARD_MINING_STATIC_FG_COEF', 1))
        d.coef_locals = float(m.model_cfg.get('LOSS_HARD_MINING_STATIC_LOCALS_COEF', 1))
        d.grad_scale = float(grad_scale)
        d.dhead, d.ld_dhead = p(out['dhead']), out['dhead'].shape[1]
        d.losses, d.labels = p(out['losses']), p(out['labels'])
        tops.hunter_losses(d, dev)
        s['loss_out'] = out
        return out

    # ---- backward --------------------------------------------------------------------------------------------------------------------
    def backward(self, dfused):
        """dfused: Act, gradient of the fused map (from the CenterHead).  Returns Act gradient of the backbone output."""
        s = self.s
        lo = s['loss_out']
        meta = s['meta']
        B, H, W, C, N = s['B'], s['H'], s['W'], s['C'], s['N']
        dev = s['cat'].device
        cat, dyn = s['cat'], s['dyn']
        if dfused.off != 0 or dfused.t.shape[-1] != C:
            raise ValueError('the fused-map gradient must be a dense (B, H, W, C) buffer')
        dcat = torch.empty((B, H, W, 2 * C), dtype=torch.float32, device=dev)
        dlogits = torch.empty((B, H, W, 16), dtype=torch.float32, device=dev)
        tops.softmax_fuse2_backward(dfused.t, cat, s['logits'], C, dcat, dlogits)
        # weightor: Conv2d(2C, 2, 3) then ConvBNAct(2C, 2C)
        hid = s['hid']
        dwfull = torch.empty((16, 2 * C, 3, 3), dtype=torch.float32, device=dev)
        tops.conv3x3_wgrad(hid.t, dlogits, 2 * C, 16, 1, dwfull)
        db = torch.empty((16,), dtype=torch.float32, device=dev)
        tops.colsum(dlogits, 16, db)
        ensure_grad(self.w1.weight).copy_(dwfull[:2])
        ensure_grad(self.w1.bias).copy_(db[:2])
        w, b, cp = self._w1_forms()['bw']
        dhid = ops.conv3x3(dlogits, w, b, 16, 2 * C, cp, stride=1, relu=False)
        dcat2 = self.w0.backward(Act(dhid))
        tops.accumulate(dcat, dcat2.t, 2 * C, src_ch_off=dcat2.off)
        # per-cell mean of the point features; rows the flow head moved were re-sampled at the moved position
        dpf = torch.zeros((N, C), dtype=torch.float32, device=dev)
        dcf = torch.empty((N, C), dtype=torch.float32, device=dev)
        tops.bev_scatter_mean_backward(self.sc_ws, B, H, W, N, dcat, C, C, dyn, dpf, dcf)
        dhead = lo['dhead']
        tops.bev_sample_bilinear_backward(dcf, s['points'], B, H, W, C, s['min_xy'], s['pix'], dcat, row_mask=dyn, bev=cat, dxyz=dhead, dxyz_ch_off=3)
        # point heads, residual, point MLP
        dfinal = self.heads.backward(dhead)
        tops.accumulate(dpf, dfinal, C)
        if meta.n_fg > 0:
            tops.rows_scatter_add(lo['dlocal_feat_fg'], meta.fg_idx, meta.n_fg, C, dfinal)
        d = dfinal
        for layer in reversed(self.mlp):
            d = layer.backward(d)
        tops.accumulate(dpf, d, C)
        # object head
        if meta.n_fg > 0:
            dg = self.tf_dec.backward(lo['dlocals_tf'])
            tops.accumulate(dg, lo['dlocals_feat'], C)
            d = dg
            for layer in reversed(self.local_enc):
                d = layer.backward(d)
            dlf0, dgf = tops.hunter_object_cat_backward(d, meta, C)
            tops.segment_max_backward(dgf, s['arg_g'], dlf0, C)
            tops.segment_max_backward(dlf0, s['arg_feat'], dpf, C, row_index=meta.fg_idx)
            de = torch.zeros((meta.n_fg, C), dtype=torch.float32, device=dev)
            tops.segment_max_backward(dlf0, s['arg_shape'], de, C)
            d = de
            for i, layer in enumerate(reversed(self.shape_enc)):
                d = layer.backward(d, need_dx=i < len(self.shape_enc) - 1)
        # first sampling (original positions), then conv_input
        tops.bev_sample_bilinear_backward(dpf, s['pts0'], B, H, W, C, s['min_xy'], s['pix'], dcat)
        return self.conv_input.backward(Act(dcat, 0, C))
