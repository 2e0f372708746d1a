"""ORACLE (test infrastructure only) -- HunterJr's TRAINING branch restated on the CPU with torch autograd (configs 1 / 2:
v2x_pointpillar_basic_car.yaml / _rsu.yaml).  Pinned against tests/golden/g12_hunter_train.npz (two iterations of the reference's own train
loop; meta, targets, predictions and the seven loss terms of iteration 0 one by one).

  forward (train)        /root/reference/pcdet/models/bev_layers/hunter_jr.py:289-375 (point head, object head, targets, correct_bev_image)
  object head            /root/reference/pcdet/models/bev_layers/hunter_jr.py:22-76
  locals / instances     /root/reference/pcdet/models/bev_layers/hunter_jr.py:165-196 (_build_meta)
  targets                /root/reference/pcdet/models/bev_layers/hunter_jr.py:198-260 (assign_target)
  losses                 /root/reference/pcdet/models/bev_layers/hunter_jr.py:106-113 (feature distillation), :401-495 (get_training_loss),
                         /root/reference/pcdet/models/loss_fnc/pcaccum_ce_lovasz_loss.py:20-71, lovasz_softmax.py:56-95,
                         /root/reference/pcdet/models/bev_layers/hunter_toolbox.py:42-62 (quat2mat), :161-184 (remove_gt_boxes_outside_range),
                         :187-219 (hard_mining_regression_loss)
  torch_scatter          third party, not vendored: scatter_mean / scatter_max restated with index_add_ / scatter_reduce (max ties share the
                         gradient here; they only occur at ReLU zeros, whose gradient is zero either way)
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import bev as obev


def _bn_train(x, st, prefix, eps=1e-3, momentum=0.01):
    return F.batch_norm(x, st[prefix + '.running_mean'], st[prefix + '.running_var'], st[prefix + '.weight'], st[prefix + '.bias'], True,
                        momentum, eps)


def mlp_train(x, st, prefix, n_layers):
    """nn_make_mlp(..., is_head=False): n_layers x (Linear no bias, BatchNorm1d eps 1e-3 momentum 0.01, ReLU)"""
    for i in range(n_layers):
        x = F.relu(_bn_train(F.linear(x, st['%s.%d.weight' % (prefix, 3 * i)]), st, '%s.%d' % (prefix, 3 * i + 1)))
    return x


def seg_max(x, seg, n):
    e = seg.view(-1, *([1] * (x.dim() - 1))).expand_as(x)
    return x.new_zeros((n,) + tuple(x.shape[1:])).scatter_reduce(0, e, x, 'amax', include_self=False)


def seg_mean(x, seg, n):
    s = x.new_zeros((n,) + tuple(x.shape[1:])).index_add_(0, seg, x)
    c = torch.bincount(seg, minlength=n).clamp_(min=1).to(x.dtype)
    return s / c.view(-1, *([1] * (x.dim() - 1)))


def build_meta(fg, max_inst, num_sweeps):
    """hunter_jr.py:165-196.  fg: (N_fg, 1 + C) rows [b, ..., sweep, inst]."""
    merged = (fg[:, 0].long() * max_inst + fg[:, -1].long()) * num_sweeps + fg[:, -2].long()
    locals_bis, locals2fg = torch.unique(merged, sorted=True, return_inverse=True)
    locals_bi = torch.div(locals_bis, num_sweeps, rounding_mode='floor')
    instance_bi, inst2locals = torch.unique(locals_bi, sorted=True, return_inverse=True)
    locals_sweep = locals_bis - locals_bi * num_sweeps
    n_inst = instance_bi.shape[0]
    pos = torch.arange(locals_sweep.shape[0])
    mx = torch.full((n_inst,), -1, dtype=torch.long).scatter_reduce(0, inst2locals, locals_sweep, 'amax', include_self=True)
    mn = torch.full((n_inst,), 1 << 40, dtype=torch.long).scatter_reduce(0, inst2locals, locals_sweep, 'amin', include_self=True)
    big = locals_sweep.shape[0]
    arg_max = torch.full((n_inst,), big, dtype=torch.long).scatter_reduce(
        0, inst2locals, torch.where(locals_sweep == mx[inst2locals], pos, big), 'amin', include_self=True)
    arg_min = torch.full((n_inst,), big, dtype=torch.long).scatter_reduce(
        0, inst2locals, torch.where(locals_sweep == mn[inst2locals], pos, big), 'amin', include_self=True)
    return dict(locals2fg=locals2fg, inst2locals=inst2locals, indices_locals_max_sweep=arg_max, indices_locals_min_sweep=arg_min,
                locals_bis=locals_bis, instance_bi=instance_bi)


def object_head(fg_xyz, fg_feat, meta, st, prefix, n_hidden):
    """hunter_jr.py:42-76"""
    l2f, i2l = meta['locals2fg'], meta['inst2locals']
    nl, ni = meta['locals_bis'].shape[0], meta['instance_bi'].shape[0]
    centroid = seg_mean(fg_xyz, l2f, nl)
    centered = fg_xyz - centroid[l2f]
    shape_enc = seg_max(mlp_train(centered, st, prefix + '.points_shape_encoder', n_hidden + 1), l2f, nl)
    lf = seg_max(fg_feat, l2f, nl) + shape_enc
    gf = seg_max(lf, i2l, ni)
    target_center = centroid[meta['indices_locals_max_sweep']]
    cat = torch.cat((lf, gf[i2l], centroid, target_center[i2l]), dim=1)
    lf = mlp_train(cat, st, prefix + '.local_feat_encoder', n_hidden + 1)
    tf = F.linear(lf, st[prefix + '.local_tf_decoder.0.weight'], st[prefix + '.local_tf_decoder.0.bias'])
    return tf, lf


def assign_target(points, mask_fg, gt_boxes, instances_tf, meta):
    """hunter_jr.py:198-260"""
    all_tf = instances_tf.reshape(-1, 3, 4)
    locals_tf = all_tf[meta['locals_bis']]
    points_cls = points.new_zeros(points.shape[0], 3)
    points_cls[~mask_fg, 0] = 1.0
    inst_mos = (torch.linalg.norm(instances_tf[:, :, 0, :, -1], dim=-1) > 0.5).reshape(-1)[meta['instance_bi']]
    locals_mos = inst_mos[meta['inst2locals']]
    fg_mos = locals_mos[meta['locals2fg']]
    if bool(mask_fg.any()):
        n_fg = int(mask_fg.sum())
        fg_cls = torch.zeros(n_fg, 2)
        fg_cls.scatter_(1, fg_mos.view(n_fg, 1).long(), 1.0)
        points_cls[mask_fg, 1:] = fg_cls
    box_xy = gt_boxes[:, :, :2].reshape(-1, 2)[meta['instance_bi']]
    fg_embedding = box_xy[meta['inst2locals']][meta['locals2fg']] - points[mask_fg, 1:3]
    fg = points[mask_fg]
    fg_offset = None
    if fg.shape[0] > 0:
        fg_tf = locals_tf[meta['locals2fg']]
        corrected = torch.matmul(fg_tf[:, :3, :3], fg[:, 1:4].unsqueeze(-1)).squeeze(-1) + fg_tf[:, :, -1]
        fg_offset = corrected - fg[:, 1:4]
    return dict(locals_tf=locals_tf, points_cls=points_cls, fg_embedding=fg_embedding, fg_offset=fg_offset, mask_locals_mos=locals_mos)


def lovasz_softmax(prob, labels):
    """lovasz_softmax.py:56-95: mean over the classes present of  <sorted errors, Lovasz gradient of the sorted ground truth>"""
    losses = []
    for c in range(prob.shape[1]):
        fg = (labels == c).float()
        if fg.sum() == 0:
            continue
        err = (fg - prob[:, c]).abs()
        err_sorted, perm = torch.sort(err, 0, descending=True)
        fg_sorted = fg[perm]
        gts = fg_sorted.sum()
        inter = gts - fg_sorted.cumsum(0)
        union = gts + (1 - fg_sorted).cumsum(0)
        jac = 1.0 - inter / union
        if jac.shape[0] > 1:
            jac[1:] = jac[1:] - jac[:-1].clone()
        losses.append(torch.dot(err_sorted, jac))
    return sum(losses) / len(losses) if losses else prob.sum() * 0.0


def ce_lovasz(logits, labels, n_cls=3, max_weight=50.0):
    """pcaccum_ce_lovasz_loss.py:27-71 (n_classes > 2 branch)"""
    counts = torch.stack([(labels == c).float().sum() for c in range(n_cls)])
    w = torch.clamp(torch.sqrt(counts.sum() / counts), 0.0, max_weight).to(logits.dtype)
    ce = F.cross_entropy(logits, labels, w)
    return ce + lovasz_softmax(torch.softmax(logits, dim=1), labels), ce


def hard_mining(loss_all, mask_pos, ratio=1, n_neg_when_no_pos=100):
    """hunter_toolbox.py:187-219"""
    n_pos = int(mask_pos.sum())
    if n_pos == 0:
        if n_neg_when_no_pos < loss_all.shape[0]:
            return torch.topk(loss_all, k=n_neg_when_no_pos)[0].mean()
        return loss_all.mean()
    lp = loss_all[mask_pos].mean()
    n_neg = loss_all.shape[0] - n_pos
    if n_neg > 0:
        k = min(n_pos * ratio, n_neg)
        neg = loss_all[~mask_pos]
        ln = (torch.topk(neg, k=k)[0] if k < n_neg else neg).mean()
    else:
        ln = loss_all.sum() * 0.0
    return lp + ln


def quat2mat(q):
    """hunter_toolbox.py:42-62 ([x, y, z, w], not normalised)"""
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w.pow(2), x.pow(2), y.pow(2), z.pow(2)
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz, 2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).reshape(-1, 3, 3)


def _sl1(a, b):
    return F.smooth_l1_loss(a, b, reduction='none').sum(dim=1)


def filter_gt_boxes(gt_boxes, pc_range):
    """remove_gt_boxes_outside_range (hunter_toolbox.py:161-184): keep rows whose centre is inside the range, pad with zeros"""
    lo, hi = torch.tensor(pc_range[:3], dtype=torch.float32), torch.tensor(pc_range[3:], dtype=torch.float32)
    kept = [gt_boxes[b][((gt_boxes[b, :, :3] >= lo) & (gt_boxes[b, :, :3] < hi)).all(dim=1)] for b in range(gt_boxes.shape[0])]
    m = max(k.shape[0] for k in kept)
    out = gt_boxes.new_zeros(gt_boxes.shape[0], m, gt_boxes.shape[2])
    for b, k in enumerate(kept):
        out[b, :k.shape[0]] = k
    return out


def hunter_train(bev_in, points, gt_boxes, instances_tf, st, arch, cfg, prefix='corrector'):
    """HunterJr.forward in training mode + get_training_loss.  points: (N, 1 + C) float32 torch (cloned; the mutated copy is returned).
    Returns fused map, loss, dict of loss terms, aux."""
    co = arch['corrector']
    dt = bev_in.dtype
    points = points.clone()
    pc_range = arch['pc_range']
    pix = [arch['voxel_size'][0] * co['bev_stride'], arch['voxel_size'][1] * co['bev_stride']]
    p = prefix + '.conv_input'
    bev = F.relu(_bn_train(F.conv2d(bev_in, st[p + '.0.weight'], None, padding=1), st, p + '.1'))

    def sample(pts):
        feat = bev.new_zeros(pts.shape[0], bev.shape[1])
        coord = (pts[:, 1:3] - torch.tensor(pc_range[:2], dtype=torch.float32)) / torch.tensor(pix, dtype=torch.float32)
        bidx = pts[:, 0].long()
        for b in range(bev.shape[0]):
            m = bidx == b
            feat[m] = obev._bilinear(bev[b].permute(1, 2, 0), coord[m, 0].to(dt), coord[m, 1].to(dt))
        return feat, coord
    pf, coord = sample(points)
    ph = prefix + '.point_head'
    n_hidden = len(co['point_hidden'])
    local_feat = mlp_train(pf, st, ph + '.local_feat_predictor', n_hidden + 1)
    final = pf + local_feat
    lin = lambda name: F.linear(final, st['%s.%s.0.weight' % (ph, name)], st['%s.%s.0.bias' % (ph, name)])
    cls_logit, flow, embed = lin('seg'), lin('reg_flow3d'), lin('instance_embedding')
    # ---- training branch: locals, object head, distillation of the local feature, targets
    mask_fg = points[:, -1] > -1
    fg = points[mask_fg]
    meta = build_meta(fg, gt_boxes.shape[1], cfg['NUM_SWEEPS'])
    meta['mask_fg'] = mask_fg
    locals_tf, locals_feat = object_head(fg[:, 1:4].to(dt), pf[mask_fg], meta, st, prefix + '.object_head', len(cfg['OBJ_HEAD_HIDDEN_CHANNELS']))
    if bool(mask_fg.any()):
        l_dtl = _sl1(local_feat[mask_fg], locals_feat[meta['locals2fg']]).mean() * 0.1
    else:
        l_dtl = pf.sum() * 0.0
    tgt = assign_target(points, mask_fg, gt_boxes, instances_tf, meta)
    # ---- correct_bev_image (hunter_jr.py:262-287); the in-place xyz update carries the gradient of the flow head into the re-sampling
    prob = torch.sigmoid(cls_logit)
    pmax, parg = torch.max(prob, dim=1)
    dyn = (pmax > co['thresh_cls']) & (parg == 2)
    pts_g = points.to(dt)
    pts_g = pts_g.clone()
    pts_g[dyn, 1:4] = pts_g[dyn, 1:4] + flow[dyn]
    points[dyn, 1:4] = points[dyn, 1:4] + flow[dyn].detach().float()
    if bool(dyn.any()):
        cf, ccoord = sample(pts_g)
        d = dyn.to(dt)[:, None]
        pf2 = pf * (1.0 - d) + cf * d
    else:
        pf2, ccoord = pf, coord
    corrected = obev.bev_scatter_mean(ccoord.detach().float(), points[:, 0].long(), pf2, bev.shape[2:])
    w = prefix + '.conv_weightor'
    y = F.relu(_bn_train(F.conv2d(torch.cat([bev, corrected], 1), st[w + '.0.0.weight'], None, padding=1), st, w + '.0.1'))
    y = torch.softmax(F.conv2d(y, st[w + '.1.weight'], st[w + '.1.bias'], padding=1), dim=1)
    fused = bev * y[:, [0]] + corrected * y[:, [1]]
    # ---- get_training_loss (hunter_jr.py:401-495)
    terms = {}
    l_cls, _ce = ce_lovasz(cls_logit, torch.argmax(tgt['points_cls'], dim=1))
    terms['l_points_cls'] = l_cls
    terms['l_points_embed'] = _sl1(embed[mask_fg], tgt['fg_embedding'].to(dt)).mean()
    zero = pf.sum() * 0.0
    if tgt['fg_offset'] is not None:
        terms['l_fg_offset'] = hard_mining(_sl1(flow[mask_fg], tgt['fg_offset'].to(dt)), tgt['points_cls'][mask_fg, 2] > 0,
                                           cfg.get('LOSS_HARD_MINING_STATIC_FG_COEF', 1))
    else:
        terms['l_fg_offset'] = zero
    mos = tgt['mask_locals_mos']
    if bool(mask_fg.any()):
        ttf = tgt['locals_tf'].to(dt)
        coef_l = cfg.get('LOSS_HARD_MINING_STATIC_LOCALS_COEF', 1)
        terms['l_locals_transl'] = hard_mining(_sl1(locals_tf[:, :3], ttf[:, :, -1]), mos, coef_l)
        rot = quat2mat(locals_tf[:, 3:])
        terms['l_locals_rot'] = hard_mining(torch.linalg.norm(rot - ttf[:, :, :3], dim=(1, 2), ord='fro'), mos, coef_l)
        fg_xyz = fg[:, 1:4].to(dt)
        fg_tf = ttf[meta['locals2fg']]
        gt_corr = torch.matmul(fg_tf[:, :3, :3], fg_xyz.unsqueeze(-1)).squeeze(-1) + fg_tf[:, :3, -1]
        ptf = torch.cat((rot, locals_tf[:, :3].unsqueeze(-1)), dim=-1)[meta['locals2fg']]
        corr = torch.matmul(ptf[:, :3, :3], fg_xyz.unsqueeze(-1)).squeeze(-1) + ptf[:, :3, -1]
        terms['l_recon'] = hard_mining(_sl1(corr, gt_corr), mos[meta['locals2fg']], cfg.get('LOSS_HARD_MINING_STATIC_FG_COEF', 1)) * 0.1
    else:
        terms['l_locals_transl'] = terms['l_locals_rot'] = terms['l_recon'] = zero
    terms['l_dtl_locals_feat'] = l_dtl
    loss = sum(terms.values())
    aux = dict(bev=bev, points_feat=pf, cls_logit=cls_logit, flow=flow, embed=embed, locals_tf=locals_tf, locals_feat=locals_feat, meta=meta,
               target=tgt, dyn=dyn, points=points, corrected=corrected, local_feat=local_feat)
    return fused, loss, terms, aux
