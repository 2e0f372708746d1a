"""ORACLE (test infrastructure only) -- AnchorHeadSingle + class-agnostic post-processing, torch CPU fp32.

  anchors               /root/reference/pcdet/models/dense_heads/target_assigner/anchor_generator.py:17-60
  head + decode         /root/reference/pcdet/models/dense_heads/anchor_head_single.py:39-66, anchor_head_template.py:225-272,
                        /root/reference/pcdet/utils/box_coder_utils.py:46-78, /root/reference/pcdet/utils/common_utils.py:25-28
  post-processing       /root/reference/pcdet/models/detectors/detector3d_template.py:262-326 (MULTI_CLASSES_NMS = False branch),
                        /root/reference/pcdet/models/model_utils/model_nms_utils.py:6-25
  training targets      /root/reference/pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py:37-210,
                        /root/reference/pcdet/utils/box_utils.py:291-340 (nearest-BEV IoU), box_coder_utils.py:13-44 (encode_torch)
  training losses       /root/reference/pcdet/models/dense_heads/anchor_head_template.py:99-216,
                        /root/reference/pcdet/utils/loss_utils.py:9-148,180-208 (focal, weighted smooth-L1, weighted cross entropy)
Pinned by tests/golden/g9_anchor_agnostic.npz (the reference's PointPillar detector run on the mini geometry) and
tests/golden/g11_anchor_train.npz (targets, loss terms and gradients of two iterations of the reference's train step).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import bev as obev
from . import nms as onms


def generate_anchors(cfgs, grid_size, pc_range):
    out = []
    rng = np.asarray(pc_range, dtype=np.float32)         # the reference's anchor_range is the float32 point_cloud_range array:
                                                           # strides / shifts are float32 numpy scalars (1-ulp effects on the anchors)
    for cfg in cfgs:
        fm = np.asarray(grid_size[:2], dtype=np.int64) // int(cfg['feature_map_stride'])      # numpy int64: float32 / int64 -> float64 strides
        if cfg.get('align_center', False):
            xs, ys = (rng[3] - rng[0]) / fm[0], (rng[4] - rng[1]) / fm[1]
            xo, yo = xs / 2, ys / 2
        else:
            xs, ys = (rng[3] - rng[0]) / (fm[0] - 1), (rng[4] - rng[1]) / (fm[1] - 1)
            xo, yo = 0, 0
        x = torch.arange(rng[0] + xo, rng[3] + 1e-5, step=xs, dtype=torch.float32)
        y = torch.arange(rng[1] + yo, rng[4] + 1e-5, step=ys, dtype=torch.float32)
        z = x.new_tensor(cfg['anchor_bottom_heights'])
        size = x.new_tensor(cfg['anchor_sizes'])
        rot = x.new_tensor(cfg['anchor_rotations'])
        xg, yg, zg = torch.meshgrid([x, y, z], indexing='ij')
        a = torch.stack((xg, yg, zg), dim=-1)[:, :, :, None, :].repeat(1, 1, 1, size.shape[0], 1)
        a = torch.cat((a, size.view(1, 1, 1, -1, 3).repeat([*a.shape[0:3], 1, 1])), dim=-1)
        a = a[:, :, :, :, None, :].repeat(1, 1, 1, 1, rot.shape[0], 1)
        a = torch.cat((a, rot.view(1, 1, 1, 1, -1, 1).repeat([*a.shape[0:3], size.shape[0], 1, 1])), dim=-1)
        a = a.permute(2, 1, 0, 3, 4, 5).contiguous()
        a[..., 2] += a[..., 5] / 2
        out.append(a)
    return torch.cat(out, dim=-3)


def limit_period(val, offset, period):
    return val - torch.floor(val / period + offset) * period


def head_forward(x, st, head_cfg, grid_size, pc_range, prefix='dense_head'):
    """x: (B, C, H, W).  Returns batch_cls_preds (B, N, ncls) logits, batch_box_preds (B, N, 7), anchors (N, 7)."""
    t = lambda k: obev._t(st, k)
    B = x.shape[0]
    cls = F.conv2d(x, t(prefix + '.conv_cls.weight'), t(prefix + '.conv_cls.bias')).permute(0, 2, 3, 1).contiguous()
    box = F.conv2d(x, t(prefix + '.conv_box.weight'), t(prefix + '.conv_box.bias')).permute(0, 2, 3, 1).contiguous()
    anchors = generate_anchors(head_cfg['ANCHOR_GENERATOR_CONFIG'], grid_size, pc_range).view(-1, 7)
    n = anchors.shape[0]
    cls_preds = cls.view(B, n, -1)
    enc = box.view(B, n, -1)
    an = anchors.view(1, n, 7).repeat(B, 1, 1)
    xa, ya, za, dxa, dya, dza, ra = torch.split(an, 1, dim=-1)
    xt, yt, zt, dxt, dyt, dzt, rt = torch.split(enc, 1, dim=-1)
    diag = torch.sqrt(dxa ** 2 + dya ** 2)
    boxes = torch.cat([xt * diag + xa, yt * diag + ya, zt * dza + za, torch.exp(dxt) * dxa, torch.exp(dyt) * dya, torch.exp(dzt) * dza,
                       rt + ra], dim=-1)
    if head_cfg.get('USE_DIRECTION_CLASSIFIER', None) is not None:
        dirp = F.conv2d(x, t(prefix + '.conv_dir_cls.weight'), t(prefix + '.conv_dir_cls.bias')).permute(0, 2, 3, 1).contiguous()
        dir_labels = torch.max(dirp.view(B, n, -1), dim=-1)[1]
        period = 2 * np.pi / head_cfg['NUM_DIR_BINS']
        dir_rot = limit_period(boxes[..., 6] - head_cfg['DIR_OFFSET'], head_cfg['DIR_LIMIT_OFFSET'], period)
        boxes[..., 6] = dir_rot + head_cfg['DIR_OFFSET'] + period * dir_labels.to(boxes.dtype)
    return cls_preds, boxes, anchors


def post_process(cls_preds, boxes, post_cfg):
    """class-agnostic branch.  Returns list of dict(boxes, scores, labels 1-based)."""
    nms = post_cfg['NMS_CONFIG']
    out = []
    for b in range(boxes.shape[0]):
        sc, lab = torch.max(torch.sigmoid(cls_preds[b]), dim=-1)
        lab = lab + 1
        bx = boxes[b]
        thr = post_cfg['SCORE_THRESH']
        idx_all = torch.arange(sc.shape[0])
        if thr is not None:
            m = sc >= thr
            sc_m, bx_m, idx_m = sc[m], bx[m], idx_all[m]
        else:
            sc_m, bx_m, idx_m = sc, bx, idx_all
        sel = np.zeros(0, dtype=np.int64)
        if sc_m.shape[0] > 0:
            k = min(nms['NMS_PRE_MAXSIZE'], sc_m.shape[0])
            _, order = obev.topk_desc(sc_m.numpy(), k)
            keep = onms.nms_sorted(bx_m[order, :7].numpy(), nms['NMS_THRESH'])
            sel = idx_m.numpy()[order[keep[:nms['NMS_POST_MAXSIZE']]]]
        out.append(dict(boxes=bx[sel].numpy(), scores=sc[sel].numpy(), labels=lab[sel].numpy()))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# training: AxisAlignedTargetAssigner (POS_FRACTION < 0, MATCH_HEIGHT False, NORM_BY_NUM_EXAMPLES False) and the three losses
# ---------------------------------------------------------------------------------------------------------------------

def aligned_bev(boxes):
    """box_utils.py:314-325 -- (N, 7) float32 torch -> (N, 4) [x1, y1, x2, y2] of the nearest axis-aligned rectangle"""
    rot = limit_period(boxes[:, 6], 0.5, np.pi).abs()
    dims = torch.where(rot[:, None] < np.pi / 4, boxes[:, [3, 4]], boxes[:, [4, 3]])
    return torch.cat((boxes[:, 0:2] - dims / 2, boxes[:, 0:2] + dims / 2), dim=1)


def nearest_bev_iou(a, b):
    """box_utils.py:291-311,328-340"""
    a, b = aligned_bev(a), aligned_bev(b)
    x_min = torch.max(a[:, 0, None], b[None, :, 0])
    x_max = torch.min(a[:, 2, None], b[None, :, 2])
    y_min = torch.max(a[:, 1, None], b[None, :, 1])
    y_max = torch.min(a[:, 3, None], b[None, :, 3])
    inter = torch.clamp_min(x_max - x_min, min=0) * torch.clamp_min(y_max - y_min, min=0)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / torch.clamp_min(area_a[:, None] + area_b[None, :] - inter, min=1e-6)


def encode_boxes(boxes, anchors):
    """box_coder_utils.py:13-44 (no sin/cos encoding)"""
    anchors = anchors.clone()
    boxes = boxes.clone()
    anchors[:, 3:6] = torch.clamp_min(anchors[:, 3:6], min=1e-5)
    boxes[:, 3:6] = torch.clamp_min(boxes[:, 3:6], min=1e-5)
    xa, ya, za, dxa, dya, dza, ra = torch.split(anchors, 1, dim=-1)
    xg, yg, zg, dxg, dyg, dzg, rg = torch.split(boxes, 1, dim=-1)
    diag = torch.sqrt(dxa ** 2 + dya ** 2)
    return torch.cat([(xg - xa) / diag, (yg - ya) / diag, (zg - za) / dza, torch.log(dxg / dxa), torch.log(dyg / dya), torch.log(dzg / dza),
                      rg - ra], dim=-1)


def assign_targets(anchor_list, gt_boxes, head_cfg, class_names):
    """anchor_list: one (ny, nx, n_size, n_rot, 7) tensor per ANCHOR_GENERATOR_CONFIG entry; gt_boxes (B, M, 8) float32 numpy / torch.
    Returns box_cls_labels (B, N) int32, box_reg_targets (B, N, 7), reg_weights (B, N) in the order of torch.cat(anchor_list, dim=-3)."""
    ta = head_cfg['TARGET_ASSIGNER_CONFIG']
    assert ta['POS_FRACTION'] < 0 and not ta['MATCH_HEIGHT'] and not ta['NORM_BY_NUM_EXAMPLES']
    gt_all = torch.as_tensor(gt_boxes, dtype=torch.float32)
    names = np.array(class_names)
    labels_b, targets_b, weights_b = [], [], []
    for k in range(gt_all.shape[0]):
        cur = gt_all[k, :, :7]
        cnt = cur.shape[0] - 1
        while cnt > 0 and cur[cnt].sum() == 0:
            cnt -= 1
        cur = cur[:cnt + 1]
        cur_cls = gt_all[k, :cnt + 1, 7].int()
        per_class = []
        for cfg, anchors5 in zip(head_cfg['ANCHOR_GENERATOR_CONFIG'], anchor_list):
            mask = torch.from_numpy(np.atleast_1d(names[cur_cls.numpy() - 1] == cfg['class_name']))      # class 0 wraps to the LAST name
            fm = anchors5.shape[:3]
            anchors = anchors5.reshape(-1, 7)
            gts, gcls = cur[mask], cur_cls[mask]
            n = anchors.shape[0]
            labels = torch.full((n,), -1, dtype=torch.int32)
            tgt = anchors.new_zeros((n, 7))
            if gts.shape[0] > 0:
                iou = nearest_bev_iou(anchors, gts)
                a_arg = iou.argmax(dim=1)
                a_max = iou[torch.arange(n), a_arg]
                g_max = iou.max(dim=0)[0].clone()
                g_max[g_max == 0] = -1
                forced = (iou == g_max).nonzero()[:, 0]
                labels[forced] = gcls[a_arg[forced]]
                pos = a_max >= cfg['matched_threshold']
                labels[pos] = gcls[a_arg[pos]]
                fg = (labels > 0).nonzero()[:, 0]
                labels[a_max < cfg['unmatched_threshold']] = 0
                labels[forced] = gcls[a_arg[forced]]
                tgt[fg] = encode_boxes(gts[a_arg[fg]], anchors[fg])
            else:
                labels[:] = 0
            w = (labels > 0).float()
            per_class.append((labels.view(*fm, -1), tgt.view(*fm, -1, 7), w.view(*fm, -1)))
        labels_b.append(torch.cat([p[0] for p in per_class], dim=-1).view(-1))
        targets_b.append(torch.cat([p[1] for p in per_class], dim=-2).view(-1, 7))
        weights_b.append(torch.cat([p[2] for p in per_class], dim=-1).view(-1))
    return torch.stack(labels_b), torch.stack(targets_b), torch.stack(weights_b)


def head_train_forward(x, st, head_cfg, prefix='dense_head'):
    """the three 1x1 convs of anchor_head_single.py:39-52: (B, H, W, A * ncls), (B, H, W, A * 7), (B, H, W, A * nbins)"""
    t = lambda k: st[k]
    cls = F.conv2d(x, t(prefix + '.conv_cls.weight'), t(prefix + '.conv_cls.bias')).permute(0, 2, 3, 1).contiguous()
    box = F.conv2d(x, t(prefix + '.conv_box.weight'), t(prefix + '.conv_box.bias')).permute(0, 2, 3, 1).contiguous()
    dirp = None
    if head_cfg.get('USE_DIRECTION_CLASSIFIER', None) is not None:
        dirp = F.conv2d(x, t(prefix + '.conv_dir_cls.weight'), t(prefix + '.conv_dir_cls.bias')).permute(0, 2, 3, 1).contiguous()
    return cls, box, dirp


def losses(cls_preds, box_preds, dir_preds, anchors, labels, reg_targets, head_cfg, num_class):
    """anchor_head_template.py:99-216 with torch autograd.  anchors (N, 7); labels (B, N) int; reg_targets (B, N, 7).
    Returns (rpn_loss, dict of the three weighted terms)."""
    lw = head_cfg['LOSS_CONFIG']['LOSS_WEIGHTS']
    B = cls_preds.shape[0]
    dt = cls_preds.dtype
    labels = labels.clone()
    cared, positives, negatives = labels >= 0, labels > 0, labels == 0
    cls_w = (negatives * 1.0 + 1.0 * positives).to(dt)
    reg_w = positives.to(dt)
    if num_class == 1:
        labels[positives] = 1
    norm = torch.clamp(positives.sum(1, keepdim=True).to(dt), min=1.0)
    reg_w = reg_w / norm
    cls_w = cls_w / norm
    cls_t = (labels * cared.type_as(labels)).long()
    one_hot = torch.zeros(*cls_t.shape, num_class + 1, dtype=dt)
    one_hot.scatter_(-1, cls_t.unsqueeze(-1), 1.0)
    one_hot = one_hot[..., 1:]
    x = cls_preds.view(B, -1, num_class)
    p = torch.sigmoid(x)
    alpha_w = one_hot * 0.25 + (1 - one_hot) * 0.75
    pt = one_hot * (1.0 - p) + (1.0 - one_hot) * p
    bce = torch.clamp(x, min=0) - x * one_hot + torch.log1p(torch.exp(-torch.abs(x)))
    cls_loss = (alpha_w * torch.pow(pt, 2.0) * bce * cls_w.unsqueeze(-1)).sum() / B * lw['cls_weight']
    n = anchors.shape[0]
    bp = box_preds.view(B, n, -1)
    tg = reg_targets.to(dt)
    bp_s = torch.cat([bp[..., :6], torch.sin(bp[..., 6:7]) * torch.cos(tg[..., 6:7])], dim=-1)
    tg_s = torch.cat([tg[..., :6], torch.cos(bp[..., 6:7]) * torch.sin(tg[..., 6:7])], dim=-1)
    tg_s = torch.where(torch.isnan(tg_s), bp_s, tg_s)
    diff = (bp_s - tg_s) * torch.tensor(np.array(lw['code_weights'], dtype=np.float32)).to(dt).view(1, 1, -1)
    a = torch.abs(diff)
    beta = 1.0 / 9.0
    sl1 = torch.where(a < beta, 0.5 * a ** 2 / beta, a - 0.5 * beta)
    loc_loss = (sl1 * reg_w.unsqueeze(-1)).sum() / B * lw['loc_weight']
    terms = dict(rpn_loss_cls=cls_loss, rpn_loss_loc=loc_loss)
    total = cls_loss + loc_loss
    if dir_preds is not None:
        nb = head_cfg['NUM_DIR_BINS']
        rot_gt = reg_targets[..., 6] + anchors.view(1, n, 7)[..., 6]
        off = limit_period(rot_gt - head_cfg['DIR_OFFSET'], 0, 2 * np.pi)
        dir_t = torch.clamp(torch.floor(off / (2 * np.pi / nb)).long(), min=0, max=nb - 1)
        w = positives.to(dt)
        w = w / torch.clamp(w.sum(-1, keepdim=True), min=1.0)
        ce = F.cross_entropy(dir_preds.view(B, n, nb).permute(0, 2, 1), dir_t, reduction='none') * w
        dir_loss = ce.sum() / B * lw['dir_weight']
        terms['rpn_loss_dir'] = dir_loss
        total = total + dir_loss
    return total, terms
