"""ORACLE (test infrastructure only) -- AnchorHeadSingle + class-agnostic post-processing, torch CPU fp32.

  anchors               /root/reference/pcdet/models/dense_heads/target_assigner/anchor_generator.py:17-60
  head + decode         /root/reference/pcdet/models/dense_heads/anchor_head_single.py:39-66, anchor_head_template.py:225-272,
                        /root/reference/pcdet/utils/box_coder_utils.py:46-78, /root/reference/pcdet/utils/common_utils.py:25-28
  post-processing       /root/reference/pcdet/models/detectors/detector3d_template.py:262-326 (MULTI_CLASSES_NMS = False branch),
                        /root/reference/pcdet/models/model_utils/model_nms_utils.py:6-25
Pinned by tests/golden/g9_anchor_agnostic.npz (the reference's PointPillar detector run on the mini geometry).
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
