"""ORACLE (test infrastructure only) -- the dense BEV stages as functional torch-CPU fp32 code driven by a state dict
with the reference's key names.

  backbone            /root/reference/pcdet/models/backbones_2d/base_bev_backbone.py:30-69, 81-112
  center_head_maps    /root/reference/pcdet/models/dense_heads/center_head.py:13-47, 75-96, 377-383
  decode_boxes        /root/reference/pcdet/models/dense_heads/center_head.py:302-357 and
                      /root/reference/pcdet/models/model_utils/centernet_utils.py:127-214
  warp_nearest        /root/reference/pcdet/models/bev_layers/v2x_fusion_disco.py:29-45
  disco_fusion        /root/reference/pcdet/models/bev_layers/v2x_fusion_disco.py:8-26, 71-126
  hunter_jr           /root/reference/pcdet/models/bev_layers/hunter_jr.py:251-312, 350-371 and hunter_toolbox.py:8-127
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import nms as onms


def _t(st, key):
    v = st[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def _bn(x, st, prefix, eps):
    return F.batch_norm(x, _t(st, prefix + '.running_mean'), _t(st, prefix + '.running_var'),
                        _t(st, prefix + '.weight'), _t(st, prefix + '.bias'), False, 0.0, eps)


def _conv(x, st, prefix, stride=1, padding=0):
    b = _t(st, prefix + '.bias') if (prefix + '.bias') in st else None
    return F.conv2d(x, _t(st, prefix + '.weight'), b, stride=stride, padding=padding)


def backbone(x, st, arch, prefix='backbone_2d'):
    """BaseBEVBackbone.forward.  Returns (spatial_features_2d, {stride: block output})."""
    bb = arch['backbone']
    ups, feats = [], {}
    h0 = x.shape[2]
    for i, (nl, s) in enumerate(zip(bb['layer_nums'], bb['strides'])):
        p = '%s.blocks.%d' % (prefix, i)
        x = F.pad(x, (1, 1, 1, 1))
        x = F.relu(_bn(_conv(x, st, p + '.1', stride=s), st, p + '.2', 1e-3))
        for k in range(nl):
            x = F.relu(_bn(_conv(x, st, '%s.%d' % (p, 4 + 3 * k), padding=1), st, '%s.%d' % (p, 5 + 3 * k), 1e-3))
        feats[int(h0 / x.shape[2])] = x
        us = bb['up_strides'][i]
        d = '%s.deblocks.%d' % (prefix, i)
        if us >= 1:
            y = F.conv_transpose2d(x, _t(st, d + '.0.weight'), None, stride=int(us))
        else:
            k = int(np.round(1 / us))
            y = F.conv2d(x, _t(st, d + '.0.weight'), None, stride=k)
        ups.append(F.relu(_bn(y, st, d + '.1', 1e-3)))
    return torch.cat(ups, dim=1), feats


def center_head_maps(x, st, arch, prefix='dense_head'):
    """shared conv + separate heads -> dict name -> (B, k, H, W) raw maps (no sigmoid / exp)."""
    x = F.relu(_bn(_conv(x, st, prefix + '.shared_conv.0', padding=1), st, prefix + '.shared_conv.1', 1e-5))
    out = {}
    for name, _k in arch['head']['heads']:
        p = '%s.heads_list.0.%s' % (prefix, name)
        y = x
        nconv = arch['head']['num_conv']
        for c in range(nconv - 1):
            y = F.relu(_bn(_conv(y, st, '%s.%d.0' % (p, c), padding=1), st, '%s.%d.1' % (p, c), 1e-5))
        out[name] = _conv(y, st, '%s.%d' % (p, nconv - 1), padding=1)
    return out


def topk_desc(scores, k):
    """torch.topk semantics with a deterministic tie rule (lower flat index first)."""
    order = np.lexsort((np.arange(scores.shape[0]), -scores.astype(np.float64)))
    order = order[:k]
    return scores[order], order


def decode_boxes(maps, arch):
    """generate_predicted_boxes for one single-class head, per batch element, WITHOUT NMS.
    Returns list of dict(boxes (n,7), scores (n,), labels (n,) 0-based, cell (n,) flat index)."""
    hd = arch['head']
    hm = torch.sigmoid(maps['hm']).numpy()
    B, C, H, W = hm.shape
    assert C == 1, 'oracle covers the single-class heads of the five configs'
    dim = torch.exp(maps['dim']).numpy()
    ctr, cz, rot = maps['center'].numpy(), maps['center_z'].numpy(), maps['rot'].numpy()
    K = min(hd['max_obj'], H * W)
    stride = np.float32(hd['stride'])
    vx, vy = np.float32(arch['voxel_size'][0]), np.float32(arch['voxel_size'][1])
    x0, y0 = np.float32(arch['pc_range'][0]), np.float32(arch['pc_range'][1])
    lim = np.asarray(hd['limit_range'], dtype=np.float32)
    res = []
    for b in range(B):
        sc, ind = topk_desc(hm[b, 0].reshape(-1), K)
        ys = (ind // W).astype(np.float32)
        xs = (ind % W).astype(np.float32)
        g = lambda m, c: m[b, c].reshape(-1)[ind]
        xs = (xs + g(ctr, 0)) * stride * vx + x0
        ys = (ys + g(ctr, 1)) * stride * vy + y0
        ang = np.arctan2(g(rot, 1), g(rot, 0)).astype(np.float32)
        boxes = np.stack([xs, ys, g(cz, 0), g(dim, 0), g(dim, 1), g(dim, 2), ang], axis=1).astype(np.float32)
        m = (boxes[:, :3] >= lim[None, :3]).all(1) & (boxes[:, :3] <= lim[None, 3:]).all(1)
        if hd['score_thresh'] is not None:
            m &= sc > np.float32(hd['score_thresh'])
        res.append(dict(boxes=boxes[m], scores=sc[m].astype(np.float32), labels=np.zeros(int(m.sum()), np.int64),
                        cell=ind[m].astype(np.int64)))
    return res


def head_postprocess(maps, arch):
    """decode + class-agnostic rotated NMS -> final_box_dicts (labels 1-based, center_head.py:336,355)."""
    hd = arch['head']
    out = []
    for d in decode_boxes(maps, arch):
        sel, sel_scores = onms.class_agnostic_nms(d['scores'], d['boxes'], hd['nms_thresh'], hd['nms_pre'], hd['nms_post'])
        out.append(dict(pred_boxes=d['boxes'][sel], pred_scores=sel_scores, pred_labels=d['labels'][sel] + 1))
    return out


# ---------------------------------------------------------------------------------------------------
# DiscoNet mid fusion
# ---------------------------------------------------------------------------------------------------

def warp_nearest(dst_se3_src, bev_chw, pc_min, pix):
    """transform_bev_img (v2x_fusion_disco.py:29-45), bev_chw: (C, H, W) torch fp32, dst_se3_src: (4,4) torch fp32."""
    rot = dst_se3_src[:2, :2]
    t = dst_se3_src[:2, [-1]]
    tn = 2.0 * ((t - pc_min) / pix) / bev_chw.shape[1] - 1.0
    theta = torch.cat([rot.T, -torch.matmul(rot.T, tn)], dim=1)[None]
    grid = F.affine_grid(theta, (1,) + tuple(bev_chw.shape), align_corners=False)
    return F.grid_sample(bev_chw[None], grid, mode='nearest', align_corners=False)[0]


def _compress(x, st, p):
    x = F.relu(_bn(_conv(x, st, p + '.0', padding=1), st, p + '.1', 1e-5))
    return _conv(x, st, p + '.3', padding=1)


def _pixel_weight(x, st, p):
    x = F.relu(_bn(_conv(x, st, p + '.conv1_1'), st, p + '.bn1_1', 1e-5))
    x = F.relu(_bn(_conv(x, st, p + '.conv1_2'), st, p + '.bn1_2', 1e-5))
    return F.relu(_conv(x, st, p + '.conv1_4'))


def disco_fusion(ego_map, agent_maps, se3_from_ego_per_batch, st, arch, prefix='v2x_mid_fusion'):
    """V2XMidFusionDisco.forward (eval).  agent_maps: ordered dict agent_id -> (B,384,H,W);
    se3_from_ego_per_batch: list (len B) of dict agent_id -> 4x4 float64."""
    fu = arch['fusion']
    ego = _compress(ego_map, st, prefix + '.compressor')
    B = ego.shape[0]
    all_bev = [ego]
    all_w = [_pixel_weight(torch.cat([ego, ego], 1), st, prefix + '.pixel_weightor')]
    for aid, m in agent_maps.items():
        cm = _compress(m, st, prefix + '.compressor')
        padded = cm.new_zeros((B,) + tuple(cm.shape[1:]))
        for b, meta in enumerate(se3_from_ego_per_batch):
            if aid not in meta:
                continue
            T = torch.from_numpy(np.linalg.inv(meta[aid])).float()
            padded[b] = padded[b] + warp_nearest(T, cm[b], fu['pc_min'], fu['pix'])
        all_bev.append(padded)
        all_w.append(_pixel_weight(torch.cat([ego, padded], 1), st, prefix + '.pixel_weightor'))
    w = F.softmax(torch.cat(all_w, dim=1), dim=1)
    fused = sum(all_bev[a] * w[:, a:a + 1] for a in range(len(all_bev)))
    p = prefix + '.decompressor'
    y = F.relu(_bn(_conv(fused, st, p + '.0', padding=1), st, p + '.1', 1e-5))
    return _conv(y, st, p + '.3', padding=1), dict(compressed_ego=ego, weights=w, fused=fused)


# ---------------------------------------------------------------------------------------------------
# HunterJr corrector (inference branch)
# ---------------------------------------------------------------------------------------------------

def _bilinear(im_hwc, x, y):
    """hunter_toolbox.py:8-39"""
    H, W = im_hwc.shape[0], im_hwc.shape[1]
    x0 = torch.floor(x).long()
    x1 = x0 + 1
    y0 = torch.floor(y).long()
    y1 = y0 + 1
    x0 = x0.clamp(0, W - 1)
    x1 = x1.clamp(0, W - 1)
    y0 = y0.clamp(0, H - 1)
    y1 = y1.clamp(0, H - 1)
    Ia, Ib, Ic, Id = im_hwc[y0, x0], im_hwc[y1, x0], im_hwc[y0, x1], im_hwc[y1, x1]
    wa = (x1.type_as(x) - x) * (y1.type_as(y) - y)
    wb = (x1.type_as(x) - x) * (y - y0.type_as(y))
    wc = (x - x0.type_as(x)) * (y1.type_as(y) - y)
    wd = (x - x0.type_as(x)) * (y - y0.type_as(y))
    return Ia * wa[:, None] + Ib * wb[:, None] + Ic * wc[:, None] + Id * wd[:, None]


def sample_point_features(bev, points, pc_range, pix_xy):
    """interpolate_points_feat_from_bev_img (hunter_toolbox.py:94-127)."""
    feat = bev.new_zeros(points.shape[0], bev.shape[1])
    coord = (points[:, 1:3] - torch.tensor(pc_range[:2], dtype=torch.float32)) / torch.tensor(pix_xy, dtype=torch.float32)
    bidx = points[:, 0].long()
    for b in range(bev.shape[0]):
        m = bidx == b
        feat[m] = _bilinear(bev[b].permute(1, 2, 0), coord[m, 0], coord[m, 1])
    return feat, coord


def _mlp_bn_relu(x, st, p_lin, p_bn):
    x = F.linear(x, _t(st, p_lin + '.weight'))
    return F.relu(F.batch_norm(x, _t(st, p_bn + '.running_mean'), _t(st, p_bn + '.running_var'), _t(st, p_bn + '.weight'),
                               _t(st, p_bn + '.bias'), False, 0.0, 1e-3))


def bev_scatter_mean(coord, bidx, feat, hw, batch_size=None):
    """bev_scatter (hunter_toolbox.py:65-91): strict float mask 0 < c < size, truncation, per-cell mean."""
    H, W = hw
    B = int(bidx.max().item()) + 1 if batch_size is None else batch_size
    m = (coord[:, 0] > 0) & (coord[:, 0] < W) & (coord[:, 1] > 0) & (coord[:, 1] < H)
    c = coord[m].long()
    merged = bidx[m] * (H * W) + c[:, 1] * W + c[:, 0]
    unq, inv = torch.unique(merged, return_inverse=True)
    acc = feat.new_zeros(unq.shape[0], feat.shape[1]).index_add_(0, inv, feat[m])
    cnt = torch.bincount(inv, minlength=unq.shape[0]).clamp(min=1).to(feat.dtype)
    img = feat.new_zeros(B * H * W, feat.shape[1])
    img[unq] = acc / cnt[:, None]
    return img.view(B, H, W, -1).permute(0, 3, 1, 2).contiguous()


def hunter_jr(bev_in, points, st, arch, prefix='corrector'):
    """HunterJr.forward eval branch.  Returns (fused map, dict of intermediates).  `points` is cloned; the
    reference mutates batch_dict['points'][:, 1:4] in place (hunter_jr.py:265) -- the mutated copy is returned."""
    co = arch['corrector']
    points = points.clone()
    pc_range = arch['pc_range']
    pix = [arch['voxel_size'][0] * co['bev_stride'], arch['voxel_size'][1] * co['bev_stride']]
    p = prefix + '.conv_input'
    bev = F.relu(_bn(_conv(bev_in, st, p + '.0', padding=1), st, p + '.1', 1e-3))
    pf, coord = sample_point_features(bev, points, pc_range, pix)
    # point head: local_feat_predictor = Linear,BN,ReLU (hidden) + Linear,BN,ReLU (out); then 3 linear heads
    ph = prefix + '.point_head'
    h = pf
    li = 0
    for _ in range(len(co['point_hidden']) + 1):
        h = _mlp_bn_relu(h, st, '%s.local_feat_predictor.%d' % (ph, li), '%s.local_feat_predictor.%d' % (ph, li + 1))
        li += 3
    final = pf + h
    lin = lambda name: F.linear(final, _t(st, '%s.%s.0.weight' % (ph, name)), _t(st, '%s.%s.0.bias' % (ph, name)))
    cls_logit, flow, embed = lin('seg'), lin('reg_flow3d'), lin('instance_embedding')
    prob = torch.sigmoid(cls_logit)
    pmax, parg = torch.max(prob, dim=1)
    dyn = (pmax > co['thresh_cls']) & (parg == 2)
    points[dyn, 1:4] = points[dyn, 1:4] + flow[dyn]
    if bool(dyn.any()):
        cf, ccoord = sample_point_features(bev, points, pc_range, pix)
        d = dyn.float()[:, None]
        pf2 = pf * (1.0 - d) + cf * d
    else:
        pf2, ccoord = pf, coord
    corrected = bev_scatter_mean(ccoord, points[:, 0].long(), pf2, bev.shape[2:])
    w = prefix + '.conv_weightor'
    y = F.relu(_bn(_conv(torch.cat([bev, corrected], 1), st, w + '.0.0', padding=1), st, w + '.0.1', 1e-3))
    y = torch.softmax(_conv(y, st, w + '.1', padding=1), dim=1)
    fused = bev * y[:, [0]] + corrected * y[:, [1]]
    return fused, dict(bev=bev, points_feat=pf, cls_logit=cls_logit, flow=flow, embed=embed, dyn=dyn,
                       corrected=corrected, weights=y, points=points)
