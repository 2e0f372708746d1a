"""ORACLE (test infrastructure only) -- the config-5 (DiscoNet) TRAINING step restated on the CPU: train-mode forward with
batch-statistics BatchNorm, CenterNet target assignment, focal / L1 / distillation losses, torch autograd for the
gradients, and the fastai-style Adam one-cycle step.  Pinned against tests/golden/g7_train.npz (two iterations of the
reference's own train loop).

  train-mode forward   /root/reference/pcdet/models/detectors/centerpoint.py:9-62 (module chain + get_training_loss)
  PFN (train)          /root/reference/pcdet/models/backbones_3d/vfe/dynamic_pillar_vfe.py:35-46 (BatchNorm1d eps 1e-3 momentum 0.01)
  backbone (train)     /root/reference/pcdet/models/backbones_2d/base_bev_backbone.py:30-112 (BatchNorm2d eps 1e-3 momentum 0.01)
  fusion (train)       /root/reference/pcdet/models/bev_layers/v2x_fusion_disco.py:71-126 (compressor BN sees six batches;
                       the warp is @torch.no_grad so agent maps carry no gradient; distillation :119-123)
  head (train)         /root/reference/pcdet/models/dense_heads/center_head.py:13-47, 75-96, 377-392
  targets              /root/reference/pcdet/models/dense_heads/center_head.py:104-164, 166-268 and
                       /root/reference/pcdet/models/model_utils/centernet_utils.py:8-68
  losses               /root/reference/pcdet/models/dense_heads/center_head.py:270-300, /root/reference/pcdet/utils/loss_utils.py:264-375
  optimizer            /root/reference/tools/train_utils/optimization/fastai_optim.py:104-152, learning_schedules_fastai.py:44-77,
                       /root/reference/tools/train_utils/train_utils.py:39-65
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import bev as obev
from . import model as omodel
from . import pillars as opil

F32 = np.float32


# ---------------------------------------------------------------------------------------------------------------------
# targets (numpy, float32 arithmetic like the reference's CPU tensors)
# ---------------------------------------------------------------------------------------------------------------------

def gaussian_radius(height, width, min_overlap):
    """centernet_utils.py:8-35 on float32 arrays."""
    height = height.astype(F32)
    width = width.astype(F32)
    mo = min_overlap
    b1 = height + width
    c1 = width * height * F32((1 - mo) / (1 + mo))
    sq1 = np.sqrt(b1 ** 2 - 4 * c1)
    r1 = (b1 + sq1) / 2
    b2 = 2 * (height + width)
    c2 = F32(1 - mo) * width * height
    sq2 = np.sqrt(b2 ** 2 - 16 * c2)
    r2 = (b2 + sq2) / 2
    a3 = F32(4 * mo)
    b3 = F32(-2 * mo) * (height + width)
    c3 = F32(mo - 1) * width * height
    sq3 = np.sqrt(b3 ** 2 - 4 * a3 * c3)
    r3 = (b3 + sq3) / 2
    return np.minimum(np.minimum(r1, r2), r3).astype(F32)


def gaussian2d(radius):
    """centernet_utils.py:38-44 with sigma = diameter / 6 (float64 numpy, like the reference)."""
    d = 2 * radius + 1
    sigma = d / 6
    m = (d - 1.) / 2.
    y, x = np.ogrid[-m:m + 1, -m:m + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def assign_targets(gt_boxes, arch, feature_hw):
    """single head, classes in config order.  gt_boxes: (B, M, 8) float32 numpy.
    Returns heatmap (B, ncls, H, W), target_boxes (B, K, 8), inds (B, K) int64, mask (B, K) int64."""
    hd = arch['head']
    K = hd['num_max_objs']
    ncls = dict(hd['heads'])['hm']
    H, W = feature_hw
    stride = hd['stride']
    vs, rng = arch['voxel_size'], arch['pc_range']
    B = gt_boxes.shape[0]
    heat = np.zeros((B, ncls, H, W), dtype=F32)
    tb = np.zeros((B, K, 8), dtype=F32)
    inds = np.zeros((B, K), dtype=np.int64)
    mask = np.zeros((B, K), dtype=np.int64)
    for b in range(B):
        g = gt_boxes[b]
        g = g[(g[:, -1] >= 1) & (g[:, -1] <= ncls)]                       # 'bg' rows (class 0 padding) never enter a head
        if g.shape[0] == 0:
            continue
        cx = (g[:, 0] - F32(rng[0])) / F32(vs[0]) / F32(stride)
        cy = (g[:, 1] - F32(rng[1])) / F32(vs[1]) / F32(stride)
        cx = np.clip(cx, F32(0), F32(W - 0.5)).astype(F32)
        cy = np.clip(cy, F32(0), F32(H - 0.5)).astype(F32)
        ix, iy = cx.astype(np.int32), cy.astype(np.int32)
        dx = g[:, 3] / F32(vs[0]) / F32(stride)
        dy = g[:, 4] / F32(vs[1]) / F32(stride)
        with np.errstate(invalid='ignore'):
            rad = gaussian_radius(dx, dy, hd['gaussian_overlap'])
        rad = np.maximum(np.nan_to_num(rad, nan=0.0).astype(np.int32), hd['min_radius'])
        for k in range(min(K, g.shape[0])):
            if dx[k] <= 0 or dy[k] <= 0:
                continue
            if not (0 <= ix[k] <= W and 0 <= iy[k] <= H):
                continue
            r = int(rad[k])
            gk = gaussian2d(r)
            x, y = int(ix[k]), int(iy[k])
            left, right = min(x, r), min(W - x, r + 1)
            top, bottom = min(y, r), min(H - y, r + 1)
            cls = int(g[k, -1]) - 1
            hm = heat[b, cls, y - top:y + bottom, x - left:x + right]
            gg = gk[r - top:r + bottom, r - left:r + right].astype(F32)
            if min(gg.shape) > 0 and min(hm.shape) > 0:
                np.maximum(hm, gg, out=hm)
            inds[b, k] = iy[k] * W + ix[k]
            mask[b, k] = 1
            tb[b, k, 0] = cx[k] - F32(ix[k])
            tb[b, k, 1] = cy[k] - F32(iy[k])
            tb[b, k, 2] = g[k, 2]
            tb[b, k, 3:6] = np.log(g[k, 3:6])
            tb[b, k, 6] = np.cos(g[k, 6])
            tb[b, k, 7] = np.sin(g[k, 6])
    return heat, tb, inds, mask


# ---------------------------------------------------------------------------------------------------------------------
# losses (torch, differentiable)
# ---------------------------------------------------------------------------------------------------------------------

def focal_loss(pred, gt):
    """neg_loss_cornernet (loss_utils.py:264-298) without mask."""
    pos = gt.eq(1).float()
    neg = gt.lt(1).float()
    negw = torch.pow(1 - gt, 4)
    pos_loss = (torch.log(pred) * torch.pow(1 - pred, 2) * pos).sum()
    neg_loss = (torch.log(1 - pred) * torch.pow(pred, 2) * negw * neg).sum()
    npos = pos.sum()
    if npos == 0:
        return -neg_loss
    return -(pos_loss + neg_loss) / npos


def reg_loss(pred_maps, mask, inds, target):
    """RegLossCenterNet (loss_utils.py:313-375): gather (B,K,8) at inds, masked L1 summed over B,K, / max(num,1) -> (8,)"""
    B, C = pred_maps.shape[0], pred_maps.shape[1]
    feat = pred_maps.permute(0, 2, 3, 1).reshape(B, -1, C)
    pred = feat.gather(1, inds[:, :, None].expand(-1, -1, C))
    num = mask.float().sum()
    m = mask[:, :, None].float().expand_as(target) * (~torch.isnan(target)).float()
    loss = torch.abs(pred * m - target * m).sum(dim=(0, 1))
    return loss / torch.clamp_min(num, 1.0)


def head_loss(maps, targets, arch):
    heat, tb, inds, mask = (torch.from_numpy(np.ascontiguousarray(t)) for t in targets)
    heat, tb = heat.to(maps['hm'].dtype), tb.to(maps['hm'].dtype)
    hd = arch['head']
    hm = torch.clamp(maps['hm'].sigmoid(), min=1e-4, max=1 - 1e-4)
    hm_loss = focal_loss(hm, heat) * hd['cls_weight']
    pred = torch.cat([maps[n] for n, _ in hd['heads'] if n != 'hm'], dim=1)
    rl = reg_loss(pred, mask, inds, tb)
    loc_loss = (rl * rl.new_tensor(hd['code_weights'])).sum() * hd['loc_weight']
    return hm_loss + loc_loss, hm_loss, loc_loss


def distill_loss(fused, early):
    return F.smooth_l1_loss(F.softmax(fused, dim=1), F.softmax(early, dim=1)) * 10.0


# ---------------------------------------------------------------------------------------------------------------------
# train-mode forward
# ---------------------------------------------------------------------------------------------------------------------

def _bn_train(x, st, prefix, eps, momentum):
    return F.batch_norm(x, st[prefix + '.running_mean'], st[prefix + '.running_var'], st[prefix + '.weight'],
                        st[prefix + '.bias'], True, momentum, eps)


def _conv(x, st, prefix, stride=1, padding=0):
    b = st.get(prefix + '.bias')
    return F.conv2d(x, st[prefix + '.weight'], b, stride=stride, padding=padding)


def pfn_train(features, inv, num_pillars, st, prefix='vfe'):
    x = features
    idx = inv
    for li in range(2):
        y = x @ st['%s.pfn_layers.%d.linear.weight' % (prefix, li)].t()
        y = F.relu(_bn_train(y, st, '%s.pfn_layers.%d.norm' % (prefix, li), 1e-3, 0.01))
        e = idx[:, None].expand_as(y)
        ymax = y.new_zeros((num_pillars, y.shape[1])).scatter_reduce(0, e, y, 'amax', include_self=False)
        if li == 1:
            return ymax
        x = torch.cat([y, ymax[idx]], dim=1)


def backbone_train(x, st, arch, prefix='backbone_2d'):
    bb = arch['backbone']
    ups = []
    for i, (nl, s) in enumerate(zip(bb['layer_nums'], bb['strides'])):
        p = '%s.blocks.%d' % (prefix, i)
        x = F.pad(x, (1, 1, 1, 1))
        x = F.relu(_bn_train(_conv(x, st, p + '.1', stride=s), st, p + '.2', 1e-3, 0.01))
        for k in range(nl):
            x = F.relu(_bn_train(_conv(x, st, '%s.%d' % (p, 4 + 3 * k), padding=1), st, '%s.%d' % (p, 5 + 3 * k), 1e-3, 0.01))
        us = bb['up_strides'][i]
        d = '%s.deblocks.%d' % (prefix, i)
        if us >= 1:
            y = F.conv_transpose2d(x, st[d + '.0.weight'], None, stride=int(us))
        else:
            y = F.conv2d(x, st[d + '.0.weight'], None, stride=int(np.round(1 / us)))
        ups.append(F.relu(_bn_train(y, st, d + '.1', 1e-3, 0.01)))
    return torch.cat(ups, dim=1)


def _compress_train(x, st, p):
    x = F.relu(_bn_train(_conv(x, st, p + '.0', padding=1), st, p + '.1', 1e-5, 0.1))
    return _conv(x, st, p + '.3', padding=1)


def _pixel_weight_train(x, st, p):
    x = F.relu(_bn_train(_conv(x, st, p + '.conv1_1'), st, p + '.bn1_1', 1e-5, 0.1))
    x = F.relu(_bn_train(_conv(x, st, p + '.conv1_2'), st, p + '.bn1_2', 1e-5, 0.1))
    return F.relu(_conv(x, st, p + '.conv1_4'))


def fusion_train(ego_map, agent_maps, se3, st, arch, prefix='v2x_mid_fusion', probe=None):
    fu = arch['fusion']
    ego = _compress_train(ego_map, st, prefix + '.compressor')
    B = ego.shape[0]
    all_bev = [ego]
    all_w = [_pixel_weight_train(torch.cat([ego, ego], 1), st, prefix + '.pixel_weightor')]
    for aid, m in agent_maps.items():
        cm = _compress_train(m, st, prefix + '.compressor').detach()      # warp is @torch.no_grad: constants downstream
        padded = cm.new_zeros((B,) + tuple(cm.shape[1:]))
        for b, meta in enumerate(se3):
            if aid not in meta:
                continue
            T = torch.from_numpy(np.linalg.inv(meta[aid])).float()
            padded[b] = padded[b] + obev.warp_nearest(T, cm[b].float(), fu['pc_min'], fu['pix']).to(cm.dtype)
        all_bev.append(padded)
        all_w.append(_pixel_weight_train(torch.cat([ego, padded], 1), st, prefix + '.pixel_weightor'))
    w = F.softmax(torch.cat(all_w, dim=1), dim=1)
    fused = sum(all_bev[a] * w[:, a:a + 1] for a in range(len(all_bev)))
    p = prefix + '.decompressor'
    y = F.relu(_bn_train(_conv(fused, st, p + '.0', padding=1), st, p + '.1', 1e-5, 0.1))
    if probe is not None:                      # intermediate tensors for gradient bisection in tests / tools
        probe.update(ego_compressed=ego, fused_in=fused, decomp_mid=y, weights=w)
        for t in (ego, fused, y):
            t.retain_grad()
    return _conv(y, st, p + '.3', padding=1)


def head_train(x, st, arch, prefix='dense_head'):
    x = F.relu(_bn_train(_conv(x, st, prefix + '.shared_conv.0', padding=1), st, prefix + '.shared_conv.1', 1e-5, 0.1))
    out = {}
    for name, _k in arch['head']['heads']:
        p = '%s.heads_list.0.%s' % (prefix, name)
        y = x
        nconv = arch['head']['num_conv']
        for c in range(nconv - 1):
            y = F.relu(_bn_train(_conv(y, st, '%s.%d.0' % (p, c), padding=1), st, '%s.%d.1' % (p, c), 1e-5, 0.1))
        out[name] = _conv(y, st, '%s.%d' % (p, nconv - 1), padding=1)
    return out


def add_train_arch(arch, model_cfg, class_names=None):
    hd = model_cfg['DENSE_HEAD']
    if model_cfg.get('CORRECTOR') is not None:
        arch['corrector_cfg'] = model_cfg['CORRECTOR']
    if arch['head'].get('kind') == 'anchor':
        arch['head']['class_names'] = list(class_names)
        return arch
    ta = hd['TARGET_ASSIGNER_CONFIG']
    lw = hd['LOSS_CONFIG']['LOSS_WEIGHTS']
    arch['head'].update(num_max_objs=ta['NUM_MAX_OBJS'], gaussian_overlap=ta['GAUSSIAN_OVERLAP'], min_radius=ta['MIN_RADIUS'],
                        cls_weight=lw['cls_weight'], loc_weight=lw['loc_weight'], code_weights=list(lw['code_weights']))
    return arch


def is_trainable(key):
    return not key.startswith('bev_maker') and not ('running_' in key or 'num_batches' in key or key == 'global_step')


def make_state(state):
    """numpy/torch state dict -> dict of torch tensors; trainable ones require grad."""
    st = {}
    for k, v in state.items():
        t = v.detach().clone() if isinstance(v, torch.Tensor) else torch.from_numpy(np.array(v))
        if t.dtype == torch.float32 and is_trainable(k):
            t.requires_grad_(True)
        st[k] = t
    return st


def train_forward(points, gt_boxes, metadata, st, arch, probe=None, instances_tf=None):
    """One train-mode forward.  st: make_state() output (running stats are updated in place).  Returns (loss, tb, aux)."""
    pts = np.ascontiguousarray(points, dtype=F32)
    st_np = {k: v.detach().numpy() for k, v in st.items()}
    st_det = {k: v.detach() for k, v in st.items()}
    bev_img, bev_early = None, None
    with torch.no_grad():
        for name in ('bev_maker_rsu', 'bev_maker_car', 'bev_maker_early'):
            if name in arch.get('makers', {}):
                r = omodel.bev_maker(pts, metadata, st_np, st_det, arch['makers'][name], name)
                if arch['makers'][name]['maker_type'] == 'early':
                    bev_early = r
                else:
                    bev_img = r
    nr = arch['num_raw']
    vox = opil.voxelize(pts, nr, arch['pc_range'], arch['voxel_size'], arch['grid_size'])
    feats, _mean = opil.point_features(pts, nr, vox, arch['pc_range'], arch['voxel_size'])
    P = vox['unq'].shape[0]
    dt = st['vfe.pfn_layers.0.linear.weight'].dtype          # float64 runs measure the fp32 noise floor of a fixture
    pf = pfn_train(torch.from_numpy(feats).to(dt), torch.from_numpy(vox['inv']), P, st)
    B = gt_boxes.shape[0]
    nx, ny = int(arch['grid_size'][0]), int(arch['grid_size'][1])
    co = torch.from_numpy(vox['coords'].astype(np.int64))
    canvas = pf.new_zeros((B, ny * nx, pf.shape[1]))
    flat = co[:, 2] * nx + co[:, 3]
    canvas = canvas.index_put((co[:, 0], flat), pf).view(B, ny, nx, -1).permute(0, 3, 1, 2)
    m = backbone_train(canvas, st, arch)
    aux = dict(backbone_out=m)
    loss_distill = None
    loss_corrector = None
    if arch.get('corrector') is not None:                  # configs 1 / 2: HunterJr between backbone and head (centerpoint.py:9-62)
        from . import hunter_train as oht
        gtt = torch.as_tensor(gt_boxes, dtype=torch.float32)
        m, loss_corrector, cterms, caux = oht.hunter_train(m, torch.from_numpy(pts), gtt, torch.as_tensor(instances_tf, dtype=torch.float32),
                                                            st, arch, arch['corrector_cfg'])
        gt_boxes = oht.filter_gt_boxes(gtt, arch['pc_range']).numpy()
        aux.update(hunter=caux, hunter_terms=cterms, gt_boxes_after=gt_boxes)
    if arch.get('fusion') is not None:
        se3 = [md['se3_from_ego'] for md in metadata]
        m = fusion_train(m, {a: t.to(dt) for a, t in bev_img.items()}, se3, st, arch, probe=probe)
        if probe is not None:
            m.retain_grad()
            aux['backbone_out'].retain_grad()
        loss_distill = distill_loss(m, bev_early.to(dt))
        aux['fused'] = m
    if arch['head'].get('kind') == 'anchor':               # MODEL.NAME PointPillar (pointpillar.py:20-33): loss = dense_head.get_loss()
        from . import anchor as oan
        hc, names = arch['head']['cfg'], arch['head']['class_names']
        cls, box, dirp = oan.head_train_forward(m, st, hc)
        anchor_list = [oan.generate_anchors([c], arch['grid_size'], arch['pc_range']) for c in hc['ANCHOR_GENERATOR_CONFIG']]
        labels, reg_t, reg_w = oan.assign_targets(anchor_list, gt_boxes, hc, names)
        anchors = torch.cat(anchor_list, dim=-3).view(-1, 7)
        loss, terms = oan.losses(cls, box, dirp, anchors, labels, reg_t, hc, len(names))
        tb = {k: float(v.detach()) for k, v in terms.items()}
        tb.update(rpn_loss=float(loss.detach()), loss_rpn=float(loss.detach()))
        aux.update(cls_preds=cls, box_preds=box, dir_cls_preds=dirp, box_cls_labels=labels, box_reg_targets=reg_t, reg_weights=reg_w,
                   pillar_features=pf)
        return loss, tb, aux
    maps = head_train(m, st, arch)
    targets = assign_targets(gt_boxes, arch, (m.shape[2], m.shape[3]))
    loss_rpn, hm_loss, loc_loss = head_loss(maps, targets, arch)
    loss = loss_rpn
    tb = dict(loss_rpn=float(loss_rpn.detach()), hm_loss_head_0=float(hm_loss.detach()), loc_loss_head_0=float(loc_loss.detach()),
              rpn_loss=float(loss_rpn.detach()))
    if loss_corrector is not None:
        loss = loss + loss_corrector
        tb.update({k: float(v.detach()) for k, v in cterms.items()})
        tb['loss_corrector'] = float(loss_corrector.detach())
    if loss_distill is not None:
        loss = loss + loss_distill
        tb['loss_mid_fusion_distill'] = float(loss_distill.detach())
    tb['loss_total'] = float(loss.detach())
    aux.update(maps=maps, targets=targets, pillar_features=pf)
    return loss, tb, aux


# ---------------------------------------------------------------------------------------------------------------------
# optimizer: Adam one-cycle with decoupled weight decay (fastai wrapper), gradient clipping
# ---------------------------------------------------------------------------------------------------------------------

def annealing_cos(start, end, pct):
    return end + (start - end) / 2 * (np.cos(np.pi * pct) + 1)


def onecycle(step, total_step, lr_max, moms, div_factor, pct_start):
    """(lr, beta1) the scheduler sets when lr_scheduler.step(step) runs (learning_schedules_fastai.py:44-77)."""
    low = lr_max / div_factor
    a1 = int(pct_start * total_step)
    lr_ph = ((0, a1, (low, lr_max)), (a1, total_step, (lr_max, low / 1e4)))
    mom_ph = ((0, a1, (moms[0], moms[1])), (a1, total_step, (moms[1], moms[0])))
    lr, mom = low, moms[0]
    for s, e, (a, b) in lr_ph:
        if step >= s:
            lr = annealing_cos(a, b, (step - s) / (e - s))
    for s, e, (a, b) in mom_ph:
        if step >= s:
            mom = annealing_cos(a, b, (step - s) / (e - s))
    return float(lr), float(mom)


def clip_coef(total_norm, max_norm):
    """torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1."""
    return min(1.0, max_norm / (total_norm + 1e-6))


class AdamOneCycle:
    """p *= (1 - wd * lr); Adam(betas=(mom, 0.99), eps 1e-8, weight_decay 0) step with bias correction (torch.optim.Adam)."""

    def __init__(self, names, wd=0.01, beta2=0.99, eps=1e-8):
        self.names = list(names)
        self.wd, self.beta2, self.eps = wd, beta2, eps
        self.m, self.v, self.t = {}, {}, 0

    def step(self, st, grads, lr, mom):
        self.t += 1
        bc1 = 1 - mom ** self.t
        bc2 = 1 - self.beta2 ** self.t
        with torch.no_grad():
            for n in self.names:
                p, g = st[n], grads[n]
                p.mul_(1 - self.wd * lr)
                if n not in self.m:
                    self.m[n] = torch.zeros_like(p)
                    self.v[n] = torch.zeros_like(p)
                self.m[n].mul_(mom).add_(g, alpha=1 - mom)
                self.v[n].mul_(self.beta2).addcmul_(g, g, value=1 - self.beta2)
                denom = (self.v[n].sqrt() / math.sqrt(bc2)).add_(self.eps)
                p.addcdiv_(self.m[n], denom, value=-lr / bc1)


def train_step(points, gt_boxes, metadata, st, arch, opt, it, total_steps, ocfg, instances_tf=None):
    """lr_scheduler.step(it); zero_grad; forward; backward; clip; step  (train_utils.py:39-65).
    Returns dict(loss, tb, grads, grad_norm, lr, mom)."""
    lr, mom = onecycle(it, total_steps, ocfg['LR'], list(ocfg['MOMS']), ocfg['DIV_FACTOR'], ocfg['PCT_START'])
    names = opt.names
    for n in names:
        st[n].grad = None
    loss, tb, aux = train_forward(points, gt_boxes, metadata, st, arch, instances_tf=instances_tf)
    loss.backward()
    grads = {n: (st[n].grad.detach().clone() if st[n].grad is not None else torch.zeros_like(st[n])) for n in names}
    norm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())))
    c = clip_coef(norm, ocfg['GRAD_NORM_CLIP'])
    clipped = {n: g * c for n, g in grads.items()} if c < 1.0 else grads
    opt.step(st, clipped, lr, mom)
    return dict(loss=float(loss.detach()), tb=tb, grads=grads, grad_norm=norm, lr=lr, mom=mom, aux=aux)
