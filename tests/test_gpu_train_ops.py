"""Training kernels (include/pcp_hip_train.h) against torch CPU autograd of the same op (floating point kernels: the torch fp32
reference is the checker here, tolerance written per test).  GPU only."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from pcp_amd import synth

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def _u(seed, col, shape, lo=-1.0, hi=1.0):
    n = int(np.prod(shape))
    return torch.from_numpy(synth.uniform(7000 + seed, col, n, lo, hi).reshape(shape).astype(np.float32))


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(DEV)


def _nchw(t):
    return t.permute(0, 3, 1, 2).cpu()


def _close(a, b, tol, what='', floor=1e-6):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    scale = max(float(b.abs().max()), floor)
    err = float((a - b).abs().max())
    assert err <= tol * scale, '%s: max err %.3e vs scale %.3e' % (what, err, scale)


@pytest.mark.parametrize('c,shape', [(64, (2, 24, 40)), (384, (2, 16, 16)), (16, (3, 9, 11)), (32, (1, 1, 3001))])
def test_bn_train_forward_backward(c, shape):
    from pcp_amd import train_ops as tops
    B, H, W = shape
    x = _u(1, c, (B, c, H, W), -2, 3)
    bn = nn.BatchNorm2d(c, eps=1e-3, momentum=0.01)
    bn.weight.data = _u(2, c, (c,), 0.5, 1.5)
    bn.bias.data = _u(3, c, (c,), -0.2, 0.2)
    bn.running_mean.data = _u(4, c, (c,), -0.1, 0.1)
    bn.running_var.data = _u(5, c, (c,), 0.5, 1.5)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    bn.train()
    xr = x.clone().requires_grad_(True)
    out = F.relu(bn(xr))
    dout = _u(6, c, tuple(out.shape))
    out.backward(dout)
    xg = _nhwc(x)
    rm, rv = rm0.to(DEV), rv0.to(DEV)
    vec = tops.bn_train_stats(xg, c, bn.weight.data.to(DEV), bn.bias.data.to(DEV), 1e-3, 0.01, rm, rv)
    og = torch.empty_like(xg)
    tops.scale_shift_act(xg, c, vec, True, og)
    _close(_nchw(og), out, 2e-5, 'bn forward')
    _close(rm, bn.running_mean, 1e-5, 'running_mean')
    _close(rv, bn.running_var, 1e-5, 'running_var')
    dg = torch.zeros(c, device=DEV)
    db = torch.zeros(c, device=DEV)
    dog = _nhwc(dout)
    tops.bn_act_backward(dog, xg, c, vec, True, dg, db)
    _close(dg, bn.weight.grad, 5e-5, 'dgamma')
    _close(db, bn.bias.grad, 5e-5, 'dbeta')
    _close(_nchw(dog), xr.grad, 1e-4, 'dx')


def test_colsum_accumulate_dilate():
    from pcp_amd import train_ops as tops
    x = _u(11, 1, (2, 6, 10, 48))
    xg = x.to(DEV)
    out = torch.zeros(32, device=DEV)
    tops.colsum(xg, 32, out, ch_off=8)
    _close(out, x[..., 8:40].reshape(-1, 32).sum(0), 1e-6)
    tops.colsum(xg, 32, out, accumulate=True, ch_off=8)
    _close(out, 2 * x[..., 8:40].reshape(-1, 32).sum(0), 1e-6)
    y = _u(11, 2, (2, 6, 10, 32)).to(DEV)
    want = xg.clone()
    want[..., 16:48] += 0.5 * y
    tops.accumulate(xg, y, 32, alpha=0.5, dst_ch_off=16)
    assert torch.equal(xg, want)
    d = tops.dilate2x(y, 32)
    ref = torch.zeros((2, 12, 20, 32), device=DEV)
    ref[:, ::2, ::2] = y
    assert torch.equal(d, ref)


@pytest.mark.parametrize('cin,cout,stride,shape', [(64, 128, 1, (2, 24, 40)), (128, 64, 2, (2, 24, 40)), (16, 16, 1, (1, 9, 21)),
                                                   (320, 16, 1, (2, 16, 16)), (64, 64, 2, (1, 64, 64))])
def test_conv3x3_wgrad(cin, cout, stride, shape):
    from pcp_amd import train_ops as tops
    B, H, W = shape
    x = _u(21, cin, (B, cin, H, W))
    w = _u(22, cout, (cout, cin, 3, 3), -0.1, 0.1).requires_grad_(True)
    y = F.conv2d(F.pad(x, (1, 1, 1, 1)), w, None, stride=stride)
    dy = _u(23, cout, tuple(y.shape))
    y.backward(dy)
    dw = torch.full((cout, cin, 3, 3), 7.0, device=DEV)
    tops.conv3x3_wgrad(_nhwc(x), _nhwc(dy), cin, cout, stride, dw)
    _close(dw, w.grad, 2e-5, 'wgrad')
    tops.conv3x3_wgrad(_nhwc(x), _nhwc(dy), cin, cout, stride, dw, accumulate=True)
    _close(dw, 2 * w.grad, 2e-5, 'wgrad accumulate')


def test_conv3x3_wgrad_channel_windows():
    from pcp_amd import train_ops as tops
    x = _u(24, 1, (2, 96, 16, 24))
    dy = _u(24, 2, (2, 80, 16, 24))
    w = _u(24, 3, (64, 64, 3, 3), -0.1, 0.1).requires_grad_(True)
    y = F.conv2d(x[:, 32:96], w, None, padding=1)
    y.backward(dy[:, 16:80])
    dw = torch.zeros((64, 64, 3, 3), device=DEV)
    tops.conv3x3_wgrad(_nhwc(x), _nhwc(dy), 64, 64, 1, dw, x_ch_off=32, dy_ch_off=16)
    _close(dw, w.grad, 2e-5)


def _layer_case(conv, bn, relu, x, seed):
    """runs torch CPU autograd and the HIP layer; returns dict of (mine, ref) pairs"""
    from pcp_amd import train_layers as tl
    conv_g = type(conv)(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, bias=conv.bias is not None)
    conv_g.load_state_dict(conv.state_dict())
    conv_g = conv_g.to(DEV)
    bn_g = None
    if bn is not None:
        bn_g = nn.BatchNorm2d(bn.num_features, eps=bn.eps, momentum=bn.momentum)
        bn_g.load_state_dict(bn.state_dict())
        bn_g = bn_g.to(DEV)
        bn.train()
    xr = x.clone().requires_grad_(True)
    xin = F.pad(xr, (1, 1, 1, 1)) if (isinstance(conv, nn.Conv2d) and conv.kernel_size[0] == 3 and conv.padding[0] == 0) else xr
    y = conv(xin)
    if bn is not None:
        y = bn(y)
    if relu:
        y = F.relu(y)
    dout = _u(seed, 9, tuple(y.shape))
    y.backward(dout)
    tl.StepClock.tick()
    layer = tl.ConvBNAct(conv_g, bn_g, relu, name='case')
    out = layer.forward(tl.Act(_nhwc(x)))
    dx = layer.backward(tl.Act(_nhwc(dout)))
    pairs = {'out': (_nchw(out.t), y), 'dx': (_nchw(dx.t), xr.grad), 'dw': (conv_g.weight.grad, conv.weight.grad)}
    if conv.bias is not None:
        pairs['dbias'] = (conv_g.bias.grad, conv.bias.grad)
    if bn is not None:
        pairs['dgamma'] = (bn_g.weight.grad, bn.weight.grad)
        pairs['dbeta'] = (bn_g.bias.grad, bn.bias.grad)
        pairs['rmean'] = (bn_g.running_mean, bn.running_mean)
        pairs['rvar'] = (bn_g.running_var, bn.running_var)
    return pairs


def _mk_conv(kind, cin, cout, bias, seed):
    if kind == 'c3s1':
        c = nn.Conv2d(cin, cout, 3, 1, 1, bias=bias)
    elif kind == 'c3s2':
        c = nn.Conv2d(cin, cout, 3, 2, 0, bias=bias)           # the backbone's ZeroPad2d(1) + stride-2 conv
    elif kind == 'plain':
        c = nn.Conv2d(cin, cout, 1, 1, 0, bias=bias)
    elif kind == 's2d':
        c = nn.Conv2d(cin, cout, 2, 2, 0, bias=bias)
    elif kind == 'plainT':
        c = nn.ConvTranspose2d(cin, cout, 1, 1, bias=bias)
    else:
        c = nn.ConvTranspose2d(cin, cout, 2, 2, bias=bias)
    c.weight.data = _u(seed, 1, tuple(c.weight.shape), -0.08, 0.08)
    if bias:
        c.bias.data = _u(seed, 2, (cout,), -0.1, 0.1)
    return c


def _mk_bn(c, seed, eps=1e-3, mom=0.01):
    bn = nn.BatchNorm2d(c, eps=eps, momentum=mom)
    bn.weight.data = _u(seed, 3, (c,), 0.5, 1.5)
    bn.bias.data = _u(seed, 4, (c,), -0.2, 0.2)
    return bn


@pytest.mark.parametrize('kind,cin,cout,bias,with_bn,relu,shape', [
    ('c3s1', 64, 64, False, True, True, (2, 24, 40)),
    ('c3s1', 128, 64, True, True, True, (2, 16, 16)),
    ('c3s1', 64, 128, True, False, False, (2, 16, 24)),
    ('c3s2', 64, 128, False, True, True, (2, 32, 48)),
    ('plain', 256, 64, True, True, True, (2, 16, 16)),
    ('plain', 64, 16, True, True, True, (2, 16, 16)),
    ('plainT', 128, 128, False, True, True, (2, 16, 24)),
    ('s2d', 64, 128, False, True, True, (2, 32, 48)),
    ('d2s', 128, 128, False, True, True, (2, 8, 12)),
])
def test_conv_bn_act_layer_forward_backward(kind, cin, cout, bias, with_bn, relu, shape, monkeypatch):
    B, H, W = shape
    seed = 31 + cin + cout
    conv = _mk_conv(kind, cin, cout, bias, seed)
    bn = _mk_bn(cout, seed) if with_bn else None
    x = _u(seed, 5, (B, cin, H, W))
    # c3s1: every forward / data-gradient kernel family (the fused F(4x4) kernels take the layer in `auto` only at full-size maps: forced
    # here); c3s2: the data gradient (a stride-1 conv on the zero-dilated gradient) through the fused F(4x4) kernel as well
    for algo in (('direct', 'winograd', 'winograd4f', 'winograd4h') if kind == 'c3s1' else ('auto', 'winograd4h') if kind == 'c3s2' else ('auto',)):
        monkeypatch.setenv('PCP_CONV_ALGO', algo)
        conv.zero_grad()
        if bn is not None:
            bn.zero_grad()
            bn.running_mean.zero_()
            bn.running_var.fill_(1.0)
        pairs = _layer_case(conv, bn, relu, x, seed)
        for k, (mine, ref) in pairs.items():
            # a conv bias in front of a BatchNorm has an analytically zero gradient: both sides hold rounding noise there
            floor = 1.0 if (k == 'dbias' and with_bn) else 1e-6
            _close(mine, ref, 3e-4, '%s/%s/%s' % (kind, algo, k), floor=floor)


def test_grouped_weight_repack_equals_the_per_layer_repack_over_several_steps(monkeypatch):
    """from the second optimizer step on every bias-free 3x3 layer is repacked by ONE launch (pcp_pack_conv3x3_group); three steps of a two-layer
    stack with an in-place SGD update give bit-identical activations and gradients with the group switched off"""
    from pcp_amd import train_layers as tl

    def run(grouping):
        monkeypatch.setattr(tl, 'PACK_GROUPING', grouping)
        monkeypatch.setattr(tl, 'PACK_GROUP', tl.tops.PackGroup())
        c1, c2 = _mk_conv('c3s1', 64, 64, False, 91).to(DEV), _mk_conv('c3s2', 64, 128, False, 92).to(DEV)
        b1, b2 = _mk_bn(64, 93).to(DEV), _mk_bn(128, 94).to(DEV)
        l1, l2 = tl.ConvBNAct(c1, b1, True, name='g1'), tl.ConvBNAct(c2, b2, True, name='g2')
        x = _nhwc(_u(95, 1, (2, 64, 32, 48)).to(DEV))
        outs = []
        for step in range(3):
            tl.StepClock.tick()
            y = l2.forward(l1.forward(tl.Act(x)))
            g = l1.backward(l2.backward(tl.Act(torch.ones_like(y.t) * 0.01)))
            outs += [y.t.clone(), g.t.clone(), c1.weight.grad.clone(), c2.weight.grad.clone()]
            with torch.no_grad():
                for c in (c1, c2):
                    c.weight.add_(c.weight.grad, alpha=-0.05)               # in place: the packed forms are stale until the next repack
        return outs, len(tl.PACK_GROUP.jobs)
    a, n_a = run(True)
    b, n_b = run(False)
    assert n_a == 4 and n_b == 0                                         # two layers x (forward, data gradient) registered
    for i, (u, v) in enumerate(zip(a, b)):
        assert torch.equal(u, v), i
    assert not torch.equal(a[0], a[4])                                   # the weights did move between the steps


# ---------------------------------------------------------------------------------------------------------------------
# a15: targets, losses
# ---------------------------------------------------------------------------------------------------------------------

def _head_arch(ncls=1, K=500, H=32, W=32):
    return dict(pc_range=[-12.8, -12.8, -8.0, 12.8, 12.8, 0.0], voxel_size=[0.2, 0.2, 8.0],
                head=dict(heads=[('center', 2), ('center_z', 1), ('dim', 3), ('rot', 2), ('hm', ncls)], stride=4, num_max_objs=K,
                          gaussian_overlap=0.1, min_radius=2, cls_weight=1.0, loc_weight=0.25, code_weights=[1.0, 1.0, 1.0, 1.0, 2.0, 1.0, 0.5, 1.0]))


def _rand_gt(seed, B, M, ncls, n_valid):
    gt = np.zeros((B, M, 8), dtype=np.float32)
    for b in range(B):
        n = n_valid[b]
        s = 8100 + 10 * seed + b
        rows = np.sort(np.argsort(synth.uniform01(s, 0, M))[:n])            # valid rows interleaved with padding rows
        gt[b, rows, 0] = synth.uniform(s, 1, n, -14.0, 14.0)                 # some centres outside the range (clamped)
        gt[b, rows, 1] = synth.uniform(s, 2, n, -14.0, 14.0)
        gt[b, rows, 2] = synth.uniform(s, 3, n, -3.0, -1.0)
        gt[b, rows, 3] = synth.uniform(s, 4, n, 0.5, 9.0)
        gt[b, rows, 4] = synth.uniform(s, 5, n, 0.5, 4.0)
        gt[b, rows, 5] = synth.uniform(s, 6, n, 1.0, 3.0)
        gt[b, rows, 6] = synth.uniform(s, 7, n, -3.14159, 3.14159)
        gt[b, rows, 7] = np.floor(synth.uniform(s, 8, n, 1.0, ncls + 0.999))
        if n > 3:
            gt[b, rows[1], 3] = 0.0                                          # degenerate box: skipped (center_head.py:143)
            gt[b, rows[2], 0:2] = gt[b, rows[0], 0:2]                        # two boxes in one cell
    return gt


def _target_desc(arch, B, H, W, ncls):
    from pcp_amd import lib
    hd = arch['head']
    return lib.Target(B, H, W, ncls, hd['num_max_objs'], float(hd['stride']), 0.2, 0.2, -12.8, -12.8, hd['gaussian_overlap'], hd['min_radius'])


@pytest.mark.parametrize('ncls,K,M,n_valid', [(1, 500, 40, [25, 0, 40]), (3, 500, 64, [64, 10, 33]), (1, 8, 30, [30, 3, 12]), (2, 500, 1, [1, 0, 1])])
def test_centerhead_targets_match_oracle(ncls, K, M, n_valid):
    from oracle import train as otr
    from pcp_amd import train_ops as tops
    H = W = 32
    arch = _head_arch(ncls, K, H, W)
    gt = _rand_gt(ncls * 7 + K, 3, M, ncls, n_valid)
    heat, tb, inds, mask = otr.assign_targets(gt, arch, (H, W))
    gh, gtb, ginds, gmask = tops.centerhead_targets(torch.from_numpy(gt).to(DEV), _target_desc(arch, 3, H, W, ncls))
    assert np.array_equal(ginds.cpu().numpy(), inds) and np.array_equal(gmask.cpu().numpy(), mask)
    np.testing.assert_allclose(gh.permute(0, 3, 1, 2).cpu().numpy(), heat, atol=1e-6)
    assert np.array_equal(gh.permute(0, 3, 1, 2).cpu().numpy() == 1.0, heat == 1.0)       # the positives of the focal loss
    np.testing.assert_allclose(gtb.cpu().numpy(), tb, atol=2e-6)


@pytest.mark.parametrize('ncls,n_valid', [(1, [25, 0, 40]), (3, [30, 10, 33]), (1, [0, 0, 0])])
def test_centerhead_loss_and_gradient_match_autograd(ncls, n_valid):
    from oracle import train as otr
    from pcp_amd import lib
    from pcp_amd import train_ops as tops
    H = W = 32
    B = 3
    arch = _head_arch(ncls, 500, H, W)
    gt = _rand_gt(50 + ncls, B, 48, ncls, n_valid)
    targets = otr.assign_targets(gt, arch, (H, W))
    nch = 8 + ncls
    ld = 16
    maps_t = _u(60 + ncls, 1, (B, nch, H, W), -3.0, 3.0)
    maps_t[:, 8:] -= 1.0
    maps_t[0, 8, 0, 0] = 12.0                                   # sigmoid clamped from above: zero gradient
    maps_t[0, 8, 0, 1] = -12.0                                  # ... and from below
    maps_t.requires_grad_(True)
    names = ['center', 'center_z', 'dim', 'rot', 'hm']
    sizes = [2, 1, 3, 2, ncls]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    maps = {n: maps_t[:, offs[i]:offs[i + 1]] for i, n in enumerate(names)}
    loss, hm_loss, loc_loss = otr.head_loss(maps, targets, arch)
    loss.backward()
    buf = torch.zeros((B, H, W, ld))
    buf[..., :nch] = maps_t.detach().permute(0, 2, 3, 1)
    buf = buf.to(DEV)
    d = lib.HeadLoss()
    d.batch, d.h, d.w, d.ld, d.ld_d, d.num_class, d.ch_hm, d.k = B, H, W, ld, ld, ncls, 8, 500
    for j in range(8):
        d.reg_ch[j] = j
        d.code_weights[j] = arch['head']['code_weights'][j]
    d.cls_weight, d.loc_weight = 1.0, 0.25
    heat, tb, inds, mask = (torch.from_numpy(np.ascontiguousarray(t)).to(DEV) for t in targets)
    dhead = torch.full((B, H, W, ld), 9.0, device=DEV)
    losses = tops.centerhead_loss(buf, d, heat.permute(0, 2, 3, 1).contiguous(), tb, inds.int(), mask.int(), dhead=dhead)
    lv = losses.cpu().numpy()
    assert abs(lv[0] - float(hm_loss.detach())) <= 2e-5 * max(abs(float(hm_loss.detach())), 1e-3)
    assert abs(lv[1] - float(loc_loss.detach())) <= 2e-5 * max(abs(float(loc_loss.detach())), 1e-3)
    assert abs(lv[2] - float(loss.detach())) <= 2e-5 * max(abs(float(loss.detach())), 1e-3)
    assert lv[3] == float((targets[0] == 1).sum())
    g = dhead.cpu()
    assert float(g[..., nch:].abs().max()) == 0.0
    _close(g[..., :nch].permute(0, 3, 1, 2), maps_t.grad, 2e-4, 'dhead')


def test_distill_loss_and_gradient():
    from oracle import train as otr
    from pcp_amd import train_ops as tops
    f = _u(71, 1, (2, 384, 12, 10), -2, 2).requires_grad_(True)
    e = _u(71, 2, (2, 384, 12, 10), -2, 2)
    loss = otr.distill_loss(f, e)
    loss.backward()
    fg, eg = _nhwc(f.detach()), _nhwc(e)
    df = torch.zeros_like(fg)
    lv = tops.distill_loss(fg, eg, 384, dfused=df)
    assert abs(float(lv) - float(loss.detach())) <= 1e-5 * float(loss.detach())
    _close(_nchw(df), f.grad, 2e-4, 'dfused')
    lv2 = tops.distill_loss(fg, eg, 384, dfused=df, accumulate=True)
    _close(_nchw(df), 2 * f.grad, 2e-4, 'dfused accumulate')


# ---------------------------------------------------------------------------------------------------------------------
# PFN (train mode) through the module-level driver
# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize('crowded', [False, True])
def test_vfe_train_forward_backward_matches_autograd(crowded):
    """crowded: cells of 17 ... 1 500 points on top of the fixture's cloud -- the pillariser lists pillars of more than 16 points and the
    training kernels give each a workgroup (k_pfnt_*_long, k_sort_long_runs) instead of one lane group walking all its points"""
    from helpers import arch_of, load_golden
    from oracle import pillars as opil
    from oracle import train as otr
    from pcdet.models import build_network_from_meta
    from pcdet.models.train_path import VFETrain
    from pcp_amd import train_layers as tl
    g = load_golden('g7_train.npz')
    meta = g['meta']
    arch = arch_of(meta)
    model = build_network_from_meta(meta)
    stf = synth.fill_state_dict(meta['state_shapes'])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in stf.items()})
    vfe = model.vfe.to(DEV).train()
    tl.StepClock.tick()
    pts = g['points']
    if crowded:
        rs = np.random.RandomState(17)
        x_lo, y_lo = float(arch['pc_range'][0]), float(arch['pc_range'][1])
        extra = []
        for i, k in enumerate((17, 16, 40, 300, 1500, 33)):
            q = np.zeros((k, pts.shape[1]), np.float32)
            q[:, 0] = i % 2
            q[:, 1] = x_lo + 0.2 * (20 + 3 * i) + rs.uniform(0.01, 0.19, k)
            q[:, 2] = y_lo + 0.2 * (31 + i) + rs.uniform(0.01, 0.19, k)
            q[:, 3] = rs.uniform(-6.0, -1.0, k)
            q[:, 4:] = pts[rs.randint(0, pts.shape[0], k), 4:]
            extra.append(q)
        pts = np.concatenate([pts] + extra, 0)
        pts = np.ascontiguousarray(pts[rs.permutation(pts.shape[0])])
    bd = {'points': torch.from_numpy(pts).to(DEV), 'batch_size': 2}
    drv = VFETrain(vfe)
    bd = drv.forward(bd)
    # oracle
    st = otr.make_state(stf)
    vox = opil.voxelize(pts, arch['num_raw'], arch['pc_range'], arch['voxel_size'], arch['grid_size'])
    feats, _ = opil.point_features(pts, arch['num_raw'], vox, arch['pc_range'], arch['voxel_size'])
    P = vox['unq'].shape[0]
    pf = otr.pfn_train(torch.from_numpy(feats), torch.from_numpy(vox['inv']), P, st)
    assert np.array_equal(bd['voxel_coords'].cpu().numpy(), vox['coords'])
    _close(bd['pillar_features'], pf, 2e-5, 'pillar_features (train-mode BN)')
    for li in range(2):
        for k in ('running_mean', 'running_var'):
            _close(getattr(vfe.pfn_layers[li].norm, k), st['vfe.pfn_layers.%d.norm.%s' % (li, k)], 1e-5, k)
    nx, ny = arch['grid_size'][0], arch['grid_size'][1]
    R = _u(81, 1, (2, ny, nx, 64))
    co = torch.from_numpy(vox['coords'].astype(np.int64))
    (pf * R[co[:, 0], co[:, 2], co[:, 3]]).sum().backward()
    drv.backward(R.to(DEV))
    for name, p in (('pfn_layers.0.linear.weight', vfe.pfn_layers[0].linear.weight), ('pfn_layers.0.norm.weight', vfe.pfn_layers[0].norm.weight),
                    ('pfn_layers.0.norm.bias', vfe.pfn_layers[0].norm.bias), ('pfn_layers.1.linear.weight', vfe.pfn_layers[1].linear.weight),
                    ('pfn_layers.1.norm.weight', vfe.pfn_layers[1].norm.weight), ('pfn_layers.1.norm.bias', vfe.pfn_layers[1].norm.bias)):
        _close(p.grad, st['vfe.' + name].grad, 5e-4, name)


# ---------------------------------------------------------------------------------------------------------------------
# DiscoNet fusion kernels, optimizer
# ---------------------------------------------------------------------------------------------------------------------

def test_disco_weight_logits_and_fuse_backward():
    import torch.nn.functional as F
    from pcp_amd import ops
    from pcp_amd import train_ops as tops
    A, B, H, W, C = 3, 2, 6, 10, 128
    maps = [_u(91, a, (B, H, W, C)).requires_grad_(a == 0) for a in range(A)]
    h2 = [_u(92, a, (B, H, W, 16), -0.5, 1.0).requires_grad_(True) for a in range(A)]
    w4 = _u(93, 1, (16,), -0.5, 0.5).requires_grad_(True)
    b4 = _u(93, 2, (1,), -0.1, 0.3).requires_grad_(True)
    logits = torch.stack([F.relu(h @ w4 + b4) for h in h2], dim=-1)
    wgt = torch.softmax(logits, dim=-1)
    fused = sum(maps[a] * wgt[..., a:a + 1] for a in range(A))
    dfused = _u(94, 1, (B, H, W, C))
    fused.backward(dfused)
    gm = [m.detach().to(DEV) for m in maps]
    gh = [h.detach().to(DEV) for h in h2]
    lg = torch.zeros((B, H, W, 8), device=DEV)
    tops.disco_weight_logits(gh, w4.detach().to(DEV), b4.detach().to(DEV), lg)
    _close(lg[..., :A], logits, 1e-5, 'logits')
    out = torch.empty((B, H, W, C), device=DEV)
    ops.softmax_fuse_raw([m.data_ptr() for m in gm], lg, C, C, out)
    _close(out, fused, 1e-5, 'fused')
    dmap0 = torch.empty((B, H, W, C), device=DEV)
    dh2 = [torch.empty((B, H, W, 16), device=DEV) for _ in range(A)]
    dw4, db4 = torch.zeros(16, device=DEV), torch.zeros(1, device=DEV)
    tops.disco_fuse_backward([m.data_ptr() for m in gm], C, C, lg, dfused.to(DEV), gh, w4.detach().to(DEV), dmap0, dh2, dw4, db4)
    _close(dmap0, maps[0].grad, 1e-5, 'dmap0')
    for a in range(A):
        _close(dh2[a], h2[a].grad, 1e-4, 'dh2[%d]' % a)
    _close(dw4, w4.grad, 1e-4, 'dw4')
    _close(db4, b4.grad, 1e-4, 'db4')


def test_fused_adam_step_matches_torch_adam_with_clipping():
    from pcp_amd import train_ops as tops
    n = 100003
    p0 = _u(95, 1, (n,), -0.5, 0.5)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=0.0, betas=(0.9, 0.99))
    p = torch.zeros(n + 1, device=DEV)[:n]
    p.copy_(p0)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    sq = torch.zeros(1, dtype=torch.float64, device=DEV)
    for step in range(1, 4):
        g = _u(95, 10 + step, (n,), -1.0, 1.0) * (5.0 if step == 2 else 0.01)       # step 2 is clipped, the others are not
        lr, mom, wd, max_norm = 1e-3 * step, 0.95 - 0.01 * step, 0.01, 10.0
        ref.grad = g.clone()
        norm = torch.nn.utils.clip_grad_norm_([ref], max_norm)
        for grp in opt.param_groups:
            grp['lr'], grp['betas'] = lr, (mom, 0.99)
        with torch.no_grad():
            ref.mul_(1 - wd * lr)
        opt.step()
        gg = g.to(DEV)
        tops.grad_sqnorm(gg, out=sq)
        assert abs(float(sq.sqrt()) - float(norm)) <= 1e-5 * float(norm)
        tops.adam_step(p, gg, m, v, lr, mom, 0.99, 1e-8, wd, step, max_norm=max_norm, sqnorm=sq)
        _close(p, ref, 2e-6, 'param after step %d' % step)


def test_plain_bf16_conv_is_within_the_bf16_error_band():
    """pcp_conv3x3_bf16 (single bf16 products, the mixed-precision training mode): error vs float64 between the split-bf16 kernel's
    and 2^-8 relative to the output scale, stride 1 and 2"""
    from pcp_amd import ops, pack
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    for stride in (1, 2):
        x = (torch.rand((2, 64, 40, 48), generator=g) - 0.3)
        w = (torch.rand((128, 64, 3, 3), generator=g) - 0.5) * 0.1
        b = (torch.rand(128, generator=g) - 0.5) * 0.2
        want = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=1)
        p3, b3, cp3 = pack.pack_conv3x3_bf16x3(w, b)
        xd = ops.as_nhwc(x.to(DEV))
        got1 = ops.conv3x3_bf16x3(xd, p3.to(DEV), b3.to(DEV), 64, 128, cp3, stride=stride, relu=False, plain=True)
        got3 = ops.conv3x3_bf16x3(xd, p3.to(DEV), b3.to(DEV), 64, 128, cp3, stride=stride, relu=False)
        torch.cuda.synchronize()
        scale = float(want.abs().max())
        e1 = float((got1.permute(0, 3, 1, 2).cpu().double() - want).abs().max()) / scale
        e3 = float((got3.permute(0, 3, 1, 2).cpu().double() - want).abs().max()) / scale
        assert e3 < 5e-5 and e3 < e1 < 2 ** -8, (stride, e1, e3)


def test_conv_bn_act_layer_with_optin_bf16x3(monkeypatch):
    """training layer forward + data gradient through the opt-in split-bf16 conv kernel (weights re-split on the device every step by
    pcp_pack_conv3x3); same 3e-4 bar as the fp32 kernels.  No ReLU here: a 1e-5 forward difference flips the mask of the few
    pre-activations within 1e-5 of zero, which moves single dx elements by O(|dout|) -- a property of the discontinuity, not of the
    kernel (the e2e training test bounds that effect globally)"""
    monkeypatch.setenv('PCP_CONV_ALGO', 'bf16x3')
    seed = 977
    conv = _mk_conv('c3s1', 64, 128, False, seed)
    bn = _mk_bn(128, seed)
    x = _u(seed, 5, (4, 64, 128, 128))
    pairs = _layer_case(conv, bn, False, x, seed)
    for k, (mine, ref) in pairs.items():
        _close(mine, ref, 3e-4, 'bf16x3/' + k)


# ---- a16: AnchorHeadSingle training ------------------------------------------------------------------------------------------

def _anchor_setup(head_cfg, class_names, grid_size, pc_range):
    from oracle import anchor as oan
    from pcp_amd import lib
    gcfg = head_cfg['ANCHOR_GENERATOR_CONFIG']
    anchor_list = [oan.generate_anchors([c], grid_size, pc_range) for c in gcfg]
    flat = torch.cat(anchor_list, dim=-3).reshape(-1, 7).contiguous()
    ny, nx = anchor_list[0].shape[1], anchor_list[0].shape[2]
    d = lib.AnchorAssign()
    d.h, d.w = ny, nx
    d.num_class, d.num_groups = len(class_names), len(gcfg)
    slot = 0
    for g, c in enumerate(gcfg):
        for _ in range(len(c['anchor_rotations']) * len(c['anchor_sizes']) * len(c['anchor_bottom_heights'])):
            d.slot_group[slot] = g
            slot += 1
        d.group_class[g] = class_names.index(c['class_name'])
        d.matched[g], d.unmatched[g] = c['matched_threshold'], c['unmatched_threshold']
    d.anchors_per_loc = slot
    return anchor_list, flat, d


def test_anchor_target_assignment_is_the_references():
    """pcp_anchor_assign_targets on the reference's own fixture (tests/golden/g11_anchor_train.npz: three anchor classes, threshold and
    forced positives, ignored anchors, an all-zero row in the middle of a frame, trailing padding): labels and weights bit exact,
    regression targets to 1e-6 (device logf)"""
    from helpers import load_golden
    from pcp_amd import train_ops as tops
    g = load_golden('g11_anchor_train.npz')
    meta = g['meta']
    hc = meta['model']['DENSE_HEAD']
    _al, flat, d = _anchor_setup(hc, meta['class_names'], [128, 128, 1], meta['pc_range'])
    assert np.array_equal(flat.numpy(), g['anchors'].reshape(-1, 7))
    d.batch = 2
    labels, reg_t, reg_w = tops.anchor_assign_targets(flat.to(DEV), torch.from_numpy(g['gt_boxes']).to(DEV), d)
    assert np.array_equal(labels.cpu().numpy(), g['box_cls_labels'])
    assert np.array_equal(reg_w.cpu().numpy(), g['reg_weights'])
    np.testing.assert_allclose(reg_t.cpu().numpy(), g['box_reg_targets'], rtol=0, atol=1e-6)


@pytest.mark.parametrize('case', ['full_single_class', 'three_classes_many_boxes', 'no_boxes', 'all_zero_rows'])
def test_anchor_target_assignment_matches_the_oracle(case):
    """full-size map (128 x 128 x 2 anchors = the shipped v2x_pointpillar_anchor.yaml), 60 boxes per frame; 6 anchors x 200 boxes with
    many ties (boxes ON anchor centres); frames without boxes (M = 0 and all-zero rows): labels bit exact against oracle/anchor.py"""
    from oracle import anchor as oan
    from pcp_amd import train_ops as tops
    rng = np.random.RandomState(11)
    if case == 'full_single_class':
        names = ['car']
        hc = {'ANCHOR_GENERATOR_CONFIG': [{'class_name': 'car', 'anchor_sizes': [[4.7, 2.1, 1.7]], 'anchor_rotations': [0, 1.57],
                                           'anchor_bottom_heights': [-3.0], 'align_center': False, 'feature_map_stride': 4,
                                           'matched_threshold': 0.6, 'unmatched_threshold': 0.45}]}
        grid, rngp, B, M = [512, 512, 1], [-51.2, -51.2, -8.0, 51.2, 51.2, 0.0], 3, 60
    else:
        names = ['car', 'pedestrian', 'cyclist']
        mk = lambda n, s, hi, lo: {'class_name': n, 'anchor_sizes': [s], 'anchor_rotations': [0, 1.57], 'anchor_bottom_heights': [-1.0],
                                   'align_center': False, 'feature_map_stride': 4, 'matched_threshold': hi, 'unmatched_threshold': lo}
        hc = {'ANCHOR_GENERATOR_CONFIG': [mk('car', [3.9, 1.6, 1.56], 0.6, 0.45), mk('pedestrian', [0.8, 0.6, 1.73], 0.5, 0.35),
                                           mk('cyclist', [1.76, 0.6, 1.73], 0.5, 0.35)]}
        grid, rngp, B, M = [256, 192, 1], [-25.6, -19.2, -8.0, 25.6, 19.2, 0.0], 2, 200
    hc['TARGET_ASSIGNER_CONFIG'] = {'POS_FRACTION': -1.0, 'MATCH_HEIGHT': False, 'NORM_BY_NUM_EXAMPLES': False}
    al, flat, d = _anchor_setup(hc, names, grid, rngp)
    d.batch = B
    gt = np.zeros((B, M, 8), dtype=np.float32)
    if case in ('full_single_class', 'three_classes_many_boxes'):
        for b in range(B):
            n = M - 7 * b
            gt[b, :n, 0] = rng.uniform(rngp[0], rngp[3], n)
            gt[b, :n, 1] = rng.uniform(rngp[1], rngp[4], n)
            gt[b, :n, 2] = rng.uniform(-3, -1, n)
            cls = rng.randint(1, len(names) + 1, n)
            base = np.array([[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]] if len(names) == 3 else [[4.7, 2.1, 1.7]], dtype=np.float32)
            gt[b, :n, 3:6] = base[cls - 1] * rng.uniform(0.8, 1.25, (n, 3))
            gt[b, :n, 6] = rng.uniform(-3.2, 3.2, n)
            gt[b, :n, 7] = cls
            # a third of the boxes sit exactly on anchor centres with anchor headings: equal overlaps between neighbours (ties)
            a7 = flat.numpy()
            pick = rng.randint(0, a7.shape[0], n // 3)
            gt[b, :n // 3, 0:2] = a7[pick, 0:2]
            gt[b, :n // 3, 6] = a7[pick, 6]
    if case == 'no_boxes':
        gt = np.zeros((B, 0, 8), dtype=np.float32)
    ref_l, ref_t, ref_w = oan.assign_targets(al, gt if gt.shape[1] else np.zeros((B, 1, 8), np.float32), hc, names)
    labels, reg_t, reg_w = tops.anchor_assign_targets(flat.to(DEV), torch.from_numpy(gt).to(DEV), d)
    assert np.array_equal(labels.cpu().numpy(), ref_l.numpy())
    assert np.array_equal(reg_w.cpu().numpy(), ref_w.numpy())
    np.testing.assert_allclose(reg_t.cpu().numpy(), ref_t.numpy(), rtol=0, atol=2e-6)
    if case in ('full_single_class', 'three_classes_many_boxes'):
        assert (ref_l.numpy() > 0).sum() > B * 20 and (ref_l.numpy() == -1).any()


def test_anchor_losses_and_head_gradient():
    """pcp_anchor_loss on the reference's own head outputs and targets (g11): the three weighted terms against the reference's tb_dict
    (1e-5 relative), dL/d(head maps) against torch autograd of oracle/anchor.py (1e-5 of the largest entry), padding channels zero"""
    import json
    from helpers import load_golden
    from oracle import anchor as oan
    from pcp_amd import lib
    from pcp_amd import train_ops as tops
    g = load_golden('g11_anchor_train.npz')
    meta = g['meta']
    hc = meta['model']['DENSE_HEAD']
    ncls, A, nb = 3, 6, 2
    cls, box, dirp = (torch.from_numpy(g[k].copy()).requires_grad_(True) for k in ('cls_preds', 'box_preds', 'dir_cls_preds'))
    anchors = torch.from_numpy(g['anchors']).reshape(-1, 7).contiguous()
    labels, reg_t = torch.from_numpy(g['box_cls_labels']), torch.from_numpy(g['box_reg_targets'])
    total, terms = oan.losses(cls, box, dirp, anchors, labels, reg_t, hc, ncls)
    total.backward()
    B, H, W = cls.shape[0], cls.shape[1], cls.shape[2]
    used = A * (ncls + 7 + nb)
    ld = 80
    head = torch.zeros((B, H, W, ld))
    head[..., :A * ncls] = cls.detach()
    head[..., A * ncls:A * ncls + A * 7] = box.detach()
    head[..., A * ncls + A * 7:used] = dirp.detach()
    d = lib.AnchorLoss()
    d.batch, d.h, d.w, d.ld, d.ld_d = B, H, W, ld, ld
    d.anchors_per_loc, d.num_class, d.num_dir_bins = A, ncls, nb
    d.ch_cls, d.ch_box, d.ch_dir = 0, A * ncls, A * ncls + A * 7
    d.dir_offset, d.dir_period = hc['DIR_OFFSET'], float(2 * np.pi / nb)
    lw = hc['LOSS_CONFIG']['LOSS_WEIGHTS']
    d.cls_weight, d.loc_weight, d.dir_weight = lw['cls_weight'], lw['loc_weight'], lw['dir_weight']
    for j in range(7):
        d.code_weights[j] = lw['code_weights'][j]
    dhead = torch.full((B, H, W, ld), 7.0, device=DEV)
    losses = tops.anchor_loss(head.to(DEV), anchors.to(DEV), labels.to(DEV), reg_t.to(DEV), d, dhead=dhead).cpu().numpy()
    ref_tb = json.loads(str(g['it0_tb_json']))
    for i, k in enumerate(('rpn_loss_cls', 'rpn_loss_loc', 'rpn_loss_dir', 'rpn_loss')):
        assert abs(losses[i] - ref_tb[k]) <= 1e-5 * abs(ref_tb[k]), (k, losses[i], ref_tb[k])
    assert losses[4] == float((g['box_cls_labels'] > 0).sum())
    dh = dhead.cpu()
    assert float(dh[..., used:].abs().max()) == 0.0
    ref = torch.cat([cls.grad, box.grad, dirp.grad], dim=-1)
    _close(dh[..., :used], ref, 1e-5, 'dhead')
    for lo, hi, name in ((0, A * ncls, 'cls'), (A * ncls, A * ncls + A * 7, 'box'), (A * ncls + A * 7, used, 'dir')):
        _close(dh[..., lo:hi], ref[..., lo:hi], 2e-5, 'dhead ' + name)
    # class-agnostic head (num_class 1): every positive trains logit 0; and grad_scale scales the gradient only
    d.num_class = 1
    d.ch_box, d.ch_dir = A, A + A * 7
    cls1 = cls.detach()[..., ::3].contiguous().requires_grad_(True)
    head1 = torch.zeros((B, H, W, ld))
    head1[..., :A], head1[..., A:A + A * 7], head1[..., A + A * 7:A + A * 9] = cls1.detach(), box.detach(), dirp.detach()
    t1, _ = oan.losses(cls1, box.detach(), dirp.detach(), anchors, labels, reg_t, hc, 1)
    t1.backward()
    l1 = tops.anchor_loss(head1.to(DEV), anchors.to(DEV), labels.to(DEV), reg_t.to(DEV), d, dhead=dhead, grad_scale=0.5).cpu().numpy()
    assert abs(l1[3] - float(t1)) <= 1e-5 * float(t1)
    _close(dhead.cpu()[..., :A] * 2.0, cls1.grad, 2e-5, 'class-agnostic dcls')


# ---- a17: HunterJr training branch -------------------------------------------------------------------------------------------

def _g12():
    from helpers import load_golden
    return load_golden('g12_hunter_train.npz')


def test_hunter_meta_equals_the_references_unique_chain():
    """pcp_hunter_meta against the reference's _build_meta (two torch.unique + scatter_max / scatter_min) on the g12 fixture: every index
    list bit exact"""
    from pcp_amd import train_ops as tops
    g = _g12()
    pts = torch.from_numpy(g['points']).to(DEV)
    m = tops.hunter_meta(pts, 2, g['gt_boxes'].shape[1], 11, -2, -1)
    mask = g['meta/mask_fg']
    assert m.n_fg == int(mask.sum()) and m.n_local == g['meta/locals_bis'].shape[0] and m.n_inst == g['meta/instance_bi'].shape[0] and m.bad_rows == 0
    assert np.array_equal(m.fg_idx[:m.n_fg].cpu().numpy(), np.nonzero(mask)[0])
    assert np.array_equal(m.fg_local[:m.n_fg].cpu().numpy(), g['meta/locals2fg'])
    assert np.array_equal(m.local_key[:m.n_local].cpu().numpy(), g['meta/locals_bis'])
    assert np.array_equal(m.local_inst[:m.n_local].cpu().numpy(), g['meta/inst2locals'])
    assert np.array_equal(m.inst_key[:m.n_inst].cpu().numpy(), g['meta/instance_bi'])
    assert np.array_equal(m.inst_last[:m.n_inst].cpu().numpy(), g['meta/indices_locals_max_sweep'])
    assert np.array_equal(m.inst_first[:m.n_inst].cpu().numpy(), g['meta/indices_locals_min_sweep'])
    # a cloud without foreground
    bg = pts.clone()
    bg[:, -1] = -1.0
    m0 = tops.hunter_meta(bg, 2, 6, 11, -2, -1)
    assert (m0.n_fg, m0.n_local, m0.n_inst) == (0, 0, 0)


def test_segment_max_and_its_routing():
    from pcp_amd import train_ops as tops
    rng = np.random.RandomState(5)
    n, c, nseg = 3000, 48, 37
    src = torch.from_numpy(rng.randn(n, 64).astype(np.float32))
    src[::7, :10] = 0.0                                       # ties (ReLU zeros): the first row wins, like torch_scatter
    rows = torch.from_numpy(np.sort(rng.choice(n, 900, replace=False)).astype(np.int32))
    seg = torch.from_numpy(rng.randint(0, nseg, 900).astype(np.int32))
    seg[:nseg] = torch.arange(nseg, dtype=torch.int32)        # no empty segment
    out, arg = tops.segment_max(src.to(DEV), seg.to(DEV), nseg, c, row_index=rows.to(DEV), rows=900)
    sel = src[rows.long(), :c]
    ref = torch.full((nseg, c), -np.inf).scatter_reduce(0, seg.long()[:, None].expand(-1, c), sel, 'amax', include_self=True)
    assert torch.equal(out.cpu(), ref)
    a = arg.cpu().long()
    assert torch.equal(sel.gather(0, a), ref)                 # arg rows hold the maxima
    first = torch.full((nseg, c), 900, dtype=torch.long).scatter_reduce(
        0, seg.long()[:, None].expand(-1, c), torch.where(sel == ref[seg.long()], torch.arange(900)[:, None].expand(-1, c), 900), 'amin')
    assert torch.equal(a, first)
    dout = torch.from_numpy(rng.randn(nseg, c).astype(np.float32))
    dsrc = torch.zeros((n, 64), device=DEV)
    tops.segment_max_backward(dout.to(DEV), arg, dsrc, c, row_index=rows.to(DEV))
    ref_d = torch.zeros((n, 64))
    for ch in range(c):
        ref_d[rows.long()[a[:, ch]], ch] += dout[:, ch]
    assert torch.equal(dsrc.cpu(), ref_d)


@pytest.mark.parametrize('case', ['mixed', 'no_positive', 'all_positive', 'ties', 'large'])
def test_hard_mining_loss_and_weights(case):
    """pcp_hunter_losses' hard-mining kernel through the fg-offset term is covered end to end below; here the selection itself: a C-ABI
    call is not exported for it, so the check runs the whole loss entry on a crafted batch where only l_fg_offset varies"""
    from oracle import hunter_train as oht
    rng = np.random.RandomState(3)
    n = {'mixed': 500, 'no_positive': 300, 'all_positive': 50, 'ties': 400, 'large': 70000}[case]
    vals = torch.from_numpy(rng.rand(n).astype(np.float32) * 3)
    pos = torch.from_numpy(rng.rand(n) < {'mixed': 0.2, 'no_positive': 0.0, 'all_positive': 1.1, 'ties': 0.1, 'large': 0.3}[case])
    if case == 'ties':
        vals[~pos] = torch.from_numpy(rng.randint(1, 5, int((~pos).sum())).astype(np.float32)) * 0.5
    v = vals.clone().requires_grad_(True)
    ref = oht.hard_mining(v, pos)
    ref.backward()
    got, w = _hard_mining_via_offsets(vals, pos)
    assert abs(got - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    if case == 'ties':                                        # tied values share the slots: compare the per-value totals
        for val in torch.unique(vals[~pos]):
            sel = (~pos) & (vals == val)
            assert abs(float(w[sel].sum()) - float(v.grad[sel].sum())) < 1e-5
        assert torch.allclose(w[pos], v.grad[pos], atol=1e-7)
    else:
        assert torch.allclose(w, v.grad, atol=1e-7)


def _hard_mining_via_offsets(vals, pos):
    """drive pcp_hunter_losses so that the per-point offset loss equals `vals` (flow = [val, 0, 0] against a zero offset target with
    val < 1 -> smooth-L1 = 0.5 val^2 ... simpler: flow = val + 0.5 with val >= 0 gives |d| - 0.5 = val when |d| >= 1, so use d = val + 0.5
    for val >= 0.5 and sqrt(2 val) below) and the motion flag equals `pos`; returns (l_fg_offset, d loss / d val)."""
    from pcp_amd import lib
    from pcp_amd import train_ops as tops
    n = vals.shape[0]
    d_abs = torch.where(vals >= 0.5, vals + 0.5, torch.sqrt(2 * vals))          # sl1(d) = vals
    # one frame, one instance per motion class, one sweep each: instance 0 static, instance 1 moving
    pts = torch.zeros((n, 8))
    pts[:, 1:4] = torch.rand(n, 3)
    pts[:, 6] = 0.0
    pts[:, 7] = pos.float()
    gt = torch.zeros((1, 2, 8))
    gt[0, :, 7] = 1.0
    itf = torch.zeros((1, 2, 1, 3, 4))
    itf[..., :3, :3] = torch.eye(3)
    itf[0, 1, 0, 0, 3] = 1.0                                   # instance 1 moves: |t| > 0.5
    meta = tops.hunter_meta(pts.to(DEV), 1, 2, 1, -2, -1)
    assert meta.n_fg == n
    head = torch.zeros((n, 16))
    off_t = torch.zeros((n, 3))
    off_t[pos, 0] = 1.0                                        # offset target = R p + t - p = t
    head[:, 3] = off_t[:, 0] + d_abs
    C = 16
    local_feat = torch.zeros((n, C))
    locals_feat = torch.zeros((meta.n_local, C))
    locals_tf = torch.zeros((meta.n_local, 16))
    locals_tf[:, 6] = 1.0
    dev_t = lambda t: t.to(DEV).contiguous()
    ten = dict(pts=dev_t(pts), gt=dev_t(gt), itf=dev_t(itf), head=dev_t(head), lf=dev_t(local_feat), lsf=dev_t(locals_feat), ltf=dev_t(locals_tf),
               dhead=torch.empty((n, 16), device=DEV), dlf=torch.empty((n, C), device=DEV), dlsf=torch.empty((meta.n_local, C), device=DEV),
               dltf=torch.empty((meta.n_local, 16), device=DEV), losses=torch.empty(8, device=DEV), labels=torch.empty(n, dtype=torch.int32, device=DEV))
    d = lib.HunterLoss()
    d.n, d.stride, d.n_fg, d.n_local, d.n_inst, d.c = n, 8, meta.n_fg, meta.n_local, meta.n_inst, C
    d.batch, d.max_inst, d.num_sweeps = 1, 2, 1
    p = lambda t: t.data_ptr()
    d.points, d.gt_boxes, d.instances_tf = p(ten['pts']), p(ten['gt']), p(ten['itf'])
    d.fg_idx, d.fg_local, d.local_key, d.local_inst, d.inst_key = p(meta.fg_idx), p(meta.fg_local), p(meta.local_key), p(meta.local_inst), p(meta.inst_key)
    d.head, d.ld_head, d.local_feat, d.ld_local_feat = p(ten['head']), 16, p(ten['lf']), C
    d.locals_feat, d.ld_locals_feat, d.locals_tf, d.ld_locals_tf = p(ten['lsf']), C, p(ten['ltf']), 16
    d.coef_fg, d.coef_locals, d.grad_scale = 1.0, 1.0, 1.0
    d.dhead, d.ld_dhead, d.dlocal_feat_fg, d.dlocals_feat, d.dlocals_tf, d.ld_dlocals_tf = p(ten['dhead']), 16, p(ten['dlf']), p(ten['dlsf']), p(ten['dltf']), 16
    d.losses, d.labels = p(ten['losses']), p(ten['labels'])
    tops.hunter_losses(d, torch.device(DEV))
    g = ten['dhead'].cpu()[:, 3]                               # = w * sl1'(d) ; sl1'(d) = min(d, 1)
    w = g / torch.clamp(d_abs, max=1.0)
    w = torch.where(d_abs > 0, w, torch.zeros_like(w))
    return float(ten['losses'][2]), w


def test_hunter_losses_and_gradients_on_the_references_predictions():
    """pcp_hunter_losses fed with the reference's own predictions (g12: point logits / flow / embedding, locals_tf) and its own points,
    boxes and instance motions: six of the seven terms against the reference's tb_dict (1e-5), targets against the reference's
    assign_target, the distillation term and every gradient against torch autograd of oracle/hunter_train.py"""
    import json
    from oracle import hunter_train as oht
    from pcp_amd import lib
    from pcp_amd import train_ops as tops
    g = _g12()
    ref_tb = json.loads(str(g['it0_tb_json']))
    pts = torch.from_numpy(g['points'])
    gt, itf = torch.from_numpy(g['gt_boxes']), torch.from_numpy(g['instances_tf'])
    N = pts.shape[0]
    mask_fg = torch.from_numpy(g['meta/mask_fg'])
    meta_ref = {k: torch.from_numpy(g['meta/' + k]) for k in ('locals2fg', 'inst2locals', 'indices_locals_max_sweep', 'locals_bis', 'instance_bi')}
    n_local = meta_ref['locals_bis'].shape[0]
    C = 32
    rng = np.random.RandomState(9)
    cls = torch.from_numpy(g['pred/points_cls_logit']).requires_grad_(True)
    flow = torch.from_numpy(g['pred/points_flow3d']).requires_grad_(True)
    emb = torch.from_numpy(g['pred/points_embedding']).requires_grad_(True)
    ltf = torch.from_numpy(g['pred/locals_tf']).requires_grad_(True)
    lf = torch.from_numpy(rng.randn(N, C).astype(np.float32)).requires_grad_(True)
    lsf = torch.from_numpy(rng.randn(n_local, C).astype(np.float32)).requires_grad_(True)
    # ---- oracle terms with autograd
    tgt = oht.assign_target(pts, mask_fg, gt, itf, meta_ref)
    terms = {}
    terms['l_points_cls'], _ = oht.ce_lovasz(cls, torch.argmax(tgt['points_cls'], dim=1))
    terms['l_points_embed'] = oht._sl1(emb[mask_fg], tgt['fg_embedding']).mean()
    terms['l_fg_offset'] = oht.hard_mining(oht._sl1(flow[mask_fg], tgt['fg_offset']), tgt['points_cls'][mask_fg, 2] > 0)
    mos = tgt['mask_locals_mos']
    terms['l_locals_transl'] = oht.hard_mining(oht._sl1(ltf[:, :3], tgt['locals_tf'][:, :, -1]), mos)
    rot = oht.quat2mat(ltf[:, 3:])
    terms['l_locals_rot'] = oht.hard_mining(torch.linalg.norm(rot - tgt['locals_tf'][:, :, :3], dim=(1, 2), ord='fro'), mos)
    fg_xyz = pts[mask_fg][:, 1:4]
    fg_tf = tgt['locals_tf'][meta_ref['locals2fg']]
    gt_corr = torch.matmul(fg_tf[:, :3, :3], fg_xyz.unsqueeze(-1)).squeeze(-1) + fg_tf[:, :3, -1]
    ptf = torch.cat((rot, ltf[:, :3].unsqueeze(-1)), dim=-1)[meta_ref['locals2fg']]
    corr = torch.matmul(ptf[:, :3, :3], fg_xyz.unsqueeze(-1)).squeeze(-1) + ptf[:, :3, -1]
    terms['l_recon'] = oht.hard_mining(oht._sl1(corr, gt_corr), mos[meta_ref['locals2fg']]) * 0.1
    terms['l_dtl_locals_feat'] = oht._sl1(lf[mask_fg], lsf[meta_ref['locals2fg']]).mean() * 0.1
    for k in ('l_points_cls', 'l_points_embed', 'l_fg_offset', 'l_locals_transl', 'l_locals_rot', 'l_recon'):
        assert abs(float(terms[k]) - ref_tb[k]) <= 1e-5 * abs(ref_tb[k]), (k, float(terms[k]), ref_tb[k])     # the oracle itself
    sum(terms.values()).backward()
    # ---- device
    dev_t = lambda t: t.detach().to(DEV).contiguous()
    meta = tops.hunter_meta(dev_t(pts), 2, gt.shape[1], 11, -2, -1)
    head = torch.zeros((N, 16))
    head[:, 0:3], head[:, 3:6], head[:, 6:8] = cls.detach(), flow.detach(), emb.detach()
    ltf16 = torch.zeros((n_local, 16))
    ltf16[:, :7] = ltf.detach()
    ten = dict(pts=dev_t(pts), gt=dev_t(gt), itf=dev_t(itf), head=dev_t(head), lf=dev_t(lf), lsf=dev_t(lsf), ltf=dev_t(ltf16),
               dhead=torch.full((N, 16), 3.0, device=DEV), dlf=torch.empty((meta.n_fg, C), device=DEV), dlsf=torch.empty((n_local, C), device=DEV),
               dltf=torch.full((n_local, 16), 3.0, device=DEV), losses=torch.empty(8, device=DEV), labels=torch.empty(N, dtype=torch.int32, device=DEV),
               te=torch.empty((meta.n_fg, 2), device=DEV), to=torch.empty((meta.n_fg, 3), device=DEV))
    d = lib.HunterLoss()
    d.n, d.stride, d.n_fg, d.n_local, d.n_inst, d.c = N, pts.shape[1], meta.n_fg, meta.n_local, meta.n_inst, C
    d.batch, d.max_inst, d.num_sweeps = 2, gt.shape[1], 11
    p = lambda t: t.data_ptr()
    d.points, d.gt_boxes, d.instances_tf = p(ten['pts']), p(ten['gt']), p(ten['itf'])
    d.fg_idx, d.fg_local, d.local_key, d.local_inst, d.inst_key = p(meta.fg_idx), p(meta.fg_local), p(meta.local_key), p(meta.local_inst), p(meta.inst_key)
    d.head, d.ld_head, d.local_feat, d.ld_local_feat = p(ten['head']), 16, p(ten['lf']), C
    d.locals_feat, d.ld_locals_feat, d.locals_tf, d.ld_locals_tf = p(ten['lsf']), C, p(ten['ltf']), 16
    d.coef_fg, d.coef_locals, d.grad_scale = 1.0, 1.0, 1.0
    d.dhead, d.ld_dhead, d.dlocal_feat_fg, d.dlocals_feat, d.dlocals_tf, d.ld_dlocals_tf = p(ten['dhead']), 16, p(ten['dlf']), p(ten['dlsf']), p(ten['dltf']), 16
    d.losses, d.labels, d.tgt_embedding, d.tgt_offset = p(ten['losses']), p(ten['labels']), p(ten['te']), p(ten['to'])
    tops.hunter_losses(d, torch.device(DEV))
    got = ten['losses'].cpu().numpy()
    names = ('l_points_cls', 'l_points_embed', 'l_fg_offset', 'l_locals_transl', 'l_locals_rot', 'l_recon', 'l_dtl_locals_feat')
    for i, k in enumerate(names):
        assert abs(got[i] - float(terms[k])) <= 1e-5 * abs(float(terms[k])), (k, got[i], float(terms[k]))
    assert abs(got[7] - sum(float(terms[k]) for k in names)) <= 1e-5 * got[7]
    assert np.array_equal(ten['labels'].cpu().numpy(), np.argmax(g['tgt/points_cls'], axis=1))
    np.testing.assert_allclose(ten['te'].cpu().numpy(), g['tgt/fg_embedding'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(ten['to'].cpu().numpy(), g['tgt/fg_offset'], rtol=0, atol=2e-6)
    dh = ten['dhead'].cpu()
    assert float(dh[:, 8:].abs().max()) == 0.0
    _close(dh[:, 0:3], cls.grad, 2e-4, 'd cls logits')
    _close(dh[:, 3:6], flow.grad, 1e-5, 'd flow')
    _close(dh[:, 6:8], emb.grad, 1e-5, 'd embedding')
    dl = ten['dltf'].cpu()
    assert float(dl[:, 7:].abs().max()) == 0.0
    _close(dl[:, :7], ltf.grad, 2e-5, 'd locals_tf')
    _close(ten['dlf'].cpu(), lf.grad[mask_fg], 1e-5, 'd local_feat (foreground rows)')
    _close(ten['dlsf'].cpu(), lsf.grad, 1e-5, 'd locals_feat')


def test_bev_correction_backward_kernels():
    """pcp_softmax_fuse2_backward, pcp_bev_scatter_mean_backward, pcp_bev_sample_bilinear_backward (incl. the position gradient) against
    torch autograd of oracle/bev.py's bilinear / scatter-mean / blend on a small map"""
    from oracle import bev as obev
    from pcp_amd import ops
    from pcp_amd import train_ops as tops
    rng = np.random.RandomState(21)
    B, H, W, C, N = 2, 12, 10, 32, 700
    pc_min, pix = [-5.0, -6.0], [1.0, 1.0]
    pts = np.zeros((N, 8), np.float32)
    pts[:, 0] = rng.randint(0, B, N)
    pts[:, 1] = rng.uniform(-5.5, 5.5, N)                       # some rows outside the map (dropped by the scatter, clamped by the sampler)
    pts[:, 2] = rng.uniform(-6.5, 6.5, N)
    bev = torch.from_numpy(rng.randn(B, C, H, W).astype(np.float32)).requires_grad_(True)
    flow = torch.from_numpy((rng.randn(N, 2) * 0.3).astype(np.float32)).requires_grad_(True)
    dyn = torch.from_numpy(rng.rand(N) < 0.4)
    logits = torch.from_numpy(rng.randn(B, 2, H, W).astype(np.float32)).requires_grad_(True)
    p0 = torch.from_numpy(pts)
    coord0 = (p0[:, 1:3] - torch.tensor(pc_min)) / torch.tensor(pix)

    def sample(coord):
        feat = bev.new_zeros(N, C)
        for b in range(B):
            mb = p0[:, 0].long() == b
            feat[mb] = obev._bilinear(bev[b].permute(1, 2, 0), coord[mb, 0], coord[mb, 1])
        return feat
    pf = sample(coord0)
    moved = p0[:, 1:3] + flow * dyn[:, None].float()
    coord1 = (moved - torch.tensor(pc_min)) / torch.tensor(pix)
    cf = sample(coord1)
    d = dyn.float()[:, None]
    pf2 = pf * (1 - d) + cf * d
    corrected = obev.bev_scatter_mean(coord1.detach(), p0[:, 0].long(), pf2, (H, W), batch_size=B)
    wts = torch.softmax(logits, dim=1)
    fused = bev * wts[:, [0]] + corrected * wts[:, [1]]
    gout = torch.from_numpy(rng.randn(B, C, H, W).astype(np.float32))
    (fused * gout).sum().backward()
    # ---- device: forward pieces with the product kernels, then the three backward kernels
    cat = torch.zeros((B, H, W, 2 * C), device=DEV)
    cat[..., :C] = bev.detach().permute(0, 2, 3, 1).to(DEV)
    pts_moved = p0.clone()
    pts_moved[:, 1:3] = moved.detach()
    pm = pts_moved.to(DEV).contiguous()
    dyn_d = dyn.to(torch.uint8).to(DEV)
    pf_d = ops.bev_sample_bilinear(cat, p0.to(DEV), pc_min, pix, channels=C)
    pf2_d = pf_d.clone()
    ops.bev_sample_bilinear(cat, pm, pc_min, pix, out=pf2_d, row_mask=dyn_d, channels=C)
    ws = torch.empty(ops._lib.load().pcp_bev_scatter_mean_workspace_bytes(B, H, W, N), dtype=torch.uint8, device=DEV)
    ops.bev_scatter_mean(pm, pf2_d, B, H, W, pc_min, pix, out=cat, out_ch_off=C, workspace=ws)
    np.testing.assert_allclose(cat[..., C:].cpu().numpy(), corrected.detach().permute(0, 2, 3, 1).numpy(), rtol=0, atol=1e-5)
    lg = logits.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    dfused = gout.permute(0, 2, 3, 1).contiguous().to(DEV)
    dcat = torch.empty((B, H, W, 2 * C), device=DEV)
    dlog = torch.empty((B, H, W, 16), device=DEV)
    tops.softmax_fuse2_backward(dfused, cat, lg, C, dcat, dlog)
    _close(dlog[..., :2].cpu(), logits.grad.permute(0, 2, 3, 1), 1e-5, 'd logits')
    assert float(dlog[..., 2:].abs().max()) == 0.0
    dpf = torch.zeros((N, C), device=DEV)
    dcf = torch.zeros((N, C), device=DEV)
    tops.bev_scatter_mean_backward(ws, B, H, W, N, dcat, C, C, dyn_d, dpf, dcf)
    dxyz = torch.zeros((N, 16), device=DEV)
    tops.bev_sample_bilinear_backward(dcf, pm, B, H, W, C, pc_min, pix, dcat, row_mask=dyn_d, bev=cat, dxyz=dxyz, dxyz_ch_off=3)
    tops.bev_sample_bilinear_backward(dpf, p0.to(DEV), B, H, W, C, pc_min, pix, dcat)
    _close(dxyz[:, 3:5].cpu(), flow.grad, 2e-5, 'd flow (position gradient of the re-sampling)')
    assert float(dxyz[:, 5:].abs().max()) == 0.0 and float(dxyz[:, :3].abs().max()) == 0.0
    _close(dcat[..., :C].cpu(), bev.grad.permute(0, 2, 3, 1), 2e-5, 'd bev')


def test_filter_gt_boxes_keeps_in_range_rows_in_order():
    from pcp_amd import train_ops as tops
    g = _g12()
    out = tops.filter_gt_boxes(torch.from_numpy(g['gt_boxes']).to(DEV), [-12.8, -12.8, -8.0, 12.8, 12.8, 0.0]).cpu().numpy()
    ref = g['gt_boxes_after']
    assert np.array_equal(out[:, :ref.shape[1]], ref) and not out[:, ref.shape[1]:].any()


@pytest.mark.parametrize('seed,B,M,S,n,frac', [(1, 1, 3, 11, 700, 0.3), (2, 4, 40, 11, 50000, 0.05), (3, 3, 7, 5, 9000, 0.9), (4, 2, 1, 1, 300, 0.5),
                                               (5, 6, 25, 11, 260000, 0.02)])
def test_hunter_meta_random_clouds_against_the_unique_chain(seed, B, M, S, n, frac):
    """pcp_hunter_meta against oracle/hunter_train.py::build_meta (torch.unique(sorted, return_inverse) twice + arg-max / arg-min sweeps) on
    random (frame, instance, sweep) columns: sparse and crowded key tables, a single-key table, 260 k rows (more than 1 024 scan blocks)"""
    from oracle import hunter_train as oht
    from pcp_amd import train_ops as tops
    rng = np.random.RandomState(seed)
    pts = np.zeros((n, 8), np.float32)
    pts[:, 0] = rng.randint(0, B, n)
    pts[:, 1:4] = rng.randn(n, 3)
    pts[:, 6] = rng.randint(0, S, n)
    inst = rng.randint(0, M, n).astype(np.float32)
    inst[rng.rand(n) > frac] = -1.0
    pts[:, 7] = inst
    p = torch.from_numpy(pts)
    m = tops.hunter_meta(p.to(DEV), B, M, S, -2, -1)
    mask = p[:, -1] > -1
    assert m.n_fg == int(mask.sum()) and m.bad_rows == 0
    if m.n_fg == 0:
        return
    ref = oht.build_meta(p[mask], M, S)
    assert m.n_local == ref['locals_bis'].shape[0] and m.n_inst == ref['instance_bi'].shape[0]
    assert torch.equal(m.fg_idx[:m.n_fg].cpu().long(), torch.nonzero(mask)[:, 0])
    assert torch.equal(m.fg_local[:m.n_fg].cpu().long(), ref['locals2fg'])
    assert torch.equal(m.local_key[:m.n_local].cpu().long(), ref['locals_bis'])
    assert torch.equal(m.local_inst[:m.n_local].cpu().long(), ref['inst2locals'])
    assert torch.equal(m.inst_key[:m.n_inst].cpu().long(), ref['instance_bi'])
    assert torch.equal(m.inst_last[:m.n_inst].cpu().long(), ref['indices_locals_max_sweep'])
    assert torch.equal(m.inst_first[:m.n_inst].cpu().long(), ref['indices_locals_min_sweep'])


def test_hunter_meta_reports_rows_outside_the_key_table():
    from pcp_amd import train_ops as tops
    pts = torch.zeros((10, 8))
    pts[:, 7] = torch.tensor([0, 1, 2, 3, -1, 0, 1, 9, 0, 1.0])         # instance 9 >= max_inst 4
    pts[:, 6] = torch.tensor([0, 1, 2, 3, 0, 11, 1, 1, 0, 1.0])         # sweep 11 >= NUM_SWEEPS 11
    m = tops.hunter_meta(pts.to(DEV), 1, 4, 11, -2, -1)
    assert m.bad_rows == 2 and m.n_fg == 7


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_filter_gt_boxes_random(seed):
    from oracle import hunter_train as oht
    from pcp_amd import train_ops as tops
    rng = np.random.RandomState(seed)
    B, M = 3, 300 if seed == 2 else 17
    gt = (rng.rand(B, M, 8).astype(np.float32) - 0.5) * np.array([130, 130, 12, 5, 3, 2, 6, 0], np.float32)
    gt[..., 2] -= 4.0
    gt[..., 7] = 1.0
    gt[1, M // 2:] = 0.0
    rngp = [-51.2, -51.2, -8.0, 51.2, 51.2, 0.0]
    ref = oht.filter_gt_boxes(torch.from_numpy(gt), rngp).numpy()
    out = tops.filter_gt_boxes(torch.from_numpy(gt).to(DEV), rngp).cpu().numpy()
    assert np.array_equal(out[:, :ref.shape[1]], ref) and not out[:, ref.shape[1]:].any()
