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
    for algo in (('direct', 'winograd') if kind == 'c3s1' else ('auto',)):
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
