"""Mixed-precision (bf16) training kernels of include/pcp_hip_mp.h against torch CPU references.  The checker computes in float64 on the
SAME bf16-rounded operands the kernel multiplies, so the only differences left are the fp32 accumulation order and the final rounding of a
bf16 output: tolerances 2e-4 of the output scale for fp32 outputs, 1 bf16 ulp (2^-8 relative) for bf16 outputs.  GPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pcp_amd import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _u(seed, col, shape, lo=-1.0, hi=1.0):
    n = int(np.prod(shape))
    return torch.from_numpy(synth.uniform(9100 + seed, col, n, lo, hi).reshape(shape).astype(np.float32))


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _nhwc(x, dtype=torch.float32):
    return x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)


def _check(got, want, out_bf16, what):
    got = got.detach().float().cpu().double()
    want = want.double()
    scale = float(want.abs().max()) + 1e-12
    err = float((got - want).abs().max())
    tol = (2.0 ** -8) * 1.01 + 2e-4 if out_bf16 else 2e-4
    assert err <= tol * scale, '%s: max err %.3e of scale %.3e (tol %.1e)' % (what, err, scale, tol)


CONV_CASES = [
    # cin, cout, stride, (B, H, W), in_bf16, out_bf16, relu        -- which kernel
    (64, 64, 1, (2, 40, 72), True, True, True),        # fast, 8-row items, ragged tile (40 = 5 x 8, 72 = 2.25 x 32)
    (64, 128, 1, (1, 16, 32), True, False, False),     # fast, fp32 output, two channel blocks
    (128, 64, 1, (3, 33, 31), True, True, False),      # fast, 4 slices, odd map
    (32, 64, 1, (1, 8, 32), True, True, True),         # fast, ONE slice per item
    (384, 64, 1, (1, 24, 40), True, True, True),       # fast, 12 slices (CenterHead shared conv)
    (64, 64, 1, (20, 64, 64), True, True, True),       # fast, 16-row items (>= 512 items), several items per workgroup
    (64, 72, 1, (1, 16, 32), True, True, False),       # fast, cout not a multiple of 64 (masked channel groups)
    (64, 64, 2, (2, 32, 48), True, True, True),        # general kernel, stride 2
    (64, 128, 2, (1, 24, 40), False, True, False),     # general kernel, fp32 input (the canvas), stride 2
    (16, 64, 1, (1, 19, 23), False, False, True),      # general kernel, cin = 16, fp32 in and out
    (48, 64, 1, (1, 16, 32), True, True, False),       # general kernel, cin % 32 == 16 with bf16 input
]


@pytest.mark.parametrize('cin,cout,stride,shape,in_bf16,out_bf16,relu', CONV_CASES)
def test_mp_conv3x3_forward_and_data_gradient_forms(cin, cout, stride, shape, in_bf16, out_bf16, relu):
    import ctypes
    from pcp_amd import lib, train_ops as tops
    B, H, W = shape
    x = _u(1, cin + cout, (B, cin, H, W))
    w = _u(2, cin * 3 + cout, (cout, cin, 3, 3), -0.2, 0.2)
    bias = _u(3, cout, (cout,))
    xq, wq = _bf(x), _bf(w)
    want = F.conv2d(xq.double(), wq.double(), bias.double(), stride=stride, padding=1)
    if relu:
        want = want.clamp_min(0)
    packed, opad = tops.mp_pack_conv3x3(w.to(DEV))
    bp = torch.zeros(opad, device=DEV)
    bp[:cout] = bias.to(DEV)
    xin = _nhwc(x, torch.bfloat16 if in_bf16 else torch.float32)
    d = lib.MpConv3x3(B, H, W, cin, cout, opad, stride, cin, cout, 1 if relu else 0, lib.DT_BF16 if in_bf16 else lib.DT_F32,
                      lib.DT_BF16 if out_bf16 else lib.DT_F32)
    fast = ctypes.c_int32(-1)
    lib.load().pcp_mp_conv3x3_plan(ctypes.byref(d), ctypes.byref(fast), None)
    assert fast.value == (1 if (stride == 1 and in_bf16 and cin % 32 == 0) else 0)
    out = tops.mp_conv3x3(xin, packed, bp, cin, cout, opad, stride=stride, relu=relu, out_dtype=torch.bfloat16 if out_bf16 else torch.float32)
    torch.cuda.synchronize()
    _check(out.permute(0, 3, 1, 2), want, out_bf16, 'forward')
    if stride == 1 and cout % 16 == 0:
        # data gradient = the same kernel with the transposed / flipped weight form: dx = conv_transpose(dy, w)
        dy = _u(4, cin + 7, (B, cout, H, W))
        dyq = _bf(dy)
        want_dx = F.conv_transpose2d(dyq.double(), wq.double(), stride=1, padding=1)
        packed_t, ipad = tops.mp_pack_conv3x3(w.to(DEV), transpose=True)
        zb = torch.zeros(ipad, device=DEV)
        dx = tops.mp_conv3x3(_nhwc(dy, torch.bfloat16 if in_bf16 else torch.float32), packed_t, zb, cout, cin, ipad, stride=1, relu=False,
                             out_dtype=torch.bfloat16 if out_bf16 else torch.float32)
        torch.cuda.synchronize()
        _check(dx.permute(0, 3, 1, 2), want_dx, out_bf16, 'data gradient')


def test_mp_conv3x3_channel_windows_of_wider_buffers():
    """input read from a channel window of a wider bf16 buffer, output written into a window of a wider buffer; the rest is untouched"""
    from pcp_amd import train_ops as tops
    B, H, W, cin, cout = 2, 16, 40, 64, 64
    x = _u(11, 1, (B, cin, H, W))
    w = _u(12, 2, (cout, cin, 3, 3), -0.2, 0.2)
    want = F.conv2d(_bf(x).double(), _bf(w).double(), None, padding=1)
    packed, opad = tops.mp_pack_conv3x3(w.to(DEV))
    wide_in = torch.full((B, H, W, 96), 7.0, dtype=torch.bfloat16, device=DEV)
    wide_in[..., 16:80] = _nhwc(x, torch.bfloat16)
    wide_out = torch.full((B, H, W, 128), -3.0, dtype=torch.bfloat16, device=DEV)
    tops.mp_conv3x3(wide_in, packed, torch.zeros(opad, device=DEV), cin, cout, opad, out=wide_out, in_ch_off=16, out_ch_off=32)
    torch.cuda.synchronize()
    _check(wide_out[..., 32:96].permute(0, 3, 1, 2), want, True, 'window')
    assert float((wide_out[..., :32].float() + 3.0).abs().max()) == 0.0 and float((wide_out[..., 96:].float() + 3.0).abs().max()) == 0.0


WGRAD_CASES = [
    # cin, cout, stride, (B, H, W)
    (64, 64, 1, (2, 24, 40)),          # one pair, ragged strip (40 = 1.25 x 32), 24 rows = 6 stages
    (128, 64, 1, (1, 16, 32)),         # two ci blocks
    (64, 128, 1, (3, 33, 31)),         # two co blocks, odd map, partial last stage
    (64, 64, 2, (2, 32, 48)),          # stride 2 (even / odd column images)
    (128, 256, 2, (1, 16, 64)),        # stride 2, several pairs
    (64, 16, 1, (1, 12, 20)),          # cout < 64 (the padded head maps): rows 16..63 of the tile are discarded
    (320, 16, 1, (1, 8, 32)),          # cin = 5 x 64 (the five CenterHead branches side by side)
    (64, 64, 1, (4, 128, 128)),        # several row splits per strip
]


@pytest.mark.parametrize('cin,cout,stride,shape', WGRAD_CASES)
def test_mp_conv3x3_wgrad(cin, cout, stride, shape):
    from pcp_amd import train_ops as tops
    B, H, W = shape
    x = _u(21, cin, (B, cin, H, W))
    dy = _u(22, cout, (B, cout, H // stride, W // stride))
    xq, dyq = _bf(x).double().requires_grad_(False), _bf(dy).double()
    wref = torch.zeros((cout, cin, 3, 3), dtype=torch.float64, requires_grad=True)
    F.conv2d(xq, wref, None, stride=stride, padding=1).backward(dyq)
    want = wref.grad
    dw = torch.full((cout, cin, 3, 3), 5.0, dtype=torch.float32, device=DEV)
    tops.mp_conv3x3_wgrad(_nhwc(x, torch.bfloat16), _nhwc(dy, torch.bfloat16), cin, cout, stride, dw)
    torch.cuda.synchronize()
    _check(dw, want, False, 'dw')
    before = dw.clone()
    tops.mp_conv3x3_wgrad(_nhwc(x, torch.bfloat16), _nhwc(dy, torch.bfloat16), cin, cout, stride, dw, accumulate=True)
    torch.cuda.synchronize()
    assert torch.equal(dw, before + before)                 # accumulate adds the same (bitwise reproducible) sums


def test_mp_wgrad_channel_windows():
    from pcp_amd import train_ops as tops
    B, H, W, cin, cout = 1, 16, 32, 64, 64
    x = _u(31, 1, (B, cin, H, W))
    dy = _u(32, 2, (B, cout, H, W))
    wref = torch.zeros((cout, cin, 3, 3), dtype=torch.float64, requires_grad=True)
    F.conv2d(_bf(x).double(), wref, None, padding=1).backward(_bf(dy).double())
    wx = torch.full((B, H, W, 128), 9.0, dtype=torch.bfloat16, device=DEV)
    wx[..., 64:] = _nhwc(x, torch.bfloat16)
    wdy = torch.full((B, H, W, 72), 9.0, dtype=torch.bfloat16, device=DEV)
    wdy[..., 8:] = _nhwc(dy, torch.bfloat16)
    dw = torch.zeros((cout, cin, 3, 3), device=DEV)
    tops.mp_conv3x3_wgrad(wx, wdy, cin, cout, 1, dw, x_ch_off=64, dy_ch_off=8)
    torch.cuda.synchronize()
    _check(dw, wref.grad, False, 'dw windows')


@pytest.mark.parametrize('xd,dd,od', [(torch.bfloat16, torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32, torch.bfloat16),
                                       (torch.float32, torch.bfloat16, torch.float32), (torch.bfloat16, torch.bfloat16, torch.float32)])
def test_mp_batchnorm_relu_forward_backward_storage_types(xd, dd, od):
    """training-mode BatchNorm + ReLU with bf16 / fp32 storage per tensor against float64 torch on the same stored values"""
    from pcp_amd import train_ops as tops
    c, rows = 64, 3000
    x = _u(41, 1, (rows, c), -2.0, 3.0)
    dy = _u(42, 2, (rows, c))
    gamma, beta = _u(43, 3, (c,), 0.5, 1.5), _u(44, 4, (c,), -0.5, 0.5)
    xs = x.to(xd)
    dys = dy.to(dd)
    xr = xs.double().requires_grad_(True)
    bn = F.batch_norm(xr, None, None, gamma.double(), beta.double(), True, 0.0, 1e-3)
    act = bn.clamp_min(0)
    act.backward(dys.double())
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    xg = xs.to(DEV).contiguous()
    vec = tops.bn_train_stats(xg, c, gamma.to(DEV), beta.to(DEV), 1e-3, 0.01, rm, rv)
    out = torch.empty((rows, c), dtype=od, device=DEV)
    tops.scale_shift_act(xg, c, vec, True, out)
    dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    dx = torch.empty((rows, c), dtype=od, device=DEV)
    tops.bn_act_backward(dys.to(DEV).contiguous(), xg, c, vec, True, dg, db, dx=dx)
    torch.cuda.synchronize()
    _check(out, act.detach(), od == torch.bfloat16, 'activation')
    _check(dx, xr.grad, od == torch.bfloat16, 'dx')
    # dgamma / dbeta: the ReLU mask is decided on fp32 fma(x, scale, shift), torch's on float64 -- elements within rounding of zero may flip
    np.testing.assert_allclose(db.cpu().double().numpy(), (dys.double() * (act.detach() > 0)).sum(0).numpy(), rtol=0, atol=2e-3 * rows ** 0.5)
    want_mean = xs.double().mean(0)
    np.testing.assert_allclose(vec.mean.cpu().double().numpy(), want_mean.numpy(), rtol=0, atol=1e-6)


def test_mp_colsum_accumulate_dilate_storage_types():
    from pcp_amd import train_ops as tops
    x = _u(51, 1, (2, 6, 10, 32))
    xb = x.to(torch.bfloat16).to(DEV)
    out = torch.zeros(32, device=DEV)
    tops.colsum(xb, 32, out)
    np.testing.assert_allclose(out.cpu().numpy(), xb.float().cpu().reshape(-1, 32).double().sum(0).numpy(), rtol=0, atol=1e-4)
    d = tops.dilate2x(xb, 32)
    assert d.dtype == torch.bfloat16 and tuple(d.shape) == (2, 12, 20, 32)
    assert torch.equal(d[:, ::2, ::2], xb) and float(d[:, 1::2].float().abs().max()) == 0.0 and float(d[:, :, 1::2].float().abs().max()) == 0.0
    a32 = _u(52, 2, (120, 32)).to(DEV)
    want = a32 + 0.5 * xb.float().reshape(-1, 32)
    tops.accumulate(a32, xb.reshape(-1, 32), 32, alpha=0.5)                    # fp32 += bf16
    assert float((a32 - want).abs().max()) <= 1e-6
    ab = _u(53, 3, (120, 32)).to(torch.bfloat16).to(DEV)
    want_b = (ab.float() + xb.float().reshape(-1, 32)).to(torch.bfloat16)
    tops.accumulate(ab, xb.reshape(-1, 32), 32)                                # bf16 += bf16 (fp32 add, one rounding)
    torch.cuda.synchronize()
    assert torch.equal(ab, want_b)


def test_mp_grouped_weight_pack_equals_the_per_layer_pack():
    """pcp_mp_pack_conv3x3_group (one launch for many layers, both forms) writes the bits of per-layer pcp_mp_pack_conv3x3 calls"""
    import torch.nn as nn
    from pcp_amd import train_ops as tops
    convs = [nn.Conv2d(64, 64, 3, bias=False), nn.Conv2d(64, 128, 3, bias=False), nn.Conv2d(384, 64, 3), nn.Conv2d(128, 48, 3)]
    grp = tops.MpPackGroup()
    single = {}
    for i, c in enumerate(convs):
        c.weight.data = _u(60 + i, 1, tuple(c.weight.shape), -0.3, 0.3)
        c.to(DEV)
        for tr in (False, True):
            if tr and c.weight.shape[0] % 16:
                continue
            ref, opad = tops.mp_pack_conv3x3(c.weight.detach(), tr)
            single[(i, tr)] = ref
            dst = torch.full_like(ref, 7.0)
            grp.add((i, tr), c, c.weight.detach(), tr, dst, opad)
    grp.run(torch.device(DEV))
    torch.cuda.synchronize()
    for key, (_o, _j, (w, dst)) in grp.jobs.items():
        assert torch.equal(dst.view(torch.int16), single[key].view(torch.int16)), key


def test_mp_sparse_first_layer_bf16_output_and_bf16_canvas_kernels():
    """storage-typed variants used by the bf16 loop outside the conv kernels: the sparse first layer writing bf16 = its fp32 output rounded
    once; the PFN's last training kernel writing a bf16 canvas = the fp32 canvas rounded; the canvas-gradient gather reading bf16"""
    from pcp_amd import ops, pack, train_ops as tops
    pts = np.concatenate([synth.agent_cloud(agent=a, n_points=6000, layout='car') for a in range(2)], 0)
    points = torch.from_numpy(synth.collate([pts])).to(DEV)
    grid = ops.make_grid([-51.2, -51.2, -8.0, 51.2, 51.2, 0.0], [0.2, 0.2, 8.0], [512, 512, 1], 1)
    vox = ops.voxelize(points, grid, want_inverse=False, want_counts=False)
    P = int(vox.counters[0].item())
    pf = _u(71, 1, (points.shape[0], 64)).to(DEV).contiguous()
    w = _u(72, 2, (64, 64, 3, 3), -0.2, 0.2)
    b = _u(73, 3, (64,))
    wsp, bsp = pack.pack_conv3x3_sparse_s2(w, b)
    o32 = ops.sparse_conv3x3_s2(pf, vox, wsp.to(DEV), bsp.to(DEV), 64, relu=True)
    o16 = ops.sparse_conv3x3_s2(pf, vox, wsp.to(DEV), bsp.to(DEV), 64, relu=True, out_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    assert o16.dtype == torch.bfloat16 and torch.equal(o16, o32.to(torch.bfloat16)) and float(o32.abs().max()) > 0
    # PFN training tail: canvas in both storage types from the same x1
    Nk = int(vox.counters[1].item())
    x1 = _u(74, 4, (Nk, 64)).to(DEV).contiguous()
    vec = tops.BNVectors(64, torch.device(DEV))
    vec.scale.fill_(0.9)
    vec.shift.fill_(0.05)
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        canvas = torch.zeros((1, 512, 512, 64), dtype=dt, device=DEV)
        pfo = torch.empty((max(P, 1), 64), device=DEV)
        arg1 = torch.empty((max(P, 1), 64), dtype=torch.int32, device=DEV)
        tops.pfn_train_out(vox, x1, vec, pfo, arg1, canvas)
        outs[dt] = (canvas, arg1)
    torch.cuda.synchronize()
    assert torch.equal(outs[torch.bfloat16][0], outs[torch.float32][0].to(torch.bfloat16)) and float(outs[torch.float32][0].abs().max()) > 0
    dcan = _u(75, 5, (1, 512, 512, 64)).to(DEV)
    dz_a, dz_b = torch.empty((Nk, 64), device=DEV), torch.empty((Nk, 64), device=DEV)
    tops.pfn_train_route_out_grad(vox, Nk, outs[torch.float32][1], dz_a, dcanvas=dcan.to(torch.bfloat16).float().contiguous())
    tops.pfn_train_route_out_grad(vox, Nk, outs[torch.float32][1], dz_b, dcanvas=dcan.to(torch.bfloat16).contiguous())
    torch.cuda.synchronize()
    assert torch.equal(dz_a, dz_b) and float(dz_a.abs().max()) > 0


def test_mp_pfn_training_kernels_with_bf16_point_rows():
    """the per-point 64-channel rows around the second PFN Linear as bf16: pfn_train_mid writes the fp32 rows rounded once; pfn_train_out,
    the routing of the canvas gradient and the routing through the pillar max read / write bf16 rows with the results of the fp32 kernels
    on the same (rounded) values"""
    from pcp_amd import ops, train_ops as tops
    pts = np.concatenate([synth.agent_cloud(agent=a, n_points=5000, layout='car') for a in range(2)], 0)
    points = torch.from_numpy(synth.collate([pts])).to(DEV)
    grid = ops.make_grid([-51.2, -51.2, -8.0, 51.2, 51.2, 0.0], [0.2, 0.2, 8.0], [512, 512, 1], 1)
    vox = ops.voxelize(points, grid, want_inverse=False, want_counts=False)
    P, Nk = int(vox.counters[0].item()), int(vox.counters[1].item())
    dev = torch.device(DEV)
    x0 = _u(81, 1, (Nk, 32)).to(DEV).contiguous()
    vec0 = tops.BNVectors(32, dev)
    vec0.scale.fill_(1.1)
    vec0.shift.fill_(0.02)
    in1 = {dt: torch.empty((Nk, 64), dtype=dt, device=DEV) for dt in (torch.float32, torch.bfloat16)}
    arg0 = {dt: torch.empty((max(P, 1), 32), dtype=torch.int32, device=DEV) for dt in in1}
    for dt in in1:
        tops.pfn_train_mid(vox, x0, vec0, in1[dt], arg0[dt])
    torch.cuda.synchronize()
    assert torch.equal(in1[torch.bfloat16], in1[torch.float32].to(torch.bfloat16)) and torch.equal(arg0[torch.float32], arg0[torch.bfloat16])
    assert float(in1[torch.float32].abs().max()) > 0
    # last kernel: bf16 x1 == fp32 kernel on the rounded x1
    x1 = _u(82, 2, (Nk, 64)).to(DEV).to(torch.bfloat16).contiguous()
    vec1 = tops.BNVectors(64, dev)
    vec1.scale.fill_(0.9)
    vec1.shift.fill_(0.05)
    res = {}
    for xin in (x1, x1.float()):
        canvas = torch.zeros((1, 512, 512, 64), dtype=torch.bfloat16, device=DEV)
        pfo = torch.empty((max(P, 1), 64), device=DEV)
        arg1 = torch.empty((max(P, 1), 64), dtype=torch.int32, device=DEV)
        tops.pfn_train_out(vox, xin, vec1, pfo, arg1, canvas)
        res[xin.dtype] = (canvas, pfo[:P].clone(), arg1[:P].clone())
    torch.cuda.synchronize()
    for a, b in zip(res[torch.bfloat16], res[torch.float32]):
        assert torch.equal(a, b)
    # canvas gradient -> dz1 as bf16 == the fp32 routing rounded; routing through the pillar max from bf16 din1 == fp32 kernel on the rounded din1
    dcan = _u(83, 3, (1, 512, 512, 64)).to(DEV).to(torch.bfloat16).contiguous()
    dz = {dt: torch.empty((Nk, 64), dtype=dt, device=DEV) for dt in (torch.float32, torch.bfloat16)}
    for dt in dz:
        tops.pfn_train_route_out_grad(vox, Nk, res[torch.float32][2], dz[dt], dcanvas=dcan)
    din1 = _u(84, 4, (Nk, 64)).to(DEV).to(torch.bfloat16).contiguous()
    da = {}
    for xin in (din1, din1.float()):
        out = torch.empty((Nk, 32), device=DEV)
        tops.pfn_train_route_mid_grad(vox, xin, arg0[torch.float32], out)
        da[xin.dtype] = out
    torch.cuda.synchronize()
    assert torch.equal(dz[torch.bfloat16], dz[torch.float32].to(torch.bfloat16)) and float(dz[torch.float32].abs().max()) > 0
    assert torch.equal(da[torch.bfloat16], da[torch.float32]) and float(da[torch.float32].abs().max()) > 0


# ---- the pointwise family (pcp_mp_pointwise, pcp_mp_pointwise_wgrad) --------------------------------------------------------------------------

PW_CASES = [
    # kind, cin, cout, (B, H, W), in_bf16, out_bf16, relu
    ('plain', 64, 128, (1, 25, 40), True, True, True),          # 1000 rows: a ragged last pixel tile
    ('plain', 256, 64, (2, 16, 16), False, False, False),       # fp32 in and out (converted while staged)
    ('plain', 128, 72, (1, 9, 32), True, False, True),          # cout not a multiple of 64: masked channel groups
    ('s2d', 64, 128, (2, 32, 48), True, True, True),            # Conv2d k2 s2
    ('s2d', 128, 128, (1, 16, 24), False, True, False),
    ('d2s', 128, 128, (2, 16, 24), True, True, True),           # ConvTranspose2d k2 s2
    ('d2s', 256, 64, (1, 8, 20), True, False, False),
]


def _pw_layer(kind, cin, cout):
    if kind == 'plain':
        return torch.nn.Conv2d(cin, cout, 1, bias=True)
    if kind == 's2d':
        return torch.nn.Conv2d(cin, cout, 2, stride=2, bias=True)
    return torch.nn.ConvTranspose2d(cin, cout, 2, stride=2, bias=True)


def _pw_pack(kind, w, b):
    from pcp_amd import pack
    f = {'plain': lambda: pack.pack_plain(w.reshape(w.shape[0], -1), b), 's2d': lambda: pack.pack_conv2x2_s2(w, b),
         'd2s': lambda: pack.pack_convT2x2_s2(w, b)}[kind]
    wp, bp, cp = f()
    return wp.to(DEV).to(torch.bfloat16).contiguous(), bp.to(DEV).float().contiguous(), cp


@pytest.mark.parametrize('kind,cin,cout,shape,in_bf16,out_bf16,relu', PW_CASES)
def test_mp_pointwise_forward_modes_and_storage_types(kind, cin, cout, shape, in_bf16, out_bf16, relu):
    from pcp_amd import lib, train_ops as tops
    B, H, W = shape
    x = _u(31, cin + cout, (B, cin, H, W))
    layer = _pw_layer(kind, cin, cout)
    w = _u(32, cin * 3 + cout, tuple(layer.weight.shape), -0.2, 0.2)
    bias = _u(33, cout, (cout,))
    with torch.no_grad():
        layer.weight.copy_(_bf(w))
        layer.bias.copy_(bias)
        want = layer.double()(_bf(x).double())
        if relu:
            want = want.clamp_min(0)
    wp, bp, cp = _pw_pack(kind, w, bias)
    mode = {'plain': lib.PW_PLAIN, 's2d': lib.PW_SPACE2DEPTH, 'd2s': lib.PW_DEPTH2SPACE}[kind]
    xin = _nhwc(x, torch.bfloat16 if in_bf16 else torch.float32)
    got = tops.mp_pointwise(xin, wp, bp, mode, cin, cout, cp, relu=relu, out_dtype=torch.bfloat16 if out_bf16 else torch.float32)
    assert got.dtype == (torch.bfloat16 if out_bf16 else torch.float32)
    _check(got.permute(0, 3, 1, 2), want, out_bf16, 'mp_pointwise %s %d->%d' % (kind, cin, cout))


def test_mp_pointwise_channel_windows_of_wider_buffers():
    from pcp_amd import lib, train_ops as tops
    cin, cout, B, H, W = 64, 128, 2, 12, 20
    x = _u(41, 1, (B, cin, H, W))
    w = _u(42, 2, (cout, cin, 1, 1), -0.2, 0.2)
    bias = _u(43, 3, (cout,))
    want = F.conv2d(_bf(x).double(), _bf(w).double(), bias.double())
    wp, bp, cp = _pw_pack('plain', w, bias)
    wide_in = torch.full((B, H, W, cin + 32), 7.0, dtype=torch.bfloat16, device=DEV)
    wide_in[..., 16:16 + cin] = _nhwc(x, torch.bfloat16)
    wide_out = torch.full((B, H, W, cout + 64), -3.0, dtype=torch.bfloat16, device=DEV)
    tops.mp_pointwise(wide_in, wp, bp, lib.PW_PLAIN, cin, cout, cp, relu=False, out=wide_out, in_ch_off=16, out_ch_off=32)
    _check(wide_out[..., 32:32 + cout].permute(0, 3, 1, 2), want, True, 'windowed mp_pointwise')
    assert float(wide_out[..., :32].float().min()) == -3.0 and float(wide_out[..., 32 + cout:].float().max()) == -3.0


PWG_CASES = [
    # kind, cin, cout, (B, H, W) of the layer input
    ('plain', 64, 128, (2, 24, 40)),
    ('plain', 256, 64, (1, 16, 16)),
    ('plain', 72, 40, (1, 13, 29)),             # channels that are not multiples of 64, ragged rows
    ('s2d', 64, 128, (2, 32, 48)),
    ('d2s', 128, 64, (2, 16, 24)),
]


@pytest.mark.parametrize('kind,cin,cout,shape', PWG_CASES)
def test_mp_pointwise_wgrad_of_the_three_layer_kinds(kind, cin, cout, shape):
    from pcp_amd import train_ops as tops
    B, H, W = shape
    layer = _pw_layer(kind, cin, cout).double()
    x = _u(51, cin, (B, cin, H, W))
    xr = _bf(x).double().requires_grad_(False)
    y = layer(xr)
    dy = _u(52, cout, tuple(y.shape))
    (y * _bf(dy).double()).sum().backward()
    want = layer.weight.grad                                     # plain / s2d: (cout, cin, k, k); d2s: (cin, cout, 2, 2)
    xd, dyd = _nhwc(x, torch.bfloat16), _nhwc(dy, torch.bfloat16)
    rows = B * H * W
    if kind == 'plain':
        got = torch.empty((cout, cin), dtype=torch.float32, device=DEV)
        tops.pointwise_wgrad(tops.rowmap(dyd, cout), tops.rowmap(xd, cin), rows, got)
        got = got.reshape(cout, cin, 1, 1)
    elif kind == 's2d':
        Ho, Wo = H // 2, W // 2
        tmp = torch.empty((4, cout, cin), dtype=torch.float32, device=DEV)
        for tap in range(4):
            tops.pointwise_wgrad(tops.rowmap(dyd, cout), tops.rowmap(xd, cin, lattice=(Ho, Wo, tap // 2, tap % 2)), B * Ho * Wo, tmp[tap])
        got = tmp.permute(1, 2, 0).reshape(cout, cin, 2, 2)
    else:
        tmp = torch.empty((4, cin, cout), dtype=torch.float32, device=DEV)
        for tap in range(4):
            tops.pointwise_wgrad(tops.rowmap(xd, cin), tops.rowmap(dyd, cout, lattice=(H, W, tap // 2, tap % 2)), rows, tmp[tap])
        got = tmp.permute(1, 2, 0).reshape(cin, cout, 2, 2)
    _check(got, want, False, 'mp_pointwise_wgrad %s' % kind)
    # accumulate: a second call adds the same gradient
    if kind == 'plain':
        acc = got.reshape(cout, cin).clone()
        tops.pointwise_wgrad(tops.rowmap(dyd, cout), tops.rowmap(xd, cin), rows, acc, accumulate=True)
        _check(acc.reshape(cout, cin, 1, 1), 2 * want, False, 'accumulating mp_pointwise_wgrad')


# ---- the fusion module's stacked maps as bf16 (pcp_mp_warp_nearest, pcp_mp_softmax_fuse, pcp_mp_disco_fuse_backward) ---------------------------

def test_mp_fusion_kernels_on_bf16_maps_equal_the_fp32_kernels_on_the_rounded_maps():
    """the storage-typed forms change the map loads / the warp's stores only: a bf16 source warped to bf16 is the fp32 warp of the same
    values (exact: nearest-neighbour copies); softmax-weighted fusion and its backward over bf16 maps are bit for bit the fp32 kernels over the
    same (rounded) maps"""
    from pcp_amd import ops, train_ops as tops
    H, W, C, n = 24, 40, 128, 4
    th = [0.92, 0.11, 0.05, -0.09, 0.97, -0.04]
    src = _u(91, 1, (H, W, C)).to(DEV)
    src_b = src.to(torch.bfloat16)
    want = torch.zeros((H, W, 2 * C), device=DEV)
    ops.warp_nearest(src_b.float().contiguous(), want, th, C, dst_ch_off=C)
    for sdt, ddt in ((torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.float32)):
        s_in = src_b if sdt == torch.bfloat16 else src_b.float().contiguous()
        dst = torch.full((H, W, 2 * C), 5.0, dtype=ddt, device=DEV)
        ops.warp_nearest(s_in, dst, th, C, dst_ch_off=C)
        torch.cuda.synchronize()
        assert torch.equal(dst[..., C:].float(), want[..., C:]) and float(dst[..., :C].float().min()) == 5.0
    assert float(want[..., C:].abs().max()) > 0
    # fusion forward / backward: maps = channel windows [C, 2C) of n (B, H, W, 2C) buffers, as FusionTrain lays them out
    B = 2
    cats_b = [_u(92 + a, 2, (B, H, W, 2 * C)).to(DEV).to(torch.bfloat16).contiguous() for a in range(n)]
    cats_f = [c.float().contiguous() for c in cats_b]
    logits = _u(97, 3, (B, H, W, 8)).to(DEV).contiguous()
    out_f, out_b = torch.empty((B, H, W, C), device=DEV), torch.empty((B, H, W, C), device=DEV)
    ops.softmax_fuse_raw([c.data_ptr() + 4 * C for c in cats_f], logits, C, 2 * C, out_f)
    ops.softmax_fuse_raw([c.data_ptr() + 2 * C for c in cats_b], logits, C, 2 * C, out_b, map_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    assert torch.equal(out_f, out_b) and float(out_f.abs().max()) > 0
    dfused = _u(98, 4, (B, H, W, C)).to(DEV).contiguous()
    h2 = [_u(99 + a, 5, (B, H, W, 16), 0.0, 1.0).to(DEV).contiguous() for a in range(n)]
    w4 = _u(105, 6, (16,), -0.5, 0.5).to(DEV)
    res = []
    for cats, esz, mdt in ((cats_f, 4, torch.float32), (cats_b, 2, torch.bfloat16)):
        d_ego = torch.empty((B, H, W, C), device=DEV)
        dh2 = [torch.empty((B, H, W, 16), device=DEV) for _ in range(n)]
        dw4, db4 = torch.zeros(16, device=DEV), torch.zeros(1, device=DEV)
        tops.disco_fuse_backward([c.data_ptr() + esz * C for c in cats], 2 * C, C, logits, dfused, h2, w4, d_ego, dh2, dw4, db4, map_dtype=mdt)
        torch.cuda.synchronize()
        res.append([d_ego] + dh2)
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert float(res[0][0].abs().max()) > 0 and float(res[0][1].abs().max()) > 0
