"""Dispatch invariance.  `PackedConv.run` picks a layer's kernel from the LAUNCH size (pcdet/models/convnet.py): the same
layer runs the direct kernel at one frame, fused Winograd F(4x4) at four, the through-memory F(4x4) GEMM at twenty.  Every choice must be
the same convolution to fp32 rounding:
  * every distinct conv layer shape of the five BASELINE configs (traced from the models themselves), at B in {1, 4, 20} under auto dispatch,
    against torch-CPU float32 (oneDNN) at 2e-4 of the output scale -- and the kernels auto picked are recorded, so the test fails if the
    batch sizes stop exercising more than one kernel per shape family;
  * the well-conditioned fixtures (g13): the final detection set of a frame is the same whether the frame is run alone (B = 1) or as one of
    four / twenty copies in a batch (different kernels per layer), exactly the reference's set."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from helpers import assert_same_final_set, load_golden
from pcp_amd import synth

pytestmark = pytest.mark.gpu
YAMLS = ['v2x_pointpillar_basic_car.yaml', 'v2x_pointpillar_basic_ego.yaml', 'v2x_pointpillar_basic_ego_early.yaml', 'v2x_pointpillar_disco.yaml']


def _trace_layer_shapes():
    """run each config's model once (one frame of 60 000 points per agent) and record, for every PackedConv launch, the layer's kind and the map it
    was applied to: {(kind, cin, cout, stride, relu, H, W)}"""
    import bench
    from pcdet.models import convnet
    shapes = set()
    orig = convnet.PackedConv.run

    def spy(self, x, out=None, in_ch_off=0, out_ch_off=0):
        shapes.add((self.kind, int(self.cin), int(self.cout), int(self.stride), bool(self.relu), int(x.shape[1]), int(x.shape[2])))
        return orig(self, x, out=out, in_ch_off=in_ch_off, out_ch_off=out_ch_off)
    convnet.PackedConv.run = spy
    try:
        for name, conf in (('car', bench.CONFIGS['car']), ('ego', bench.CONFIGS['ego']), ('early', bench.CONFIGS['early']), ('disco', bench.CONFIGS['disco'])):
            cfg = bench.load_cfg(conf['yaml'])
            model, _state, _ds = bench.build_model(cfg)
            model = model.cuda().eval()
            pts, metas = bench.make_points(conf, 1, 0)
            with torch.no_grad():
                model({'points': torch.from_numpy(pts).cuda(), 'batch_size': 1, 'metadata': metas})
            del model
            torch.cuda.empty_cache()
    finally:
        convnet.PackedConv.run = orig
    return sorted(shapes)


_SHAPES = None


def _shapes():
    global _SHAPES
    if _SHAPES is None:
        _SHAPES = _trace_layer_shapes()
    return _SHAPES


def _module_for(kind, cin, cout, stride):
    if kind == '3x3':
        return nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=True)
    if kind == 'plain':
        return nn.Conv2d(cin, cout, 1, bias=True)
    if kind == 's2d':
        return nn.Conv2d(cin, cout, 2, stride=2, bias=True)
    if kind == 'd2s':
        return nn.ConvTranspose2d(cin, cout, 2, stride=2, bias=True)
    raise AssertionError(kind)


def test_every_layer_shape_of_the_five_configs_at_batch_1_4_20_under_auto_dispatch(monkeypatch):
    monkeypatch.delenv('PCP_CONV_ALGO', raising=False)
    from pcdet.models import convnet
    from pcp_amd import ops
    shapes = _shapes()
    # the trace must have seen the whole path: the backbone's three resolutions, the head, HunterJr's 768-wide layer, the fusion's compressor
    assert any(s[0] == '3x3' and s[3] == 2 for s in shapes) and any(s[1] == 768 for s in shapes) and any(s[0] == 'd2s' for s in shapes)
    assert any(s[:3] == ('3x3', 384, 128) for s in shapes) and len(shapes) >= 20, shapes
    picked = {}
    names = ('conv3x3', 'conv3x3_winograd', 'conv3x3_winograd4', 'conv3x3_winograd4f', 'conv3x3_winograd4h', 'conv3x3_winograd4c', 'pointwise')
    origs = {n: getattr(ops, n) for n in names}
    current = {}
    for n in names:
        monkeypatch.setattr(ops, n, (lambda n_: lambda *a, **k: (current.__setitem__('k', n_), origs[n_](*a, **k))[1])(n))
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    for si, (kind, cin, cout, stride, relu, H, W) in enumerate(shapes):
        conv = _module_for(kind, cin, cout, stride)
        fan = cin * (9 if kind == '3x3' else 4 if kind == 's2d' else 1)
        with torch.no_grad():
            conv.weight.copy_(torch.from_numpy(synth.uniform(5000 + si, 1, conv.weight.numel(), -1.0, 1.0).reshape(tuple(conv.weight.shape))) * (3.0 / fan) ** 0.5)
            conv.bias.copy_(torch.from_numpy(synth.uniform(5000 + si, 2, cout, -0.2, 0.2)))
        pc = convnet.pack_conv_module(conv.cuda(), None, relu=relu)
        conv = conv.cpu()
        for B in (1, 4, 20):
            x = torch.from_numpy(synth.uniform(6000 + si, B, B * cin * H * W, -1.0, 1.0).reshape(B, H, W, cin))
            got = pc.run(x.cuda())
            with torch.no_grad():
                want = conv(x.permute(0, 3, 1, 2))
                if relu:
                    want = F.relu(want)
            scale = max(1.0, float(want.abs().max()))
            np.testing.assert_allclose(got.float().permute(0, 3, 1, 2).cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-4 * scale,
                                       err_msg='%s at B = %d via %s' % ((kind, cin, cout, stride, relu, H, W), B, current.get('k')))
            picked.setdefault((kind, cin, cout, stride, H, W), []).append(current.get('k'))
    # the three batch sizes do exercise different kernels for the same layer (otherwise this test pins nothing about dispatch)
    multi = [k for k, v in picked.items() if len(set(v)) > 1]
    assert len(multi) >= 4, picked
    used = {k for v in picked.values() for k in v}
    assert {'conv3x3', 'conv3x3_winograd4c', 'conv3x3_winograd4f', 'pointwise'} <= used, used


@pytest.mark.parametrize('case', ['car', 'ego', 'early', 'disco'])
def test_final_set_does_not_depend_on_the_batch_a_frame_travels_in(case, monkeypatch):
    """g13 mini fixtures: frame 0 of the fixture alone, then as every element of a batch of 4 and of 20 copies (auto dispatch picks kernels by
    launch size): the reference's exact final set every time"""
    monkeypatch.delenv('PCP_CONV_ALGO', raising=False)
    from test_gpu_e2e import _g13_model, _g13_points
    g = load_golden('g13_conditioned.npz')
    model = _g13_model(g, case)
    pts, B0 = _g13_points(case)
    frame0 = pts[pts[:, 0] == 0]
    rb, rs = g['%s_boxes_0' % case], g['%s_scores_0' % case]
    meta0 = {'se3_from_ego': {0: g['disco_pose_0'], 2: g['disco_pose_2']}} if case == 'disco' else {}
    for B in (1, 4, 20):
        rows = []
        for b in range(B):
            f = frame0.copy()
            f[:, 0] = b
            rows.append(f)
        batch = {'points': torch.from_numpy(np.concatenate(rows, 0)).cuda(), 'batch_size': B, 'metadata': [dict(meta0) for _ in range(B)]}
        with torch.no_grad():
            pred, _ = model(batch)
        for b in range(B):
            assert_same_final_set(rb, rs, pred[b]['pred_boxes'].cpu().numpy(), pred[b]['pred_scores'].cpu().numpy(), tol=1e-3)
