"""A/B of the pointwise kernels on the deblock shapes of the DiscoNet step (PCP_PW_ALGO=tile | stream), HIP events around 20 launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pcp_amd import lib, ops, pack  # noqa: E402

SHAPES = [  # (mode, cin, cout, H, W of the INPUT)
    ('plain', 64, 128, 128, 128), ('s2d', 64, 128, 128, 128), ('plain', 128, 128, 64, 64), ('d2s', 256, 128, 32, 32),
]


def run(mode, cin, cout, h, w, B, algo, reps=20):
    os.environ['PCP_PW_ALGO'] = algo
    d = torch.device('cuda:0')
    g = torch.Generator().manual_seed(1)
    x = torch.rand((B, h, w, cin), generator=g).to(d)
    bias = torch.rand((cout,), generator=g)
    if mode == 'plain':
        packed, bp, cpad = pack.pack_plain(torch.rand((cout, cin), generator=g) - 0.5, bias)
        m, oshape = lib.PW_PLAIN, (B, h, w, 384)
        k, n = cin, cout
        rows = B * h * w
    elif mode == 's2d':
        packed, bp, cpad = pack.pack_conv2x2_s2(torch.rand((cout, cin, 2, 2), generator=g) - 0.5, bias)
        m, oshape = lib.PW_SPACE2DEPTH, (B, h // 2, w // 2, 384)
        k, n = 4 * cin, cout
        rows = B * h * w // 4
    else:
        packed, bp, cpad = pack.pack_convT2x2_s2(torch.rand((cin, cout, 2, 2), generator=g) - 0.5, bias)
        m, oshape = lib.PW_DEPTH2SPACE, (B, 2 * h, 2 * w, 384)
        k, n = cin, 4 * cout
        rows = B * h * w
    packed, bp = packed.to(d), bp.to(d)
    out = torch.empty(oshape, device=d)
    for _ in range(3):
        ops.pointwise(x, packed, bp, m, cin, cout, cpad, relu=True, out=out, out_ch_off=128)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.pointwise(x, packed, bp, m, cin, cout, cpad, relu=True, out=out, out_ch_off=128)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000.0 / reps
    flops = 2.0 * rows * k * n
    byts = 4.0 * (x.numel() + rows * n)
    return us, flops / us * 1e-6, byts / us * 1e-6, out


if __name__ == '__main__':
    for B in (4, 20):
        for shp in SHAPES:
            res = {}
            for algo in ('tile', 'stream'):
                us, tf, tb, out = run(*shp, B, algo)
                res[algo] = (us, tf, tb, out.clone())
            diff = float((res['tile'][3] - res['stream'][3]).abs().max())
            print('B=%2d %-6s %3d->%3d @%3dx%3d  tile %7.1f us (%5.1f TFLOP/s %4.2f TB/s)   stream %7.1f us (%5.1f TFLOP/s %4.2f TB/s)   x%.2f  maxdiff %.2e'
                  % ((B,) + shp + res['tile'][:3] + res['stream'][:3] + (res['tile'][0] / res['stream'][0], diff)))
