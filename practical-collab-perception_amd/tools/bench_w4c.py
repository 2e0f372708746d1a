"""k_wino4c (waves split over output channels, output transform in registers) against k_wino4h (waves split over positions, accumulator image
in LDS) on the layer shapes of the DiscoNet step: python tools/bench_w4c.py [reps] -- interleaved timing in one process, bitwise comparison
of the two outputs, max |difference| against torch on the small shapes, ragged sizes / channel windows."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from pcp_amd import ops, pack  # noqa: E402

SHAPES = [(20, 128, 128, 128, 128), (4, 128, 128, 128, 128), (20, 256, 256, 64, 64), (4, 256, 256, 64, 64), (4, 128, 128, 384, 128),
          (4, 128, 128, 64, 320), (20, 64, 64, 128, 128), (4, 64, 64, 256, 256), (1, 128, 128, 64, 64), (1, 256, 256, 64, 64)]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = 'cuda:0'
    torch.manual_seed(0)
    for (B, H, W, cin, cout) in SHAPES:
        x = torch.randn((B, H, W, cin), device=dev)
        w = torch.randn((cout, cin, 3, 3), device=dev) / (3.0 * cin ** 0.5)
        b = torch.randn((cout,), device=dev)
        uh, bh, cph = pack.pack_conv3x3_winograd4h(w, b)
        uc, bc, cpc = pack.pack_conv3x3_winograd4c(w, b)
        oh = ops.conv3x3_winograd4h(x, uh, bh, cin, cout, cph, relu=True)
        oc = ops.conv3x3_winograd4c(x, uc, bc, cin, cout, cpc, relu=True)
        torch.cuda.synchronize()
        ts = {'4h': [], '4c': []}
        for _ in range(reps):
            for name, fn, args, o in (('4h', ops.conv3x3_winograd4h, (x, uh, bh, cin, cout, cph), oh), ('4c', ops.conv3x3_winograd4c, (x, uc, bc, cin, cout, cpc), oc)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(*args, relu=True, out=o)
                e1.record()
                torch.cuda.synchronize()
                ts[name].append(e0.elapsed_time(e1) * 1e3)
        m = lambda v: sorted(v)[len(v) // 2]
        fl = 2.0 * 36 * B * ((H + 15) // 16) * ((W + 15) // 16) * 16 * cin * ((cout + 63) // 64 * 64)
        print('B%-2d %3dx%-3d %3d->%-3d  4h %7.1f us  4c %7.1f us  (x%.3f, %5.1f TF executed)   bitwise equal: %s' % (
            B, H, W, cin, cout, m(ts['4h']), m(ts['4c']), m(ts['4h']) / m(ts['4c']), fl / m(ts['4c']) / 1e6, bool(torch.equal(oh, oc))), flush=True)
    for (B, H, W, cin, cout) in [(2, 37, 50, 16, 52), (1, 16, 16, 8, 64), (3, 20, 100, 72, 132)]:
        x = torch.randn((B, H, W, cin + 8), device=dev)
        w = torch.randn((cout, cin, 3, 3), device=dev) / (3.0 * cin ** 0.5)
        b = torch.randn((cout,), device=dev)
        uc, bc, cpc = pack.pack_conv3x3_winograd4c(w, b)
        out = torch.full((B, H, W, cout + 12), -7.0, device=dev)
        ops.conv3x3_winograd4c(x, uc, bc, cin, cout, cpc, relu=False, out=out, in_ch_off=4, out_ch_off=8)
        ref = torch.nn.functional.conv2d(x[..., 4:4 + cin].permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1).float()
        print('ragged B%d %dx%d %d->%d  |4c-ref| %.2e  untouched margins: %s' % (B, H, W, cin, cout, (out[..., 8:8 + cout] - ref).abs().max().item(),
              bool((out[..., :8] == -7.0).all() and (out[..., 8 + cout:] == -7.0).all())))


if __name__ == '__main__':
    main()
