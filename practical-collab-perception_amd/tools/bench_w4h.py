"""k_wino4h (two four-wave workgroups per CU) against k_wino4f (one eight-wave workgroup) on the layer shapes of the DiscoNet step:
python tools/bench_w4h.py [reps]  -- interleaved timing in one process + max |difference| of the two outputs and against torch."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from pcp_amd import ops, pack  # noqa: E402

SHAPES = [(20, 128, 128, 128, 128), (4, 128, 128, 128, 128), (20, 256, 256, 64, 64), (4, 256, 256, 64, 64), (4, 128, 128, 384, 128),
          (4, 128, 128, 384, 384), (4, 128, 128, 64, 320), (20, 64, 64, 128, 128), (4, 64, 64, 256, 256), (1, 128, 128, 64, 64)]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = 'cuda:0'
    torch.manual_seed(0)
    for (B, H, W, cin, cout) in SHAPES:
        x = torch.randn((B, H, W, cin), device=dev)
        w = torch.randn((cout, cin, 3, 3), device=dev) / (3.0 * cin ** 0.5)
        b = torch.randn((cout,), device=dev)
        uf, bf, cpf = pack.pack_conv3x3_winograd4f(w, b)
        uh, bh, cph = pack.pack_conv3x3_winograd4h(w, b)
        of = ops.conv3x3_winograd4f(x, uf, bf, cin, cout, cpf, relu=True)
        oh = ops.conv3x3_winograd4h(x, uh, bh, cin, cout, cph, relu=True)
        ref = None
        if B * H * W <= 4 * 128 * 128:
            ref = torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1)).permute(0, 2, 3, 1).float()
        torch.cuda.synchronize()
        ts = {'4f': [], '4h': []}
        for _ in range(reps):
            for name, fn, args in (('4f', ops.conv3x3_winograd4f, (x, uf, bf, cin, cout, cpf)), ('4h', ops.conv3x3_winograd4h, (x, uh, bh, cin, cout, cph))):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(*args, relu=True, out=of if name == '4f' else oh)
                e1.record()
                torch.cuda.synchronize()
                ts[name].append(e0.elapsed_time(e1) * 1e3)
        m = lambda v: sorted(v)[len(v) // 2]
        fl = 2.0 * 36 * B * ((H + 15) // 16) * ((W + 15) // 16) * 16 * cin * ((cout + 63) // 64 * 64)
        print('B%-2d %3dx%-3d %3d->%-3d  4f %7.1f us  4h %7.1f us  (x%.3f, %5.1f TF executed)   |4h-4f| %.2e%s' % (
            B, H, W, cin, cout, m(ts['4f']), m(ts['4h']), m(ts['4f']) / m(ts['4h']), fl / m(ts['4h']) / 1e6, (oh - of).abs().max().item(),
            '' if ref is None else '   |4h-ref| %.2e |4f-ref| %.2e' % ((oh - ref).abs().max().item(), (of - ref).abs().max().item())), flush=True)
    # ragged sizes / channel windows
    for (B, H, W, cin, cout) in [(2, 37, 50, 16, 52), (1, 16, 16, 8, 64), (3, 20, 100, 72, 132)]:
        x = torch.randn((B, H, W, cin), device=dev)
        w = torch.randn((cout, cin, 3, 3), device=dev) / (3.0 * cin ** 0.5)
        b = torch.randn((cout,), device=dev)
        uh, bh, cph = pack.pack_conv3x3_winograd4h(w, b)
        oh = ops.conv3x3_winograd4h(x, uh, bh, cin, cout, cph, relu=False)
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1).float()
        print('ragged B%d %dx%d %d->%d  |4h-ref| %.2e' % (B, H, W, cin, cout, (oh - ref).abs().max().item()))


if __name__ == '__main__':
    main()
