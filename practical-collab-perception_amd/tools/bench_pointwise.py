"""Micro-benchmark of the pointwise family (pcp_pointwise: 1x1 conv / Conv2d k2 s2 / ConvTranspose2d k2 s2 of the backbone's deblocks and the
fusion module) on the shapes of the five configs.  Prints time and fp32 MFMA TFLOP/s; checks the result against torch on one frame."""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcp_amd import lib, ops, pack  # noqa: E402

SHAPES = [  # (name, mode, cin, cout, H, W of the INPUT)
    ('deblock0 Conv k2s2 64->128 @256', 's2d', 64, 128, 256, 256), ('deblock1 1x1 128->128 @128', 'plain', 128, 128, 128, 128),
    ('deblock2 ConvT k2s2 128->128 @64', 'd2s', 128, 128, 64, 64), ('deblock2 ConvT k2s2 256->128 @64', 'd2s', 256, 128, 64, 64),
    ('fusion 1x1 256->128 @128', 'plain', 256, 128, 128, 128), ('fusion 1x1 128->32 @128', 'plain', 128, 32, 128, 128),
]


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    d = torch.device('cuda:0')
    for B in (4, 16):
        for name, mode, cin, cout, H, W in SHAPES:
            x = torch.randn((B, H, W, cin), device=d)
            if mode == 'plain':
                w = torch.randn((cout, cin)) * 0.05
                pw, pb, cp = pack.pack_plain(w, torch.zeros(cout))
                ref = F.conv2d(x[:1].permute(0, 3, 1, 2).cpu(), w.view(cout, cin, 1, 1)).permute(0, 2, 3, 1)
                m, oshape, flops = lib.PW_PLAIN, (B, H, W, cout), 2.0 * B * H * W * cin * cout
            elif mode == 's2d':
                w = torch.randn((cout, cin, 2, 2)) * 0.05
                pw, pb, cp = pack.pack_conv2x2_s2(w, torch.zeros(cout))
                ref = F.conv2d(x[:1].permute(0, 3, 1, 2).cpu(), w, stride=2).permute(0, 2, 3, 1)
                m, oshape, flops = lib.PW_SPACE2DEPTH, (B, H // 2, W // 2, cout), 2.0 * B * (H // 2) * (W // 2) * 4 * cin * cout
            else:
                w = torch.randn((cin, cout, 2, 2)) * 0.05
                pw, pb, cp = pack.pack_convT2x2_s2(w, torch.zeros(cout))
                ref = F.conv_transpose2d(x[:1].permute(0, 3, 1, 2).cpu(), w, stride=2).permute(0, 2, 3, 1)
                m, oshape, flops = lib.PW_DEPTH2SPACE, (B, 2 * H, 2 * W, cout), 2.0 * B * H * W * cin * 4 * cout
            pw, pb = pw.to(d), pb.to(d)
            out = torch.empty(oshape, device=d)
            f = lambda: ops.pointwise(x, pw, pb, m, cin, cout, cp, relu=False, out=out)
            t = timeit(f)
            err = float((out[:1].cpu() - ref).abs().max() / ref.abs().max())
            print('B=%2d %-34s %8.1f us %6.1f TF   (err %.1e)' % (B, name, t * 1e6, flops / t / 1e12, err))


if __name__ == '__main__':
    main()
