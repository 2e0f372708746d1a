"""Micro-benchmark of the training kernels on config-5 layer shapes: python tools/bench_wgrad.py [B]"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from pcp_amd import train_ops as tops  # noqa: E402

LAYERS = [('backbone.b0 64->64 @256', 64, 64, 256, 1), ('backbone.b0.0 64->64 @512 s2', 64, 64, 512, 2), ('backbone.b1 128->128 @128', 128, 128, 128, 1),
          ('backbone.b2 128->128 @64', 128, 128, 64, 1), ('compress 384->128 @128', 384, 128, 128, 1), ('decompress 128->384 @128', 128, 384, 128, 1),
          ('decompress 384->384 @128', 384, 384, 128, 1), ('head.shared 384->64 @128', 384, 64, 128, 1), ('head.stage1 64->64 @128', 64, 64, 128, 1),
          ('head.final 320->16 @128', 320, 16, 128, 1)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = 'cuda:0'
    for name, cin, cout, hw, stride in LAYERS:
        x = torch.randn((B, hw, hw, cin), device=dev)
        dy = torch.randn((B, hw // stride, hw // stride, cout), device=dev)
        dw = torch.empty((cout, cin, 3, 3), device=dev)
        for _ in range(3):
            tops.conv3x3_wgrad(x, dy, cin, cout, stride, dw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 10
        for _ in range(n):
            tops.conv3x3_wgrad(x, dy, cin, cout, stride, dw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        fl = 2.0 * B * (hw // stride) ** 2 * cout * cin * 9
        print('%-34s wgrad %9.1f us  %6.1f TF' % (name, us, fl / us / 1e6))
        c = cout
        vec = tops.bn_train_stats(dy, c, torch.ones(c, device=dev), torch.zeros(c, device=dev), 1e-3, 0.01, None, None)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            tops.bn_train_stats(dy, c, torch.ones(c, device=dev), torch.zeros(c, device=dev), 1e-3, 0.01, None, None, vec=vec)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print('%-34s bn_stats %6.1f us  %6.2f TB/s' % ('', us, dy.numel() * 4 / us / 1e6))


if __name__ == '__main__':
    main()
