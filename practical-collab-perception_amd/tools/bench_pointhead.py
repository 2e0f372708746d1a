"""Micro-benchmark of the fused HunterJr point head: python tools/bench_pointhead.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from pcp_amd import ops, synth  # noqa: E402


def main():
    dev = 'cuda:0'
    B, H, W, C = 4, 128, 128, 384
    cat = torch.randn((B, H, W, 2 * C), device=dev)
    pts = torch.from_numpy(synth.collate([synth.agent_cloud(agent=f, n_points=60000, layout='car') for f in range(B)])).to(dev)
    grid = ops.make_grid([-51.2, -51.2, -8, 51.2, 51.2, 0], [0.2, 0.2, 8.0], [512, 512, 1], B)
    vox = ops.voxelize(pts, grid, want_inverse=False, want_counts=False)
    order = ops.voxelize_row_order(vox)
    w = [torch.randn(s, device=dev) * 0.05 for s in ((32, C), (32,), (C, 32), (C,), (8, C), (8,))]
    for name, od in (('index order', None), ('bucket order', order)):
        for _ in range(3):
            ops.hunter_point_head(cat, pts, [-51.2, -51.2], [0.8, 0.8], *w, channels=C, order=od)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.hunter_point_head(cat, pts, [-51.2, -51.2], [0.8, 0.8], *w, channels=C, order=od)
        e1.record()
        torch.cuda.synchronize()
        print('%-14s %8.1f us' % (name, e0.elapsed_time(e1) * 100))


if __name__ == '__main__':
    main()
