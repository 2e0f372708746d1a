"""Micro-benchmark of the pillariser + fused PFN/scatter (pcp_voxelize, pcp_pfn_scatter) on the synthetic clouds of bench.py:
usage: bench_pfn.py [frames=4] [agents=1|6] [dist=uniform|ring]   (agents = 6: the merged 360k-point cloud of early fusion / DiscoNet's main branch)"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcp_amd import ops, pack, synth  # noqa: E402


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    agents = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dist = sys.argv[3] if len(sys.argv) > 3 else 'uniform'
    d = torch.device('cuda:0')
    frames = []
    for b in range(B):
        clouds = [synth.agent_cloud(a, 60000, 'car', seed=synth.SEED_BASE + b, dist=dist) for a in range(agents)]
        frames.append(np.concatenate(clouds, 0))
    pts = torch.from_numpy(synth.collate(frames)).to(d)
    grid = ops.make_grid([-51.2, -51.2, -8.0, 51.2, 51.2, 0.0], [0.2, 0.2, 8.0], [512, 512, 1], B)
    g = torch.Generator().manual_seed(1)
    w0 = (torch.rand(32, 11, generator=g) - 0.5).to(d)
    b0 = (torch.rand(32, generator=g) - 0.5).to(d)
    w1 = ((torch.rand(64, 64, generator=g) - 0.5) * 0.3).to(d)
    b1 = (torch.rand(64, generator=g) - 0.5).to(d)
    canvas = torch.zeros((B, 512, 512, 64), device=d)
    vox = ops.voxelize(pts, grid, want_inverse=False, want_counts=False)
    torch.cuda.synchronize()
    P = int(vox.counters[0])
    tv = timeit(lambda: ops.voxelize(pts, grid, want_inverse=False, want_counts=False, workspace=vox.workspace))
    tp = timeit(lambda: ops.pfn_scatter(pts, vox, 5, w0, b0, w1, b1, canvas=canvas))
    # first backbone layer: dense stride-2 conv on the canvas vs the sparse form on the pillar list
    pf = torch.empty((pts.shape[0], 64), device=d)
    ops.pfn_scatter(pts, vox, 5, w0, b0, w1, b1, canvas=canvas, pillar_features=pf)
    wc = (torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.1
    bc = torch.zeros(64)
    pd, bd, cpd = pack.pack_conv3x3(wc, bc)
    pd, bd = pd.to(d), bd.to(d)
    wsp, bsp = pack.pack_conv3x3_sparse_s2(wc, bc)
    wsp, bsp = wsp.to(d), bsp.to(d)
    out = torch.empty((B, 256, 256, 64), device=d)
    out2 = torch.empty_like(out)
    td = timeit(lambda: ops.conv3x3(canvas, pd, bd, 64, 64, cpd, stride=2, out=out))
    ts = timeit(lambda: ops.sparse_conv3x3_s2(pf, vox, wsp, bsp, 64, out=out2))
    tpf = timeit(lambda: ops.pfn_scatter(pts, vox, 5, w0, b0, w1, b1, canvas=None, pillar_features=pf))
    print('first layer 64->64 s2: dense %.1f us | sparse %.1f us (max diff %.2e) | pfn writing pillar rows only %.1f us' % (
        td, ts, float((out - out2).abs().max()), tpf))
    n = pts.shape[0]
    print('B %d agents %d: N %d P %d | voxelize %.1f us | pfn+scatter %.1f us = %.2f G points/s | canvas checksum %.6e' % (
        B, agents, n, P, tv, tp, n / tp * 1e-3, float(canvas.double().sum())))


if __name__ == '__main__':
    main()
