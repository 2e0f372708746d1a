"""Micro-benchmark of the VFE stage, rounds 1-4 form against the round-5 form, on the synthetic clouds of bench.py:
  old: pcp_voxelize + pcp_pfn_scatter (+ pcp_canvas_clear for a dense canvas)
  new: pcp_pillarise_rows + pcp_pfn_rows (the canvas is written completely: no clear)
usage: bench_frontend.py [frames=4] [agents=1|6] [dense=0|1] [dist=uniform|ring]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcp_amd import ops, synth  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(3):
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
    dense = int(sys.argv[3]) if len(sys.argv) > 3 else (1 if agents > 1 else 0)
    dist = sys.argv[4] if len(sys.argv) > 4 else 'uniform'
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
    n = pts.shape[0]
    canvas = torch.zeros((B, 512, 512, 64), device=d) if dense else None
    pf = None if dense else torch.empty((n, 64), device=d)
    vox = ops.voxelize(pts, grid, want_inverse=False, want_counts=False)
    rows = ops.pillarise_rows(pts, grid, 5)
    torch.cuda.synchronize()
    P, Nv = int(rows.counters[0]), int(rows.counters[1])
    alg = 4.0 * n * pts.shape[1] + 16.0 * P + (4.0 * B * 512 * 512 * 64 if dense else 256.0 * P)
    t_vox = timeit(lambda: ops.voxelize(pts, grid, want_inverse=False, want_counts=False, workspace=vox.workspace))
    t_pfn = timeit(lambda: ops.pfn_scatter(pts, vox, 5, w0, b0, w1, b1, canvas=canvas, pillar_features=pf))
    t_clr = timeit(lambda: ops.canvas_clear(vox, canvas)) if dense else 0.0
    t_rows = timeit(lambda: ops.pillarise_rows(pts, grid, 5, workspace=rows.workspace))
    canvas2 = torch.empty((B, 512, 512, 64), device=d) if dense else None
    pf2 = None if dense else torch.empty((n, 64), device=d)
    t_pfn2 = timeit(lambda: ops.pfn_rows(rows, w0, b0, w1, b1, canvas=canvas2, pillar_features=pf2))
    if dense:
        ops.pfn_scatter(pts, vox, 5, w0, b0, w1, b1, canvas=canvas)
        torch.cuda.synchronize()
        err = float((canvas - canvas2).abs().max())
    else:
        err = float((pf[:P] - pf2[:P]).abs().max())

    def whole_old():
        v = ops.voxelize(pts, grid, want_inverse=False, want_counts=False, workspace=vox.workspace)
        ops.pfn_scatter(pts, v, 5, w0, b0, w1, b1, canvas=canvas, pillar_features=pf)
        if dense:
            ops.canvas_clear(v, canvas)

    def whole_new():
        v = ops.pillarise_rows(pts, grid, 5, workspace=rows.workspace)
        ops.pfn_rows(v, w0, b0, w1, b1, canvas=canvas2, pillar_features=pf2)
    t_old, t_new = timeit(whole_old), timeit(whole_new)
    print('B=%d agents=%d %s: n=%d P=%d N\'=%d algorithmic %.1f MB' % (B, agents, 'dense canvas' if dense else 'pillar rows', n, P, Nv, alg / 1e6))
    print('  old: voxelize %.1f us + pfn %.1f us + clear %.1f us; stage %.1f us = %.0f GB/s' % (t_vox, t_pfn, t_clr, t_old, alg / t_old / 1e3))
    print('  new: pillarise_rows %.1f us + pfn_rows %.1f us;              stage %.1f us = %.0f GB/s   (max |new - old| %.2e)'
          % (t_rows, t_pfn2, t_new, alg / t_new / 1e3, err))


if __name__ == '__main__':
    main()
