"""Interleaved A/B timing of pcp_pfn_scatter builds in ONE process (cdna_hip_programming.md rule 24): the shipped library and every variant
under lib/variants (PCP_DIAG_VARIANTS = name prefix filter), on the pillar list the shipped pcp_voxelize leaves in the workspace.
usage: bench_pfn_ab.py [frames=4] [agents=1|6]"""
import ctypes
import glob
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcp_amd import lib as plib, ops, synth  # noqa: E402

HERE = Path(__file__).resolve().parent.parent


def load(path):
    L = ctypes.CDLL(str(path))
    for name in ('pcp_pfn_scatter', 'pcp_sparse_conv3x3_s2'):
        res, args = plib.SYMBOLS[name]
        getattr(L, name).restype, getattr(L, name).argtypes = res, args
    return L


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    agents = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    d = torch.device('cuda:0')
    frames = [np.concatenate([synth.agent_cloud(a, 60000, 'car', seed=synth.SEED_BASE + b) for a in range(agents)], 0) for b in range(B)]
    pts = torch.from_numpy(synth.collate(frames)).to(d)
    grid = ops.make_grid([-51.2, -51.2, -8.0, 51.2, 51.2, 0.0], [0.2, 0.2, 8.0], [512, 512, 1], B)
    g = torch.Generator().manual_seed(1)
    w0 = (torch.rand(32, 11, generator=g) - 0.5).to(d)
    b0 = (torch.rand(32, generator=g) - 0.5).to(d)
    w1 = ((torch.rand(64, 64, generator=g) - 0.5) * 0.3).to(d)
    b1 = (torch.rand(64, generator=g) - 0.5).to(d)
    vox = ops.voxelize(pts, grid, want_inverse=False, want_counts=False)
    torch.cuda.synchronize()
    libs = {'shipped': load(HERE / 'lib' / 'libpcp_hip.so')}
    pat = os.environ.get('PCP_DIAG_VARIANTS', '')
    for p in sorted(glob.glob(str(HERE / 'lib' / 'variants' / 'libpcp_hip_*.so'))):
        name = os.path.basename(p)[len('libpcp_hip_'):-3]
        if name.startswith(pat):
            libs[name] = load(p)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for label, canvas, pf in (('dense canvas', torch.zeros((B, 512, 512, 64), device=d), None),
                              ('pillar rows only', None, torch.empty((pts.shape[0], 64), device=d))):
        def call(L):
            return L.pcp_pfn_scatter(pts.data_ptr(), vox.n, vox.row_stride, 5, ctypes.byref(vox.grid), vox.workspace.data_ptr(), w0.data_ptr(),
                                     b0.data_ptr(), w1.data_ptr(), b1.data_ptr(), pf.data_ptr() if pf is not None else None,
                                     canvas.data_ptr() if canvas is not None else None, st)
        times = {k: [] for k in libs}
        sums = {}
        for rnd in range(7):
            for k, L in libs.items():
                assert call(L) == 0
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    call(L)
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / 10 * 1e3)
                sums[k] = float((canvas if canvas is not None else pf[:int(vox.counters[0])]).double().sum())
        print('B %d agents %d N %d P %d, %s' % (B, agents, pts.shape[0], int(vox.counters[0]), label))
        for k, v in times.items():
            v = sorted(v)
            print('   %-14s median %8.1f us  min %8.1f us   checksum %.9e' % (k, v[len(v) // 2], v[0], sums[k]))


def sparse_ab(B=4):
    """the first backbone layer from the pillar list (pcp_sparse_conv3x3_s2), every library on the same pillar rows"""
    from pcp_amd import pack
    d = torch.device('cuda:0')
    frames = [synth.agent_cloud(0, 60000, 'car', seed=synth.SEED_BASE + b) for b in range(B)]
    pts = torch.from_numpy(synth.collate(frames)).to(d)
    grid = ops.make_grid([-51.2, -51.2, -8.0, 51.2, 51.2, 0.0], [0.2, 0.2, 8.0], [512, 512, 1], B)
    g = torch.Generator().manual_seed(1)
    w0 = (torch.rand(32, 11, generator=g) - 0.5).to(d)
    b0 = (torch.rand(32, generator=g) - 0.5).to(d)
    w1 = ((torch.rand(64, 64, generator=g) - 0.5) * 0.3).to(d)
    b1 = (torch.rand(64, generator=g) - 0.5).to(d)
    vox = ops.voxelize(pts, grid, want_inverse=False, want_counts=False)
    pf = torch.empty((pts.shape[0], 64), device=d)
    ops.pfn_scatter(pts, vox, 5, w0, b0, w1, b1, canvas=None, pillar_features=pf)
    wc = (torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.1
    wsp, bsp = pack.pack_conv3x3_sparse_s2(wc, torch.zeros(64))
    wsp, bsp = wsp.to(d), bsp.to(d)
    libs = {'shipped': load(HERE / 'lib' / 'libpcp_hip.so')}
    pat = os.environ.get('PCP_DIAG_VARIANTS', '')
    for p in sorted(glob.glob(str(HERE / 'lib' / 'variants' / 'libpcp_hip_*.so'))):
        name = os.path.basename(p)[len('libpcp_hip_'):-3]
        if name.startswith(pat):
            libs[name] = load(p)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = torch.empty((B, 256, 256, 64), device=d)
    times, sums = {k: [] for k in libs}, {}
    for rnd in range(7):
        for k, L in libs.items():
            call = lambda: L.pcp_sparse_conv3x3_s2(pf.data_ptr(), ctypes.byref(vox.grid), vox.workspace.data_ptr(), vox.n, wsp.data_ptr(), bsp.data_ptr(),
                                                   64, 1, out.data_ptr(), 64, st)
            assert call() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 10 * 1e3)
            sums[k] = float(out.double().sum())
    print('sparse first layer, B %d x 60k points' % B)
    for k, v in times.items():
        v = sorted(v)
        print('   %-14s median %8.1f us  min %8.1f us   checksum %.12e' % (k, v[len(v) // 2], v[0], sums[k]))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'sparse':
        sparse_ab(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
        sys.exit(0)
    main()
