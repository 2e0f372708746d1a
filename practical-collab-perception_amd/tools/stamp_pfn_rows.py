"""In-kernel s_memtime stamps of one k_pfn_rows wave (build: csrc/build_variant.sh prstamp "-DPR_STAMP=<workgroup>"): shader cycles per phase,
summed over the wave's tiles.  usage: stamp_pfn_rows.py [frames=4] [agents=6] [dense=1] [dist=uniform|ring].  Diagnostic tool, not part of the product path."""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np
import torch

PKG = Path(__file__).resolve().parent.parent
os.environ['PCP_HIP_LIB'] = str(PKG / 'lib' / 'variants' / ('libpcp_hip_prstamp%s.so' % os.environ.get('PCP_STAMP_VARIANT', '')))
sys.path.insert(0, str(PKG))
from pcp_amd import ops, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
agents = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dense = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dist = sys.argv[4] if len(sys.argv) > 4 else 'uniform'
d = torch.device('cuda:0')
frames = [np.concatenate([synth.agent_cloud(a, 60000, 'car', seed=synth.SEED_BASE + b, dist=dist) for a in range(agents)], 0) for b in range(B)]
pts = torch.from_numpy(synth.collate(frames)).to(d)
grid = ops.make_grid([-51.2, -51.2, -8.0, 51.2, 51.2, 0.0], [0.2, 0.2, 8.0], [512, 512, 1], B)
w0 = torch.randn(32, 11, device=d) * 0.1
b0 = torch.zeros(32, device=d)
w1 = torch.randn(64, 64, device=d) * 0.1
b1 = torch.zeros(64, device=d)
rows = ops.pillarise_rows(pts, grid, 5)
canvas = torch.empty((B, 512, 512, 64), device=d) if dense else None
pf = None if dense else torch.empty((pts.shape[0], 64), device=d)
raw = ctypes.CDLL(os.environ['PCP_HIP_LIB'])
names = ['prefetch issue', 'A sums', 'B means', 'C layers', 'D epilogue', 'E gap fill']
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.pfn_rows(rows, w0, b0, w1, b1, canvas=canvas, pillar_features=pf)
    e1.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    raw.pcp_debug_read_pfn_rows(buf, 16 * 8)
    tiles = max(int(buf[8]), 1)
    tot = sum(buf[k] for k in range(6))
    print('rep %d: %.1f us, %d tiles of the stamped wave, %d cycles per tile: ' % (rep, e0.elapsed_time(e1) * 1e3, tiles, tot // tiles)
          + '  '.join('%s %d' % (names[k], buf[k] // tiles) for k in range(6)) + '   [%d column tiles, %d pillars in those tiles]' % (buf[9], buf[10]))

# every wave's cycles in its three parts (tiles | singles | canvas fill): how even is the work?
wc = (ctypes.c_ulonglong * (4096 * 3))()
raw.pcp_debug_read_pfn_wave_cycles(wc, 4096 * 3 * 8)
w = np.array(list(wc), dtype=np.float64).reshape(4096, 3)
w = w[w.sum(1) > 0]
tot = w.sum(1)
print('%d waves: cycles per wave min %.0f  median %.0f  mean %.0f  p99 %.0f  max %.0f   (tiles mean %.0f max %.0f | singles mean %.0f max %.0f | fill mean %.0f max %.0f)'
      % (w.shape[0], tot.min(), np.median(tot), tot.mean(), np.percentile(tot, 99), tot.max(), w[:, 0].mean(), w[:, 0].max(), w[:, 1].mean(), w[:, 1].max(),
         w[:, 2].mean(), w[:, 2].max()))
w0 = np.array(list(wc), dtype=np.float64).reshape(4096, 3)[::4]          # wave 0 of every workgroup: the one a -DPR_STAMP=<wg> build stamps
order = np.argsort(-w0[:, 0])
print('workgroups whose wave 0 spends most cycles in its tiles:', [(int(i), int(w0[i, 0])) for i in order[:6]], ' median', int(np.median(w0[w0[:, 0] > 0, 0])))

# every tile's cycles against what it holds: which tiles are the expensive ones?
tc = (ctypes.c_uint * (65536 * 4))()
raw.pcp_debug_read_pfn_tile_cycles(tc, 65536 * 4 * 4)
t = np.array(list(tc), dtype=np.int64).reshape(65536, 4)
t = t[t[:, 0] > 0]
print('%d tiles: cycles mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %d' % ((t.shape[0], t[:, 0].mean()) + tuple(np.percentile(t[:, 0], [50, 90, 99])) + (t[:, 0].max(),)))
for lo, hi in [(0, 1), (1, 17), (17, 33), (33, 49), (49, 65), (65, 129), (129, 257), (257, 1 << 30)]:
    m = (t[:, 1] >= lo) & (t[:, 1] < hi)
    if m.any():
        print('  records [%d, %d): %6d tiles, cycles mean %7.0f (sum %5.1f %%), pillars mean %.1f' % (lo, hi, m.sum(), t[m, 0].mean(), 100.0 * t[m, 0].sum() / t[:, 0].sum(), t[m, 2].mean()))
for lo, hi in [(0, 1), (1, 2), (2, 4), (4, 8), (8, 16), (16, 33)]:
    m = (t[:, 2] >= lo) & (t[:, 2] < hi) & (t[:, 1] >= 17) & (t[:, 1] < 49)
    if m.any():
        print('  17..48 records, pillars [%d, %d): %6d tiles, cycles mean %7.0f' % (lo, hi, m.sum(), t[m, 0].mean()))
# where the tiles of the slowest decile lie
idx = np.nonzero(np.array(list(tc), dtype=np.int64).reshape(65536, 4)[:, 0] > np.percentile(t[:, 0], 95))[0]
print('  tiles above p95: histogram of tile index / 1024:', np.bincount(idx // 1024).tolist())
