"""in-kernel s_memtime stamps of csrc/wino4f.hip (diagnostic build -DF4_STAMP=<workgroup>, selected with PCP_HIP_LIB): where one workgroup's
time goes: main loop | accumulator dump | barrier | LDS reads + first output-transform stage | second stage + stores."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, 'practical-collab-perception_amd')
from pcp_amd import lib, ops, pack  # noqa: E402

d = torch.device('cuda:0')
B, H, W, cin, cout = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (20, 128, 128, 128, 128)
x = torch.randn((B, H, W, cin), device=d)
w = torch.randn((cout, cin, 3, 3)) * 0.05
pw, bw, cpw = pack.pack_conv3x3_winograd4f(w, torch.zeros(cout))
pw, bw = pw.to(d), bw.to(d)
out = torch.empty((B, H, W, cout), device=d)
for _ in range(3):
    ops.conv3x3_winograd4f(x, pw, bw, cin, cout, cpw, out=out)
torch.cuda.synchronize()
L = lib.load()
L.pcp_debug_read_f4.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(8 * 32, dtype=np.uint64)
print('rc', L.pcp_debug_read_f4(buf.ctypes.data, buf.nbytes))
t = buf.reshape(8, 32).astype(np.int64)
t0 = t[:, 0].min()
for wv in range(8):
    r = t[wv] - t0
    half = wv >> 2
    if half == 0:
        print('wave %d: loop %6d | dump %5d | barrier %5d | finish: stage1 %5d stage2+stores %5d | tail barriers %5d' %
              (wv, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[6] - r[3], r[4] - r[6], r[5] - r[4]))
    else:
        print('wave %d: loop %6d | wait for half 0 %5d | dump %5d | barrier %5d | finish: stage1 %5d stage2+stores %5d' %
              (wv, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[8] - r[4], r[5] - r[8]))
print('slice 4 of the main loop (nine fenced blocks; cycles from the slice start to the END of each block, then the barrier):')
for wv in range(8):
    r = t[wv] - t[:, 16].min()
    ends = [int(r[17 + k]) for k in range(9)]
    print('  wave %d: start %5d | block ends %s | barrier done %5d' % (wv, r[16], ' '.join('%5d' % e for e in ends), r[26]))
