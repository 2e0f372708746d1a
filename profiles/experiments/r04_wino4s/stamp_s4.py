"""in-kernel stamps of k_wino4s (library built with -DS4_STAMP): per workgroup, for the first consumer and the first producer wave, the
shader-clock time from start to first step, the length of the step loop, and the time spent waiting at the step barrier"""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from pcp_amd import lib, ops, pack  # noqa: E402

B, H, W, cin, cout = [int(v) for v in (sys.argv[1:6] if len(sys.argv) >= 6 else (20, 128, 128, 128, 128))]
dev = 'cuda:0'
torch.manual_seed(0)
x = torch.randn((B, H, W, cin), device=dev)
w = torch.randn((cout, cin, 3, 3), device=dev) / (3.0 * cin ** 0.5)
b = torch.randn((cout,), device=dev)
uc, bc, cpc = pack.pack_conv3x3_winograd4c(w, b)
for _ in range(3):
    out = ops.conv3x3_winograd4s(x, uc, bc, cin, cout, cpc, relu=True)
torch.cuda.synchronize()
L = lib.load()
buf = np.zeros(256 * 16, dtype=np.uint64)
L.pcp_debug_read_s4.restype = ctypes.c_int
L.pcp_debug_read_s4.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert L.pcp_debug_read_s4(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
t = buf.reshape(256, 2, 8).astype(np.int64)
items = B * ((H + 15) // 16) * ((W + 15) // 16) * ((cout + 63) // 64)
n_wg = min(items, 256)
steps = -(-items // n_wg) * (cin // 8)
for role, name in ((0, 'consumer'), (1, 'producer')):
    tt = t[:n_wg, role]
    pro = tt[:, 1] - tt[:, 0]
    loop = tt[:, 2] - tt[:, 1]
    wait = tt[:, 3]
    print('%s: prologue %7.0f  loop %9.0f (%6.0f per step, %d steps)  waiting at the barrier %9.0f (%4.1f %% of the loop)' % (
        name, np.median(pro), np.median(loop), np.median(loop) / steps, steps, np.median(wait), 100.0 * np.median(wait) / np.median(loop)))
