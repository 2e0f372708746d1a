"""Per-workgroup timeline of k_mp_wgrad3x3 (diagnostic build -DWG_STAMP): PCP_HIP_LIB=<variant> python tools/stamp_wg.py B HW cin cout
stamps (shader cycles): 0 start; per stage s < 10: 1+2s wait + barrier passed, 2+2s products done; 21 loop end; 22 partial stores issued; 23 = stages."""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcp_amd import train_ops as tops  # noqa: E402

B, hw, cin, cout = [int(v) for v in sys.argv[1:5]]
dev = 'cuda:0'
x = torch.randn((B, hw, hw, cin), device=dev).to(torch.bfloat16)
dy = torch.randn((B, hw, hw, cout), device=dev).to(torch.bfloat16)
dw = torch.zeros((cout, cin, 3, 3), device=dev)
for _ in range(5):
    tops.mp_conv3x3_wgrad(x, dy, cin, cout, 1, dw)
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ['PCP_HIP_LIB'])
buf = np.zeros(1024 * 24, dtype=np.uint64)
L.pcp_debug_read_wg.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert L.pcp_debug_read_wg(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(1024, 24).astype(np.int64)
t = t[t[:, 23] > 0]
ns = int(np.median(t[:, 23]))
print('%d workgroups, %d stages (median)' % (t.shape[0], ns))
prev = t[:, 0]
print('%-28s mean %7.0f' % ('start -> stage 0 landed', (t[:, 1] - prev).mean()))
for s in range(min(ns, 10)):
    print('  s%d products %7.0f   then wait + barrier %7.0f' % (s, (t[:, 2 + 2 * s] - t[:, 1 + 2 * s]).mean(),
                                                          ((t[:, 3 + 2 * s] if s + 1 < min(ns, 10) else t[:, 21]) - t[:, 2 + 2 * s]).mean()))
print('%-28s mean %7.0f' % ('partial stores', (t[:, 22] - t[:, 21]).mean()))
print('%-28s mean %7.0f' % ('whole workgroup', (t[:, 22] - t[:, 0]).mean()))
