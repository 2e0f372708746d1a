"""Per-workgroup timeline of k_mp_conv3x3_s1 (diagnostic build -DMC_STAMP): python tools/stamp_mc.py B HW cin cout
stamps (s_memtime ticks, 100 MHz): 0 start; per stage q < 4: 1+3q barrier passed, 2+3q next copy issued, 3+3q products done; 13 after stage 3
(incl. its epilogue); 14 end; 15 = number of stages."""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault('PCP_HIP_LIB', str(Path(__file__).resolve().parent.parent / 'lib' / 'variants' / 'libpcp_hip_stamp.so'))
from pcp_amd import lib as plib, train_ops as tops  # noqa: E402

B, hw, cin, cout = [int(v) for v in sys.argv[1:5]]
dev = 'cuda:0'
x = torch.randn((B, hw, hw, cin), device=dev).to(torch.bfloat16)
w = torch.randn((cout, cin, 3, 3), device=dev) * 0.05
packed, opad = tops.mp_pack_conv3x3(w)
bias = torch.zeros(opad, device=dev)
out = torch.empty((B, hw, hw, cout), dtype=torch.bfloat16, device=dev)
for _ in range(5):
    tops.mp_conv3x3(x, packed, bias, cin, cout, opad, relu=True, out=out)
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ['PCP_HIP_LIB'])
buf = np.zeros(1024 * 16, dtype=np.uint64)
L.pcp_debug_read_mc.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert L.pcp_debug_read_mc(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(1024, 16).astype(np.int64)
t = t[t[:, 15] > 0]
n = t.shape[0]
ns = int(np.median(t[:, 15]))
print('%d workgroups, %d stages each (median); ticks are 10 ns' % (n, ns))
t0 = t[:, 0].min()
print('launch: first start -> last start %d ticks, first start -> last end %d ticks' % (t[:, 0].max() - t0, t[:, 14].max() - t0))
names = ['start -> stage0 data landed'] + sum([['  s%d: issue next copy' % q, '  s%d: products' % q, '  s%d: (epilogue) + wait + barrier' % q] for q in range(4)], [])
seq = [0] + [i for q in range(min(ns, 4)) for i in (1 + 3 * q, 2 + 3 * q, 3 + 3 * q)] 
prev = t[:, 0]
for k, idx in enumerate(seq[1:]):
    d = t[:, idx] - prev
    print('%-36s mean %7.0f  median %7.0f  p90 %7.0f' % (names[k], d.mean(), np.median(d), np.percentile(d, 90)))
    prev = t[:, idx]
print('%-36s mean %7.0f' % ('last stamped -> end', (t[:, 14] - prev).mean()))
print('%-36s mean %7.0f' % ('whole workgroup', (t[:, 14] - t[:, 0]).mean()))
