import sys, ctypes, numpy as np, torch
sys.path.insert(0, 'practical-collab-perception_amd')
from pcp_amd import ops, pack, lib
d = torch.device('cuda:0')
B, H, W, cin, cout = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (4, 128, 128, 768, 768)
x = torch.randn((B, H, W, cin), device=d); w = torch.randn((cout, cin, 3, 3)) * 0.05
pw, bw, cpw = pack.pack_conv3x3_winograd(w, torch.zeros(cout)); pw, bw = pw.to(d), bw.to(d)
out = torch.empty((B, H, W, cout), device=d)
for _ in range(3): ops.conv3x3_winograd(x, pw, bw, cin, cout, cpw, out=out)
torch.cuda.synchronize()
L = lib.load(); L.pcp_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(8 * 8 * 8, dtype=np.uint64)
print('rc', L.pcp_debug_read(buf.ctypes.data, buf.nbytes))
t = buf.reshape(8, 8, 8).astype(np.int64)
t0 = t[0, :, 0].min()
np.set_printoptions(linewidth=200)
for s in range(8):
    print('slice', s)
    for wv in range(8):
        r = t[s, wv, :6] - t0
        print('  wave', wv, 'start %6d | xform-first %5d | mult %5d | store+loads %5d | xform-last %5d | barrier %5d' % (r[0], r[1]-r[0], r[2]-r[1], r[3]-r[2], r[4]-r[3], r[5]-r[4]))
