"""Per-workgroup timeline of k_wino4h (diagnostic build -DH4_STAMP): python tools/stamp_h4.py B H W cin cout
Stamps (s_memtime): 0 start, 1 main loop start, 2 main loop end, 3 half 0 read, 4 half 0 stores issued, 5 half 1 read, 6 end; 7 = HW_ID | XCC_ID << 32.
Prints the mean phase lengths and, per CU, how the two resident workgroups' phases overlap."""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcp_amd import lib as plib, pack  # noqa: E402


def main():
    B, H, W, cin, cout = [int(v) for v in sys.argv[1:6]]
    L = ctypes.CDLL(os.environ['PCP_HIP_LIB'])
    d = torch.device('cuda:0')
    x = torch.randn((B, H, W, cin), device=d)
    w = torch.randn((cout, cin, 3, 3)) * 0.05
    pw, bw, cp = pack.pack_conv3x3_winograd4h(w, torch.zeros(cout))
    pw, bw = pw.to(d), bw.to(d)
    out = torch.empty((B, H, W, cout), device=d)
    desc = plib.Conv3x3(B, H, W, cin, cout, cp, 1, cin, cout, 1)
    fn = L.pcp_conv3x3_winograd4h
    fn.restype = ctypes.c_int32
    fn.argtypes = [ctypes.POINTER(plib.Conv3x3)] + [ctypes.c_void_p] * 5
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        fn(ctypes.byref(desc), x.data_ptr(), pw.data_ptr(), bw.data_ptr(), out.data_ptr(), st)
    torch.cuda.synchronize()
    n = min(8192, B * ((H + 15) // 16) * ((W + 15) // 16) * (cp // 64))
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    L.pcp_debug_read_h4.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert L.pcp_debug_read_h4(buf.ctypes.data, buf.nbytes) == 0
    t = buf.reshape(8192, 8)[:n].astype(np.int64)
    hw = t[:, 7]
    cu = ((hw >> 32) & 15) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 15)      # xcc, se, sh, cu
    t0 = t[:, 0].min()
    ph = np.diff(t[:, :7], axis=1)
    names = ['prologue', 'main loop', 'dump0+read0', 'dump1+stores0', 'barrier+read1', 'stores1']
    print('%d workgroups on %d CUs, launch span %d ticks' % (n, len(set(cu.tolist())), t[:, 6].max() - t0))
    for i, nm in enumerate(names):
        print('  %-16s mean %8.0f  median %8.0f  p90 %8.0f' % (nm, ph[:, i].mean(), np.median(ph[:, i]), np.percentile(ph[:, i], 90)))
    print('  %-16s mean %8.0f' % ('whole item', (t[:, 6] - t[:, 0]).mean()))
    # overlap: for every workgroup, the share of its non-main-loop time during which a co-resident workgroup was in its main loop
    cov, tot = 0.0, 0.0
    alone = 0.0
    for c in set(cu.tolist()):
        idx = np.nonzero(cu == c)[0]
        for i in idx:
            segs = [(t[i, 0], t[i, 1]), (t[i, 2], t[i, 6])]
            for (a, b) in segs:
                tot += b - a
                for j in idx:
                    if j != i:
                        lo, hi = max(a, t[j, 1]), min(b, t[j, 2])
                        if hi > lo:
                            cov += hi - lo
        # time the CU had exactly one / zero workgroups in a main loop: sample
    print('  prologue + epilogue time covered by a co-resident main loop: %.1f %%' % (100.0 * cov / tot))
    c0 = cu[0]
    idx = np.nonzero(cu == c0)[0]
    idx = idx[np.argsort(t[idx, 0])]
    print('  CU %d timeline (start, main start, main end, end; slot):' % c0)
    for i in idx[:12]:
        print('    wg %5d  %8d %8d %8d %8d   slot %d' % (i, t[i, 0] - t0, t[i, 1] - t0, t[i, 2] - t0, t[i, 6] - t0, hw[i] & 15))
    slices(L, cin // 8)


def slices(L, n_slices):
    buf = np.zeros(4 * 64 * 12, dtype=np.uint64)
    L.pcp_debug_read_h4_slices.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert L.pcp_debug_read_h4_slices(buf.ctypes.data, buf.nbytes) == 0
    t = buf.reshape(4, 64, 12).astype(np.int64)[:, :n_slices]
    print('  one workgroup, per wave: mean ticks per slice (last slice excluded)')
    print('  wave   blk0  blk1  blk2  blk3  blk4  blk5  blk6  blk7  blk8  lds-drain  barrier-wait   slice')
    for w in range(4):
        d = t[w, :n_slices - 1]
        rel = (d[:, 1:] - d[:, :1]).mean(axis=0)
        blocks = np.diff(np.concatenate([[0], rel[:9]]))
        print('  %4d  ' % w + ' '.join('%5.0f' % v for v in blocks) + '   %7.0f  %10.0f  %8.0f' % (rel[9] - rel[8], rel[10] - rel[9], rel[10]))
    print('  slice start times of wave 0 (relative):', ' '.join('%d' % (v - t[0, 0, 0]) for v in t[0, :, 0]))


if __name__ == '__main__':
    main()
