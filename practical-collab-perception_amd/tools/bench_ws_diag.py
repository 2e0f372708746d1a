"""Interleaved A/B timing of csrc/wino_ws.hip builds in ONE process (cdna_hip_programming.md rule 24): the shipped library and the
timing-only diagnostic variants under lib/variants (phase-skipping builds: outputs are wrong by construction, only the time matters).
usage: bench_ws_diag.py [B]"""
import ctypes
import glob
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcp_amd import lib as plib, pack  # noqa: E402

HERE = Path(__file__).resolve().parent.parent


ENTRY = os.environ.get('PCP_DIAG_ENTRY', 'pcp_conv3x3_winograd_ws')       # or pcp_conv3x3_winograd4f


def load(path):
    L = ctypes.CDLL(str(path))
    fn = getattr(L, ENTRY)
    fn.restype = ctypes.c_int32
    fn.argtypes = [ctypes.POINTER(plib.Conv3x3)] + [ctypes.c_void_p] * 5
    L.entry = fn
    return L


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    d = torch.device('cuda:0')
    libs = {'shipped': load(HERE / 'lib' / 'libpcp_hip.so')}
    pat = os.environ.get('PCP_DIAG_VARIANTS', '')
    for p in sorted(glob.glob(str(HERE / 'lib' / 'variants' / 'libpcp_hip_*.so'))):
        name = os.path.basename(p)[len('libpcp_hip_'):-3]
        if name.startswith(pat):
            libs[name] = load(p)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    direct = ENTRY == 'pcp_conv3x3'                 # the direct implicit-GEMM kernel on the backbone's three stride-2 layers
    shapes = [(64, 64, 512, 512), (64, 128, 256, 256), (128, 256, 128, 128)] if direct else [(128, 128, 128, 128), (64, 64, 256, 256), (384, 128, 128, 128)]
    if os.environ.get('PCP_DIAG_SHAPES'):
        shapes = [tuple(int(v) for v in t.split(',')) for t in os.environ['PCP_DIAG_SHAPES'].split(';')]
    for (cin, cout, H, W) in shapes:
        x = torch.randn((B, H, W, cin), device=d)
        w = torch.randn((cout, cin, 3, 3)) * 0.05
        f4 = ENTRY.endswith('4f') or ENTRY.endswith('4h')
        if direct:
            pw, bw, cp = pack.pack_conv3x3(w, torch.zeros(cout))
            out = torch.empty((B, H // 2, W // 2, cout), device=d)
            desc = plib.Conv3x3(B, H, W, cin, cout, cp, 2, cin, cout, 1)
            flops_exec = 2.0 * B * (H // 2) * (W // 2) * cp * 9 * cin
        else:
            pk = pack.pack_conv3x3_winograd4h if ENTRY.endswith('4h') else pack.pack_conv3x3_winograd4f if f4 else pack.pack_conv3x3_winograd_ws
            pw, bw, cp = pk(w, torch.zeros(cout))
            out = torch.empty((B, H, W, cout), device=d)
            desc = plib.Conv3x3(B, H, W, cin, cout, cp, 1, cin, cout, 1)
            flops_exec = 2.0 * (36 * (B * H * W / 16) if f4 else 16 * (B * H * W / 4)) * cin * cp
        pw, bw = pw.to(d), bw.to(d)
        times = {k: [] for k in libs}
        if ENTRY.endswith('4h'):                     # the eight-wave kernel of the shipped library as the yardstick
            if 'k_wino4f' not in libs:
                L4 = ctypes.CDLL(str(HERE / 'lib' / 'libpcp_hip.so'))
                L4.entry0 = L4.pcp_conv3x3_winograd4f
                L4.entry0.restype = ctypes.c_int32
                L4.entry0.argtypes = [ctypes.POINTER(plib.Conv3x3)] + [ctypes.c_void_p] * 5
                libs['k_wino4f'] = L4
            pf = pack.pack_conv3x3_winograd4f(w, torch.zeros(cout))[0].to(d)
            libs['k_wino4f'].entry = lambda dd, xx, ww, bb, oo, ss, _f=libs['k_wino4f'].entry0, _pf=pf: _f(dd, xx, _pf.data_ptr(), bb, oo, ss)
            times['k_wino4f'] = []
        for rnd in range(7):
            for k, L in libs.items():
                for _ in range(2):
                    L.entry(ctypes.byref(desc), x.data_ptr(), pw.data_ptr(), bw.data_ptr(), out.data_ptr(), st)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    L.entry(ctypes.byref(desc), x.data_ptr(), pw.data_ptr(), bw.data_ptr(), out.data_ptr(), st)
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / 10 * 1e3)
        print('%d->%d @%dx%d B=%d   (executed %.2f GFLOP)' % (cin, cout, H, W, B, flops_exec / 1e9))
        for k, v in times.items():
            v = sorted(v)
            print('   %-12s median %8.1f us  min %8.1f us   executed %6.1f TFLOP/s (median)' % (k, v[len(v) // 2], v[0], flops_exec / v[len(v) // 2] / 1e6))


if __name__ == '__main__':
    main()
