"""pcp_pointwise_wgrad (fp32) on the PFN's shape of the training step (rows = 1.4 M, 64 x 64) and a weightor shape; median of 10 launches"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd'))
from pcp_amd import train_ops as tops
dev = 'cuda:0'
torch.manual_seed(0)
for rows, n, k in [(1396074, 64, 64), (1396074, 32, 16), (65536, 64, 256), (65536, 16, 64)]:
    a = torch.randn((rows, n), device=dev)
    b = torch.randn((rows, k), device=dev)
    out = torch.empty((n, k), device=dev)
    m = min(rows, 200000)
    tops.pointwise_wgrad(tops.rowmap(a, n), tops.rowmap(b, k), rows, out)
    ref = (a[:m].double().t() @ b[:m].double())
    o2 = torch.empty((n, k), device=dev)
    tops.pointwise_wgrad(tops.rowmap(a[:m].contiguous(), n), tops.rowmap(b[:m].contiguous(), k), m, o2)
    err = float((o2.double() - ref).abs().max() / ref.abs().max())
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tops.pointwise_wgrad(tops.rowmap(a, n), tops.rowmap(b, k), rows, out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t = sorted(ts)[5]
    print('rows %8d  %3d x %-3d  %7.1f us  %5.1f TFLOP/s  %5.2f TB/s   rel err %.1e' % (rows, n, k, t, 2.0 * rows * n * k / t / 1e6, rows * (n + k) * 4 / t / 1e6, err))
