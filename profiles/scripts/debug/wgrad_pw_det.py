import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd'))
from pcp_amd import train_ops as tops
dev = 'cuda:0'
torch.manual_seed(0)
for rows, n, k in [(1396074, 64, 64), (1396074, 32, 16), (65536, 64, 256), (300000, 64, 64)]:
    a = torch.randn((rows, n), device=dev)
    b = torch.randn((rows, k), device=dev)
    outs = []
    for rep in range(6):
        out = torch.empty((n, k), device=dev)
        tops.pointwise_wgrad(tops.rowmap(a, n), tops.rowmap(b, k), rows, out)
        torch.cuda.synchronize()
        outs.append(out.clone())
    same = [bool(torch.equal(outs[0], o)) for o in outs[1:]]
    print(rows, n, k, 'repeat equal:', same, 'max diff', max(float((outs[0] - o).abs().max()) for o in outs[1:]))
