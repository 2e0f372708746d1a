import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd'))
from pcp_amd import ops, pack
dev = 'cuda:0'
torch.manual_seed(0)
for (B, H, W, cin, cout) in [(1, 16, 16, 16, 128), (1, 16, 16, 8, 128), (1, 32, 32, 64, 128)]:
    x = torch.randn((B, H, W, cin), device=dev)
    w = torch.randn((cout, cin, 3, 3), device=dev) / (3.0 * cin ** 0.5)
    b = torch.randn((cout,), device=dev)
    uc, bc, cpc = pack.pack_conv3x3_winograd4c(w, b)
    out = ops.conv3x3_winograd4c(x, uc, bc, cin, cout, cpc, relu=False)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1).float()
    err = (out - ref).abs()
    print((B, H, W, cin, cout), 'max err', float(err.max()))
    bad_ch = (err.amax(dim=(0, 1, 2)) > 1e-3).nonzero().flatten().tolist()
    bad_px = (err.amax(dim=3) > 1e-3)[0]
    print('  bad channels:', bad_ch[:40], ' bad pixels:', int(bad_px.sum()), 'of', bad_px.numel())
    if bad_px.any():
        ys, xs = bad_px.nonzero(as_tuple=True)
        print('  bad rows', sorted(set(ys.tolist()))[:20], 'cols', sorted(set(xs.tolist()))[:20])
