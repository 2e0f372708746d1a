import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, 'practical-collab-perception_amd')
from pcp_amd import ops, pack
d = torch.device('cuda:0')
for (cin, cout, h, w, batch) in [(64, 64, 32, 32, 1), (128, 128, 16, 64, 2), (128, 128, 16, 64, 1), (128, 64, 16, 32, 1), (64, 64, 16, 64, 2), (72, 64, 16, 32, 1),
                                 (16, 64, 16, 32, 1), (8, 64, 16, 32, 1), (24, 64, 16, 32, 1), (128, 128, 64, 64, 3)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn((batch, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, 3, 3), generator=g) * 0.05
    b = torch.randn((cout,), generator=g) * 0.1
    want = F.conv2d(x, wt, b, padding=1)
    pk, bp, cp = pack.pack_conv3x3_winograd4f(wt, b)
    got = ops.conv3x3_winograd4f(ops.as_nhwc(x.to(d)), pk.to(d), bp.to(d), cin, cout, cp, relu=False).permute(0, 3, 1, 2).cpu()
    err = (got - want).abs()
    bad = (err > 1e-3)
    print((cin, cout, h, w, batch), 'max err %.3g' % float(err.max()), 'bad frac %.4f' % float(bad.float().mean()),
          'bad per batch', [float(bad[i].float().mean()) for i in range(batch)],
          'bad channels', bad.any(dim=(0, 2, 3)).nonzero().flatten().tolist()[:8], 'rows', bad.any(dim=(0, 1, 3)).nonzero().flatten().tolist()[:10],
          'cols', bad.any(dim=(0, 1, 2)).nonzero().flatten().tolist()[:10])
