"""stride-2 direct 3x3 layers of the DiscoNet step on pcp_conv3x3 (interleaving is done by running the script once per library build)"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd'))
from pcp_amd import ops, pack  # noqa: E402

SHAPES = [(20, 256, 256, 64, 128), (4, 512, 512, 64, 64), (4, 256, 256, 64, 128), (20, 128, 128, 128, 128), (4, 128, 128, 128, 256), (4, 128, 128, 16, 320)]
dev = 'cuda:0'
torch.manual_seed(0)
for (B, H, W, cin, cout) in SHAPES:
    stride = 2 if cin != 16 else 1
    x = torch.randn((B, H, W, cin), device=dev)
    w = torch.randn((cout, cin, 3, 3)) / (3.0 * cin ** 0.5)
    b = torch.randn((cout,))
    pw, pb, cp = pack.pack_conv3x3(w, b)
    pw, pb = pw.to(dev), pb.to(dev)
    out = ops.conv3x3(x, pw, pb, cin, cout, cp, stride=stride, relu=True)
    ref = torch.relu(torch.nn.functional.conv2d(x[:1].permute(0, 3, 1, 2).double().cpu(), w.double(), b.double(), stride=stride, padding=1)).permute(0, 2, 3, 1).float()
    err = float((out[:1].cpu() - ref).abs().max())
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv3x3(x, pw, pb, cin, cout, cp, stride=stride, relu=True, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t = sorted(ts)[len(ts) // 2]
    fl = 2.0 * B * (H // stride) * (W // stride) * cp * 9 * cin
    print('B%-2d %3dx%-3d %3d->%-3d s%d  %7.1f us  %5.1f TF   |err| %.1e' % (B, H, W, cin, cout, stride, t, fl / t / 1e6, err), flush=True)
