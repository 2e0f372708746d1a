"""One fused-Winograd conv shape launched a few times (a target for rocprofv3 --pmc passes): run_conv_once.py 4f|4h B H W cin cout [reps]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcp_amd import ops, pack  # noqa: E402


def main():
    kern = sys.argv[1]
    B, H, W, cin, cout = [int(v) for v in sys.argv[2:7]]
    reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
    d = torch.device('cuda:0')
    x = torch.randn((B, H, W, cin), device=d)
    w = torch.randn((cout, cin, 3, 3)) * 0.05
    pk = pack.pack_conv3x3_winograd4h if kern == '4h' else pack.pack_conv3x3_winograd4f
    fn = ops.conv3x3_winograd4h if kern == '4h' else ops.conv3x3_winograd4f
    u, b, cp = pk(w, torch.zeros(cout))
    u, b = u.to(d), b.to(d)
    out = torch.empty((B, H, W, cout), device=d)
    for _ in range(reps):
        fn(x, u, b, cin, cout, cp, relu=True, out=out)
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
