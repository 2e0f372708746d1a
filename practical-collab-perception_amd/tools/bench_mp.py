"""per-layer timing of the bf16 training kernels (pcp_mp_conv3x3, pcp_mp_conv3x3_wgrad) on the DiscoNet shapes:
python tools/bench_mp.py [frames]   -> us per launch, executed TFLOP/s, algorithmic GB/s"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pcp_amd import train_ops as tops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = 'cuda:0'
SHAPES = [(64, 64, 256, 1), (128, 128, 128, 1), (256, 256, 64, 1), (384, 64, 256, 1), (64, 128, 256, 2), (128, 256, 128, 2), (64, 64, 512, 2)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print('frames %d' % B)
for cin, cout, hw, s in SHAPES:
    x = torch.randn((B, hw, hw, cin), device=dev).to(torch.bfloat16)
    w = torch.randn((cout, cin, 3, 3), device=dev) * 0.05
    packed, opad = tops.mp_pack_conv3x3(w)
    bias = torch.zeros(opad, device=dev)
    ho = hw // s
    out = torch.empty((B, ho, ho, cout), dtype=torch.bfloat16, device=dev)
    us = timeit(lambda: tops.mp_conv3x3(x, packed, bias, cin, cout, opad, stride=s, relu=True, out=out))
    fl = 2.0 * B * ho * ho * cout * 9 * cin
    by = (x.numel() + out.numel()) * 2
    print('conv  %3d->%3d @%3d s%d  %8.1f us  %7.1f TF  %6.2f TB/s' % (cin, cout, hw, s, us, fl / us / 1e6, by / us / 1e6))
    dy = torch.randn((B, ho, ho, cout), device=dev).to(torch.bfloat16)
    dw = torch.zeros((cout, cin, 3, 3), device=dev)
    us = timeit(lambda: tops.mp_conv3x3_wgrad(x, dy, cin, cout, s, dw))
    by = (x.numel() + dy.numel()) * 2
    print('wgrad %3d->%3d @%3d s%d  %8.1f us  %7.1f TF  %6.2f TB/s' % (cin, cout, hw, s, us, fl / us / 1e6, by / us / 1e6))
