"""Micro-benchmark of the 3x3 convolution kernels on the layer shapes of the five configs (B frames per launch):
direct implicit GEMM (pcp_conv3x3) vs fused Winograd F(2x2,3x3) (pcp_conv3x3_winograd).  Prints algorithmic TFLOP/s
(2*B*H*W*Cout*9*Cin / time) for both; used to choose the per-layer algorithm in pcdet/models/convnet.py."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcp_amd import ops, pack  # noqa: E402

LAYERS = [  # (name, cin, cout, H, W)
    ('backbone.b0 64->64 @256', 64, 64, 256, 256), ('backbone.b1 128->128 @128', 128, 128, 128, 128),
    ('backbone.b2 128->128 @64', 128, 128, 64, 64), ('backbone.b2 256->256 @64', 256, 256, 64, 64),
    ('head.shared 384->64 @128', 384, 64, 128, 128), ('head.stage1 64->320 @128', 64, 320, 128, 128),
    ('hunter.conv_input 384->384 @128', 384, 384, 128, 128), ('hunter.weightor 768->768 @128', 768, 768, 128, 128),
    ('disco.compress 384->128 @128', 384, 128, 128, 128), ('disco.decompress 128->384 @128', 128, 384, 128, 128),
]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    only = int(sys.argv[2]) if len(sys.argv) > 2 else -1            # layer index (for rocprof --pmc runs)
    d = torch.device('cuda:0')
    print('B = %d' % B)
    for li, (name, cin, cout, H, W) in enumerate(LAYERS):
        if only >= 0 and li != only:
            continue
        x = torch.randn((B, H, W, cin), device=d)
        w = torch.randn((cout, cin, 3, 3)) * 0.05
        b = torch.zeros(cout)
        pd, bd, cpd = pack.pack_conv3x3(w, b)
        pw, bw, cpw = pack.pack_conv3x3_winograd(w, b)
        p3, b3, cp3 = pack.pack_conv3x3_bf16x3(w, b)
        pd, bd, pw, bw, p3, b3 = pd.to(d), bd.to(d), pw.to(d), bw.to(d), p3.to(d), b3.to(d)
        out = torch.empty((B, H, W, cout), device=d)
        flops = 2.0 * B * H * W * cout * 9 * cin
        td = timeit(lambda: ops.conv3x3(x, pd, bd, cin, cout, cpd, out=out))
        tw = timeit(lambda: ops.conv3x3_winograd(x, pw, bw, cin, cout, cpw, out=out))
        t3 = timeit(lambda: ops.conv3x3_bf16x3(x, p3, b3, cin, cout, cp3, out=out))
        tws = None
        if cin % 32 == 0:
            pws, bws, cpws = pack.pack_conv3x3_winograd_ws(w, b)
            pws, bws = pws.to(d), bws.to(d)
            ref = torch.empty_like(out)
            ops.conv3x3_winograd(x, pw, bw, cin, cout, cpw, out=ref)
            tws = timeit(lambda: ops.conv3x3_winograd_ws(x, pws, bws, cin, cout, cpws, out=out))
            errws = float((out - ref).abs().max() / ref.abs().max())
        line = ('%-36s direct %8.1f us %6.1f TF | winograd %8.1f us %6.1f TF (algorithmic) | x%.2f | bf16x3 (opt-in) %8.1f us %6.1f TF' %
                (name, td * 1e6, flops / td / 1e12, tw * 1e6, flops / tw / 1e12, td / tw, t3 * 1e6, flops / t3 / 1e12))
        if tws is not None:
            line += ' | WS %8.1f us %6.1f TF (x%.2f vs winograd, diff %.1e)' % (tws * 1e6, flops / tws / 1e12, tw / tws, errws)
        p4f, b4f, cp4f = pack.pack_conv3x3_winograd4f(w, b)
        p4f, b4f = p4f.to(d), b4f.to(d)
        ref = torch.empty_like(out)
        ops.conv3x3_winograd(x, pw, bw, cin, cout, cpw, out=ref)
        t4f = timeit(lambda: ops.conv3x3_winograd4f(x, p4f, b4f, cin, cout, cp4f, out=out))
        err4f = float((out - ref).abs().max() / ref.abs().max())
        line += ' | F(4x4) FUSED %8.1f us %6.1f TF (x%.2f vs winograd, diff %.1e)' % (t4f * 1e6, flops / t4f / 1e12, tw / t4f, err4f)
        p4h = pack.repack_winograd4f_to_4h(p4f)
        t4h = timeit(lambda: ops.conv3x3_winograd4h(x, p4h, b4f, cin, cout, cp4f, out=out))
        err4h = float((out - ref).abs().max() / ref.abs().max())
        line += ' | F(4x4) FUSED 2 WG/CU %8.1f us %6.1f TF (x%.2f vs the 8-wave kernel, diff %.1e)' % (t4h * 1e6, flops / t4h / 1e12, t4f / t4h, err4h)
        if cin % pack.WINO4_CK == 0 and cout % 4 == 0 and cin >= 128:
            ref = out.clone()                                    # bf16x3 result ran last; recompute the fp32 Winograd result as the yardstick
            ops.conv3x3_winograd(x, pw, bw, cin, cout, cpw, out=ref)
            p4, b4, cp4 = pack.pack_conv3x3_winograd4(w, b)
            p4, b4 = p4.to(d), b4.to(d)
            t4 = timeit(lambda: ops.conv3x3_winograd4(x, p4, b4, cin, cout, cp4, out=out))
            err = float((out - ref).abs().max() / ref.abs().max())
            st = []
            for _ in range(5):
                ops.conv3x3_winograd4(x, p4, b4, cin, cout, cp4, out=out, stage_times=st)
            st = [min(r[i] for r in st) * 1e3 for i in range(3)]
            line += ' | F(4x4) %8.1f us %6.1f TF  (max diff vs F(2x2) %.1e of scale; input / gemm / output %.0f / %.0f / %.0f us)' % (
                t4 * 1e6, flops / t4 / 1e12, err, st[0], st[1], st[2])
        print(line)


if __name__ == '__main__':
    main()
