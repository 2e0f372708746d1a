"""Shared plumbing for the dense conv stacks: turns (Conv2d | ConvTranspose2d) [+ BatchNorm2d] parameter containers into
packed HIP weights and launches them on NHWC buffers."""
import os

import torch
import torch.nn as nn

from pcp_amd import lib, ops, pack


# fused Winograd F(2x2,3x3) needs enough workgroups to fill the 256 CUs; below that the direct kernel's smaller tiles win
# (measured on MI355X, tools/bench_conv.py: >= 256 workgroups -> x1.35 .. x2.0 over the direct kernel)
WINOGRAD_MIN_WORKGROUPS = 256
# F(4x4,3x3) through memory (three launches, csrc/wino4.hip) wins on the wide layers once its batched GEMM fills the chip twice over
# (tools/bench_conv.py on MI355X, 4 frames: 768->768 @128 x1.50, 384->384 @128 x1.26, 128->384 @128 x1.20, 256->256 @64 x1.16 over the fused
# F(2x2) kernel; narrower outputs or fewer tiles lose)
B3_MIN_WORKGROUPS = 256
WINOGRAD4_MIN_COUT = 256
WINOGRAD4_MIN_WORKGROUPS = 512
CONV_ALGO = os.environ.get('PCP_CONV_ALGO', 'auto')          # auto | direct | winograd (F(2x2) only) | winograd4 | winograd4f | winograd4h | winograd4c | bf16x3 (opt-in: split-bf16 products)


_ENV_DATA = getattr(os.environ, '_data', None)           # CPython's backing dict of os.environ (bytes keys on POSIX)
_ENV_KEY = os.environ.encodekey('PCP_CONV_ALGO') if hasattr(os.environ, 'encodekey') else None


def conv_algo():
    """'bf16' (plain bf16 products) is the mixed-precision TRAINING mode (bench.py refuses it without --train): like autocast it also
    covers the frozen teachers' forward passes inside a training iteration; in this module it selects the same launches as 'bf16x3'
    with single products.
    Read from the environment at every call (tests and bench.py --optin switch it between forwards) -- through the backing dict:
    os.environ.get() encodes the key and decodes the value every time, ~450 calls and half a millisecond of host time per DiscoNet step."""
    if _ENV_DATA is not None and _ENV_KEY is not None:
        v = _ENV_DATA.get(_ENV_KEY)
        if v is None:
            return CONV_ALGO
        return os.environ.decodevalue(v) if isinstance(v, bytes) else v
    return os.environ.get('PCP_CONV_ALGO', CONV_ALGO)


def _plain_bf16():
    return conv_algo() == 'bf16'


# fused F(4x4,3x3) (csrc/wino4f.hip: one workgroup per CU = 16 x 32 pixels x 64 channels): measured against the fused F(2x2) kernel on MI355X
# (tools/bench_conv.py, 4 / 20 frames): 64->64 @256 x1.21 / x1.12, 128->128 @128 x1.34 / x1.36, 384->64 @128 - / x1.41, 384->128 @128 x1.41 /
# x1.46, 128->384 x1.37 / x1.47; it loses when its grid does not fill the chip (128->128 @64 at 4 frames: 64 workgroups) or covers it unevenly
# (320 workgroups on 256 CUs), and the through-memory F(4x4) path keeps the very wide layers (768 -> 768: x0.84)
WINOGRAD4F_MIN_WORKGROUPS = 256
WINOGRAD4F_MAX_CIN = 448
WINOGRAD4H = os.environ.get('PCP_WINO4H', 'auto')          # auto | 0 (never dispatch k_wino4h)
WINOGRAD4H_MAX_CIN = 128
WINOGRAD4H_MIN_WORKGROUPS = int(os.environ.get('PCP_WINO4H_MIN_WGS', '256'))
WINOGRAD4F_MAX_INPUT_BYTES = 0x7fffffff                    # buffer-descriptor addressing (tests lower it to exercise the fallback)


class PackedConv:
    """One fused conv(+BN)(+ReLU) launch description."""
    __slots__ = ('kind', 'w', 'b', 'cin', 'cout', 'cout_pad', 'stride', 'relu', 'wino', 'b3', 'w4', 'w4f', 'w4h', 'w4c', 'mp')

    def _use_winograd4f(self, x, out, out_ch_off, in_ch_off=0):
        algo = conv_algo()
        if self.kind != '3x3' or getattr(self, 'w4f', None) is None or algo in ('direct', 'winograd', 'winograd4', 'bf16x3', 'bf16'):
            return False
        if out is not None and (out.shape[-1] % 4 != 0 or out_ch_off % 4 != 0):
            return False                                   # 16-byte output stores
        # the kernel's own limits (csrc/wino4f.hip f4_geom): 16-byte input loads through a buffer descriptor with 32-bit byte offsets.
        # Outside them the launch returns PCP_ERR_UNSUPPORTED / PCP_ERR_ARG, so the dispatch falls through to the other kernels instead
        if x.shape[-1] % 4 != 0 or in_ch_off % 4 != 0 or x.numel() * 4 > WINOGRAD4F_MAX_INPUT_BYTES:
            return False
        if algo in ('winograd4f', 'winograd4h', 'winograd4c'):
            return True
        B, H, W, _ = x.shape
        wgs = B * ((H + 15) // 16) * ((W + 31) // 32) * (self.w4f[2] // 64)
        if self.cin > WINOGRAD4F_MAX_CIN and getattr(self, 'w4', None) is not None:
            return False
        if self._prefer_winograd4h(x):
            return True                                    # the half-size items of k_wino4h also cover grids k_wino4f fills unevenly
        if H * W <= 64 * 64 and self.cin <= 128:
            return False                                   # 8 spatial tiles per frame, 16 K slices: F(2x2) wins (profiles/r02_bench_conv_b*.txt)
        return wgs >= WINOGRAD4F_MIN_WORKGROUPS and (wgs % 256 == 0 or wgs >= 512)

    def _prefer_winograd4h(self, x):
        """which fused F(4x4) kernel: k_wino4h (two four-wave workgroups per CU, 16 x 16-pixel items: one workgroup's prologue / epilogue
        under the other's MFMAs) or k_wino4f (one eight-wave workgroup, 16 x 32-pixel items: half the weight traffic per product).
        Interleaved A/B on MI355X (tools/bench_w4h.py, profiles/r03_wino4h_ab.txt): 4h wins up to 128 input channels wherever its
        grid covers the chip (>= 256 workgroups), by 3-4 % on full grids and 20-40 % on the grids 4f fills unevenly; 4f keeps cin >= 256."""
        algo = conv_algo()
        if algo in ('winograd4h', 'winograd4c'):
            return True
        if algo == 'winograd4f' or WINOGRAD4H == '0':
            return False
        B, H, W, _ = x.shape
        nb = self.w4f[2] // 64
        wgs = B * ((H + 15) // 16) * ((W + 15) // 16) * nb
        if wgs < WINOGRAD4H_MIN_WORKGROUPS:
            return False
        # wider layers only where the eight-wave kernel's 16 x 32-pixel items leave CUs idle (CenterHead's 384 -> 64 conv at 4 frames: 128
        # items; k_wino4h 95 us against 145 us on the fused F(2x2) kernel that used to take it, tools/bench_conv.py)
        return self.cin <= WINOGRAD4H_MAX_CIN or B * ((H + 15) // 16) * ((W + 31) // 32) * nb < WINOGRAD4F_MIN_WORKGROUPS

    def _w4h(self):
        if getattr(self, 'w4h', None) is None:
            self.w4h = (pack.repack_winograd4f_to_4h(self.w4f[0]), self.w4f[1], self.w4f[2])
        return self.w4h

    def _w4c(self):
        if getattr(self, 'w4c', None) is None:
            self.w4c = (pack.repack_winograd4f_to_4c(self.w4f[0]), self.w4f[1], self.w4f[2])
        return self.w4c

    def _use_winograd4(self, x):
        algo = conv_algo()
        if self.kind != '3x3' or getattr(self, 'w4', None) is None or algo in ('direct', 'winograd', 'bf16x3', 'bf16'):
            return False
        if algo == 'winograd4':
            return True
        B, H, W, _ = x.shape
        tiles = B * ((H + 3) // 4) * ((W + 3) // 4)
        # the through-memory GEMM pays two extra passes over V and M: measured a win only from 256 input channels and a full 128-wide N tile
        # (128 -> 128 @64^2 B = 4: 46 vs 32 us fused F(2x2); 384 -> 64 @128^2: 225 vs 145 us; profiles/r02_bench_conv_b*.txt)
        if self.cin < 256 or self.cout < 128:
            return False
        return 36 * ((tiles + 127) // 128) * (self.w4[2] // 128) >= WINOGRAD4_MIN_WORKGROUPS

    def _use_winograd(self, x):
        algo = conv_algo()
        if self.kind != '3x3' or getattr(self, 'wino', None) is None or algo == 'direct':
            return False
        if algo == 'winograd':
            return True
        B, H, W, _ = x.shape
        # workgroups of the 32-tile instantiation (8 rows x 16 columns of pixels x 64 channels), the finest the library uses
        return B * ((H + 7) // 8) * ((W + 15) // 16) * (self.wino[2] // 64) >= WINOGRAD_MIN_WORKGROUPS

    def _use_bf16x3(self, x):
        """opt-in only (PCP_CONV_ALGO=bf16x3), and only where the launch fills the chip (>= 256 workgroups of 16x16 px x 64 ch)"""
        if self.kind != '3x3' or getattr(self, 'b3', None) is None or conv_algo() not in ('bf16x3', 'bf16'):
            return False
        B, H, W, _ = x.shape
        Ho, Wo = (H - 1) // self.stride + 1, (W - 1) // self.stride + 1
        th = 16 if self.stride == 1 else 8
        return B * ((Ho + th - 1) // th) * ((Wo + 15) // 16) * (self.b3[2] // 64) >= B3_MIN_WORKGROUPS

    def run(self, x, out=None, in_ch_off=0, out_ch_off=0):
        if getattr(self, 'mp', None) is not None and self.kind == '3x3' and _plain_bf16():
            # the bf16 loop of config 5 (include/pcp_hip_mp.h): frozen teachers inside a training iteration run the bf16 kernels on bf16
            # activations -- BatchNorm folded into the bf16 weights + fp32 bias; an fp32 input (the sparse first layer's output, a canvas)
            # is cast once, the output is bf16 unless the caller's buffer says float32
            from pcp_amd import train_ops as tops
            if x.dtype != torch.bfloat16 or in_ch_off % 8 or x.shape[-1] % 8:
                x, in_ch_off = x[..., in_ch_off:in_ch_off + self.cin].to(torch.bfloat16).contiguous(), 0
            wp, bp, cp = self.mp
            return tops.mp_conv3x3(x, wp, bp, self.cin, self.cout, cp, stride=self.stride, relu=self.relu, out=out, in_ch_off=in_ch_off,
                                   out_ch_off=out_ch_off)
        if (getattr(self, 'mp', None) is not None and self.kind != '3x3' and _plain_bf16() and x.dtype == torch.bfloat16 and in_ch_off % 8 == 0
                and x.shape[-1] % 8 == 0 and (out is None or (out.shape[-1] % 8 == 0 and out_ch_off % 8 == 0))):
            # bf16 loop: a pointwise layer behind a bf16 3x3 layer reads the bf16 map as it lies (pcp_mp_pointwise); its output takes the
            # storage type of the caller's buffer (float32 when it allocates here: the consumers outside the conv stacks are fp32 kernels)
            from pcp_amd import train_ops as tops
            mode = {'plain': lib.PW_PLAIN, 's2d': lib.PW_SPACE2DEPTH, 'd2s': lib.PW_DEPTH2SPACE}[self.kind]
            wp, bp, cp = self.mp
            return tops.mp_pointwise(x, wp, bp, mode, self.cin, self.cout, cp, relu=self.relu, out=out, in_ch_off=in_ch_off,
                                     out_ch_off=out_ch_off, out_dtype=torch.float32)
        if x.dtype != torch.float32:
            x, in_ch_off = x[..., in_ch_off:in_ch_off + self.cin].float().contiguous(), 0      # the fp32 kernels' view of a bf16 activation
        if self._use_bf16x3(x):
            w3, b3, cp3 = self.b3
            return ops.conv3x3_bf16x3(x, w3, b3, self.cin, self.cout, cp3, stride=self.stride, relu=self.relu, out=out,
                                      in_ch_off=in_ch_off, out_ch_off=out_ch_off, plain=_plain_bf16())
        if self._use_winograd4f(x, out, out_ch_off, in_ch_off):
            if self._prefer_winograd4h(x):
                # the half-size items: k_wino4c (round 4: waves split over the output channels, output transform in registers; the bits of
                # k_wino4h, 2 - 6 % faster on every shape of the step, profiles/r04_wino4c_ab.txt) unless k_wino4h is asked for by name
                if conv_algo() != 'winograd4h':
                    u, ub, ucp = self._w4c()
                    return ops.conv3x3_winograd4c(x, u, ub, self.cin, self.cout, ucp, relu=self.relu, out=out, in_ch_off=in_ch_off,
                                                  out_ch_off=out_ch_off)
                u, ub, ucp = self._w4h()
                return ops.conv3x3_winograd4h(x, u, ub, self.cin, self.cout, ucp, relu=self.relu, out=out, in_ch_off=in_ch_off,
                                              out_ch_off=out_ch_off)
            u, ub, ucp = self.w4f
            return ops.conv3x3_winograd4f(x, u, ub, self.cin, self.cout, ucp, relu=self.relu, out=out, in_ch_off=in_ch_off,
                                          out_ch_off=out_ch_off)
        if self._use_winograd4(x):
            u, ub, ucp = self.w4
            return ops.conv3x3_winograd4(x, u, ub, self.cin, self.cout, ucp, relu=self.relu, out=out, in_ch_off=in_ch_off,
                                         out_ch_off=out_ch_off)
        if self._use_winograd(x):
            u, ub, ucp = self.wino
            return ops.conv3x3_winograd(x, u, ub, self.cin, self.cout, ucp, relu=self.relu, out=out, in_ch_off=in_ch_off,
                                        out_ch_off=out_ch_off)
        if self.kind == '3x3':
            return ops.conv3x3(x, self.w, self.b, self.cin, self.cout, self.cout_pad, stride=self.stride, relu=self.relu, out=out,
                               in_ch_off=in_ch_off, out_ch_off=out_ch_off)
        mode = {'plain': lib.PW_PLAIN, 's2d': lib.PW_SPACE2DEPTH, 'd2s': lib.PW_DEPTH2SPACE}[self.kind]
        return ops.pointwise(x, self.w, self.b, mode, self.cin, self.cout, self.cout_pad, relu=self.relu, out=out,
                             in_ch_off=in_ch_off, out_ch_off=out_ch_off)


def _winograd4_shape(cin, cout, stride):
    return (stride == 1 and cin % pack.WINO4_CK == 0 and cin >= 128 and cout % 4 == 0 and cout >= WINOGRAD4_MIN_COUT
            and conv_algo() not in ('direct', 'winograd', 'bf16x3', 'bf16'))


def _winograd4f_shape(cin, cout, stride):
    # layers auto dispatch never sends to the fused kernel (cin above its cap with the through-memory form available) do not get the
    # 4x-sized fused weight form packed at all; PCP_CONV_ALGO=winograd4f packs it for every eligible layer
    if conv_algo() not in ('winograd4f', 'winograd4h', 'winograd4c') and cin > WINOGRAD4F_MAX_CIN and _winograd4_shape(cin, cout, stride):
        return False
    return (stride == 1 and cin % 8 == 0 and cout % 4 == 0 and cout >= 48 and conv_algo() not in ('direct', 'winograd', 'winograd4', 'bf16x3', 'bf16'))


def _fold(conv, bn, out_axis):
    w = conv.weight.detach().float()
    cb = conv.bias.detach().float() if conv.bias is not None else None
    if bn is None:
        n_out = w.shape[out_axis]
        return w, (cb if cb is not None else torch.zeros(n_out, dtype=torch.float32, device=w.device))
    return pack.fold_bn(w, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps, conv_bias=cb,
                        out_axis=out_axis)


def _pack_mp(w, b):
    """(folded) fp32 weights -> the bf16 form of pcp_mp_conv3x3 + padded fp32 bias"""
    from pcp_amd import train_ops as tops
    wp, opad = tops.mp_pack_conv3x3(w.contiguous().float())
    bp = torch.zeros(opad, dtype=torch.float32, device=w.device)
    bp[:b.numel()] = b
    return wp, bp, opad


def _pack_mp_pointwise(pc):
    """bf16 copy of a pointwise layer's packed weights for pcp_mp_pointwise (PCP_CONV_ALGO=bf16 only, shapes the kernel takes)"""
    if conv_algo() == 'bf16' and pc.w.is_cuda and pc.cin % 32 == 0 and pc.cout % 8 == 0 and pc.cout_pad % 64 == 0:
        pc.mp = (pc.w.to(torch.bfloat16).contiguous(), pc.b, pc.cout_pad)


def pack_conv_module(conv, bn=None, relu=True):
    """conv: nn.Conv2d (3x3 s1/s2 p1 | 1x1 | k2 s2) or nn.ConvTranspose2d (k1 s1 | k2 s2)."""
    pc = PackedConv()
    pc.relu = relu
    pc.stride = 1
    pc.wino = None
    pc.b3 = None
    pc.w4 = None
    pc.w4f = None
    pc.w4h = None
    pc.w4c = None
    pc.mp = None
    if isinstance(conv, nn.ConvTranspose2d):
        w, b = _fold(conv, bn, out_axis=1)
        k, s = conv.kernel_size[0], conv.stride[0]
        pc.cin, pc.cout = w.shape[0], w.shape[1]
        if k == 1 and s == 1:
            pc.kind = 'plain'
            pc.w, pc.b, pc.cout_pad = pack.pack_convT1x1(w, b)
        elif k == 2 and s == 2:
            pc.kind = 'd2s'
            pc.w, pc.b, pc.cout_pad = pack.pack_convT2x2_s2(w, b)
        else:
            raise NotImplementedError('ConvTranspose2d k=%d s=%d has no HIP kernel in this build' % (k, s))
        _pack_mp_pointwise(pc)
        return pc
    w, b = _fold(conv, bn, out_axis=0)
    k, s = conv.kernel_size[0], conv.stride[0]
    pc.cin, pc.cout = w.shape[1], w.shape[0]
    if k == 3 and s in (1, 2):
        pc.kind = '3x3'
        pc.stride = s
        pc.w, pc.b, pc.cout_pad = pack.pack_conv3x3(w, b)
        if s == 1 and pc.cin % pack.WINO_CK == 0 and pc.cout >= 48:
            pc.wino = pack.pack_conv3x3_winograd(w, b)
        if conv_algo() == 'bf16x3' and pc.cin % pack.CK == 0 and pc.cout >= 48:
            pc.b3 = pack.pack_conv3x3_bf16x3(w, b)
        if conv_algo() == 'bf16' and pc.cin % 16 == 0 and pc.cout % 8 == 0 and w.is_cuda:
            pc.mp = _pack_mp(w, b)
        if _winograd4_shape(pc.cin, pc.cout, s):
            pc.w4 = pack.pack_conv3x3_winograd4(w, b)
        if _winograd4f_shape(pc.cin, pc.cout, s):
            pc.w4f = pack.pack_conv3x3_winograd4f(w, b)
    elif k == 1 and s == 1:
        pc.kind = 'plain'
        pc.w, pc.b, pc.cout_pad = pack.pack_plain(w, b)
    elif k == 2 and s == 2:
        pc.kind = 's2d'
        pc.w, pc.b, pc.cout_pad = pack.pack_conv2x2_s2(w, b)
    else:
        raise NotImplementedError('Conv2d k=%d s=%d has no HIP kernel in this build' % (k, s))
    if pc.kind != '3x3':
        _pack_mp_pointwise(pc)
    return pc


def pack_conv_raw(w, b, relu, stride=1):
    """3x3 conv from an explicit (already folded) weight/bias pair, e.g. the fused CenterHead branches."""
    pc = PackedConv()
    pc.kind = '3x3'
    pc.relu = relu
    pc.stride = stride
    pc.cin, pc.cout = w.shape[1], w.shape[0]
    pc.w, pc.b, pc.cout_pad = pack.pack_conv3x3(w, b)
    pc.wino = pack.pack_conv3x3_winograd(w, b) if (stride == 1 and pc.cin % pack.WINO_CK == 0 and pc.cout >= 48) else None
    pc.b3 = pack.pack_conv3x3_bf16x3(w, b) if (conv_algo() == 'bf16x3' and pc.cin % pack.CK == 0 and pc.cout >= 48) else None
    pc.mp = _pack_mp(w, b) if (conv_algo() == 'bf16' and pc.cin % 16 == 0 and pc.cout % 8 == 0 and w.is_cuda) else None
    pc.w4 = pack.pack_conv3x3_winograd4(w, b) if _winograd4_shape(pc.cin, pc.cout, stride) else None
    pc.w4f = pack.pack_conv3x3_winograd4f(w, b) if _winograd4f_shape(pc.cin, pc.cout, stride) else None
    pc.w4h = None
    pc.w4c = None
    return pc
