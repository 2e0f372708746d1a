"""Hand-rolled forward/backward of the dense layers for the training path (no torch.autograd inside): every layer saves what its
backward needs, `backward(dout)` writes parameter gradients straight into `param.grad` and returns the input gradient.

A conv layer of the trainable branch is  y = conv(x) [+ bias];  a = relu?(bn_train?(y)).
  forward   conv    -> pcp_conv3x3 / pcp_conv3x3_winograd / pcp_pointwise with the RAW weights (no BN folding, relu = 0)
            bn      -> pcp_bn_train_stats (batch statistics, running stats updated) + pcp_scale_shift_act
  backward  bn+relu -> pcp_bn_act_backward (in place on the incoming gradient)
            bias    -> pcp_colsum
            wgrad   -> pcp_conv3x3_wgrad / pcp_pointwise_wgrad (pixel-contraction MFMA GEMMs)
            dgrad   -> the FORWARD kernels again with flipped / transposed weights (stride-2 3x3: on the zero-dilated gradient)

Reference modules these stand for: nn.Conv2d / nn.ConvTranspose2d + nn.BatchNorm2d + nn.ReLU stacks of
pcdet/models/backbones_2d/base_bev_backbone.py:30-69, dense_heads/center_head.py:24-29,75-82, bev_layers/v2x_fusion_disco.py:11-17,51-63.
"""
import os

import torch
import torch.nn as nn

from . import lib, ops, pack
from . import train_ops as tops


B3_MIN_WORKGROUPS = 256        # the bf16 / bf16x3 conv kernels are used where a launch fills the chip (tests lower it)


def fused_f4_choice(B, H, W, cin, cout):
    """which fused Winograd F(4x4,3x3) kernel a stride-1 3x3 conv cin -> cout on a (B, H, W) map goes to in the training step: None | '4f' |
    '4h'.  The rule of the inference dispatch (pcdet/models/convnet.py::PackedConv._use_winograd4f / _prefer_winograd4h): k_wino4h up to 128
    input channels or where k_wino4f's 16 x 32-pixel items leave CUs idle, k_wino4f for the wider layers, neither below one item per CU
    slot.  PCP_CONV_ALGO = winograd4f | winograd4h forces a kernel (tests), direct | winograd | bf16* switch both off."""
    algo = os.environ.get('PCP_CONV_ALGO', 'auto')
    if cin % 8 or cout % 4 or cout < 48 or algo in ('direct', 'winograd', 'winograd4', 'bf16x3', 'bf16'):
        return None
    if algo == 'winograd4f':
        return '4f'
    if algo == 'winograd4h':
        return '4h'
    nb = pack.round_up(cout, 64) // 64
    w4h = B * ((H + 15) // 16) * ((W + 15) // 16) * nb
    w4f = B * ((H + 15) // 16) * ((W + 31) // 32) * nb
    if w4h >= 256 and (cin <= 128 or w4f < 256):
        return '4h'
    if cin <= 448 and w4f >= 256 and (w4f % 256 == 0 or w4f >= 512) and not (H * W <= 64 * 64 and cin <= 128):
        return '4f'
    return None


def mp_mode():
    """PCP_CONV_ALGO=bf16: the mixed-precision training loop (include/pcp_hip_mp.h) -- the 3x3 layers store their activations and
    gradients as bf16 and run forward / data-gradient / weight-gradient on the bf16 matrix cores; master weights, BatchNorm, losses
    and the optimizer stay fp32"""
    return os.environ.get('PCP_CONV_ALGO', 'auto') == 'bf16'


def as_f32(t, off=0, c=None):
    """contiguous float32 copy of a channel window (the fp32-only kernels' view of a bf16 activation); fp32 full-width input: itself"""
    c = t.shape[-1] - off if c is None else c
    if t.dtype == torch.float32 and off == 0 and c == t.shape[-1]:
        return t
    return t[..., off:off + c].float().contiguous()


def as_bf16(t, off=0, c=None):
    c = t.shape[-1] - off if c is None else c
    if t.dtype == torch.bfloat16 and off == 0 and c == t.shape[-1]:
        return t
    return t[..., off:off + c].to(torch.bfloat16).contiguous()


PACK_GROUP = tops.PackGroup()
MP_PACK_GROUP = tops.MpPackGroup()
PACK_GROUPING = os.environ.get('PCP_PACK_GROUP', '1') != '0'


class StepClock:
    """Packed weights are rebuilt when the optimizer has stepped (weights change through raw pointers, torch cannot tell)."""
    step = 0

    @classmethod
    def tick(cls):
        cls.step += 1
        flush_batches_tracked()


# nn.BatchNorm's num_batches_tracked += 1 is one tiny launch per BatchNorm per forward (47 per DiscoNet iteration): the increments are
# collected and applied by ONE multi-tensor add per optimizer step, and before any state_dict(): PackedModule and Detector3DTemplate register
# flush_batches_tracked() as a state_dict pre-hook, so a checkpoint taken between a forward and optimizer.step() carries current counters.
_PENDING_NBT = {}


def bump_batches_tracked(bn):
    t = bn.num_batches_tracked
    if t is None:
        return
    e = _PENDING_NBT.get(id(t))
    if e is None:
        _PENDING_NBT[id(t)] = [t, 1]
    else:
        e[1] += 1
    if len(_PENDING_NBT) > 4096:
        flush_batches_tracked()


def drop_pending_batches_tracked(module):
    """forget the collected increments of `module`'s BatchNorm counters (they were just replaced by load_state_dict: adding increments
    from before the load would corrupt the loaded values)"""
    if not _PENDING_NBT:
        return
    for name, buf in module.named_buffers():
        if name.endswith('num_batches_tracked'):
            _PENDING_NBT.pop(id(buf), None)


def flush_batches_tracked():
    if not _PENDING_NBT:
        return
    by_count = {}
    for t, n in _PENDING_NBT.values():
        by_count.setdefault((n, t.device), []).append(t)
    _PENDING_NBT.clear()
    for (n, _dev), ts in by_count.items():
        torch._foreach_add_(ts, n)


def ensure_grad(p):
    if p.grad is None or p.grad.shape != p.shape or p.grad.device != p.device or not p.grad.is_contiguous():
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


class Act:
    """a channel window of an NHWC (or row-major) buffer"""
    __slots__ = ('t', 'off', 'c')

    def __init__(self, t, off=0, c=None):
        self.t, self.off, self.c = t, off, (t.shape[-1] - off if c is None else c)

    @property
    def rows(self):
        return self.t.numel() // self.t.shape[-1]


def _zeros_like_cache(cache, key, n, device):
    z = cache.get(key)
    if z is None or z.numel() < n or z.device != device:
        z = torch.zeros(n, dtype=torch.float32, device=device)
        cache[key] = z
    return z


_ZERO_BIAS = {}


class ConvBNAct:
    """kind: '3x3' (stride 1 | 2, padding 1), 'plain' (Conv2d 1x1), 'plainT' (ConvTranspose2d 1x1), 's2d' (Conv2d k2 s2),
    'd2s' (ConvTranspose2d k2 s2)."""

    def __init__(self, conv, bn=None, relu=True, name=''):
        self.conv, self.bn, self.relu, self.name = conv, bn, relu, name
        k, s = conv.kernel_size[0], conv.stride[0]
        if isinstance(conv, nn.ConvTranspose2d):
            self.cin, self.cout = conv.weight.shape[0], conv.weight.shape[1]
            self.kind = {(1, 1): 'plainT', (2, 2): 'd2s'}.get((k, s))
        else:
            self.cout, self.cin = conv.weight.shape[0], conv.weight.shape[1]
            self.kind = {(3, 1): '3x3', (3, 2): '3x3', (1, 1): 'plain', (2, 2): 's2d'}.get((k, s))
        if self.kind is None:
            raise NotImplementedError('%s: conv k=%d s=%d has no training kernel' % (name, k, s))
        self.stride = s if self.kind == '3x3' else 1
        self._fw = self._bw = None
        self._fw_mp = self._bw_mp = None
        self.saved = None
        self.vec = None

    # ---- weight forms ----------------------------------------------------------------------------------------------------
    # cached ON THE CONV MODULE (several layer instances may share one module: the weightor runs once per agent) and rebuilt
    # once per optimizer step; 3x3 layouts are written by pcp_pack_conv3x3 into persistent buffers (one launch per direction)
    def _repack(self, bhw=None):
        """bhw: (B, H, W) of the layer input -- decides, once, whether the fused F(4x4) forms are packed (forward: the conv on that map;
        backward: the data-gradient conv on the same map, for stride 2 on the zero-dilated gradient)"""
        cache = getattr(self.conv, '_pcp_train_pack', None)
        if cache is None:
            cache = self.conv._pcp_train_pack = dict(step=-1)
        if bhw is not None and self.kind == '3x3' and cache.get('f4_key') != (tuple(bhw), os.environ.get('PCP_CONV_ALGO', 'auto')):
            cache['f4_key'] = (tuple(bhw), os.environ.get('PCP_CONV_ALGO', 'auto'))
            f4_new = (fused_f4_choice(*bhw, self.cin, self.cout) if self.stride == 1 else None, fused_f4_choice(*bhw, self.cout, self.cin))
            if f4_new != cache.get('f4'):
                # another fused-F(4x4) form is needed (another batch / map shape, e.g. the last partial batch): take the slow path once so
                # the new form is packed AND registered -- the group launch only repacks the forms it was given
                cache['f4'] = f4_new
                cache['step'] = -1
                if cache.get('grouped'):
                    PACK_GROUP.drop((id(self.conv), False))
                    PACK_GROUP.drop((id(self.conv), True))
                    cache['grouped'] = False
        if cache['step'] == StepClock.step:
            self._fw, self._bw = cache['fw'], cache['bw']
            self._fw_mp, self._bw_mp = cache.get('fw_mp'), cache.get('bw_mp')
            return
        self._fw_mp = self._bw_mp = None
        if self.kind == '3x3' and mp_mode() and self.cin % 16 == 0 and self.cout % 16 == 0:
            # bf16 weight forms (forward + data gradient) from the fp32 master weights, persistent buffers, two small launches per step
            w = self.conv.weight.detach()
            wc = w if w.is_contiguous() else w.contiguous()
            mpb = cache.setdefault('mp', {})
            grouped = (PACK_GROUPING and wc.data_ptr() == w.data_ptr() and 'fw' in mpb and MP_PACK_GROUP.has((id(self.conv), False), w.data_ptr())
                       and MP_PACK_GROUP.has((id(self.conv), True), w.data_ptr()))
            if grouped:
                # steps after the first: ONE launch repacks every registered layer, issued by whichever layer asks first in a step
                if MP_PACK_GROUP.step != StepClock.step:
                    MP_PACK_GROUP.run(w.device)
                    MP_PACK_GROUP.step = StepClock.step
                fwp, bwp = mpb['fw'], mpb['bw']
                fo, bo = pack.round_up(self.cout, 64), pack.round_up(self.cin, 64)
            else:
                fwp, fo = tops.mp_pack_conv3x3(wc, False, out=mpb.get('fw'))
                bwp, bo = tops.mp_pack_conv3x3(wc, True, out=mpb.get('bw'))
                mpb['fw'], mpb['bw'] = fwp, bwp
                if PACK_GROUPING and wc.data_ptr() == w.data_ptr():          # the parameter's own storage: later steps go through the group launch
                    MP_PACK_GROUP.add((id(self.conv), False), self.conv, wc, False, fwp, fo)
                    MP_PACK_GROUP.add((id(self.conv), True), self.conv, wc, True, bwp, bo)
            zeros = _zeros_like_cache(_ZERO_BIAS, w.device, 2048, w.device)
            b = self.conv.bias.detach() if self.conv.bias is not None else None
            fb = zeros if b is None else (b if fo == self.cout else pack.pad_bias(b, fo))
            cache['fw'], cache['bw'], cache['step'] = dict(mp=(fwp, fb, fo)), dict(mp=(bwp, zeros, bo)), StepClock.step
            self._fw, self._bw = cache['fw'], cache['bw']
            return
        if (self.kind == '3x3' and cache.get('grouped') and PACK_GROUP.has((id(self.conv), False))
                and cache.get('w_ptr') == self.conv.weight.data_ptr()):
            # steps after the first: ONE launch repacks every registered 3x3 layer, issued by whichever layer asks first in a step
            if PACK_GROUP.step != StepClock.step:
                PACK_GROUP.run(self.conv.weight.device)
                PACK_GROUP.step = StepClock.step
            cache['step'] = StepClock.step
            self._fw, self._bw = cache['fw'], cache['bw']
            return
        w = self.conv.weight.detach()
        dev = w.device
        zeros = _zeros_like_cache(_ZERO_BIAS, dev, 2048, dev)
        b = self.conv.bias.detach() if self.conv.bias is not None else None
        k = self.kind

        def bias_for(n_pad, n):
            if b is None:
                return zeros
            return b if n_pad == n else pack.pad_bias(b, n_pad)
        if k == '3x3':
            if 'buf' not in cache:
                rp = lambda o: pack.round_up(o, 64 if o > 32 else 32)
                bufs = dict(fw_d=torch.empty((self.cin // 16, 9, rp(self.cout), 16), dtype=torch.float32, device=dev),
                            bw_d=torch.empty((self.cout // 16, 9, rp(self.cin), 16), dtype=torch.float32, device=dev))
                if self.stride == 1 and self.cin % pack.WINO_CK == 0 and self.cout >= 48:
                    bufs['fw_w'] = torch.empty((self.cin // 8, 16, pack.round_up(self.cout, 64), 8), dtype=torch.float32, device=dev)
                if self.cout % pack.WINO_CK == 0 and self.cin >= 48:
                    bufs['bw_w'] = torch.empty((self.cout // 8, 16, pack.round_up(self.cin, 64), 8), dtype=torch.float32, device=dev)
                if os.environ.get('PCP_CONV_ALGO', 'auto') in ('bf16x3', 'bf16'):  # opt-in split / plain bf16 arithmetic (conv_bf16x3.hip)
                    if self.cin % 16 == 0 and self.cout >= 48:
                        bufs['fw_3'] = torch.empty((self.cin // 16) * pack.round_up(self.cout, 64) * 9 * 16 * 2, dtype=torch.int16, device=dev)
                    if self.cout % 16 == 0 and self.cin >= 48:
                        bufs['bw_3'] = torch.empty((self.cout // 16) * pack.round_up(self.cin, 64) * 9 * 16 * 2, dtype=torch.int16, device=dev)
                cache['buf'] = bufs
            bufs = cache['buf']
            wc = w.contiguous()
            fo, bo = bufs['fw_d'].shape[2], bufs['bw_d'].shape[2]
            fww, bww = bufs.get('fw_w'), bufs.get('bw_w')
            f3, b3 = bufs.get('fw_3'), bufs.get('bw_3')
            tops.pack_conv3x3(wc, False, bufs['fw_d'], fo, fww, fww.shape[2] if fww is not None else 0, f3, pack.round_up(self.cout, 64))
            tops.pack_conv3x3(wc, True, bufs['bw_d'], bo, bww, bww.shape[2] if bww is not None else 0, b3, pack.round_up(self.cin, 64))
            f4 = cache.get('f4', (None, None))
            f4fw = f4bw = None
            for tag, kind, transpose, o in (('fw_4', f4[0], False, self.cout), ('bw_4', f4[1], True, self.cin)):
                if kind is None:
                    continue
                i_ch = self.cin if not transpose else self.cout
                op4 = pack.round_up(o, 64)
                key = tag + kind
                if key not in bufs:
                    bufs[key] = torch.empty((i_ch // 8) * 36 * op4 * 8, dtype=torch.float32, device=dev)
                tops.pack_conv3x3_winograd4(wc, transpose, bufs[key] if kind == '4f' else None, bufs[key] if kind == '4h' else None, op4)
                form = (kind, bufs[key], bias_for(op4, self.cout) if not transpose else zeros, op4)
                if transpose:
                    f4bw = form
                else:
                    f4fw = form
            fw = dict(direct=(bufs['fw_d'], bias_for(fo, self.cout), fo))
            if fww is not None:
                fw['wino'] = (fww, bias_for(fww.shape[2], self.cout), fww.shape[2])
            if f3 is not None:
                fw['b3'] = (f3, bias_for(pack.round_up(self.cout, 64), self.cout), pack.round_up(self.cout, 64))
            cache['grouped'] = False
            if f3 is None and b3 is None and b is None and wc.data_ptr() == w.data_ptr() and PACK_GROUPING:      # (a padded bias is a per-step copy)
                # the buffers are persistent and the weight tensor is the parameter's own storage: later steps repack through the group launch
                k4 = lambda form, kind: form[1] if (form is not None and form[0] == kind) else None
                PACK_GROUP.add((id(self.conv), False), self.conv, wc, False, bufs['fw_d'], fo, fww, fww.shape[2] if fww is not None else 0,
                               k4(f4fw, '4f'), k4(f4fw, '4h'), f4fw[3] if f4fw is not None else 0)
                PACK_GROUP.add((id(self.conv), True), self.conv, wc, True, bufs['bw_d'], bo, bww, bww.shape[2] if bww is not None else 0,
                               k4(f4bw, '4f'), k4(f4bw, '4h'), f4bw[3] if f4bw is not None else 0)
                cache['grouped'] = True
                cache['w_ptr'] = wc.data_ptr()
            if f4fw is not None:
                fw['f4'] = f4fw
            bw = dict(direct=(bufs['bw_d'], zeros, bo))
            if f4bw is not None:
                bw['f4'] = f4bw
            if bww is not None:
                bw['wino'] = (bww, zeros, bww.shape[2])
            if b3 is not None:
                bw['b3'] = (b3, zeros, pack.round_up(self.cin, 64))
        else:
            zb = lambda n: zeros[:n]
            fb = lambda n: (b if b is not None else zeros[:n])
            if k == 'plain':
                m = w.reshape(self.cout, self.cin)
                fw = pack.pack_plain(m, fb(self.cout))
                bw = pack.pack_plain(m.t().contiguous(), zb(self.cin))
            elif k == 'plainT':
                m = w.reshape(self.cin, self.cout)
                fw = pack.pack_plain(m.t().contiguous(), fb(self.cout))
                bw = pack.pack_plain(m.contiguous(), zb(self.cin))
            elif k == 's2d':
                fw = pack.pack_conv2x2_s2(w, fb(self.cout))
                bw = pack.pack_convT2x2_s2(w, zb(self.cin))             # (cout, cin, 2, 2) read as a ConvTranspose2d weight
            else:  # d2s
                fw = pack.pack_convT2x2_s2(w, fb(self.cout))
                bw = pack.pack_conv2x2_s2(w, zb(self.cin))              # (cin, cout, 2, 2) read as a Conv2d weight
        cache['fw'], cache['bw'], cache['step'] = fw, bw, StepClock.step
        self._fw, self._bw = fw, bw
        cache['fw_mp'] = cache['bw_mp'] = None
        if k != '3x3' and mp_mode() and dev.type == 'cuda':
            # bf16 loop: the same packed forms as bf16 for pcp_mp_pointwise (used when the layer's input arrives as bf16)
            if tops.mp_pointwise_ok(None, self.cin, self.cout, fw[2], 8, 8):
                cache['fw_mp'] = (fw[0].to(torch.bfloat16), fw[1], fw[2])
            if tops.mp_pointwise_ok(None, self.cout, self.cin, bw[2], 8, 8):
                cache['bw_mp'] = (bw[0].to(torch.bfloat16), bw[1], bw[2])
        self._fw_mp, self._bw_mp = cache['fw_mp'], cache['bw_mp']

    # ---- launches --------------------------------------------------------------------------------------------------------
    @staticmethod
    def _run3x3(forms, x, cin, cout, stride, out, in_off, out_off):
        B, H, W, _ = x.shape
        algo = os.environ.get('PCP_CONV_ALGO', 'auto')                  # auto | direct | winograd | bf16x3 | bf16 (same switch as inference)
        if algo in ('bf16x3', 'bf16') and 'b3' in forms:
            Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
            th = 16 if stride == 1 else 8
            if B * ((Ho + th - 1) // th) * ((Wo + 15) // 16) * (forms['b3'][2] // 64) >= B3_MIN_WORKGROUPS:
                w3, b3, cp3 = forms['b3']
                return ops.conv3x3_bf16x3(x, w3, b3, cin, cout, cp3, stride=stride, relu=False, out=out, in_ch_off=in_off, out_ch_off=out_off,
                                          plain=(algo == 'bf16'))
        if (stride == 1 and 'f4' in forms and x.shape[-1] % 4 == 0 and in_off % 4 == 0 and out.shape[-1] % 4 == 0 and out_off % 4 == 0
                and x.numel() * 4 <= 0x7fffffff):           # the fused kernels' own limits (16-byte windows, 32-bit buffer offsets)
            kind, u, ub, ucp = forms['f4']
            run = ops.conv3x3_winograd4h if kind == '4h' else ops.conv3x3_winograd4f
            return run(x, u, ub, cin, cout, ucp, relu=False, out=out, in_ch_off=in_off, out_ch_off=out_off)
        big = B * ((H + 7) // 8) * ((W + 15) // 16) * (forms['wino'][2] // 64) >= 256 if 'wino' in forms else False
        if stride == 1 and 'wino' in forms and algo != 'direct' and (big or algo == 'winograd'):
            u, ub, ucp = forms['wino']
            return ops.conv3x3_winograd(x, u, ub, cin, cout, ucp, relu=False, out=out, in_ch_off=in_off, out_ch_off=out_off)
        w, b, cp = forms['direct']
        return ops.conv3x3(x, w, b, cin, cout, cp, stride=stride, relu=False, out=out, in_ch_off=in_off, out_ch_off=out_off)

    def mp_pointwise_capable(self):
        """a pointwise layer whose forward AND data-gradient shapes pcp_mp_pointwise takes (it then reads and writes bf16 maps)"""
        rp = lambda o: pack.round_up(o, 64 if o > 32 else 32)
        return (self.kind != '3x3' and mp_mode() and tops.mp_pointwise_ok(None, self.cin, self.cout, rp(self.cout), 8, 8)
                and tops.mp_pointwise_ok(None, self.cout, self.cin, rp(self.cin), 8, 8))

    def out_shape(self, x):
        B, H, W = x.t.shape[0], x.t.shape[1], x.t.shape[2]
        if self.kind == '3x3':
            return (B, H // self.stride, W // self.stride)
        if self.kind == 's2d':
            return (B, H // 2, W // 2)
        if self.kind == 'd2s':
            return (B, 2 * H, 2 * W)
        return (B, H, W)

    def forward(self, x, out=None, out_dtype=None):
        """x: Act.  out: Act to write the activation into (a channel window of a wider buffer) or None.  Returns Act.
        Mixed-precision mode (mp_mode()): a 3x3 layer takes fp32 or bf16 input, keeps its conv output as bf16 and writes its activation
        as bf16 unless `out` / `out_dtype` say float32 (a consumer that is an fp32-only kernel); the other kinds compute in fp32."""
        self._repack(tuple(x.t.shape[:3]))
        dev = x.t.device
        shp = self.out_shape(x)
        need_post = self.bn is not None or self.relu
        k = self.kind
        mp = k == '3x3' and 'mp' in self._fw
        # a pointwise layer joins the bf16 loop when its input ARRIVES as bf16 (the up / down-sampling layers behind the bf16 3x3 blocks);
        # fp32 inputs (the fusion module's concatenated maps) keep the fp32 kernels
        mpw = (k != '3x3' and self._fw_mp is not None and self._bw_mp is not None and x.t.dtype == torch.bfloat16 and x.off % 8 == 0
               and x.t.shape[-1] % 8 == 0 and (out is None or (out.off % 8 == 0 and out.t.shape[-1] % 8 == 0)))
        if mp:
            if x.t.dtype != torch.bfloat16 or x.off % 8 or x.t.shape[-1] % 8:
                x = Act(as_bf16(x.t, x.off, x.c).contiguous(), 0, x.c)   # one cast; the fast kernel and the weight gradient both read the bf16 copy
            act_dtype = out.t.dtype if out is not None else (out_dtype or torch.bfloat16)
            y_dtype = torch.bfloat16 if need_post else act_dtype
        elif mpw:
            act_dtype = out.t.dtype if out is not None else (out_dtype or torch.bfloat16)
            y_dtype = torch.bfloat16 if need_post else act_dtype
        else:
            if x.t.dtype != torch.float32:
                x = Act(as_f32(x.t, x.off, x.c), 0, x.c)
            act_dtype = y_dtype = torch.float32
            if out is not None and out.t.dtype != torch.float32:
                raise NotImplementedError('%s: fp32 layer kind %s cannot write a bf16 buffer' % (self.name, k))
        y_t = torch.empty(shp + (self.cout,), dtype=y_dtype, device=dev) if (need_post or out is None) else None
        y = Act(y_t, 0, self.cout) if y_t is not None else out
        if mp:
            wp, bp, cp = self._fw['mp']
            tops.mp_conv3x3(x.t, wp, bp, self.cin, self.cout, cp, stride=self.stride, relu=False, out=y.t, in_ch_off=x.off, out_ch_off=y.off)
        elif k == '3x3':
            self._run3x3(self._fw, x.t, self.cin, self.cout, self.stride, y.t, x.off, y.off)
        else:
            mode = {'plain': lib.PW_PLAIN, 'plainT': lib.PW_PLAIN, 's2d': lib.PW_SPACE2DEPTH, 'd2s': lib.PW_DEPTH2SPACE}[k]
            if mpw:
                w, b, cp = self._fw_mp
                tops.mp_pointwise(x.t, w, b, mode, self.cin, self.cout, cp, relu=False, out=y.t, in_ch_off=x.off, out_ch_off=y.off)
            else:
                w, b, cp = self._fw
                ops.pointwise(x.t, w, b, mode, self.cin, self.cout, cp, relu=False, out=y.t, in_ch_off=x.off, out_ch_off=y.off)
        self._mpw = mpw
        if not need_post:
            self.saved = (x, y)
            return y
        if out is None:
            out = Act(torch.empty(shp + (self.cout,), dtype=act_dtype, device=dev), 0, self.cout)
        if self.bn is not None:
            bn = self.bn
            self.vec = tops.bn_train_stats(y.t, self.cout, bn.weight.detach(), bn.bias.detach(), bn.eps, bn.momentum,
                                           bn.running_mean, bn.running_var, vec=self.vec, ch_off=y.off)
            bump_batches_tracked(bn)
        else:
            if self.vec is None:
                self.vec = tops.BNVectors(self.cout, dev)
                self.vec.scale.fill_(1.0)
                self.vec.shift.zero_()
        tops.scale_shift_act(y.t, self.cout, self.vec, self.relu, out.t, in_ch_off=y.off, out_ch_off=out.off)
        self.saved = (x, y)
        return out

    def backward(self, dout, need_dx=True, accumulate=False, dx_out=None, dx_dtype=None):
        """dout: Act (gradient of the layer output; overwritten in place by the gradient of the conv output when the storage types allow).
        Returns Act dx (or None).  accumulate: add to param.grad instead of overwriting (a module applied several times).
        Mixed-precision 3x3 layers take fp32 or bf16 gradients, keep the conv-output gradient as bf16 (the operand of both gradient
        GEMMs) and return dx as bf16 unless dx_out / dx_dtype say float32."""
        x, y = self.saved
        dev = dout.t.device
        k = self.kind
        mp = (k == '3x3' and 'mp' in self._fw) or (k != '3x3' and getattr(self, '_mpw', False))     # bf16 operands for both gradient GEMMs
        if not mp and dout.t.dtype != torch.float32:
            dout = Act(as_f32(dout.t, dout.off, dout.c), 0, dout.c)
        dy = dout
        if self.bn is not None:
            g_w, g_b = ensure_grad(self.bn.weight), ensure_grad(self.bn.bias)
            # in place on dout, whatever its storage type (callers rely on it: the CenterHead runs ONE data-gradient conv over the five
            # branch gradients it handed to five layers)
            tops.bn_act_backward(dout.t, y.t, self.cout, self.vec, self.relu, g_w, g_b, accumulate=accumulate, dout_ch_off=dout.off,
                                 x_ch_off=y.off)
            if mp and (dout.t.dtype != torch.bfloat16 or dout.off % 8 or dout.t.shape[-1] % 8):
                dy = Act(as_bf16(dout.t, dout.off, dout.c).contiguous(), 0, dout.c)
        elif self.relu:
            raise NotImplementedError('%s: ReLU without BatchNorm is handled by the fused fusion kernels' % self.name)
        elif mp and (dout.t.dtype != torch.bfloat16 or dout.off % 8 or dout.t.shape[-1] % 8):
            dy = Act(as_bf16(dout.t, dout.off, dout.c).contiguous(), 0, dout.c)
        if mp and (dy.off % 8 or dy.t.shape[-1] % 8):
            dy = Act(dy.t[..., dy.off:dy.off + dy.c].contiguous(), 0, dy.c)
        if self.conv.bias is not None:
            tops.colsum(dy.t, self.cout, ensure_grad(self.conv.bias), accumulate=accumulate, ch_off=dy.off)
        gw = ensure_grad(self.conv.weight)
        rows = x.rows
        if mp and k == '3x3':
            tops.mp_conv3x3_wgrad(x.t, dy.t, self.cin, self.cout, self.stride, gw, accumulate=accumulate, x_ch_off=x.off, dy_ch_off=dy.off)
        elif k == '3x3':
            tops.conv3x3_wgrad(x.t, dy.t, self.cin, self.cout, self.stride, gw, accumulate=accumulate, x_ch_off=x.off, dy_ch_off=dy.off)
        elif k == 'plain':
            tops.pointwise_wgrad(tops.rowmap(dy.t, self.cout, dy.off), tops.rowmap(x.t, self.cin, x.off), rows,
                                 gw.view(self.cout, self.cin), accumulate=accumulate)
        elif k == 'plainT':
            tops.pointwise_wgrad(tops.rowmap(x.t, self.cin, x.off), tops.rowmap(dy.t, self.cout, dy.off), rows,
                                 gw.view(self.cin, self.cout), accumulate=accumulate)
        elif k == 's2d':
            Ho, Wo = dy.t.shape[1], dy.t.shape[2]
            tmp = torch.empty((4, self.cout, self.cin), dtype=torch.float32, device=dev)
            for tap in range(4):
                tops.pointwise_wgrad(tops.rowmap(dy.t, self.cout, dy.off),
                                     tops.rowmap(x.t, self.cin, x.off, lattice=(Ho, Wo, tap // 2, tap % 2)), dy.rows, tmp[tap])
            g = tmp.permute(1, 2, 0).reshape(self.cout, self.cin, 2, 2)
            gw.add_(g) if accumulate else gw.copy_(g)
        else:  # d2s
            H, W = x.t.shape[1], x.t.shape[2]
            tmp = torch.empty((4, self.cin, self.cout), dtype=torch.float32, device=dev)
            for tap in range(4):
                tops.pointwise_wgrad(tops.rowmap(x.t, self.cin, x.off),
                                     tops.rowmap(dy.t, self.cout, dy.off, lattice=(H, W, tap // 2, tap % 2)), rows, tmp[tap])
            g = tmp.permute(1, 2, 0).reshape(self.cin, self.cout, 2, 2)
            gw.add_(g) if accumulate else gw.copy_(g)
        if not need_dx:
            return None
        self._repack(tuple(x.t.shape[:3]))
        if dx_out is None:
            dt = (dx_dtype or torch.bfloat16) if mp else torch.float32
            dx_out = Act(torch.empty(tuple(x.t.shape[:-1]) + (self.cin,), dtype=dt, device=dev), 0, self.cin)
        if mp and k != '3x3':
            mode = {'plain': lib.PW_PLAIN, 'plainT': lib.PW_PLAIN, 's2d': lib.PW_DEPTH2SPACE, 'd2s': lib.PW_SPACE2DEPTH}[k]
            w, b, cp = self._bw_mp
            tops.mp_pointwise(dy.t, w, b, mode, self.cout, self.cin, cp, relu=False, out=dx_out.t, in_ch_off=dy.off, out_ch_off=dx_out.off)
        elif mp:
            src = dy
            if self.stride == 2:
                src = Act(tops.dilate2x(dy.t, self.cout, ch_off=dy.off), 0, self.cout)
            wp, bp, cp = self._bw['mp']
            tops.mp_conv3x3(src.t, wp, bp, self.cout, self.cin, cp, stride=1, relu=False, out=dx_out.t, in_ch_off=src.off, out_ch_off=dx_out.off)
        elif k == '3x3':
            src = dy
            if self.stride == 2:
                src = Act(tops.dilate2x(dy.t, self.cout, ch_off=dy.off), 0, self.cout)
            self._run3x3(self._bw, src.t, self.cout, self.cin, 1, dx_out.t, src.off, dx_out.off)
        else:
            mode = {'plain': lib.PW_PLAIN, 'plainT': lib.PW_PLAIN, 's2d': lib.PW_DEPTH2SPACE, 'd2s': lib.PW_SPACE2DEPTH}[k]
            w, b, cp = self._bw
            ops.pointwise(dy.t, w, b, mode, self.cout, self.cin, cp, relu=False, out=dx_out.t, in_ch_off=dy.off, out_ch_off=dx_out.off)
        return dx_out


class LinearBNAct:
    """nn.Linear (no bias) + BatchNorm1d (batch statistics) + ReLU on rows (N, cin) -- the blocks of hunter_toolbox.py:130-158 (nn_make_mlp).
    cin is padded to a multiple of 16 with zero weight columns (the 3-d and 774-d inputs of the object head)."""

    def __init__(self, linear, bn, relu=True, name=''):
        self.lin, self.bn, self.relu, self.name = linear, bn, relu, name
        self.cout, self.cin = linear.weight.shape
        self.cin_pad = pack.round_up(self.cin, 16)
        assert linear.bias is None and bn is not None
        self._step = -1
        self.vec = None
        self.saved = None

    def _forms(self):
        if self._step == StepClock.step:
            return self._f
        w = self.lin.weight.detach().float()
        dev = w.device
        wp = w.new_zeros((self.cout, self.cin_pad))
        wp[:, :self.cin] = w
        self._f = dict(fw=pack.pack_plain(wp, w.new_zeros(self.cout)), bw=pack.pack_plain(wp.t().contiguous(), w.new_zeros(self.cin_pad)))
        self._step = StepClock.step
        return self._f

    def forward(self, x):
        """x: (rows, ld >= cin_pad) contiguous, columns [cin, cin_pad) zero.  Returns (rows, cout)."""
        f = self._forms()
        rows = x.shape[0]
        w, b, cp = f['fw']
        y = torch.empty((rows, self.cout), dtype=torch.float32, device=x.device)
        ops.pointwise(x, w, b, lib.PW_PLAIN, self.cin_pad, self.cout, cp, relu=False, out=y)
        bn = self.bn
        self.vec = tops.bn_train_stats(y, self.cout, bn.weight.detach(), bn.bias.detach(), bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                                       vec=self.vec)
        bump_batches_tracked(bn)
        out = torch.empty_like(y)
        tops.scale_shift_act(y, self.cout, self.vec, self.relu, out)
        self.saved = (x, y)
        return out

    def backward(self, dout, need_dx=True):
        """dout: (rows, cout), overwritten.  Returns (rows, cin_pad) or None."""
        x, y = self.saved
        tops.bn_act_backward(dout, y, self.cout, self.vec, self.relu, ensure_grad(self.bn.weight), ensure_grad(self.bn.bias))
        rows = x.shape[0]
        dw = torch.empty((self.cout, self.cin_pad), dtype=torch.float32, device=x.device)
        tops.pointwise_wgrad(tops.rowmap(dout, self.cout), tops.rowmap(x, self.cin_pad), rows, dw)
        ensure_grad(self.lin.weight).copy_(dw[:, :self.cin])
        if not need_dx:
            return None
        w, b, cp = self._forms()['bw']
        dx = torch.empty((rows, self.cin_pad), dtype=torch.float32, device=x.device)
        ops.pointwise(dout, w, b, lib.PW_PLAIN, self.cout, self.cin_pad, cp, relu=False, out=dx)
        return dx


class MultiLinear:
    """several nn.Linear (with bias) reading the same rows, as ONE pointwise GEMM over the concatenated outputs (the three point heads
    seg / reg_flow3d / instance_embedding of hunter_jr.py:93-96, the local_tf_decoder of :40).  The output / gradient buffers are 16 columns
    wide per 16 outputs so that the gradient is a legal contraction operand of the data-gradient GEMM."""

    def __init__(self, linears, name=''):
        self.lins, self.name = list(linears), name
        self.outs = [l.weight.shape[0] for l in self.lins]
        self.offs = [0]
        for o in self.outs:
            self.offs.append(self.offs[-1] + o)
        self.cin = self.lins[0].weight.shape[1]
        self.cout = self.offs[-1]
        self.ld = pack.round_up(self.cout, 16)
        self._step = -1
        self.saved = None

    def _forms(self):
        if self._step == StepClock.step:
            return self._f
        w = torch.cat([l.weight.detach().float() for l in self.lins], 0)
        b = torch.cat([l.bias.detach().float() for l in self.lins], 0)
        wt = w.new_zeros((self.cin, self.ld))
        wt[:, :self.cout] = w.t()
        self._f = dict(fw=pack.pack_plain(w, b), bw=pack.pack_plain(wt, w.new_zeros(self.cin)))
        self._step = StepClock.step
        return self._f

    def forward(self, x):
        """x: (rows, cin) contiguous.  Returns (rows, ld); columns past cout are zero."""
        w, b, cp = self._forms()['fw']
        out = torch.zeros((x.shape[0], self.ld), dtype=torch.float32, device=x.device)
        ops.pointwise(x, w, b, lib.PW_PLAIN, self.cin, self.cout, cp, relu=False, out=out)
        self.saved = x
        return out

    def backward(self, dout):
        """dout: (rows, ld) with zero padding columns.  Returns (rows, cin)."""
        x = self.saved
        rows = x.shape[0]
        dw = torch.empty((self.ld, self.cin), dtype=torch.float32, device=x.device)
        tops.pointwise_wgrad(tops.rowmap(dout, self.ld), tops.rowmap(x, self.cin), rows, dw)
        db = torch.empty((self.ld,), dtype=torch.float32, device=x.device)
        tops.colsum(dout, self.ld, db)
        for i, l in enumerate(self.lins):
            ensure_grad(l.weight).copy_(dw[self.offs[i]:self.offs[i + 1]])
            ensure_grad(l.bias).copy_(db[self.offs[i]:self.offs[i + 1]])
        w, b, cp = self._forms()['bw']
        dx = torch.empty((rows, self.cin), dtype=torch.float32, device=x.device)
        ops.pointwise(dout, w, b, lib.PW_PLAIN, self.ld, self.cin, cp, relu=False, out=dx)
        return dx
