"""Tensor-level wrappers over the training half of the C ABI (include/pcp_hip_train.h).  Same rules as ops.py: CUDA tensors
only, torch supplies memory and the stream, no eager fallback."""
import ctypes

import torch

from . import lib as _lib
from .lib import Conv3x3, RowMap, check
from .ops import _chan_ptr, _need_cuda, _p, _stream


class Scratch:
    """grow-only byte buffer reused by every call that needs a workspace (all calls are stream ordered)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        return self.buf


_BN_WS = Scratch()
_WG_WS = Scratch()


def _rows_c(x):
    """(..., ld) contiguous tensor -> (rows, ld)"""
    assert x.is_contiguous() and x.dtype in (torch.float32, torch.bfloat16)
    return x.numel() // x.shape[-1], x.shape[-1]


def _dt(t):
    """storage-type code of include/pcp_hip_mp.h"""
    if t.dtype == torch.float32:
        return _lib.DT_F32
    if t.dtype == torch.bfloat16:
        return _lib.DT_BF16
    raise _lib.PcpError('tensor of %s: the training kernels store float32 or bfloat16' % t.dtype)


def _all_f32(*ts):
    return all(t.dtype == torch.float32 for t in ts)


class BNVectors:
    __slots__ = ('scale', 'shift', 'mean', 'invstd')

    def __init__(self, c, device):
        buf = torch.empty((4, c), dtype=torch.float32, device=device)
        self.scale, self.shift, self.mean, self.invstd = buf[0], buf[1], buf[2], buf[3]


# Cross-rank BatchNorm (tools/train.py --sync_bn == nn.SyncBatchNorm.convert_sync_batchnorm of the reference, tools/train.py:128-129):
# when SYNC_BN is set and a process group with more than one rank exists, every training-mode BatchNorm all-reduces its per-channel
# float64 sums (+ the row count) in the forward and its two gradient sums in the backward.
SYNC_BN = False


def _sync_group():
    import torch.distributed as dist
    if SYNC_BN and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


def _all_reduce_sums(dist, sums_and_count):
    """sum over the ranks of a float64 device vector; gloo (the CPU-test / shared-device backend) reduces a host copy"""
    if dist.get_backend() == 'nccl':
        dist.all_reduce(sums_and_count)
        return sums_and_count
    host = sums_and_count.cpu()
    dist.all_reduce(host)
    return host.to(sums_and_count.device)


def bn_train_stats(x, c, gamma, beta, eps, momentum, running_mean, running_var, vec=None, ch_off=0):
    """x: (..., ld) NHWC / row-major.  Returns BNVectors; running stats are updated in place (pass None to skip)."""
    _need_cuda(x, gamma, beta)
    L = _lib.load()
    rows, ld = _rows_c(x)
    if vec is None:
        vec = BNVectors(c, x.device)
    ws = _BN_WS.get(L.pcp_bn_workspace_bytes(c), x.device)
    dist = _sync_group()
    if dist is not None:
        buf = torch.empty((2 * c + 1,), dtype=torch.float64, device=x.device)
        buf[2 * c:].fill_(float(rows))
        check(L.pcp_mp_bn_train_sums(_chan_ptr(x, ch_off), _dt(x), rows, c, ld, _p(ws), _p(buf), _stream()), 'pcp_mp_bn_train_sums')
        buf = _all_reduce_sums(dist, buf)
        total = int(round(float(buf[2 * c].item())))
        check(L.pcp_bn_train_stats_from_sums(_p(buf), total, c, _p(gamma), _p(beta), float(eps), float(momentum), _p(running_mean),
                                             _p(running_var), _p(vec.scale), _p(vec.shift), _p(vec.mean), _p(vec.invstd), _stream()),
              'pcp_bn_train_stats_from_sums')
        return vec
    if not _all_f32(x):
        check(L.pcp_mp_bn_train_stats(_chan_ptr(x, ch_off), _dt(x), rows, c, ld, _p(gamma), _p(beta), float(eps), float(momentum),
                                      _p(running_mean), _p(running_var), _p(ws), _p(vec.scale), _p(vec.shift), _p(vec.mean), _p(vec.invstd),
                                      _stream()), 'pcp_mp_bn_train_stats')
        return vec
    check(L.pcp_bn_train_stats(_chan_ptr(x, ch_off), rows, c, ld, _p(gamma), _p(beta), float(eps), float(momentum), _p(running_mean),
                               _p(running_var), _p(ws), _p(vec.scale), _p(vec.shift), _p(vec.mean), _p(vec.invstd), _stream()),
          'pcp_bn_train_stats')
    return vec


def scale_shift_act(x, c, vec, relu, out, in_ch_off=0, out_ch_off=0):
    _need_cuda(x, out)
    L = _lib.load()
    rows, ld = _rows_c(x)
    rows_o, ld_o = _rows_c(out)
    assert rows == rows_o
    if not _all_f32(x, out):
        check(L.pcp_mp_scale_shift_act(_chan_ptr(x, in_ch_off), _dt(x), rows, c, ld, _p(vec.scale), _p(vec.shift), 1 if relu else 0,
                                       _chan_ptr(out, out_ch_off), _dt(out), ld_o, _stream()), 'pcp_mp_scale_shift_act')
        return out
    check(L.pcp_scale_shift_act(_chan_ptr(x, in_ch_off), rows, c, ld, _p(vec.scale), _p(vec.shift), 1 if relu else 0,
                                _chan_ptr(out, out_ch_off), ld_o, _stream()), 'pcp_scale_shift_act')
    return out


def bn_act_backward(dout, x, c, vec, relu, dgamma, dbeta, accumulate=False, dx=None, dout_ch_off=0, x_ch_off=0, dx_ch_off=0):
    """dx defaults to dout (in place).  dgamma / dbeta: (c,) float32."""
    _need_cuda(dout, x, dgamma, dbeta)
    L = _lib.load()
    rows, ld_d = _rows_c(dout)
    rows_x, ld_x = _rows_c(x)
    assert rows == rows_x
    if dx is None:
        dx, dx_ch_off = dout, dout_ch_off
    ld_dx = dx.shape[-1]
    ws = _BN_WS.get(L.pcp_bn_workspace_bytes(c), x.device)
    dist = _sync_group()
    if dist is not None:
        local = torch.empty((2 * c + 1,), dtype=torch.float64, device=x.device)
        local[2 * c:].fill_(float(rows))
        check(L.pcp_mp_bn_bwd_sums(_chan_ptr(dout, dout_ch_off), _dt(dout), ld_d, _chan_ptr(x, x_ch_off), _dt(x), ld_x, rows, c, _p(vec.scale),
                                   _p(vec.shift), _p(vec.mean), _p(vec.invstd), 1 if relu else 0, _p(ws), _p(local), _stream()), 'pcp_mp_bn_bwd_sums')
        glob = _all_reduce_sums(dist, local.clone())
        total = int(round(float(glob[2 * c].item())))
        check(L.pcp_mp_bn_bwd_apply_from_sums(_chan_ptr(dout, dout_ch_off), _dt(dout), ld_d, _chan_ptr(x, x_ch_off), _dt(x), ld_x, rows, c,
                                              _p(vec.scale), _p(vec.shift), _p(vec.mean), _p(vec.invstd), 1 if relu else 0, _p(local), _p(glob),
                                              total, _p(ws), _p(dgamma), _p(dbeta), 1 if accumulate else 0, _chan_ptr(dx, dx_ch_off), _dt(dx),
                                              ld_dx, _stream()), 'pcp_mp_bn_bwd_apply_from_sums')
        return dx
    if not _all_f32(dout, x, dx):
        check(L.pcp_mp_bn_act_backward(_chan_ptr(dout, dout_ch_off), _dt(dout), ld_d, _chan_ptr(x, x_ch_off), _dt(x), ld_x, rows, c,
                                       _p(vec.scale), _p(vec.shift), _p(vec.mean), _p(vec.invstd), 1 if relu else 0, _p(ws), _p(dgamma),
                                       _p(dbeta), 1 if accumulate else 0, _chan_ptr(dx, dx_ch_off), _dt(dx), ld_dx, _stream()),
              'pcp_mp_bn_act_backward')
        return dx
    check(L.pcp_bn_act_backward(_chan_ptr(dout, dout_ch_off), ld_d, _chan_ptr(x, x_ch_off), ld_x, rows, c, _p(vec.scale), _p(vec.shift),
                                _p(vec.mean), _p(vec.invstd), 1 if relu else 0, _p(ws), _p(dgamma), _p(dbeta), 1 if accumulate else 0,
                                _chan_ptr(dx, dx_ch_off), ld_dx, _stream()), 'pcp_bn_act_backward')
    return dx


def colsum(x, c, out, accumulate=False, ch_off=0):
    _need_cuda(x, out)
    L = _lib.load()
    rows, ld = _rows_c(x)
    ws = _BN_WS.get(L.pcp_bn_workspace_bytes(c), x.device)
    if not _all_f32(x):
        check(L.pcp_mp_colsum(_chan_ptr(x, ch_off), _dt(x), rows, c, ld, _p(ws), _p(out), 1 if accumulate else 0, _stream()), 'pcp_mp_colsum')
        return out
    check(L.pcp_colsum(_chan_ptr(x, ch_off), rows, c, ld, _p(ws), _p(out), 1 if accumulate else 0, _stream()), 'pcp_colsum')
    return out


def accumulate(dst, src, c, alpha=1.0, dst_ch_off=0, src_ch_off=0):
    _need_cuda(dst, src)
    L = _lib.load()
    rows, ld_d = _rows_c(dst)
    rows_s, ld_s = _rows_c(src)
    assert rows == rows_s
    if not _all_f32(dst, src):
        check(L.pcp_mp_accumulate(_chan_ptr(dst, dst_ch_off), _dt(dst), ld_d, _chan_ptr(src, src_ch_off), _dt(src), ld_s, rows, c, float(alpha),
                                  _stream()), 'pcp_mp_accumulate')
        return dst
    check(L.pcp_accumulate(_chan_ptr(dst, dst_ch_off), ld_d, _chan_ptr(src, src_ch_off), ld_s, rows, c, float(alpha), _stream()),
          'pcp_accumulate')
    return dst


def dilate2x(x, c, out=None, ch_off=0):
    _need_cuda(x, out)
    L = _lib.load()
    B, H, W, ld = x.shape
    if out is None:
        out = torch.empty((B, 2 * H, 2 * W, c), dtype=x.dtype, device=x.device)
    if not _all_f32(x, out):
        assert out.dtype == x.dtype
        check(L.pcp_mp_dilate2x(_chan_ptr(x, ch_off), _dt(x), B, H, W, c, ld, _p(out), out.shape[-1], _stream()), 'pcp_mp_dilate2x')
        return out
    check(L.pcp_dilate2x(_chan_ptr(x, ch_off), B, H, W, c, ld, _p(out), out.shape[-1], _stream()), 'pcp_dilate2x')
    return out


def conv3x3_wgrad(x, dy, cin, cout, stride, dw, accumulate=False, x_ch_off=0, dy_ch_off=0):
    """x: (B, H, W, ld_x) input of the conv, dy: (B, Ho, Wo, ld_dy) gradient of its output; dw: (cout, cin, 3, 3) contiguous."""
    _need_cuda(x, dy, dw)
    L = _lib.load()
    B, H, W, ld_x = x.shape
    assert dw.shape == (cout, cin, 3, 3) and dw.is_contiguous() and x.is_contiguous() and dy.is_contiguous()
    assert dy.shape[0] == B and dy.shape[1] == H // stride and dy.shape[2] == W // stride
    d = Conv3x3(B, H, W, cin, cout, 0, stride, ld_x, dy.shape[3], 0)
    need = L.pcp_conv3x3_wgrad_workspace_bytes(ctypes.byref(d))
    ws = _WG_WS.get(need, x.device)
    check(L.pcp_conv3x3_wgrad(ctypes.byref(d), _chan_ptr(x, x_ch_off), _chan_ptr(dy, dy_ch_off), _p(ws), ws.numel(), _p(dw),
                              1 if accumulate else 0, _stream()), 'pcp_conv3x3_wgrad')
    return dw


# ---------------------------------------------------------------------------------------------------------------------
# mixed-precision (bf16) convolution kernels of the training loop (include/pcp_hip_mp.h)
# ---------------------------------------------------------------------------------------------------------------------

def mp_pack_conv3x3(w, transpose=False, out=None, fold_scale=None):
    """(cout, cin, 3, 3) float32 -> the bf16 weight form of pcp_mp_conv3x3 (a bfloat16 tensor); transpose: the data-gradient form.
    Returns (packed, out_pad)."""
    _need_cuda(w, out, fold_scale)
    L = _lib.load()
    cout, cin = int(w.shape[0]), int(w.shape[1])
    K, O = (cout, cin) if transpose else (cin, cout)
    opad = (O + 63) // 64 * 64
    n = L.pcp_mp_conv3x3_packed_bytes(K, opad) // 2
    if n == 0:
        raise _lib.PcpError('pcp_mp_pack_conv3x3: %d contraction channels (need a multiple of 16)' % K)
    if out is None:
        out = torch.empty(n, dtype=torch.bfloat16, device=w.device)
    assert out.numel() == n and out.dtype == torch.bfloat16 and w.is_contiguous() and w.dtype == torch.float32
    check(L.pcp_mp_pack_conv3x3(_p(w), cout, cin, 1 if transpose else 0, _p(fold_scale), _p(out), opad, _stream()), 'pcp_mp_pack_conv3x3')
    return out, opad


class MpPackGroup:
    """the bf16 weight forms of every 3x3 layer (forward + data gradient) rebuilt by ONE launch per optimizer step (pcp_mp_pack_conv3x3_group:
    55 four-microsecond launches per DiscoNet iteration otherwise).  Jobs hold weak references to their conv modules."""

    def __init__(self):
        self.jobs = {}
        self.dirty = True
        self.table = None
        self.total = 0
        self.step = -1

    def add(self, key, owner, w, transpose, packed, out_pad):
        import weakref
        j = _lib.MpPackJob(w.data_ptr(), packed.data_ptr(), int(w.shape[0]), int(w.shape[1]), 1 if transpose else 0, int(out_pad), 0, 0)
        if _lib.load().pcp_mp_pack_conv3x3_group_blocks(ctypes.byref(j)) <= 0:
            raise _lib.PcpError('pcp_mp_pack_conv3x3_group: invalid job')
        self.jobs[key] = (weakref.ref(owner), j, (w, packed))
        self.dirty = True

    def has(self, key, w_ptr):
        e = self.jobs.get(key)
        return e is not None and e[0]() is not None and e[1].w == w_ptr

    def run(self, device):
        L = _lib.load()
        dead = [k for k, e in self.jobs.items() if e[0]() is None]
        for k in dead:
            del self.jobs[k]
            self.dirty = True
        if not self.jobs:
            return
        if self.dirty or self.table is None or self.table.device != device:
            arr = (_lib.MpPackJob * len(self.jobs))()
            start = 0
            for i, (_o, j, _t) in enumerate(self.jobs.values()):
                j.block_start = start
                start += L.pcp_mp_pack_conv3x3_group_blocks(ctypes.byref(j))
                arr[i] = j
            self.total = start
            self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
            self.dirty = False
        check(L.pcp_mp_pack_conv3x3_group(_p(self.table), len(self.jobs), self.total, _stream()), 'pcp_mp_pack_conv3x3_group')


def mp_conv3x3(x, packed, bias, cin, cout, cout_pad, stride=1, relu=False, out=None, out_dtype=torch.bfloat16, in_ch_off=0, out_ch_off=0):
    """x: (B, H, W, ld) float32 | bfloat16 NHWC; out: same for the output (allocated (B, Ho, Wo, cout) of out_dtype when None)."""
    _need_cuda(x, packed, bias, out)
    L = _lib.load()
    B, H, W, ld_in = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None:
        out = torch.empty((B, Ho, Wo, cout), dtype=out_dtype, device=x.device)
    assert tuple(out.shape[:3]) == (B, Ho, Wo) and x.is_contiguous() and out.is_contiguous()
    assert in_ch_off + cin <= ld_in and out_ch_off + cout <= out.shape[3] and bias.numel() >= cout_pad
    d = _lib.MpConv3x3(B, H, W, cin, cout, cout_pad, stride, ld_in, out.shape[3], 1 if relu else 0, _dt(x), _dt(out))
    check(L.pcp_mp_conv3x3(ctypes.byref(d), _chan_ptr(x, in_ch_off), _p(packed), _p(bias), _chan_ptr(out, out_ch_off), _stream()), 'pcp_mp_conv3x3')
    return out


def mp_conv3x3_wgrad(x, dy, cin, cout, stride, dw, accumulate=False, x_ch_off=0, dy_ch_off=0):
    """x: (B, H, W, ld) bfloat16, dy: (B, H/stride, W/stride, ld) bfloat16, dw: (cout, cin, 3, 3) float32"""
    _need_cuda(x, dy, dw)
    L = _lib.load()
    B, H, W, ld_x = x.shape
    assert x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and dw.dtype == torch.float32
    assert dw.shape == (cout, cin, 3, 3) and dw.is_contiguous() and x.is_contiguous() and dy.is_contiguous()
    assert dy.shape[0] == B and dy.shape[1] == H // stride and dy.shape[2] == W // stride
    d = _lib.MpWgrad3x3(B, H, W, cin, cout, stride, ld_x, dy.shape[3], _lib.DT_BF16, _lib.DT_BF16, 1 if accumulate else 0)
    need = L.pcp_mp_conv3x3_wgrad_workspace_bytes(ctypes.byref(d))
    if need == 0:
        # a shape the bf16 pixel-contraction kernel has no plan for: the fp32 weight-gradient kernel on fp32 copies of the
        # window -- slower, same contract (it raises itself where it has no kernel either)
        xf = x[..., x_ch_off:x_ch_off + cin].float().contiguous()
        dyf = dy[..., dy_ch_off:dy_ch_off + cout].float().contiguous()
        return conv3x3_wgrad(xf, dyf, cin, cout, stride, dw, accumulate=accumulate)
    ws = _WG_WS.get(need, x.device)
    check(L.pcp_mp_conv3x3_wgrad(ctypes.byref(d), _chan_ptr(x, x_ch_off), _chan_ptr(dy, dy_ch_off), _p(dw), _p(ws), ws.numel(), _stream()),
          'pcp_mp_conv3x3_wgrad')
    return dw


def rowmap(t, channels, ch_off=0, lattice=None):
    """lattice: None (row r = pixel r) or (grid_h, grid_w, ky, kx).  A bfloat16 tensor gives the row map of the bf16 kernel
    (pcp_mp_rowmap_t, include/pcp_hip_mp.h)."""
    gh, gw, ky, kx = lattice if lattice is not None else (0, 0, 0, 0)
    if t.dtype == torch.bfloat16:
        return _lib.MpRowMap(t.data_ptr() + 2 * ch_off, t.shape[-1], channels, 0 if lattice is None else 1, gh, gw, ky, kx, _lib.DT_BF16,
                             2 * (t.numel() - ch_off))
    assert t.dtype == torch.float32
    return RowMap(t.data_ptr() + 4 * ch_off, t.shape[-1], channels, 0 if lattice is None else 1, gh, gw, ky, kx)


def pointwise_wgrad(a, b, rows, out, accumulate=False):
    """out[n, k] (+)= sum_r a[map(r), n] * b[map(r), k];  a, b: row maps of ONE storage type (see rowmap());  out: (n, k) row-major
    (ld = out.stride(0)).  bf16 operands run pcp_mp_pointwise_wgrad (bf16 products), fp32 ones pcp_pointwise_wgrad."""
    L = _lib.load()
    assert out.dtype == torch.float32 and out.stride(-1) == 1 and out.shape == (a.channels, b.channels)
    mp = isinstance(a, _lib.MpRowMap)
    if mp != isinstance(b, _lib.MpRowMap):
        raise _lib.PcpError('pointwise_wgrad: the two operands must share one storage type')
    if mp:
        need = L.pcp_mp_pointwise_wgrad_workspace_bytes(rows, a.channels, b.channels)
        ws = _WG_WS.get(need, out.device)
        check(L.pcp_mp_pointwise_wgrad(ctypes.byref(a), ctypes.byref(b), rows, _p(ws), ws.numel(), _p(out), out.stride(0),
                                       1 if accumulate else 0, _stream()), 'pcp_mp_pointwise_wgrad')
        return out
    need = L.pcp_pointwise_wgrad_workspace_bytes(rows, a.channels, b.channels)
    ws = _WG_WS.get(need, out.device)
    check(L.pcp_pointwise_wgrad(ctypes.byref(a), ctypes.byref(b), rows, _p(ws), ws.numel(), _p(out), out.stride(0),
                                1 if accumulate else 0, _stream()), 'pcp_pointwise_wgrad')
    return out


def mp_pointwise_ok(mode, cin, cout, cout_pad, ld_in, ld_out):
    """the shapes pcp_mp_pointwise takes (anything else stays on the fp32 entry point)"""
    return cin % 32 == 0 and cout % 8 == 0 and cout_pad % 64 == 0 and ld_in % 8 == 0 and ld_out % 8 == 0


def mp_pointwise(x, w_bf16, bias, mode, cin, cout, cout_pad, relu=True, out=None, in_ch_off=0, out_ch_off=0, out_dtype=torch.bfloat16):
    """pointwise family with bf16 products (include/pcp_hip_mp.h: pcp_mp_pointwise).  x: bf16 or fp32 NHWC map / rows; w_bf16: the fp32
    pack of pack.pack_plain / pack_conv2x2_s2 / pack_convT2x2_s2 cast to bfloat16; out: bf16 or fp32 (allocated as out_dtype when None)."""
    _need_cuda(x, w_bf16, bias, out)
    L = _lib.load()
    assert w_bf16.dtype == torch.bfloat16 and bias.dtype == torch.float32 and x.is_contiguous()
    ld_in = x.shape[-1]
    if mode == _lib.PW_PLAIN:
        rows = x.numel() // ld_in
        B = H = W = 0
        oshape = tuple(x.shape[:-1])
    else:
        B, H, W, _ = x.shape
        rows = 0
        oshape = (B, H // 2, W // 2) if mode == _lib.PW_SPACE2DEPTH else (B, 2 * H, 2 * W)
    if out is None:
        out = torch.empty(oshape + (cout,), dtype=out_dtype, device=x.device)
    assert out.is_contiguous() and tuple(out.shape[:-1]) == oshape
    d = _lib.MpPointwise(mode, rows, B, H, W, cin, cout, cout_pad, ld_in, out.shape[-1], 1 if relu else 0, _dt(x), _dt(out))
    check(L.pcp_mp_pointwise(ctypes.byref(d), _chan_ptr(x, in_ch_off), _p(w_bf16), _p(bias), _chan_ptr(out, out_ch_off), _stream()),
          'pcp_mp_pointwise')
    return out


# ---------------------------------------------------------------------------------------------------------------------
# a15: targets + losses
# ---------------------------------------------------------------------------------------------------------------------

_LOSS_WS = {}


def _loss_ws(device):
    ws = _LOSS_WS.get(device)
    if ws is None:
        ws = torch.zeros(_lib.load().pcp_loss_workspace_bytes(), dtype=torch.uint8, device=device)
        _LOSS_WS[device] = ws
    return ws


def centerhead_targets(gt_boxes, desc):
    """gt_boxes: (B, M, 8) float32 CUDA.  desc: lib.Target.  Returns heatmap (B,H,W,ncls), target_boxes (B,K,8), inds, mask (B,K) i32."""
    _need_cuda(gt_boxes)
    L = _lib.load()
    assert gt_boxes.dtype == torch.float32 and gt_boxes.dim() == 3 and gt_boxes.shape[2] == 8 and gt_boxes.is_contiguous()
    B, M, _ = gt_boxes.shape
    assert B == desc.batch
    dev = gt_boxes.device
    heat = torch.empty((B, desc.h, desc.w, desc.num_class), dtype=torch.float32, device=dev)
    tb = torch.empty((B, desc.k, 8), dtype=torch.float32, device=dev)
    inds = torch.empty((B, desc.k), dtype=torch.int32, device=dev)
    mask = torch.empty((B, desc.k), dtype=torch.int32, device=dev)
    check(L.pcp_centerhead_targets(ctypes.byref(desc), _p(gt_boxes), M, _p(heat), _p(tb), _p(inds), _p(mask), _stream()),
          'pcp_centerhead_targets')
    return heat, tb, inds, mask


def centerhead_loss(head, desc, heat, tb, inds, mask, dhead=None, grad_scale=1.0):
    """head: (B,H,W,ld) raw maps.  Returns losses (4,) float32 device [hm, loc, hm+loc, num_pos]; fills dhead (B,H,W,ld_d) if given."""
    _need_cuda(head, heat, tb, inds, mask, dhead)
    L = _lib.load()
    losses = torch.empty(4, dtype=torch.float32, device=head.device)
    check(L.pcp_centerhead_loss(ctypes.byref(desc), _p(head), _p(heat), _p(tb), _p(inds), _p(mask), float(grad_scale),
                                _p(_loss_ws(head.device)), _p(losses), _p(dhead), _stream()), 'pcp_centerhead_loss')
    return losses


_ANCHOR_WS = {}


def _anchor_ws(device, nbytes):
    ws = _ANCHOR_WS.get(device)
    if ws is None or ws.numel() < nbytes:
        ws = torch.zeros(max(nbytes, 4096), dtype=torch.uint8, device=device)
        _ANCHOR_WS[device] = ws
    return ws


def anchor_assign_targets(anchors, gt_boxes, desc):
    """anchors (N, 7) flat in torch.cat(anchors, dim=-3) order, gt_boxes (B, M, 8) float32 device.
    Returns labels (B, N) int32, reg_targets (B, N, 7), reg_weights (B, N)  (axis_aligned_target_assigner.py:37-132)."""
    _need_cuda(anchors, gt_boxes)
    L = _lib.load()
    B, M = gt_boxes.shape[0], gt_boxes.shape[1]
    N = anchors.shape[0]
    assert anchors.is_contiguous() and gt_boxes.is_contiguous() and gt_boxes.shape[-1] == 8 and desc.batch == B
    assert N == desc.h * desc.w * desc.anchors_per_loc
    dev = anchors.device
    labels = torch.empty((B, N), dtype=torch.int32, device=dev)
    reg_t = torch.empty((B, N, 7), dtype=torch.float32, device=dev)
    reg_w = torch.empty((B, N), dtype=torch.float32, device=dev)
    need = L.pcp_anchor_assign_workspace_bytes(ctypes.byref(desc), M)
    ws = _anchor_ws(dev, need)
    check(L.pcp_anchor_assign_targets(ctypes.byref(desc), _p(anchors), _p(gt_boxes), M, _p(ws), ws.numel(), _p(labels), _p(reg_t), _p(reg_w),
                                      _stream()), 'pcp_anchor_assign_targets')
    return labels, reg_t, reg_w


def anchor_loss(head, anchors, labels, reg_targets, desc, dhead=None, grad_scale=1.0):
    """head (B, H, W, ld).  Returns losses (5,) float32 device [cls, loc, dir, total, positives]  (anchor_head_template.py:99-216)."""
    _need_cuda(head, anchors, labels, reg_targets, dhead)
    L = _lib.load()
    losses = torch.empty(5, dtype=torch.float32, device=head.device)
    need = L.pcp_anchor_loss_workspace_bytes(desc.batch)
    ws = _anchor_ws(head.device, need)
    assert labels.dtype == torch.int32 and labels.is_contiguous() and reg_targets.is_contiguous()
    check(L.pcp_anchor_loss(ctypes.byref(desc), _p(head), _p(anchors), _p(labels), _p(reg_targets), float(grad_scale), _p(ws), ws.numel(),
                            _p(losses), _p(dhead), _stream()), 'pcp_anchor_loss')
    return losses


def distill_loss(fused, early, c, weight=10.0, dfused=None, accumulate=False, grad_scale=1.0):
    """fused, early: (B,H,W,ld) NHWC.  Returns loss (1,) float32 device."""
    _need_cuda(fused, early, dfused)
    L = _lib.load()
    pixels = fused.numel() // fused.shape[-1]
    loss = torch.empty(1, dtype=torch.float32, device=fused.device)
    check(L.pcp_distill_loss(_p(fused), fused.shape[-1], _p(early), early.shape[-1], pixels, c, float(weight), float(grad_scale),
                             _p(_loss_ws(fused.device)), _p(loss), _p(dfused), dfused.shape[-1] if dfused is not None else 0,
                             1 if accumulate else 0, _stream()), 'pcp_distill_loss')
    return loss


def masked_smooth_l1_rows(fused, teacher, c, thresh=1e-3):
    """HunterJr's teacher-BEV term (hunter_jr.py:352-365), value only.  fused, teacher: (B,H,W,ld) NHWC.  Returns loss (1,) device."""
    _need_cuda(fused, teacher)
    L = _lib.load()
    pixels = fused.numel() // fused.shape[-1]
    assert fused.is_contiguous() and teacher.is_contiguous() and teacher.numel() // teacher.shape[-1] == pixels
    loss = torch.empty(1, dtype=torch.float32, device=fused.device)
    check(L.pcp_masked_smooth_l1_rows(_p(fused), fused.shape[-1], _p(teacher), teacher.shape[-1], pixels, c, float(thresh),
                                      _p(_loss_ws(fused.device)), _p(loss), _stream()), 'pcp_masked_smooth_l1_rows')
    return loss


# ---------------------------------------------------------------------------------------------------------------------
# PFN (train mode)
# ---------------------------------------------------------------------------------------------------------------------

def pfn_train_features(points, vox, num_raw, fbuf, slot_pillar):
    L = _lib.load()
    check(L.pcp_pfn_train_features(_p(points), vox.n, vox.row_stride, num_raw, ctypes.byref(vox.grid), _p(vox.workspace), _p(fbuf),
                                   _p(slot_pillar), _stream()), 'pcp_pfn_train_features')


def pfn_train_mid(vox, x0, vec0, in1, arg0):
    """in1: (N, 64) float32, or bfloat16 in the bf16 loop"""
    L = _lib.load()
    if in1.dtype == torch.bfloat16:
        check(L.pcp_mp_pfn_train_mid(ctypes.byref(vox.grid), _p(vox.workspace), vox.n, _p(x0), _p(vec0.scale), _p(vec0.shift), _p(in1), _lib.DT_BF16,
                                     _p(arg0), _stream()), 'pcp_mp_pfn_train_mid')
        return
    check(L.pcp_pfn_train_mid(ctypes.byref(vox.grid), _p(vox.workspace), vox.n, _p(x0), _p(vec0.scale), _p(vec0.shift), _p(in1), _p(arg0),
                              _stream()), 'pcp_pfn_train_mid')


def pfn_train_out(vox, x1, vec1, pillar_features, arg1, canvas):
    """canvas: (B, ny, nx, 64) float32, or bfloat16 in the bf16 loop (pre-zeroed either way); x1: (N, 64) float32 or bfloat16"""
    L = _lib.load()
    if (canvas is not None and canvas.dtype == torch.bfloat16) or x1.dtype == torch.bfloat16:
        check(L.pcp_mp_pfn_train_out(ctypes.byref(vox.grid), _p(vox.workspace), vox.n, _p(x1), _dt(x1), _p(vec1.scale), _p(vec1.shift),
                                     _p(pillar_features), _p(arg1), _p(canvas), _dt(canvas) if canvas is not None else _lib.DT_F32, _stream()),
              'pcp_mp_pfn_train_out')
        return
    check(L.pcp_pfn_train_out(ctypes.byref(vox.grid), _p(vox.workspace), vox.n, _p(x1), _p(vec1.scale), _p(vec1.shift),
                              _p(pillar_features), _p(arg1), _p(canvas), _stream()), 'pcp_pfn_train_out')


def pfn_train_route_out_grad(vox, kept_rows, arg1, dz1, dcanvas=None, dpillar=None):
    """dz1: (N, 64) float32 or bfloat16 (zero-filled here); dcanvas: float32 or bfloat16"""
    L = _lib.load()
    if (dcanvas is not None and dcanvas.dtype == torch.bfloat16) or dz1.dtype == torch.bfloat16:
        assert dcanvas is None or dcanvas.is_contiguous()
        check(L.pcp_mp_pfn_train_route_out_grad(ctypes.byref(vox.grid), _p(vox.workspace), vox.n, kept_rows, _p(dcanvas),
                                                _dt(dcanvas) if dcanvas is not None else _lib.DT_F32, _p(dpillar), _p(arg1), _p(dz1), _dt(dz1),
                                                _stream()), 'pcp_mp_pfn_train_route_out_grad')
        return
    check(L.pcp_pfn_train_route_out_grad(ctypes.byref(vox.grid), _p(vox.workspace), vox.n, kept_rows, _p(dcanvas), _p(dpillar), _p(arg1),
                                         _p(dz1), _stream()), 'pcp_pfn_train_route_out_grad')


def pfn_train_route_mid_grad(vox, din1, arg0, da0):
    L = _lib.load()
    if din1.dtype == torch.bfloat16:
        check(L.pcp_mp_pfn_train_route_mid_grad(ctypes.byref(vox.grid), _p(vox.workspace), vox.n, _p(din1), _lib.DT_BF16, _p(arg0), _p(da0),
                                                _stream()), 'pcp_mp_pfn_train_route_mid_grad')
        return
    check(L.pcp_pfn_train_route_mid_grad(ctypes.byref(vox.grid), _p(vox.workspace), vox.n, _p(din1), _p(arg0), _p(da0), _stream()),
          'pcp_pfn_train_route_mid_grad')


# ---------------------------------------------------------------------------------------------------------------------
# DiscoNet fusion (train mode)
# ---------------------------------------------------------------------------------------------------------------------

def _ptr_array(ptrs):
    return (ctypes.c_void_p * len(ptrs))(*ptrs)


def disco_weight_logits(h2_list, w4, b4, logits):
    """h2_list: list of (B,H,W,16) tensors; logits: (B,H,W,ld_w) -> column a = relu(conv1_4(h2_a))"""
    L = _lib.load()
    pixels = logits.numel() // logits.shape[-1]
    check(L.pcp_disco_weight_logits(_ptr_array([t.data_ptr() for t in h2_list]), len(h2_list), h2_list[0].shape[-1], _p(w4), _p(b4), pixels,
                                    _p(logits), logits.shape[-1], _stream()), 'pcp_disco_weight_logits')


def disco_fuse_backward(map_ptrs, ld_map, c, logits, dfused, h2_list, w4, dmap0, dh2_list, dw4, db4, accumulate=False, map_dtype=torch.float32):
    """map_dtype bfloat16: the stacked maps of the bf16 loop (pcp_mp_disco_fuse_backward; ld_map counts elements)"""
    L = _lib.load()
    pixels = logits.numel() // logits.shape[-1]
    ws = _BN_WS.get(L.pcp_disco_fuse_backward_workspace_bytes(), logits.device)
    if map_dtype == torch.bfloat16:
        check(L.pcp_mp_disco_fuse_backward(_ptr_array(map_ptrs), 1, len(map_ptrs), ld_map, c, _p(logits), logits.shape[-1], _p(dfused),
                                           dfused.shape[-1], _ptr_array([t.data_ptr() for t in h2_list]), h2_list[0].shape[-1], _p(w4), pixels,
                                           _p(dmap0), dmap0.shape[-1], _ptr_array([t.data_ptr() for t in dh2_list]), _p(ws), _p(dw4), _p(db4),
                                           1 if accumulate else 0, _stream()), 'pcp_mp_disco_fuse_backward')
        return
    check(L.pcp_disco_fuse_backward(_ptr_array(map_ptrs), len(map_ptrs), ld_map, c, _p(logits), logits.shape[-1], _p(dfused),
                                    dfused.shape[-1], _ptr_array([t.data_ptr() for t in h2_list]), h2_list[0].shape[-1], _p(w4), pixels,
                                    _p(dmap0), dmap0.shape[-1], _ptr_array([t.data_ptr() for t in dh2_list]), _p(ws), _p(dw4), _p(db4),
                                    1 if accumulate else 0, _stream()), 'pcp_disco_fuse_backward')


# ---------------------------------------------------------------------------------------------------------------------
# optimizer
# ---------------------------------------------------------------------------------------------------------------------

def grad_sqnorm(flat_grad, out=None, accumulate=False):
    L = _lib.load()
    if out is None:
        out = torch.zeros(1, dtype=torch.float64, device=flat_grad.device)
    check(L.pcp_grad_sqnorm(_p(flat_grad), flat_grad.numel(), _p(out), 1 if accumulate else 0, _stream()), 'pcp_grad_sqnorm')
    return out


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, max_norm=0.0, sqnorm=None, grad_scale=1.0):
    L = _lib.load()
    check(L.pcp_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), float(lr), float(beta1), float(beta2), float(eps),
                          float(weight_decay), int(step), float(max_norm), _p(sqnorm), float(grad_scale), _stream()), 'pcp_adam_step')


# ---------------------------------------------------------------------------------------------------------------------
# per-step weight repacking
# ---------------------------------------------------------------------------------------------------------------------

def pack_conv3x3(w, transpose, direct=None, direct_opad=0, wino=None, wino_opad=0, b3=None, b3_opad=0):
    """w: (cout, cin, 3, 3) contiguous CUDA.  Fills the given pre-allocated layout buffers in one launch."""
    _need_cuda(w, direct, wino, b3)
    L = _lib.load()
    assert w.is_contiguous() and w.dtype == torch.float32
    check(L.pcp_pack_conv3x3(_p(w), w.shape[0], w.shape[1], 1 if transpose else 0, _p(direct), direct_opad, _p(wino), wino_opad, _p(b3), b3_opad,
                             _stream()), 'pcp_pack_conv3x3')


def pack_conv3x3_winograd4(w, transpose, u4f=None, u4h=None, opad=0):
    """per-step repack of a 3x3 weight into the forms of the fused F(4x4,3x3) kernels (pcp_pack_conv3x3_winograd4); u4f / u4h: persistent
    float32 buffers of (I/8) * 36 * opad * 8 elements or None"""
    L = _lib.load()
    _need_cuda(w, u4f, u4h)
    assert w.is_contiguous() and w.dtype == torch.float32
    check(L.pcp_pack_conv3x3_winograd4(_p(w), w.shape[0], w.shape[1], 1 if transpose else 0, _p(u4f), _p(u4h), opad, _stream()),
          'pcp_pack_conv3x3_winograd4')


class PackGroup:
    """every 3x3 layer of the trainable branch repacked by ONE launch per optimizer step (pcp_pack_conv3x3_group).  A job = (weight, direction)
    with its persistent destination buffers; `owner` is weakly referenced (the conv module that caches the buffers): jobs of collected
    modules are dropped before the next launch."""

    def __init__(self):
        self.jobs = {}              # key -> (weakref to owner, PackJob, tensors kept alive)
        self.dirty = True
        self.table = None
        self.total = 0
        self.step = -1

    def add(self, key, owner, w, transpose, direct=None, direct_opad=0, wino=None, wino_opad=0, u4f=None, u4h=None, f4_opad=0):
        import weakref
        j = _lib.PackJob(_p(w), w.shape[0], w.shape[1], 1 if transpose else 0, direct_opad, _p(direct), _p(wino), wino_opad, f4_opad,
                         _p(u4f), _p(u4h), 0, 0)
        if _lib.load().pcp_pack_conv3x3_group_blocks(ctypes.byref(j)) <= 0:
            raise _lib.PcpError('pcp_pack_conv3x3_group: invalid job')
        self.jobs[key] = (weakref.ref(owner), j, (w, direct, wino, u4f, u4h))
        self.dirty = True

    def has(self, key):
        e = self.jobs.get(key)
        return e is not None and e[0]() is not None

    def drop(self, key):
        if self.jobs.pop(key, None) is not None:
            self.dirty = True

    def run(self, device):
        L = _lib.load()
        dead = [k for k, e in self.jobs.items() if e[0]() is None]
        for k in dead:
            del self.jobs[k]
            self.dirty = True
        if not self.jobs:
            return
        if self.dirty or self.table is None or self.table.device != device:
            arr = (_lib.PackJob * len(self.jobs))()
            start = 0
            for i, (_o, j, _t) in enumerate(self.jobs.values()):
                j.block_start = start
                start += L.pcp_pack_conv3x3_group_blocks(ctypes.byref(j))
                arr[i] = j
            self.total = start
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self.table = host.to(device)
            self.dirty = False
        check(L.pcp_pack_conv3x3_group(_p(self.table), len(self.jobs), self.total, _stream()), 'pcp_pack_conv3x3_group')


# ---------------------------------------------------------------------------------------------------------------------
# a17: HunterJr training branch
# ---------------------------------------------------------------------------------------------------------------------

_HUNTER_WS = Scratch()


class HunterMetaResult:
    """device tensors at their capacity; n_fg / n_local / n_inst read back once (the reference syncs several times here: torch.unique,
    .item())"""
    __slots__ = ('fg_idx', 'fg_local', 'local_key', 'local_inst', 'inst_key', 'inst_first', 'inst_last', 'n_fg', 'n_local', 'n_inst', 'bad_rows')


def hunter_meta(points, batch, max_inst, num_sweeps, sweep_col, inst_col):
    """hunter_jr.py:165-196 (_build_meta) + the foreground mask of :323"""
    _need_cuda(points)
    L = _lib.load()
    n, stride = points.shape
    d = _lib.HunterMeta(batch, max_inst, num_sweeps, sweep_col % stride, inst_col % stride)
    dev = points.device
    T, BM = batch * max_inst * num_sweeps, batch * max_inst
    i32 = lambda k: torch.empty(max(k, 1), dtype=torch.int32, device=dev)
    r = HunterMetaResult()
    r.fg_idx, r.fg_local, r.local_key, r.local_inst = i32(n), i32(n), i32(T), i32(T)
    r.inst_key, r.inst_first, r.inst_last = i32(BM), i32(BM), i32(BM)
    counts = i32(4)
    ws = _HUNTER_WS.get(L.pcp_hunter_meta_workspace_bytes(ctypes.byref(d), n), dev)
    check(L.pcp_hunter_meta(ctypes.byref(d), _p(points), n, stride, _p(ws), ws.numel(), _p(r.fg_idx), _p(r.fg_local), _p(r.local_key),
                            _p(r.local_inst), _p(r.inst_key), _p(r.inst_first), _p(r.inst_last), _p(counts), _stream()), 'pcp_hunter_meta')
    r.n_fg, r.n_local, r.n_inst, r.bad_rows = [int(v) for v in counts.tolist()]
    return r


def segment_max(src, seg, n_seg, c, row_index=None, rows=None):
    """torch_scatter.scatter_max(src[row_index], seg, dim=0): returns (out (n_seg, c), arg (n_seg, c) int32)"""
    _need_cuda(src, seg, row_index)
    L = _lib.load()
    rows = (row_index.shape[0] if row_index is not None else src.shape[0]) if rows is None else rows
    out = torch.empty((n_seg, c), dtype=torch.float32, device=src.device)
    arg = torch.empty((n_seg, c), dtype=torch.int32, device=src.device)
    check(L.pcp_segment_max(_p(src), src.shape[-1], _p(row_index), rows, _p(seg), n_seg, c, _p(out), c, _p(arg), _stream()), 'pcp_segment_max')
    return out, arg


def segment_max_backward(dout, arg, dsrc, c, row_index=None):
    L = _lib.load()
    check(L.pcp_segment_max_backward(_p(dout), dout.shape[-1], _p(arg), arg.shape[0], c, _p(row_index), _p(dsrc), dsrc.shape[-1], _stream()),
          'pcp_segment_max_backward')
    return dsrc


def rows_scatter_add(src, row_index, rows, c, dst):
    L = _lib.load()
    check(L.pcp_rows_scatter_add(_p(src), src.shape[-1], _p(row_index), rows, c, _p(dst), dst.shape[-1], _stream()), 'pcp_rows_scatter_add')
    return dst


def hunter_local_centroids(points, meta, ld_centered=16):
    L = _lib.load()
    dev = points.device
    centroid = torch.empty((meta.n_local, 3), dtype=torch.float32, device=dev)
    centered = torch.empty((meta.n_fg, ld_centered), dtype=torch.float32, device=dev)
    ws = _HUNTER_WS.get(32 * meta.n_local, dev)
    check(L.pcp_hunter_local_centroids(_p(points), points.shape[1], _p(meta.fg_idx), _p(meta.fg_local), meta.n_fg, meta.n_local, _p(ws),
                                       ws.numel(), _p(centroid), _p(centered), ld_centered, _stream()), 'pcp_hunter_local_centroids')
    return centroid, centered


def hunter_object_cat(lf0, gf, centroid, meta, c, ld_out):
    L = _lib.load()
    out = torch.empty((meta.n_local, ld_out), dtype=torch.float32, device=lf0.device)
    check(L.pcp_hunter_object_cat(_p(lf0), _p(gf), _p(centroid), _p(meta.local_inst), _p(meta.inst_last), meta.n_local, c, _p(out), ld_out,
                                  _stream()), 'pcp_hunter_object_cat')
    return out


def hunter_object_cat_backward(dcat, meta, c):
    L = _lib.load()
    dlf0 = torch.empty((meta.n_local, c), dtype=torch.float32, device=dcat.device)
    dgf = torch.empty((meta.n_inst, c), dtype=torch.float32, device=dcat.device)
    check(L.pcp_hunter_object_cat_backward(_p(dcat), dcat.shape[-1], _p(meta.inst_first), _p(meta.inst_last), meta.n_local, meta.n_inst, c,
                                           _p(dlf0), _p(dgf), _stream()), 'pcp_hunter_object_cat_backward')
    return dlf0, dgf


def hunter_losses(desc, device):
    """desc: lib.HunterLoss with every pointer set (see include/pcp_hip_train.h); runs the loss + gradient kernels"""
    L = _lib.load()
    need = L.pcp_hunter_loss_workspace_bytes(desc.n, desc.n_fg, desc.n_local, desc.c)
    ws = _HUNTER_WS2.get(need, device)
    check(L.pcp_hunter_losses(ctypes.byref(desc), _p(ws), ws.numel(), _stream()), 'pcp_hunter_losses')


_HUNTER_WS2 = Scratch()


def softmax_fuse2_backward(dfused, cat, logits, c, dcat, dlogits):
    L = _lib.load()
    pixels = dfused.numel() // dfused.shape[-1]
    check(L.pcp_softmax_fuse2_backward(_p(dfused), dfused.shape[-1], _p(cat), cat.shape[-1], _p(logits), logits.shape[-1], pixels, c, _p(dcat),
                                       dcat.shape[-1], _p(dlogits), dlogits.shape[-1], _stream()), 'pcp_softmax_fuse2_backward')


def bev_scatter_mean_backward(workspace, batch, h, w, n, dmap, dmap_ch_off, c, dyn_mask, dfeat_acc, dfeat_dyn):
    L = _lib.load()
    check(L.pcp_bev_scatter_mean_backward(_p(workspace), batch, h, w, n, _chan_ptr(dmap, dmap_ch_off), dmap.shape[-1], c, _p(dyn_mask),
                                          _p(dfeat_acc), dfeat_acc.shape[-1], _p(dfeat_dyn), dfeat_dyn.shape[-1], _stream()),
          'pcp_bev_scatter_mean_backward')


def bev_sample_bilinear_backward(dfeat, points, batch, h, w, c, min_xy, pix_xy, dbev, row_mask=None, bev=None, dxyz=None, dxyz_ch_off=0):
    L = _lib.load()
    n, stride = points.shape
    check(L.pcp_bev_sample_bilinear_backward(_p(dfeat), dfeat.shape[-1], _p(row_mask), _p(points), n, stride, _p(bev),
                                             bev.shape[-1] if bev is not None else 0, batch, h, w, c, float(min_xy[0]), float(min_xy[1]),
                                             float(pix_xy[0]), float(pix_xy[1]), _p(dbev), dbev.shape[-1],
                                             _chan_ptr(dxyz, dxyz_ch_off) if dxyz is not None else ctypes.c_void_p(0),
                                             dxyz.shape[-1] if dxyz is not None else 0, _stream()), 'pcp_bev_sample_bilinear_backward')


def filter_gt_boxes(gt_boxes, pc_range):
    L = _lib.load()
    B, M, _ = gt_boxes.shape
    out = torch.empty_like(gt_boxes)
    rng = (ctypes.c_float * 6)(*[float(v) for v in pc_range])
    check(L.pcp_filter_gt_boxes(_p(gt_boxes), B, M, rng, _p(out), _stream()), 'pcp_filter_gt_boxes')
    return out
