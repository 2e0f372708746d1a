"""Tensor-level wrappers over the C ABI: argument checks, workspace allocation through torch (device memory and streams are
torch's job; the arithmetic is not), and NHWC bookkeeping.  Every function requires CUDA tensors and raises otherwise --
there is no CPU or eager-PyTorch fallback.
"""
import ctypes

import torch

from . import lib as _lib
from .lib import Conv3x3, Decode, DetHead, Grid, Pointwise, check


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_CUR_DEVICE = getattr(torch._C, '_cuda_getDevice', None)


def current_stream_handle():
    """the hipStream_t (as an int) torch launches on right now.  torch.cuda.current_stream() builds a Stream object (~8 us); a forward
    asks ~120 times per batch, which was a millisecond of host time per DiscoNet step -- the raw query is 20 x cheaper"""
    if _RAW_STREAM is not None and _CUR_DEVICE is not None:
        return int(_RAW_STREAM(_CUR_DEVICE()))
    return int(torch.cuda.current_stream().cuda_stream)


def _stream():
    return ctypes.c_void_p(current_stream_handle())


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _need_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.PcpError('the HIP hot path needs CUDA/ROCm tensors (got a %s tensor); no CPU fallback exists' % t.device)


def _need_f32(what, *tensors):
    """the fp32 kernels read their maps as float: a bf16 tensor (an output of the bf16 loop, include/pcp_hip_mp.h) handed to one of them
    would be read past its end"""
    for t in tensors:
        if t is not None and t.dtype != torch.float32:
            raise _lib.PcpError('%s reads float32 maps, got %s (cast the bf16 activation first)' % (what, t.dtype))


def _zeros_views(device, specs):
    """several zero-initialised output tensors carved out of ONE allocation (one fill launch instead of one per tensor).
    specs: [(shape, dtype)], all 4-byte dtypes; every view starts 16-byte aligned."""
    sizes = []
    for shape, _dt in specs:
        n = 1
        for v in shape:
            n *= int(v)
        sizes.append((n + 3) // 4 * 4)
    buf = torch.zeros((max(sum(sizes), 1),), dtype=torch.int32, device=device)
    out, off = [], 0
    for (shape, dt), sz in zip(specs, sizes):
        n = 1
        for v in shape:
            n *= int(v)
        out.append(buf[off:off + n].view(dt).view(*shape))
        off += sz
    return out


def make_grid(pc_range, voxel_size, grid_size, batch_size):
    import numpy as np
    r = np.asarray(pc_range, dtype=np.float32)
    v = np.asarray(voxel_size, dtype=np.float32)
    return Grid(float(r[0]), float(r[1]), float(r[2]), float(v[0]), float(v[1]), float(v[2]), int(grid_size[0]),
                int(grid_size[1]), int(batch_size))


# ---------------------------------------------------------------------------------------------------------------------
# NHWC helpers: a "map" is a contiguous (B, H, W, C) tensor; callers see it as an NCHW-shaped channels_last view
# ---------------------------------------------------------------------------------------------------------------------

def as_nhwc(x):
    """(B, C, H, W) tensor of any layout -> contiguous (B, H, W, C) storage (no copy when already channels_last)."""
    return x.permute(0, 2, 3, 1).contiguous()


def nchw_view(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2)


# ---------------------------------------------------------------------------------------------------------------------
# a1-a5
# ---------------------------------------------------------------------------------------------------------------------

class VoxelizeResult:
    __slots__ = ('workspace', 'voxel_coords', 'unq_inv', 'unq_cnt', 'counters', 'n', 'grid', 'row_stride', 'num_raw', 'has_bucket_order')


def voxelize_workspace(grid, n, device, workspace=None):
    """a pillariser workspace large enough for (grid, n rows); `workspace` is returned unchanged when it already is"""
    need = _lib.load().pcp_voxelize_workspace_bytes(ctypes.byref(grid), n)
    if workspace is None or workspace.numel() < need or workspace.device != device:
        workspace = torch.empty(need, dtype=torch.uint8, device=device)
    return workspace


def voxelize(points, grid, want_inverse=True, want_counts=True, workspace=None, cells_ready=False):
    """points: (N, 1+C) float32 CUDA.  Returns VoxelizeResult with max-size outputs; counters = [P, N', 0, 0] on device.
    cells_ready: `workspace` already holds the rows' cell ids and the per-cell histogram (select_transform_compact with a grid): the
    pillariser's first pass is skipped (no unq_inv in that mode)."""
    _need_cuda(points)
    L = _lib.load()
    assert points.dtype == torch.float32 and points.dim() == 2 and points.is_contiguous()
    n, stride = points.shape
    need = L.pcp_voxelize_workspace_bytes(ctypes.byref(grid), n)
    if cells_ready:
        assert workspace is not None and workspace.numel() >= need and not want_inverse
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=points.device)
    res = VoxelizeResult()
    res.workspace = workspace
    res.n = n
    res.grid = grid
    res.row_stride = stride
    res.num_raw = None
    res.has_bucket_order = True
    cap = max(n, 1)
    res.voxel_coords = torch.empty((cap, 4), dtype=torch.int32, device=points.device)
    res.unq_inv = torch.empty((cap,), dtype=torch.int64, device=points.device) if want_inverse else None
    res.unq_cnt = torch.empty((cap,), dtype=torch.int32, device=points.device) if want_counts else None
    res.counters = torch.empty((4,), dtype=torch.int32, device=points.device)      # all four written by the scan kernel (no fill launch)
    if cells_ready:
        check(L.pcp_voxelize_cells_ready(_p(points), n, stride, ctypes.byref(grid), _p(workspace), workspace.numel(), _p(res.voxel_coords),
                                         _p(res.unq_cnt), _p(res.counters), _stream()), 'pcp_voxelize_cells_ready')
        return res
    check(L.pcp_voxelize(_p(points), n, stride, ctypes.byref(grid), _p(workspace), workspace.numel(), _p(res.voxel_coords),
                         _p(res.unq_inv), _p(res.unq_cnt), _p(res.counters), _stream()), 'pcp_voxelize')
    return res


def pfn_scatter(points, vox, num_raw, w0, b0, w1, b1, canvas=None, pillar_features=None):
    """Runs the fused PFN on the buckets left in vox.workspace.  canvas: (B, ny, nx, 64) NHWC, pre-zeroed."""
    _need_cuda(points, w0, b0, w1, b1, canvas, pillar_features)
    L = _lib.load()
    for t in (w0, b0, w1, b1):
        assert t.dtype == torch.float32 and t.is_contiguous()
    assert w0.shape == (32, num_raw + 6) and w1.shape == (64, 64) and b0.shape == (32,) and b1.shape == (64,)
    check(L.pcp_pfn_scatter(_p(points), vox.n, vox.row_stride, num_raw, ctypes.byref(vox.grid), _p(vox.workspace), _p(w0), _p(b0),
                            _p(w1), _p(b1), _p(pillar_features), _p(canvas), _stream()), 'pcp_pfn_scatter')


ROWS_CELLS_READY, ROWS_BUCKET_ORDER = 1, 2


def rows_workspace(grid, n, num_raw, device, workspace=None):
    """a pcp_pillarise_rows workspace large enough for (grid, n rows, num_raw); `workspace` is returned unchanged when it already is"""
    need = _lib.load().pcp_pillarise_rows_workspace_bytes(ctypes.byref(grid), n, num_raw)
    if workspace is None or workspace.numel() < need or workspace.device != device:
        workspace = torch.empty(need, dtype=torch.uint8, device=device)
    return workspace


def pillarise_rows(points, grid, num_raw, want_inverse=False, want_counts=False, want_coords=False, workspace=None, cells_ready=False,
                   bucket_order=False):
    """round 5 pillariser (four launches): pcp_voxelize's outputs (each optional) + the kept rows in pillar order + the wave-tile
    descriptors pfn_rows reads, all in the workspace.  Returns a VoxelizeResult (res.num_raw set: the PFN must use the same)."""
    _need_cuda(points)
    L = _lib.load()
    assert points.dtype == torch.float32 and points.dim() == 2 and points.is_contiguous()
    n, stride = points.shape
    need = L.pcp_pillarise_rows_workspace_bytes(ctypes.byref(grid), n, num_raw)
    if need == 0:
        raise _lib.PcpError('pcp_pillarise_rows: unsupported (grid, num_raw=%d)' % num_raw)
    if cells_ready:
        assert workspace is not None and workspace.numel() >= need and not want_inverse
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=points.device)
    res = VoxelizeResult()
    res.workspace = workspace
    res.n = n
    res.grid = grid
    res.row_stride = stride
    res.num_raw = num_raw
    res.has_bucket_order = bool(bucket_order)
    cap = max(n, 1)
    dev = points.device
    res.voxel_coords = torch.empty((cap, 4), dtype=torch.int32, device=dev) if want_coords else None
    res.unq_inv = torch.empty((cap,), dtype=torch.int64, device=dev) if want_inverse else None
    res.unq_cnt = torch.empty((cap,), dtype=torch.int32, device=dev) if want_counts else None
    res.counters = torch.empty((4,), dtype=torch.int32, device=dev)
    flags = (ROWS_CELLS_READY if cells_ready else 0) | (ROWS_BUCKET_ORDER if bucket_order else 0)
    check(L.pcp_pillarise_rows(_p(points), n, stride, num_raw, ctypes.byref(grid), _p(workspace), workspace.numel(), _p(res.voxel_coords),
                               _p(res.unq_inv), _p(res.unq_cnt), _p(res.counters), flags, _stream()), 'pcp_pillarise_rows')
    return res


def pillar_index_export_async(vox, want_records=False):
    """pcp_pillar_index_export on the current stream, no host read: (voxel_coords (cap, 4), row_rank (cap,), counters (4,), slot_rank,
    slot_canvas_row) as device tensors (the last two None unless want_records on a rows workspace); only the first P / n / N' entries hold data"""
    L = _lib.load()
    dev = vox.workspace.device
    cap = max(vox.n, 1)
    num_raw = int(getattr(vox, 'num_raw', 0) or 0)
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    row_rank = torch.empty((cap,), dtype=torch.int32, device=dev)
    counters = torch.empty((4,), dtype=torch.int32, device=dev)
    slot_rank = torch.empty((cap,), dtype=torch.int32, device=dev) if (want_records and num_raw) else None
    slot_row = torch.empty((cap,), dtype=torch.int32, device=dev) if (want_records and num_raw) else None
    check(L.pcp_pillar_index_export(ctypes.byref(vox.grid), _p(vox.workspace), vox.n, num_raw, _p(coords), _p(row_rank), _p(slot_rank),
                                    _p(slot_row), _p(counters), _stream()), 'pcp_pillar_index_export')
    return coords, row_rank, counters, slot_rank, slot_row


def pillar_index_export(vox, want_records=False):
    """the reference's index tensors from the workspace of a pillariser call that did not write them (pipeline mode): returns
    (voxel_coords (P, 4) int32 [b, 0, y, x], unq_inv (N',) int64, counters (4,) numpy) -- one host read -- and, with want_records, the
    (slot_rank, slot_canvas_row) int32 tensors of the pillar-ordered records pcp_pfn_rows consumes (rows workspaces only)."""
    coords, row_rank, counters, slot_rank, slot_row = pillar_index_export_async(vox, want_records)
    cnt = counters.cpu().numpy()
    num_pillars, kept = int(cnt[0]), int(cnt[1])
    rr = row_rank[:vox.n]
    inv = rr[rr >= 0].to(torch.int64)
    assert inv.numel() == kept, (inv.numel(), kept)
    if want_records:
        return coords[:num_pillars], inv, cnt, (None if slot_rank is None else slot_rank[:kept]), (None if slot_row is None else slot_row[:kept])
    return coords[:num_pillars], inv, cnt


def set_option(name, value):
    """pcp_set_option: override (or with None restore) a built-in launch rule; returns the previous override"""
    return _lib.set_option(name, value)


def pfn_rows(vox, w0, b0, w1, b1, canvas=None, pillar_features=None):
    """fused PFN + scatter on the records pillarise_rows left in vox.workspace.  canvas (B, ny, nx, 64) is written completely (pillar
    rows and zero rows): torch.empty is enough."""
    _need_cuda(w0, b0, w1, b1, canvas, pillar_features)
    _need_f32('pcp_pfn_rows', canvas, pillar_features)
    L = _lib.load()
    num_raw = vox.num_raw
    for t in (w0, b0, w1, b1):
        assert t.dtype == torch.float32 and t.is_contiguous()
    assert w0.shape == (32, num_raw + 6) and w1.shape == (64, 64) and b0.shape == (32,) and b1.shape == (64,)
    if canvas is not None:
        assert canvas.is_contiguous() and tuple(canvas.shape) == (vox.grid.batch_size, vox.grid.ny, vox.grid.nx, 64)
    check(L.pcp_pfn_rows(ctypes.byref(vox.grid), _p(vox.workspace), vox.n, num_raw, _p(w0), _p(b0), _p(w1), _p(b1), _p(pillar_features),
                         _p(canvas), _stream()), 'pcp_pfn_rows')


def pfn_features(points, vox, num_raw, use_absolute_xyz=True, with_distance=False):
    """feature rows of dynamic_pillar_vfe.py:110-126 in the bucket order `voxelize` left in vox.workspace, composition chosen at run time.
    Returns (fbuf (N, fw) zero padded to a multiple of 16 floats, slot_pillar (N,) int32, true width)."""
    _need_cuda(points)
    L = _lib.load()
    assert vox.has_bucket_order and points.dtype == torch.float32 and points.is_contiguous()
    f = num_raw - (0 if use_absolute_xyz else 3) + 6 + (1 if with_distance else 0)
    fw = (f + 15) // 16 * 16
    rows = max(vox.n, 1)
    fbuf = torch.empty((rows, fw), dtype=torch.float32, device=points.device)
    slot_pillar = torch.empty((rows,), dtype=torch.int32, device=points.device)
    flags = (1 if use_absolute_xyz else 0) | (2 if with_distance else 0)
    check(L.pcp_pfn_features(_p(points), vox.n, vox.row_stride, num_raw, flags, ctypes.byref(vox.grid), _p(vox.workspace), fw, _p(fbuf),
                             _p(slot_pillar), _stream()), 'pcp_pfn_features')
    return fbuf, slot_pillar, f


def pfn_cat_pillar_max(y, pillar_max, slot_pillar, rows, c):
    """[y[:, :c], pillar_max[slot_pillar, :c]] padded with zeros to a multiple of 16 floats: the input rows of the next PFNLayerV2"""
    _need_cuda(y, pillar_max, slot_pillar)
    L = _lib.load()
    ld_out = (2 * c + 15) // 16 * 16
    out = torch.empty((max(rows, 1), ld_out), dtype=torch.float32, device=y.device)
    check(L.pcp_pfn_cat_pillar_max(_p(y), y.shape[-1], _p(pillar_max), pillar_max.shape[-1], _p(slot_pillar), rows, c, _p(out), ld_out,
                                   _stream()), 'pcp_pfn_cat_pillar_max')
    return out


def canvas_clear(vox, canvas):
    L = _lib.load()
    check(L.pcp_canvas_clear(ctypes.byref(vox.grid), _p(vox.workspace), vox.n, _p(canvas), _stream()), 'pcp_canvas_clear')


def fill_zero(t):
    L = _lib.load()
    check(L.pcp_fill_zero(_p(t), t.numel() * t.element_size(), _stream()), 'pcp_fill_zero')


# ---------------------------------------------------------------------------------------------------------------------
# a6 / a7 convolutions.  x: (B, H, W, ld) NHWC storage; channel windows are expressed by (tensor, channel offset, channels)
# ---------------------------------------------------------------------------------------------------------------------

def _chan_ptr(t, ch_off):
    return ctypes.c_void_p(t.data_ptr() + t.element_size() * ch_off)


def conv3x3(x, packed, bias, cin, cout, cout_pad, stride=1, relu=True, out=None, in_ch_off=0, out_ch_off=0):
    """x: (B, H, W, ld_in) float32 NHWC.  out: (B, Ho, Wo, ld_out) or None (allocated with ld_out = cout)."""
    _need_cuda(x, packed, bias, out)
    _need_f32('pcp_conv3x3', x, out)
    L = _lib.load()
    B, H, W, ld_in = x.shape
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    if out is None:
        out = torch.empty((B, Ho, Wo, cout), dtype=torch.float32, device=x.device)
    assert out.shape[:3] == (B, Ho, Wo) and x.is_contiguous() and out.is_contiguous()
    assert in_ch_off + cin <= ld_in and out_ch_off + cout <= out.shape[3]
    assert bias.numel() >= cout_pad, 'bias must hold cout_pad values (the epilogue reads it 16 bytes at a time)'
    d = Conv3x3(B, H, W, cin, cout, cout_pad, stride, ld_in, out.shape[3], 1 if relu else 0)
    check(L.pcp_conv3x3(ctypes.byref(d), _chan_ptr(x, in_ch_off), _p(packed), _p(bias), _chan_ptr(out, out_ch_off), _stream()),
          'pcp_conv3x3')
    return out


def sparse_conv3x3_s2(pillar_features, vox, w_packed, bias, cout, relu=True, out=None, out_dtype=torch.float32):
    """first backbone layer (ZeroPad2d(1) + 3x3 stride-2 conv + folded BN + ReLU) straight from the pillar list: pillar_features (>= P, 64)
    in pillar-rank order and the voxelize result whose workspace still holds the cell -> rank table.  Returns (B, ny/2, nx/2, cout)."""
    _need_cuda(pillar_features, w_packed, bias, out)
    L = _lib.load()
    g = vox.grid
    B, ho, wo = g.batch_size, (g.ny - 1) // 2 + 1, (g.nx - 1) // 2 + 1
    if out is None:
        out = torch.empty((B, ho, wo, cout), dtype=out_dtype, device=pillar_features.device)
    assert out.shape[:3] == (B, ho, wo) and out.is_contiguous() and pillar_features.is_contiguous() and pillar_features.shape[1] == 64
    if out.dtype == torch.bfloat16:      # bf16 training loop: the teacher's second layer reads bf16 (include/pcp_hip_mp.h)
        check(L.pcp_mp_sparse_conv3x3_s2(_p(pillar_features), ctypes.byref(g), _p(vox.workspace), vox.n, _p(w_packed), _p(bias), cout,
                                         1 if relu else 0, _p(out), _lib.DT_BF16, out.shape[3], _stream()), 'pcp_mp_sparse_conv3x3_s2')
        return out
    check(L.pcp_sparse_conv3x3_s2(_p(pillar_features), ctypes.byref(g), _p(vox.workspace), vox.n, _p(w_packed), _p(bias), cout,
                                  1 if relu else 0, _p(out), out.shape[3], _stream()), 'pcp_sparse_conv3x3_s2')
    return out


def conv3x3_winograd(x, u_packed, bias, cin, cout, cout_pad, relu=True, out=None, in_ch_off=0, out_ch_off=0):
    """stride-1 3x3 conv through the fused Winograd F(2x2,3x3) kernel; same tensor contract as conv3x3."""
    _need_cuda(x, u_packed, bias, out)
    L = _lib.load()
    B, H, W, ld_in = x.shape
    if out is None:
        out = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device)
    assert out.shape[:3] == (B, H, W) and x.is_contiguous() and out.is_contiguous()
    assert in_ch_off + cin <= ld_in and out_ch_off + cout <= out.shape[3]
    d = Conv3x3(B, H, W, cin, cout, cout_pad, 1, ld_in, out.shape[3], 1 if relu else 0)
    check(L.pcp_conv3x3_winograd(ctypes.byref(d), _chan_ptr(x, in_ch_off), _p(u_packed), _p(bias), _chan_ptr(out, out_ch_off),
                                 _stream()), 'pcp_conv3x3_winograd')
    return out


def conv3x3_winograd_ws(x, u_packed, bias, cin, cout, cout_pad, relu=True, out=None, in_ch_off=0, out_ch_off=0):
    """stride-1 3x3 conv through the wave-stationary fused Winograd F(2x2,3x3) kernel (csrc/wino_ws.hip; weights from
    pack.pack_conv3x3_winograd_ws); same tensor contract as conv3x3.  cin % 32 == 0, cout_pad % 64 == 0."""
    _need_cuda(x, u_packed, bias, out)
    L = _lib.load()
    B, H, W, ld_in = x.shape
    if out is None:
        out = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device)
    assert out.shape[:3] == (B, H, W) and x.is_contiguous() and out.is_contiguous()
    assert in_ch_off + cin <= ld_in and out_ch_off + cout <= out.shape[3]
    d = Conv3x3(B, H, W, cin, cout, cout_pad, 1, ld_in, out.shape[3], 1 if relu else 0)
    check(L.pcp_conv3x3_winograd_ws(ctypes.byref(d), _chan_ptr(x, in_ch_off), _p(u_packed), _p(bias), _chan_ptr(out, out_ch_off),
                                    _stream()), 'pcp_conv3x3_winograd_ws')
    return out


def conv3x3_winograd4f(x, u_packed, bias, cin, cout, cout_pad, relu=True, out=None, in_ch_off=0, out_ch_off=0):
    """stride-1 3x3 conv through the FUSED Winograd F(4x4,3x3) kernel (csrc/wino4f.hip; weights from pack.pack_conv3x3_winograd4f); same
    tensor contract as conv3x3.  cin % 8 == 0, cout_pad % 64 == 0."""
    _need_cuda(x, u_packed, bias, out)
    L = _lib.load()
    B, H, W, ld_in = x.shape
    if out is None:
        out = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device)
    assert out.shape[:3] == (B, H, W) and x.is_contiguous() and out.is_contiguous()
    assert in_ch_off + cin <= ld_in and out_ch_off + cout <= out.shape[3]
    d = Conv3x3(B, H, W, cin, cout, cout_pad, 1, ld_in, out.shape[3], 1 if relu else 0)
    check(L.pcp_conv3x3_winograd4f(ctypes.byref(d), _chan_ptr(x, in_ch_off), _p(u_packed), _p(bias), _chan_ptr(out, out_ch_off),
                                   _stream()), 'pcp_conv3x3_winograd4f')
    return out


def conv3x3_winograd4h(x, u_packed, bias, cin, cout, cout_pad, relu=True, out=None, in_ch_off=0, out_ch_off=0):
    """stride-1 3x3 conv through the fused F(4x4,3x3) kernel with two four-wave workgroups per CU (csrc/wino4h.hip; weights from
    pack.pack_conv3x3_winograd4h); same tensor contract as conv3x3_winograd4f."""
    _need_cuda(x, u_packed, bias, out)
    L = _lib.load()
    B, H, W, ld_in = x.shape
    if out is None:
        out = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device)
    assert out.shape[:3] == (B, H, W) and x.is_contiguous() and out.is_contiguous()
    assert in_ch_off + cin <= ld_in and out_ch_off + cout <= out.shape[3]
    d = Conv3x3(B, H, W, cin, cout, cout_pad, 1, ld_in, out.shape[3], 1 if relu else 0)
    check(L.pcp_conv3x3_winograd4h(ctypes.byref(d), _chan_ptr(x, in_ch_off), _p(u_packed), _p(bias), _chan_ptr(out, out_ch_off),
                                   _stream()), 'pcp_conv3x3_winograd4h')
    return out


def conv3x3_winograd4c(x, u_packed, bias, cin, cout, cout_pad, relu=True, out=None, in_ch_off=0, out_ch_off=0):
    """stride-1 3x3 conv through the fused F(4x4,3x3) kernel whose waves split the output channels (csrc/wino4c.hip: the output transform
    in registers; weights from pack.pack_conv3x3_winograd4c); same tensor contract and the same bits as conv3x3_winograd4h."""
    _need_cuda(x, u_packed, bias, out)
    _need_f32('pcp_conv3x3_winograd4c', x, out)
    L = _lib.load()
    B, H, W, ld_in = x.shape
    if out is None:
        out = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device)
    assert out.shape[:3] == (B, H, W) and x.is_contiguous() and out.is_contiguous()
    assert in_ch_off + cin <= ld_in and out_ch_off + cout <= out.shape[3]
    d = Conv3x3(B, H, W, cin, cout, cout_pad, 1, ld_in, out.shape[3], 1 if relu else 0)
    check(L.pcp_conv3x3_winograd4c(ctypes.byref(d), _chan_ptr(x, in_ch_off), _p(u_packed), _p(bias), _chan_ptr(out, out_ch_off),
                                   _stream()), 'pcp_conv3x3_winograd4c')
    return out


_W4_WORKSPACE = {}
_W4_RETIRED = []


def _w4_workspace(device, nbytes):
    """one grow-only scratch buffer per (device, launch stream) for the F(4x4,3x3) transforms (V and M, ~0.9 GB for 768 -> 768 at 4
    frames); the three launches of a call consume it in stream order, so consecutive calls on the same stream share it and concurrent
    streams (CenterPoint.overlap_makers) each get their own.  A buffer that is outgrown stays referenced: a captured hipGraph may have
    its address baked in."""
    key = (device, current_stream_handle())      # one buffer per launch stream: concurrent streams never share it
    buf = _W4_WORKSPACE.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _W4_RETIRED.append(buf)
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _W4_WORKSPACE[key] = buf
    return buf


def conv3x3_winograd4(x, u_packed, bias, cin, cout, cout_pad, relu=True, out=None, in_ch_off=0, out_ch_off=0, stage_times=None):
    """stride-1 3x3 conv of a wide layer through Winograd F(4x4,3x3) (transform, 36 batched MFMA GEMMs, transform); same tensor
    contract as conv3x3.  stage_times: optional list; when given the call is synchronous and appends
    (input_ms, gemm_ms, output_ms, gemm_flops) measured with HIP events on the launch stream."""
    _need_cuda(x, u_packed, bias, out)
    L = _lib.load()
    B, H, W, ld_in = x.shape
    if out is None:
        out = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device)
    assert out.shape[:3] == (B, H, W) and x.is_contiguous() and out.is_contiguous()
    assert in_ch_off + cin <= ld_in and out_ch_off + cout <= out.shape[3]
    d = Conv3x3(B, H, W, cin, cout, cout_pad, 1, ld_in, out.shape[3], 1 if relu else 0)
    nbytes = ctypes.c_size_t(0)
    check(L.pcp_conv3x3_winograd4_workspace_bytes(ctypes.byref(d), ctypes.byref(nbytes)), 'pcp_conv3x3_winograd4_workspace_bytes')
    ws = _w4_workspace(x.device, nbytes.value)
    if stage_times is not None:
        ms = (ctypes.c_float * 3)()
        fl = ctypes.c_double(0.0)
        check(L.pcp_conv3x3_winograd4_timed(ctypes.byref(d), _chan_ptr(x, in_ch_off), _p(u_packed), _p(bias), _chan_ptr(out, out_ch_off),
                                            _p(ws), _stream(), ms, ctypes.byref(fl)), 'pcp_conv3x3_winograd4_timed')
        stage_times.append((ms[0], ms[1], ms[2], fl.value))
        return out
    check(L.pcp_conv3x3_winograd4(ctypes.byref(d), _chan_ptr(x, in_ch_off), _p(u_packed), _p(bias), _chan_ptr(out, out_ch_off), _p(ws),
                                  _stream()), 'pcp_conv3x3_winograd4')
    return out


def conv3x3_bf16x3(x, packed, bias, cin, cout, cout_pad, stride=1, relu=True, out=None, in_ch_off=0, out_ch_off=0, plain=False):
    """opt-in split-bf16 arithmetic (see include/pcp_hip.h); same tensor contract as conv3x3.  plain=True: single bf16 products
    (pcp_conv3x3_bf16, the mixed-precision training mode)"""
    _need_cuda(x, packed, bias, out)
    L = _lib.load()
    B, H, W, ld_in = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None:
        out = torch.empty((B, Ho, Wo, cout), dtype=torch.float32, device=x.device)
    assert out.shape[:3] == (B, Ho, Wo) and x.is_contiguous() and out.is_contiguous()
    assert in_ch_off + cin <= ld_in and out_ch_off + cout <= out.shape[3]
    d = Conv3x3(B, H, W, cin, cout, cout_pad, stride, ld_in, out.shape[3], 1 if relu else 0)
    fn = L.pcp_conv3x3_bf16 if plain else L.pcp_conv3x3_bf16x3
    check(fn(ctypes.byref(d), _chan_ptr(x, in_ch_off), _p(packed), _p(bias), _chan_ptr(out, out_ch_off), _stream()),
          'pcp_conv3x3_bf16' if plain else 'pcp_conv3x3_bf16x3')
    return out


def conv3x3_grouped_small(x, weights, bias, offsets, out):
    """x: (B, H, W, ld_in); weights (n_out, 9, cin_per_group), cin_per_group % 64 == 0; offsets: python list of groups+1 ints."""
    _need_cuda(x, weights, bias, out)
    L = _lib.load()
    B, H, W, ld_in = x.shape
    offs = (ctypes.c_int32 * len(offsets))(*[int(v) for v in offsets])
    check(L.pcp_conv3x3_grouped_small(_p(x), B, H, W, ld_in, len(offsets) - 1, weights.shape[2], offs, _p(weights), _p(bias), _p(out), out.shape[-1],
                                      _stream()), 'pcp_conv3x3_grouped_small')
    return out


def pointwise(x, packed, bias, mode, cin, cout, cout_pad, relu=True, out=None, in_ch_off=0, out_ch_off=0, x2=None,
              k_split=0, x2_ch_off=0, residual=None, res_ch_off=0):
    """mode PW_PLAIN: x (..., ld_in) rows; PW_SPACE2DEPTH / PW_DEPTH2SPACE: x (B, H, W, ld_in).
    PLAIN extras: x2 supplies contraction channels [k_split, cin) (a cat without the copy); residual is added last."""
    _need_cuda(x, packed, bias, out, x2, residual)
    _need_f32('pcp_pointwise', x, out, x2, residual)
    L = _lib.load()
    ld_in = x.shape[-1]
    if mode == _lib.PW_PLAIN:
        rows = x.numel() // ld_in
        B = H = W = 0
        if out is None:
            out = torch.empty(tuple(x.shape[:-1]) + (cout,), dtype=torch.float32, device=x.device)
    else:
        B, H, W, _ = x.shape
        rows = 0
        if out is None:
            shp = (B, H // 2, W // 2, cout) if mode == _lib.PW_SPACE2DEPTH else (B, 2 * H, 2 * W, cout)
            out = torch.empty(shp, dtype=torch.float32, device=x.device)
    assert bias.numel() >= cout_pad, 'bias must hold cout_pad values (the epilogue reads it 16 bytes at a time)'
    assert x.is_contiguous() and out.is_contiguous()
    d = Pointwise(mode, rows, B, H, W, cin, cout, cout_pad, ld_in, out.shape[-1], 1 if relu else 0)
    if x2 is not None:
        assert mode == _lib.PW_PLAIN and x2.is_contiguous() and x2.numel() // x2.shape[-1] == rows
        d.in2 = x2.data_ptr() + 4 * x2_ch_off
        d.ld_in2 = x2.shape[-1]
        d.k_split = k_split
    if residual is not None:
        assert mode == _lib.PW_PLAIN and residual.is_contiguous() and residual.numel() // residual.shape[-1] == rows
        d.residual = residual.data_ptr() + 4 * res_ch_off
        d.ld_res = residual.shape[-1]
    check(L.pcp_pointwise(ctypes.byref(d), _chan_ptr(x, in_ch_off), _p(packed), _p(bias), _chan_ptr(out, out_ch_off), _stream()),
          'pcp_pointwise')
    return out


# ---------------------------------------------------------------------------------------------------------------------
# a8 / a9
# ---------------------------------------------------------------------------------------------------------------------

def centerhead_decode(head, desc_kwargs):
    """head: (B, H, W, ld) NHWC.  Returns (boxes (B,K,7), scores (B,K), labels (B,K) i32, cell (B,K) i32, count (B,) i32)."""
    _need_cuda(head)
    L = _lib.load()
    B, H, W, ld = head.shape
    k = desc_kwargs['k']
    d = Decode()
    d.batch, d.h, d.w, d.ld = B, H, W, ld
    d.num_class = desc_kwargs.get('num_class', 1)
    d.ch_center, d.ch_z, d.ch_dim, d.ch_rot, d.ch_hm = (desc_kwargs[n] for n in ('ch_center', 'ch_z', 'ch_dim', 'ch_rot', 'ch_hm'))
    d.k = k
    d.stride = float(desc_kwargs['stride'])
    d.voxel_x, d.voxel_y = desc_kwargs['voxel_x'], desc_kwargs['voxel_y']
    d.min_x, d.min_y = desc_kwargs['min_x'], desc_kwargs['min_y']
    for i, v in enumerate(desc_kwargs['limit']):
        d.limit[i] = float(v)
    st = desc_kwargs.get('score_thresh', None)
    d.use_score_thresh = 0 if st is None else 1
    d.score_thresh = 0.0 if st is None else float(st)
    d.activated = 1 if desc_kwargs.get('activated', False) else 0
    dev = head.device
    boxes, scores, labels, cell, count = _zeros_views(dev, [((B, k, 7), torch.float32), ((B, k), torch.float32), ((B, k), torch.int32),
                                                            ((B, k), torch.int32), ((B,), torch.int32)])
    check(L.pcp_centerhead_decode(ctypes.byref(d), _p(head), ctypes.c_void_p(0), 0, _p(boxes), _p(scores), _p(labels), _p(cell),
                                  _p(count), _stream()), 'pcp_centerhead_decode')
    return boxes, scores, labels, cell, count


def nms_normal(boxes, scores, thresh, pre_max, post_max, n_dev=None, workspace=None):
    """nms_rotated with the axis-aligned IoU of the reference's nms_normal_gpu (heading ignored); same contract"""
    return nms_rotated(boxes, scores, thresh, pre_max, post_max, n_dev=n_dev, workspace=workspace, normal=True)


def nms_rotated(boxes, scores, thresh, pre_max, post_max, n_dev=None, workspace=None, normal=False):
    """boxes (n_max, 7) or (B, n_max, 7) float32; scores matching or None (= already sorted); n_dev (B,) int32 or None.
    Returns (keep (post_max,) | (B, post_max) int32 indices into each frame's input order, count (1,) | (B,) int32)."""
    _need_cuda(boxes, scores, n_dev)
    L = _lib.load()
    single = boxes.dim() == 2
    B = 1 if single else boxes.shape[0]
    n_max = boxes.shape[-2]
    assert boxes.dtype == torch.float32 and boxes.is_contiguous() and boxes.shape[-1] == 7
    need = L.pcp_nms_workspace_bytes(n_max, B)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=boxes.device)
    keep, cnt = _zeros_views(boxes.device, [((B, post_max), torch.int32), ((B,), torch.int32)])
    fn = L.pcp_nms_normal if normal else L.pcp_nms_rotated
    check(fn(_p(boxes), _p(scores), B, n_max, _p(n_dev), float(thresh), int(pre_max), int(post_max), _p(workspace),
             workspace.numel(), _p(keep), _p(cnt), _stream()), 'pcp_nms_normal' if normal else 'pcp_nms_rotated')
    return (keep[0], cnt) if single else (keep, cnt)


def gather_detections(heads, batch):
    """heads: list of dicts(boxes (B,k,7), scores (B,k), labels (B,k) int32 | None, keep (B,keep_max) int32, keep_count (B,) int32,
    class_map int32 device tensor | None).  One launch for all frames and heads; returns (boxes (B,M,7), scores (B,M), labels (B,M) int64
    1-based, count (B,) int32) with M = sum of keep_max; rows beyond count[b] are zero."""
    L = _lib.load()
    arr = (DetHead * len(heads))()
    out_max = 0
    for i, h in enumerate(heads):
        _need_cuda(h['boxes'], h['scores'], h['labels'], h['keep'], h['keep_count'], h.get('class_map'))
        assert h['boxes'].is_contiguous() and h['scores'].is_contiguous() and h['keep'].is_contiguous() and h['keep'].dtype == torch.int32
        assert h['boxes'].shape[0] == batch and h['keep'].shape[0] == batch
        arr[i].boxes, arr[i].scores, arr[i].labels = _p(h['boxes']), _p(h['scores']), _p(h['labels'])
        arr[i].keep, arr[i].keep_count, arr[i].class_map = _p(h['keep']), _p(h['keep_count']), _p(h.get('class_map'))
        arr[i].k, arr[i].keep_max = int(h['boxes'].shape[1]), int(h['keep'].shape[1])
        out_max += int(h['keep'].shape[1])
    dev = heads[0]['boxes'].device
    ob, os_, cnt = _zeros_views(dev, [((batch, out_max, 7), torch.float32), ((batch, out_max), torch.float32), ((batch,), torch.int32)])
    ol = torch.zeros((batch, out_max), dtype=torch.int64, device=dev)
    check(L.pcp_gather_detections(arr, len(heads), batch, out_max, _p(ob), _p(os_), _p(ol), _p(cnt), _stream()), 'pcp_gather_detections')
    return ob, os_, ol, cnt


def boxes_bev_pairwise(a, b, mode):
    _need_cuda(a, b)
    L = _lib.load()
    a = a[:, :7].contiguous()
    b = b[:, :7].contiguous()
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    check(L.pcp_boxes_bev_pairwise(_p(a), a.shape[0], _p(b), b.shape[0], mode, _p(out), _stream()), 'pcp_boxes_bev_pairwise')
    return out


# ---------------------------------------------------------------------------------------------------------------------
# a11 / a12 / a14
# ---------------------------------------------------------------------------------------------------------------------

def _dtc(t):
    return 1 if t.dtype == torch.bfloat16 else 0


def warp_nearest(src, dst, theta, channels, accumulate=False, src_ch_off=0, dst_ch_off=0):
    """src, dst: (H, W, ld) NHWC single-frame maps, float32 -- or bfloat16 on either side in the bf16 training loop (pcp_mp_warp_nearest);
    theta: 6 python floats (row-major 2x3)."""
    _need_cuda(src, dst)
    L = _lib.load()
    H, W, ld_s = src.shape
    th = (ctypes.c_float * 6)(*[float(v) for v in theta])
    if src.dtype == torch.float32 and dst.dtype == torch.float32:
        check(L.pcp_warp_nearest(_chan_ptr(src, src_ch_off), _chan_ptr(dst, dst_ch_off), H, W, channels, ld_s, dst.shape[2], th,
                                 1 if accumulate else 0, _stream()), 'pcp_warp_nearest')
        return
    for t in (src, dst):
        if t.dtype not in (torch.float32, torch.bfloat16):
            raise _lib.PcpError('pcp_mp_warp_nearest stores float32 or bfloat16 maps, got %s' % t.dtype)
    check(L.pcp_mp_warp_nearest(_chan_ptr(src, src_ch_off), _dtc(src), _chan_ptr(dst, dst_ch_off), _dtc(dst), H, W, channels, ld_s, dst.shape[2],
                                th, 1 if accumulate else 0, _stream()), 'pcp_mp_warp_nearest')


def warp_nearest_batch(jobs, channels, accumulate=False, theta_dev=None):
    """jobs: list of (src, dst, theta) with (H, W, ld) single-frame maps of ONE geometry; all the warps in one launch (pcp_warp_nearest_batch),
    pixel for pixel what warp_nearest gives job by job"""
    if not jobs:
        return
    L = _lib.load()
    _need_cuda(*[t for s_, d_, _t in jobs for t in (s_, d_)])
    _need_f32('pcp_warp_nearest_batch', *[t for s_, d_, _t in jobs for t in (s_, d_)])
    H, W, ld_s = jobs[0][0].shape
    ld_d = jobs[0][1].shape[2]
    for s_, d_, _t in jobs:
        assert tuple(s_.shape) == (H, W, ld_s) and tuple(d_.shape[:2]) == (H, W) and d_.shape[2] == ld_d
    n = len(jobs)
    src = (ctypes.c_void_p * n)(*[s_.data_ptr() for s_, _d, _t in jobs])
    dst = (ctypes.c_void_p * n)(*[d_.data_ptr() for _s, d_, _t in jobs])
    if theta_dev is not None:
        # the affines in device memory (graph mode: pcp_warp_nearest_batch_dev), job-major like `jobs`
        _need_cuda(theta_dev)
        assert theta_dev.dtype == torch.float32 and theta_dev.is_contiguous() and theta_dev.numel() == 6 * n
        check(L.pcp_warp_nearest_batch_dev(src, dst, _p(theta_dev), n, H, W, channels, ld_s, ld_d, 1 if accumulate else 0, _stream()),
              'pcp_warp_nearest_batch_dev')
        return
    th = (ctypes.c_float * (6 * n))(*[float(v) for _s, _d, t in jobs for v in t])
    check(L.pcp_warp_nearest_batch(src, dst, th, n, H, W, channels, ld_s, ld_d, 1 if accumulate else 0, _stream()), 'pcp_warp_nearest_batch')


def softmax_fuse(maps, weights, channels, out):
    """maps: list of (B, H, W, ld_map) tensors (same ld); weights: (B, H, W, ld_w) logits, column a <-> maps[a]."""
    _need_cuda(weights, out, *maps)
    _need_f32('pcp_softmax_fuse', weights, out, *maps)
    L = _lib.load()
    n = len(maps)
    arr = (ctypes.c_void_p * n)(*[m.data_ptr() for m in maps])
    pixels = weights.numel() // weights.shape[-1]
    check(L.pcp_softmax_fuse(arr, n, _p(weights), weights.shape[-1], pixels, channels, maps[0].shape[-1], out.shape[-1], _p(out),
                             _stream()), 'pcp_softmax_fuse')
    return out


def disco_weight_fuse(maps, w1, b1, w2, b2, w3, b3, channels, out, logits=None, live_index=None, live=None):
    """the DiscoNet pixel weightor + softmax over the maps + weighted sum as one launch (include/pcp_hip.h: pcp_disco_weight_fuse).
    maps: list of (..., ld_map) tensors sharing one pixel stride, maps[0] = ego; BN-folded float32 weights; out (..., ld_out).
    live_index / live: map a leaves the softmax when live_index[a] >= 0 and the device flag live[live_index[a]] is 0 (hipGraph mode)."""
    _need_cuda(w1, b1, w2, b2, w3, b3, out, logits, live, *maps)
    _need_f32('pcp_disco_weight_fuse', out, logits, *maps)
    L = _lib.load()
    n = len(maps)
    arr = (ctypes.c_void_p * n)(*[m.data_ptr() for m in maps])
    ld_map = maps[0].shape[-1]
    pixels = maps[0].numel() // ld_map
    for m in maps:
        assert m.is_contiguous() and m.shape[-1] == ld_map and m.numel() // ld_map == pixels
    for t in (w1, b1, w2, b2, w3, b3):
        assert t.dtype == torch.float32 and t.is_contiguous()
    assert tuple(w1.shape) == (64, 2 * channels) and tuple(w2.shape) == (16, 64) and w3.numel() == 16 and b3.numel() == 1
    if live_index is not None:
        assert logits is None and live is not None and len(live_index) == n
        idx = (ctypes.c_int32 * n)(*[int(v) for v in live_index])
        check(L.pcp_disco_weight_fuse_live(arr, n, channels, ld_map, pixels, _p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(out),
                                           out.shape[-1], idx, _p(live), _stream()), 'pcp_disco_weight_fuse_live')
        return out
    check(L.pcp_disco_weight_fuse(arr, n, channels, ld_map, pixels, _p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(out), out.shape[-1],
                                  _p(logits), logits.shape[-1] if logits is not None else 0, _stream()), 'pcp_disco_weight_fuse')
    return out


def voxelize_sort_pillar_rows(vox):
    """the points of every pillar in ascending row order inside vox's bucket order (include/pcp_hip.h: pcp_voxelize_sort_pillar_rows) --
    the training path calls it so that its per-point sums run in a reproducible order"""
    L = _lib.load()
    check(L.pcp_voxelize_sort_pillar_rows(ctypes.byref(vox.grid), _p(vox.workspace), vox.n, _stream()), 'pcp_voxelize_sort_pillar_rows')


def voxelize_row_order(vox):
    """spatially sorted visiting order of ALL rows of the cloud pcp_voxelize just bucketed (see include/pcp_hip.h)"""
    L = _lib.load()
    dev = vox.workspace.device
    order = torch.empty((max(vox.n, 1),), dtype=torch.int32, device=dev)
    cur = torch.empty((1,), dtype=torch.int32, device=dev)
    check(L.pcp_voxelize_row_order(ctypes.byref(vox.grid), _p(vox.workspace), vox.n, _p(order), _p(cur), _stream()), 'pcp_voxelize_row_order')
    return order


def hunter_point_head(bev, points, min_xy, pix_xy, w1, b1, w2, b2, wh, bh, channels, bev_ch_off=0, order=None, flow_thresh=None):
    """fused sample -> MLP -> heads.  bev: (B, H, W, ld) NHWC (channel window [bev_ch_off, +channels)).  Returns (pf (N, C), head (N, 8))
    or, with flow_thresh, (pf, head, dyn_mask): the dynamic-foreground correction (points[:, 1:4] += flow IN PLACE) and the re-sampling
    of the corrected rows run inside the same kernel.  order: optional int32 permutation of the rows (voxelize_row_order)."""
    _need_cuda(bev, points, w1, b1, w2, b2, wh, bh, order)
    L = _lib.load()
    B, H, W, ld_bev = bev.shape
    n, stride = points.shape
    pf = torch.empty((max(n, 1), channels), dtype=torch.float32, device=bev.device)
    head = torch.empty((max(n, 1), wh.shape[0]), dtype=torch.float32, device=bev.device)
    if order is not None or flow_thresh is not None:
        dyn = torch.zeros((max(n, 1),), dtype=torch.uint8, device=bev.device) if flow_thresh is not None else None
        check(L.pcp_hunter_point_head_ex(_chan_ptr(bev, bev_ch_off), B, H, W, channels, ld_bev, _p(points), n, stride, float(min_xy[0]),
                                         float(min_xy[1]), float(pix_xy[0]), float(pix_xy[1]), _p(w1), _p(b1), _p(w2), _p(b2), _p(wh),
                                         _p(bh), w1.shape[0], wh.shape[0], _p(pf), channels, _p(head), _p(order), ctypes.c_void_p(0),
                                         0 if flow_thresh is None else 1, 0.0 if flow_thresh is None else float(flow_thresh), _p(dyn),
                                         _stream()), 'pcp_hunter_point_head_ex')
        return (pf[:n], head[:n]) if flow_thresh is None else (pf[:n], head[:n], dyn[:n])
    check(L.pcp_hunter_point_head(_chan_ptr(bev, bev_ch_off), B, H, W, channels, ld_bev, _p(points), n, stride, float(min_xy[0]),
                                  float(min_xy[1]), float(pix_xy[0]), float(pix_xy[1]), _p(w1), _p(b1), _p(w2), _p(b2), _p(wh), _p(bh),
                                  w1.shape[0], wh.shape[0], _p(pf), channels, _p(head), _stream()), 'pcp_hunter_point_head')
    return pf[:n], head[:n]


def hunter_apply_flow(points, head, thresh):
    """in-place xyz += flow for predicted dynamic-foreground rows; returns the uint8 row mask."""
    _need_cuda(points, head)
    L = _lib.load()
    n, stride = points.shape
    mask = torch.zeros((max(n, 1),), dtype=torch.uint8, device=points.device)
    check(L.pcp_hunter_apply_flow(_p(points), n, stride, _p(head), head.shape[1], float(thresh), _p(mask), _stream()),
          'pcp_hunter_apply_flow')
    return mask[:n]


def column_ids(points, col):
    """sorted distinct values of an id column (torch.unique(points[:, col].long()) of bev_maker.py:153-156) as a numpy int64 array: one
    presence-mask launch + one 16-byte read-back.  Ids must lie in 0..63 (V2X-Sim has six agents); anything else raises -- there is no
    torch fallback on the product path."""
    import numpy as np
    _need_cuda(points)
    L = _lib.load()
    n, stride = points.shape
    out = torch.empty((2,), dtype=torch.int64, device=points.device)
    check(L.pcp_column_id_mask(_p(points), n, stride, col % stride, _p(out), _stream()), 'pcp_column_id_mask')
    mask, bad = [int(v) for v in out.cpu().tolist()]
    if bad:
        raise _lib.PcpError('column %d holds %d agent ids outside 0..63: the BEV maker supports up to 64 agents' % (col, bad))
    mask &= (1 << 64) - 1
    return np.asarray([i for i in range(64) if (mask >> i) & 1], dtype=np.int64)


def column_id_counts(points, col):
    """column_ids plus the number of rows per id: (ids int64 numpy, {id: rows}); one launch + one 528-byte read-back.  The counts size
    every per-agent selection of the forward on the host (select_transform_compact)."""
    import numpy as np
    _need_cuda(points)
    L = _lib.load()
    n, stride = points.shape
    out = torch.empty((66,), dtype=torch.int64, device=points.device)
    check(L.pcp_column_id_counts(_p(points), n, stride, col % stride, _p(out), _stream()), 'pcp_column_id_counts')
    vals = [int(v) for v in out.cpu().tolist()]
    mask, bad = vals[0] & ((1 << 64) - 1), vals[1]
    if bad:
        raise _lib.PcpError('column %d holds %d agent ids outside 0..63: the BEV maker supports up to 64 agents' % (col, bad))
    ids = [i for i in range(64) if (mask >> i) & 1]
    return np.asarray(ids, dtype=np.int64), {i: vals[2 + i] for i in ids}


def agent_frame_live(points, col, batch):
    """device flags (64 * batch int32): agent a is encoded for frame b by the reference's BEV maker (include/pcp_hip.h); no host read"""
    _need_cuda(points)
    L = _lib.load()
    n, stride = points.shape
    live = torch.empty((64 * batch,), dtype=torch.int32, device=points.device)
    check(L.pcp_agent_frame_live(_p(points), n, stride, col % stride, batch, _p(live), _stream()), 'pcp_agent_frame_live')
    return live


def zero_maps_unless(maps, flag_index, live):
    """maps: contiguous (M, ...) float32; map m is zero-filled unless live[flag_index[m]] != 0 (flag_index[m] < 0: left alone)"""
    _need_cuda(maps, live)
    _need_f32('pcp_zero_maps_unless', maps)
    L = _lib.load()
    assert maps.is_contiguous() and len(flag_index) == maps.shape[0]
    idx = (ctypes.c_int32 * len(flag_index))(*[int(v) for v in flag_index])
    check(L.pcp_zero_maps_unless(_p(maps), maps[0].numel(), maps.shape[0], idx, _p(live), _stream()), 'pcp_zero_maps_unless')


_STC_SCRATCH = {}


def select_transform_compact(points, agent_col, agents, poses, present, out_rows, out=None, vox_grid=None, vox_workspace=None,
                             slot_start=None, poses_dev=None, present_dev=None):
    """Stable compaction of the rows of `agents` (list of ids, <= 8 slots) into one stacked cloud (include/pcp_hip.h:
    pcp_select_transform_compact).  poses: (slots, B, 12) float32 numpy; present: (slots, B) uint8; out_rows: capacity of `out`
    (>= the rows that will be kept: the sum of the agents' row counts).  Returns out (out_rows, C): slot s's rows behind slot s-1's,
    frame index + s * B; rows past the total carry frame index -1.  vox_grid (+ vox_workspace sized for (vox_grid, out_rows)): also emit
    the pillariser's cell ids / histogram, to be followed by voxelize(..., cells_ready=True)."""
    import numpy as np
    _need_cuda(points, out, vox_workspace, slot_start)
    L = _lib.load()
    n, stride = points.shape
    assert points.dtype == torch.float32 and points.is_contiguous()
    S = len(agents)
    poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(S, -1, 12)
    B = poses.shape[1]
    pres = np.ascontiguousarray(present, dtype=np.uint8).reshape(S, B)
    ag = np.ascontiguousarray(np.asarray(agents, dtype=np.float32))
    if out is None:
        out = torch.empty((max(int(out_rows), 1), stride), dtype=torch.float32, device=points.device)
    assert out.is_contiguous() and out.dtype == torch.float32 and out.shape[1] == stride and out.shape[0] >= out_rows
    need = L.pcp_select_transform_compact_workspace_bytes(n, S)
    key = (points.device, current_stream_handle())      # one scratch per launch stream (overlapped makers)
    ws = _STC_SCRATCH.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=points.device)
        _STC_SCRATCH[key] = ws
    gref = ctypes.byref(vox_grid) if vox_grid is not None else None
    if poses_dev is not None:
        # pose table / presence flags in device memory (graph mode: pcp_select_transform_compact_dev); `poses` / `present` only give the shapes
        _need_cuda(poses_dev, present_dev)
        assert poses_dev.dtype == torch.float32 and poses_dev.is_contiguous() and poses_dev.numel() == S * B * 12
        assert present_dev.dtype == torch.uint8 and present_dev.is_contiguous() and present_dev.numel() == S * B
        check(L.pcp_select_transform_compact_dev(_p(points), n, stride, agent_col % stride, S, ag.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), B,
                                                 _p(poses_dev), _p(present_dev), _p(out), int(out_rows), _p(ws), ws.numel(), _p(slot_start), gref,
                                                 _p(vox_workspace), vox_workspace.numel() if vox_workspace is not None else 0, _stream()),
              'pcp_select_transform_compact_dev')
        return out
    check(L.pcp_select_transform_compact(_p(points), n, stride, agent_col % stride, S, ag.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), B,
                                         poses.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                         pres.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), _p(out), int(out_rows), _p(ws), ws.numel(),
                                         _p(slot_start), gref, _p(vox_workspace), vox_workspace.numel() if vox_workspace is not None else 0,
                                         _stream()), 'pcp_select_transform_compact')
    return out


def select_transform_points(points, agent_col, agent, poses, present, out=None, batch_offset=0):
    """poses: (B, 12) float32 numpy (row-major R|t); present: (B,) bool.  Returns a same-shape copy of `points` where rows of
    other agents / absent frames carry batch index -1 and the rows kept carry frame index + batch_offset (out: optional (N, C)
    destination, e.g. a slice of a buffer that stacks several agents for one pass of a shared chain)."""
    import numpy as np
    _need_cuda(points, out)
    L = _lib.load()
    n, stride = points.shape
    if out is None:
        out = torch.empty_like(points)
    assert out.shape == points.shape and out.is_contiguous() and out.dtype == torch.float32
    poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1)
    pres = np.ascontiguousarray(present, dtype=np.uint8)
    B = pres.shape[0]
    check(L.pcp_select_transform_points(_p(points), n, stride, agent_col % stride, float(agent), B,
                                        poses.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                        pres.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), _p(out), int(batch_offset), _stream()),
          'pcp_select_transform_points')
    return out


def softmax_fuse_raw(map_ptrs, weights, channels, ld_map, out, map_dtype=torch.float32):
    """like softmax_fuse but the maps are raw device addresses sharing one pixel stride (channel windows of one buffer); map_dtype
    bfloat16: the bf16 training loop's stacked maps (pcp_mp_softmax_fuse; ld_map counts elements)."""
    _need_cuda(weights, out)
    L = _lib.load()
    n = len(map_ptrs)
    arr = (ctypes.c_void_p * n)(*map_ptrs)
    pixels = weights.numel() // weights.shape[-1]
    if map_dtype == torch.bfloat16:
        check(L.pcp_mp_softmax_fuse(arr, 1, n, _p(weights), weights.shape[-1], pixels, channels, ld_map, out.shape[-1], _p(out), _stream()),
              'pcp_mp_softmax_fuse')
        return out
    check(L.pcp_softmax_fuse(arr, n, _p(weights), weights.shape[-1], pixels, channels, ld_map, out.shape[-1], _p(out),
                             _stream()), 'pcp_softmax_fuse')
    return out


def bev_sample_bilinear(bev, points, min_xy, pix_xy, out=None, row_mask=None, channels=None, bev_ch_off=0):
    _need_cuda(bev, points, out, row_mask)
    L = _lib.load()
    B, H, W, ld_bev = bev.shape
    C = ld_bev if channels is None else channels
    n, stride = points.shape
    if out is None:
        out = torch.empty((n, C), dtype=torch.float32, device=bev.device)
    check(L.pcp_bev_sample_bilinear(_chan_ptr(bev, bev_ch_off), B, H, W, C, ld_bev, _p(points), n, stride, float(min_xy[0]), float(min_xy[1]),
                                    float(pix_xy[0]), float(pix_xy[1]), _p(row_mask), _p(out), out.shape[1], _stream()),
          'pcp_bev_sample_bilinear')
    return out


def bev_scatter_mean(points, feat, batch, h, w, min_xy, pix_xy, out=None, out_ch_off=0, workspace=None):
    _need_cuda(points, feat, out)
    L = _lib.load()
    n, stride = points.shape
    C = feat.shape[1]
    need = L.pcp_bev_scatter_mean_workspace_bytes(batch, h, w, n)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=feat.device)
    if out is None:
        out = torch.empty((batch, h, w, C), dtype=torch.float32, device=feat.device)
    check(L.pcp_bev_scatter_mean(_p(points), n, stride, _p(feat), feat.shape[1], C, batch, h, w, float(min_xy[0]), float(min_xy[1]),
                                 float(pix_xy[0]), float(pix_xy[1]), _p(workspace), workspace.numel(), _chan_ptr(out, out_ch_off),
                                 out.shape[-1], _stream()), 'pcp_bev_scatter_mean')
    return out


# ---------------------------------------------------------------------------------------------------------------------
# exchange producer / consumer
# ---------------------------------------------------------------------------------------------------------------------

def points_in_boxes(points, boxes):
    """points (B, M, 3+), boxes (B, T, 7+) float32 CUDA -> (B, M) int32 (first containing box, -1 = none)"""
    _need_cuda(points, boxes)
    L = _lib.load()
    points = points.float().contiguous()
    boxes = boxes.float().contiguous()
    B, M, ps = points.shape
    T, bs = boxes.shape[1], boxes.shape[2]
    out = torch.full((B, M), -1, dtype=torch.int32, device=points.device)
    if M and T:
        check(L.pcp_points_in_boxes(_p(boxes), B, T, bs, _p(points), M, ps, _p(out), _stream()), 'pcp_points_in_boxes')
    return out


def hunter_foreground_rows(points, head, thresh_bg=0.3, sync=True):
    """points (N, 1+F) with the frame index in column 0, head (N, >=6) = [cls logits(3), flow(3), ...].
    Returns (rows (n_send, F+6), row_batch (n_send,) int32) in the original row order; one host sync for n_send.
    sync=False: no host sync -- returns the full-capacity buffers and the device count: (rows (N, F+6), row_batch (N,), count (1,) int32)."""
    _need_cuda(points, head)
    L = _lib.load()
    n, stride = points.shape
    dev = points.device
    ws = torch.empty(L.pcp_hunter_foreground_workspace_bytes(n), dtype=torch.uint8, device=dev)
    rows = torch.empty((max(n, 1), stride - 1 + 6), dtype=torch.float32, device=dev)
    rb = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    check(L.pcp_hunter_foreground_rows(_p(points), n, stride, _p(head), head.shape[1], float(thresh_bg), _p(ws), ws.numel(), _p(rows), _p(rb),
                                       _p(cnt), _stream()), 'pcp_hunter_foreground_rows')
    if not sync:
        return rows, rb, cnt
    k = int(cnt.item())
    return rows[:k], rb[:k]


def modar_ingest_batched(det, foreground, fg_group, fg_count, poses, max_sweep_idx, frame_of_group, out=None):
    """Device-driven ingestion of every (frame, remote agent) group at once (include/pcp_hip.h: pcp_modar_ingest_batched).
    det = (boxes (G, M, 7), scores (G, M), labels (G, M) int64, count (G,) int32) as ops.gather_detections returns them;
    foreground (cap, cols), fg_group (cap,) int32, fg_count (1,) int32 as hunter_foreground_rows(sync=False) returns them (or None);
    poses (G, 12) float64 / max_sweep_idx (G,) float32 / frame_of_group (G,) int32 DEVICE tensors.  Returns (G * M, 14) rows."""
    boxes, scores, labels, count = det
    _need_cuda(boxes, scores, labels, count, foreground, fg_group, fg_count, poses, max_sweep_idx, frame_of_group, out)
    L = _lib.load()
    G, M = int(boxes.shape[0]), int(boxes.shape[1])
    dev = boxes.device
    assert boxes.is_contiguous() and scores.is_contiguous() and labels.is_contiguous() and labels.dtype == torch.int64
    assert poses.dtype == torch.float64 and poses.is_contiguous() and tuple(poses.shape) == (G, 12)
    assert max_sweep_idx.dtype == torch.float32 and frame_of_group.dtype == torch.int32
    cap, cols = (0, 0) if foreground is None else (int(foreground.shape[0]), int(foreground.shape[1]))
    if foreground is None:
        fg_count = torch.zeros(1, dtype=torch.int32, device=dev)
    need = L.pcp_modar_ingest_batched_workspace_bytes(G, cap)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    if out is None:
        out = torch.empty((G * M, 14), dtype=torch.float32, device=dev)
    assert out.is_contiguous() and tuple(out.shape) == (G * M, 14)
    check(L.pcp_modar_ingest_batched(_p(boxes), _p(scores), _p(labels), _p(count), G, M, _p(foreground), cols, _p(fg_group), _p(fg_count), cap,
                                     _p(poses), _p(max_sweep_idx), _p(frame_of_group), _p(ws), ws.numel(), _p(out), _stream()),
          'pcp_modar_ingest_batched')
    return out


def modar_ingest(modar, foreground, target_se3_lidar, max_sweep_idx):
    """modar (n, 9) CUDA [box7, score, label]; foreground (m, cols) CUDA or None; target_se3_lidar (4, 4) float64 numpy.
    Returns (n, 13) CUDA float32 rows in the ego frame (the columns config 3's VFE reads)."""
    import numpy as np
    _need_cuda(modar, foreground)
    L = _lib.load()
    modar = modar.float().contiguous()
    n = modar.shape[0]
    rows = torch.empty((n, 13), dtype=torch.float32, device=modar.device)
    T = np.ascontiguousarray(np.asarray(target_se3_lidar, dtype=np.float64)[:3, :4]).reshape(-1)
    m, cols = (0, 0)
    if foreground is not None and foreground.shape[0] > 0:
        foreground = foreground.float().contiguous()
        m, cols = foreground.shape
    check(L.pcp_modar_ingest(_p(modar), n, _p(foreground) if m else ctypes.c_void_p(0), m, cols,
                             T.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), float(max_sweep_idx), _p(rows), _stream()), 'pcp_modar_ingest')
    return rows


# ---------------------------------------------------------------------------------------------------------------------
# AnchorHeadSingle decode + candidate selection
# ---------------------------------------------------------------------------------------------------------------------

def anchor_decode(head, anchors, desc):
    """head (B, H, W, ld) NHWC; anchors (N, 7) CUDA; desc: lib.Anchor.  Returns boxes (B, N, 7), cls logits (B, N, ncls), score keys
    (B, N) int32 storage of uint32 bits, labels (B, N) int32."""
    _need_cuda(head, anchors)
    L = _lib.load()
    B = head.shape[0]
    N = anchors.shape[0]
    dev = head.device
    boxes = torch.empty((B, N, 7), dtype=torch.float32, device=dev)
    cls = torch.empty((B, N, desc.num_class), dtype=torch.float32, device=dev)
    keys = torch.empty((B, N), dtype=torch.int32, device=dev)
    labels = torch.empty((B, N), dtype=torch.int32, device=dev)
    check(L.pcp_anchor_decode(ctypes.byref(desc), _p(head), _p(anchors), _p(boxes), _p(cls), _p(keys), _p(labels), _stream()),
          'pcp_anchor_decode')
    return boxes, cls, keys, labels


def topk_boxes(keys, labels, boxes, k):
    """Per frame the k best candidates (descending score, ties to the lower index), gathered.  Returns boxes (B, k, 7), scores (B, k),
    labels (B, k) int32, index (B, k) int32, count (B,) int32."""
    _need_cuda(keys, labels, boxes)
    L = _lib.load()
    B, N = keys.shape
    dev = keys.device
    ob = torch.zeros((B, k, 7), dtype=torch.float32, device=dev)
    os_ = torch.zeros((B, k), dtype=torch.float32, device=dev)
    ol = torch.zeros((B, k), dtype=torch.int32, device=dev)
    oi = torch.zeros((B, k), dtype=torch.int32, device=dev)
    cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    check(L.pcp_topk_boxes(_p(keys), _p(labels), _p(boxes), B, N, k, _p(ob), _p(os_), _p(ol), _p(oi), _p(cnt), _stream()), 'pcp_topk_boxes')
    return ob, os_, ol, oi, cnt
