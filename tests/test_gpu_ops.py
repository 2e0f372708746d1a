"""Parity tests proper: every HIP kernel, called through the C ABI, against the oracle on the same seeded inputs.
Integer / index outputs must be bit exact; floating point within the tolerance written next to each assert
(north_star: 1e-3 on boxes / scores).  Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import arch_of, load_golden
from oracle import bev as obev
from oracle import nms as onms
from oracle import pillars as opil
from pcp_amd import synth

pytestmark = pytest.mark.gpu

PC_RANGE = [-51.2, -51.2, -8.0, 51.2, 51.2, 0.0]
VOXEL = [0.2, 0.2, 8.0]
GRID = [512, 512, 1]


def dev():
    assert torch.cuda.is_available(), 'gpu-marked tests need the MI355X'
    return torch.device('cuda:0')


def _ops():
    from pcp_amd import ops
    return ops


def _rand(seed, shape, lo=-1.0, hi=1.0):
    n = int(np.prod(shape))
    return synth.uniform(seed, 5, n, lo, hi).reshape(shape)


# ---------------------------------------------------------------------------------------------------------------------
# a1 / a4
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('case', ['car_1x60k', 'early_6x60k', 'batch3_ragged', 'empty', 'all_masked', 'edges'])
def test_voxelize_bit_exact(case):
    ops = _ops()
    B = 1
    if case == 'car_1x60k':
        pts = synth.collate([synth.agent_cloud(0, 60000, 'car')])
    elif case == 'early_6x60k':
        pts = synth.collate([np.concatenate([synth.agent_cloud(a, 60000, 'early') for a in range(6)], 0)])
    elif case == 'batch3_ragged':
        B = 3
        pts = synth.collate([synth.agent_cloud(1, 5000, 'car'), synth.agent_cloud(2, 17, 'car'), synth.agent_cloud(3, 30001, 'car', dist='ring')])
    elif case == 'empty':
        pts = np.zeros((0, 8), np.float32)
    elif case == 'all_masked':
        pts = synth.collate([synth.agent_cloud(0, 1000, 'car')])
        pts[:, 1] += 500.0
    else:  # edges: exact cell boundaries, range ends, NaN / inf rows
        xs = np.array([-51.2, -51.200001, 51.2, 51.199997, 0.0, 0.2, 0.19999999, -0.2, 0.6000000238, 0.6, np.nan, np.inf, -np.inf, 10.0],
                      np.float32)
        pts = np.zeros((xs.shape[0] * 2, 8), np.float32)
        pts[:xs.shape[0], 1] = xs
        pts[:xs.shape[0], 2] = 1.0
        pts[xs.shape[0]:, 1] = 1.0
        pts[xs.shape[0]:, 2] = xs
        pts[:, 3] = -1.0
    ref = opil.voxelize(pts, 5, PC_RANGE, VOXEL, GRID) if pts.shape[0] else None
    g = ops.make_grid(PC_RANGE, VOXEL, GRID, B)
    res = ops.voxelize(torch.from_numpy(pts).to(dev()), g)
    torch.cuda.synchronize()
    P, Nv = [int(v) for v in res.counters[:2].cpu()]
    if ref is None:
        assert P == 0 and Nv == 0
        return
    assert P == ref['unq'].shape[0] and Nv == ref['inv'].shape[0]
    assert np.array_equal(res.voxel_coords[:P].cpu().numpy(), ref['coords'])
    assert np.array_equal(res.unq_inv[:Nv].cpu().numpy(), ref['inv'])
    assert np.array_equal(res.unq_cnt[:P].cpu().numpy().astype(np.int64), ref['cnt'])


def test_voxelize_is_idempotent_and_sorted_at_full_size():
    """size-independent properties at BASELINE full size (6 x 60k points, B = 4 frames)."""
    ops = _ops()
    clouds = [np.concatenate([synth.agent_cloud(a + 10 * b, 60000, 'early') for a in range(6)], 0) for b in range(4)]
    pts = torch.from_numpy(synth.collate(clouds)).to(dev())
    g = ops.make_grid(PC_RANGE, VOXEL, GRID, 4)
    r1 = ops.voxelize(pts, g)
    r2 = ops.voxelize(pts, g)
    torch.cuda.synchronize()
    P, Nv = [int(v) for v in r1.counters[:2].cpu()]
    assert torch.equal(r1.voxel_coords[:P], r2.voxel_coords[:P]) and torch.equal(r1.unq_inv[:Nv], r2.unq_inv[:Nv])
    vc = r1.voxel_coords[:P].long()
    merged = vc[:, 0] * 262144 + vc[:, 3] * 512 + vc[:, 2]
    assert bool((merged[1:] > merged[:-1]).all())                 # strictly ascending == torch.unique order
    assert int(r1.unq_cnt[:P].sum()) == Nv                        # counts partition the kept points
    assert int(r1.unq_inv[:Nv].max()) == P - 1
    assert torch.equal(torch.bincount(r1.unq_inv[:Nv], minlength=P).int(), r1.unq_cnt[:P])


# ---------------------------------------------------------------------------------------------------------------------
# a2 / a3 / a5
# ---------------------------------------------------------------------------------------------------------------------
def _vfe_weights(num_raw, seed=11):
    shapes = {'vfe.pfn_layers.0.linear.weight': (32, num_raw + 6), 'vfe.pfn_layers.1.linear.weight': (64, 64)}
    for li, c in ((0, 32), (1, 64)):
        for leaf in ('weight', 'bias', 'running_mean', 'running_var'):
            shapes['vfe.pfn_layers.%d.norm.%s' % (li, leaf)] = (c,)
    return synth.fill_state_dict(shapes, seed=seed)


@pytest.mark.parametrize('layout,num_raw,n', [('car', 5, 60000), ('lately', 11, 20000), ('car', 5, 300)])
def test_pfn_scatter_matches_oracle(layout, num_raw, n):
    ops = _ops()
    from pcp_amd import pack
    B = 2
    pts = synth.collate([synth.agent_cloud(3, n, layout), synth.agent_cloud(4, n // 2, layout, dist='ring')])
    st = _vfe_weights(num_raw)
    arch = dict(num_raw=num_raw, pc_range=PC_RANGE, voxel_size=VOXEL, grid_size=GRID, vfe_filters=[64, 64])
    ref = opil.vfe_forward(pts, st, arch)
    t = lambda k: torch.from_numpy(st[k])
    w0, b0 = pack.fold_bn(t('vfe.pfn_layers.0.linear.weight'), t('vfe.pfn_layers.0.norm.weight'), t('vfe.pfn_layers.0.norm.bias'),
                          t('vfe.pfn_layers.0.norm.running_mean'), t('vfe.pfn_layers.0.norm.running_var'), 1e-3)
    w1, b1 = pack.fold_bn(t('vfe.pfn_layers.1.linear.weight'), t('vfe.pfn_layers.1.norm.weight'), t('vfe.pfn_layers.1.norm.bias'),
                          t('vfe.pfn_layers.1.norm.running_mean'), t('vfe.pfn_layers.1.norm.running_var'), 1e-3)
    d = dev()
    pd = torch.from_numpy(pts).to(d)
    g = ops.make_grid(PC_RANGE, VOXEL, GRID, B)
    vox = ops.voxelize(pd, g)
    P = int(vox.counters[0])
    canvas = torch.zeros((B, 512, 512, 64), device=d)
    pf = torch.zeros((max(P, 1), 64), device=d)
    ops.pfn_scatter(pd, vox, num_raw, w0.to(d).contiguous(), b0.to(d), w1.to(d).contiguous(), b1.to(d), canvas=canvas, pillar_features=pf)
    torch.cuda.synchronize()
    assert P == ref['pillar_features'].shape[0]
    # fp32 with a different summation order and folded BN: 2e-5 absolute on O(1) features
    np.testing.assert_allclose(pf.cpu().numpy(), ref['pillar_features'], rtol=1e-4, atol=2e-5)
    want = torch.from_numpy(ref['spatial_features']).permute(0, 2, 3, 1)
    np.testing.assert_allclose(canvas.cpu().numpy(), want.numpy(), rtol=1e-4, atol=2e-5)
    # the canvas must hold exactly the pillar rows, bit for bit
    vc = vox.voxel_coords[:P].long()
    assert torch.equal(canvas[vc[:, 0], vc[:, 2], vc[:, 3]], pf[:P])
    # clear-by-pillar-list restores an all-zero canvas (the per-frame reset used by the pipeline)
    ops.canvas_clear(vox, canvas)
    torch.cuda.synchronize()
    assert float(canvas.abs().max()) == 0.0
    # determinism (fixed-point means, order-independent maxima)
    pf2 = torch.zeros_like(pf)
    vox2 = ops.voxelize(pd, g)
    ops.pfn_scatter(pd, vox2, num_raw, w0.to(d).contiguous(), b0.to(d), w1.to(d).contiguous(), b1.to(d), canvas=None, pillar_features=pf2)
    torch.cuda.synchronize()
    assert torch.equal(pf, pf2)


# ---------------------------------------------------------------------------------------------------------------------
# round 5: rows in pillar order + the wave-autonomous PFN (pcp_pillarise_rows, pcp_pfn_rows)
# ---------------------------------------------------------------------------------------------------------------------
def _folded_vfe(num_raw, seed=11):
    from pcp_amd import pack
    st = _vfe_weights(num_raw, seed)
    t = lambda k: torch.from_numpy(st[k])
    w0, b0 = pack.fold_bn(t('vfe.pfn_layers.0.linear.weight'), t('vfe.pfn_layers.0.norm.weight'), t('vfe.pfn_layers.0.norm.bias'),
                          t('vfe.pfn_layers.0.norm.running_mean'), t('vfe.pfn_layers.0.norm.running_var'), 1e-3)
    w1, b1 = pack.fold_bn(t('vfe.pfn_layers.1.linear.weight'), t('vfe.pfn_layers.1.norm.weight'), t('vfe.pfn_layers.1.norm.bias'),
                          t('vfe.pfn_layers.1.norm.running_mean'), t('vfe.pfn_layers.1.norm.running_var'), 1e-3)
    d = dev()
    return st, (w0.to(d).contiguous(), b0.to(d).contiguous(), w1.to(d).contiguous(), b1.to(d).contiguous())


@pytest.mark.parametrize('case', ['car_1x60k', 'early_6x60k', 'batch3_ragged', 'empty', 'all_masked', 'edges'])
def test_pillarise_rows_bit_exact(case):
    """the round-5 pillariser against the oracle (same cases as pcp_voxelize) + the records it leaves: row k of pillar r holds the raw
    columns of one of r's points, r itself, its cell and its canvas row; every kept point exactly once."""
    ops = _ops()
    B = 1
    if case == 'car_1x60k':
        pts = synth.collate([synth.agent_cloud(0, 60000, 'car')])
    elif case == 'early_6x60k':
        pts = synth.collate([np.concatenate([synth.agent_cloud(a, 60000, 'early') for a in range(6)], 0)])
    elif case == 'batch3_ragged':
        B = 3
        pts = synth.collate([synth.agent_cloud(1, 5000, 'car'), synth.agent_cloud(2, 17, 'car'), synth.agent_cloud(3, 30001, 'car', dist='ring')])
    elif case == 'empty':
        pts = np.zeros((0, 8), np.float32)
    elif case == 'all_masked':
        pts = synth.collate([synth.agent_cloud(0, 1000, 'car')])
        pts[:, 1] += 500.0
    else:
        xs = np.array([-51.2, -51.200001, 51.2, 51.199997, 0.0, 0.2, 0.19999999, -0.2, 0.6000000238, 0.6, np.nan, np.inf, -np.inf, 10.0],
                      np.float32)
        pts = np.zeros((xs.shape[0] * 2, 8), np.float32)
        pts[:xs.shape[0], 1] = xs
        pts[:xs.shape[0], 2] = 1.0
        pts[xs.shape[0]:, 1] = 1.0
        pts[xs.shape[0]:, 2] = xs
        pts[:, 3] = -1.0
    ref = opil.voxelize(pts, 5, PC_RANGE, VOXEL, GRID) if pts.shape[0] else None
    g = ops.make_grid(PC_RANGE, VOXEL, GRID, B)
    res = ops.pillarise_rows(torch.from_numpy(pts).to(dev()), g, 5, want_inverse=True, want_counts=True, want_coords=True)
    torch.cuda.synchronize()
    P, Nv = [int(v) for v in res.counters[:2].cpu()]
    if ref is None:
        assert P == 0 and Nv == 0
        return
    assert P == ref['unq'].shape[0] and Nv == ref['inv'].shape[0]
    assert np.array_equal(res.voxel_coords[:P].cpu().numpy(), ref['coords'])
    assert np.array_equal(res.unq_inv[:Nv].cpu().numpy(), ref['inv'])
    assert np.array_equal(res.unq_cnt[:P].cpu().numpy().astype(np.int64), ref['cnt'])
    if Nv == 0:
        return
    # the records: decode the workspace with the layout the header documents (front part = pcp_voxelize's, then n rows of 8 floats)
    L = __import__('pcp_amd.lib', fromlist=['lib']).load()
    import ctypes
    front = L.pcp_voxelize_workspace_bytes(ctypes.byref(g), max(pts.shape[0], 1))
    rows = res.workspace[front:front + max(pts.shape[0], 1) * 32].view(torch.float32).view(-1, 8)[:Nv].cpu()
    tagged = rows[:, 5].view(torch.int32).numpy()
    rank = tagged & 0x7fffffff
    # the sign bit marks the records of crowded pillars (>= PCP_PFN_CROWD = 192 records: pfn_crowd_run runs those), and only those
    assert np.array_equal(tagged < 0, ref['cnt'][rank] >= 192)
    # slot order: the multi-point pillars ascending (runs of cnt records), then the single-point pillars ascending
    multi, single = np.nonzero(ref['cnt'] > 1)[0], np.nonzero(ref['cnt'] == 1)[0]
    assert np.array_equal(rank, np.concatenate([np.repeat(multi, ref['cnt'][multi]), single]))
    assert [int(v) for v in res.counters[2:].cpu()] == [int(ref['cnt'][multi].sum()), int(single.shape[0])]
    cxcy = rows[:, 6].view(torch.int32).numpy()
    crow = rows[:, 7].view(torch.int32).numpy()                                        # row of the (B, ny, nx, 64) canvas
    vc = ref['coords'][rank]
    assert np.array_equal(np.stack([cxcy & 0xffff, cxcy >> 16], 1), vc[:, 2:]) and np.array_equal(crow, (vc[:, 0] * 512 + vc[:, 2]) * 512 + vc[:, 3])
    # every kept input row appears exactly once (compare as multisets of the 5 raw columns + pillar)
    keep = np.nonzero(ref['keep'])[0]
    src = pts[:, 1:6]
    assert keep.shape[0] == Nv
    a = np.concatenate([src[keep].view(np.int32), ref['inv'][:, None].astype(np.int32)], 1)
    b = np.concatenate([rows[:, :5].numpy().view(np.int32), rank[:, None]], 1)
    assert np.array_equal(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])])


def test_pillarise_rows_with_bucket_order_keeps_the_plain_pillar_order():
    """PCP_ROWS_BUCKET_ORDER (HunterJr's configs): records and bucket order in pillar-rank order, single-point pillars NOT set apart (the
    point head walks that order for the locality of its BEV gathers); the row order derived from it is a permutation of all rows"""
    ops = _ops()
    pts = synth.collate([synth.agent_cloud(1, 5000, 'car'), synth.agent_cloud(3, 30001, 'car', dist='ring')])
    ref = opil.voxelize(pts, 5, PC_RANGE, VOXEL, GRID)
    g = ops.make_grid(PC_RANGE, VOXEL, GRID, 2)
    res = ops.pillarise_rows(torch.from_numpy(pts).to(dev()), g, 5, bucket_order=True)
    torch.cuda.synchronize()
    P, Nv, Nm, S = [int(v) for v in res.counters.cpu()]
    assert (P, Nv) == (ref['unq'].shape[0], ref['inv'].shape[0]) and (Nm, S) == (0, 0) and res.has_bucket_order
    import ctypes
    L = __import__('pcp_amd.lib', fromlist=['lib']).load()
    front = L.pcp_voxelize_workspace_bytes(ctypes.byref(g), pts.shape[0])
    rows = res.workspace[front:front + pts.shape[0] * 32].view(torch.float32).view(-1, 8)[:Nv].cpu()
    tagged = rows[:, 5].view(torch.int32).numpy()                                      # sign bit: records of crowded pillars (>= 192 records)
    assert np.array_equal(tagged & 0x7fffffff, np.repeat(np.arange(P), ref['cnt']))
    assert np.array_equal(tagged < 0, np.repeat(ref['cnt'] >= 192, ref['cnt'])) and bool((tagged < 0).any())
    order = ops.voxelize_row_order(res).cpu().numpy()
    assert np.array_equal(np.sort(order), np.arange(pts.shape[0]))
    kept = np.nonzero(ref['keep'])[0]
    assert np.array_equal(np.sort(order[:Nv]), kept) and np.array_equal(ref['inv'][np.searchsorted(kept, order[:Nv])], np.repeat(np.arange(P), ref['cnt']))


def _crowded_cloud(num_cols=8):
    """a cloud with pillars of every size: one cell with 700 points, one with 33, runs of 1 .. 17, isolated points, an empty frame in the
    middle of the batch and points in the very first / very last cell of the grid"""
    rs = np.random.RandomState(5)
    rows = []
    def cell_points(b, cx, cy, k):
        q = np.zeros((k, num_cols), np.float32)
        q[:, 0] = b
        q[:, 1] = -51.2 + 0.2 * cx + rs.uniform(0.01, 0.19, k)
        q[:, 2] = -51.2 + 0.2 * cy + rs.uniform(0.01, 0.19, k)
        q[:, 3] = rs.uniform(-8, 0, k)
        q[:, 4:] = rs.uniform(0, 1, (k, num_cols - 4))
        rows.append(q)
    cell_points(0, 0, 0, 3)
    cell_points(0, 100, 7, 700)
    cell_points(0, 100, 8, 33)
    for k in range(1, 18):
        cell_points(0, 200 + k, 300, k)
    for k in range(40):
        cell_points(2, 3 * k, 5 * k + 1, 1)
    cell_points(2, 511, 511, 2)
    cell_points(3, 250, 250, 64)
    cell_points(3, 250, 251, 31)
    cell_points(3, 250, 252, 30)
    cell_points(3, 250, 253, 29)
    pts = np.concatenate(rows, 0)
    return pts[rs.permutation(pts.shape[0])]


@pytest.mark.parametrize('layout,num_raw,n', [('car', 5, 60000), ('lately', 11, 20000), ('car', 5, 300), ('crowded', 5, 0), ('crowded', 11, 0)])
def test_pfn_rows_matches_oracle(layout, num_raw, n):
    ops = _ops()
    B = 2
    if layout == 'crowded':
        B = 4
        pts = _crowded_cloud(8 if num_raw == 5 else 14)
    else:
        pts = synth.collate([synth.agent_cloud(3, n, layout), synth.agent_cloud(4, n // 2, layout, dist='ring')])
    st, (w0, b0, w1, b1) = _folded_vfe(num_raw)
    arch = dict(num_raw=num_raw, pc_range=PC_RANGE, voxel_size=VOXEL, grid_size=GRID, vfe_filters=[64, 64])
    ref = opil.vfe_forward(pts, st, arch)
    d = dev()
    pd = torch.from_numpy(pts).to(d)
    g = ops.make_grid(PC_RANGE, VOXEL, GRID, B)
    vox = ops.pillarise_rows(pd, g, num_raw, want_coords=True)
    P = int(vox.counters[0])
    canvas = torch.full((B, 512, 512, 64), float('nan'), device=d)              # the kernel must write EVERY row: pillar rows and zero rows
    pf = torch.full((max(P, 1), 64), float('nan'), device=d)
    ops.pfn_rows(vox, w0, b0, w1, b1, canvas=canvas, pillar_features=pf)
    torch.cuda.synchronize()
    assert P == ref['pillar_features'].shape[0]
    # fp32 with a different summation order and folded BN: 2e-5 absolute on O(1) features
    np.testing.assert_allclose(pf.cpu().numpy(), ref['pillar_features'], rtol=1e-4, atol=2e-5)
    want = torch.from_numpy(ref['spatial_features']).permute(0, 2, 3, 1)
    np.testing.assert_allclose(canvas.cpu().numpy(), want.numpy(), rtol=1e-4, atol=2e-5)
    vc = vox.voxel_coords[:P].long()
    assert torch.equal(canvas[vc[:, 0], vc[:, 2], vc[:, 3]], pf[:P])            # the canvas holds exactly the pillar rows, bit for bit
    assert int((canvas != 0).any(-1).sum()) <= P                                # and zeros everywhere else
    # against the rounds 1-4 kernel on the same cloud (same means, same features; only the summation order inside the products differs)
    vox_old = ops.voxelize(pd, g)
    pf_old = torch.zeros_like(pf)
    ops.pfn_scatter(pd, vox_old, num_raw, w0, b0, w1, b1, canvas=None, pillar_features=pf_old)
    torch.cuda.synchronize()
    assert float((pf - pf_old).abs().max()) < 1e-5
    # determinism (fixed-point means, order-independent maxima, fixed product order): a second pillarisation has another arrival order
    pf2 = torch.zeros_like(pf)
    vox2 = ops.pillarise_rows(pd, g, num_raw)
    ops.pfn_rows(vox2, w0, b0, w1, b1, canvas=None, pillar_features=pf2)
    torch.cuda.synchronize()
    assert torch.equal(pf, pf2)


@pytest.mark.parametrize('num_raw,bucket_order', [(5, False), (5, True), (11, False)])
def test_pfn_rows_crowded_pillars_on_their_own_workgroups_give_the_same_bits(num_raw, bucket_order, lib_option):
    """pillars of at least PCP_PFN_CROWD records (default 192) are listed by the pillariser, passed over by the wave tiles of k_pfn_rows and
    run by the front workgroups of the same grid, a workgroup per pillar: the same operations per point, exact sums, order-free maxima -- bit for bit what the wave tiles
    compute when nothing is listed (option pfn_crowd = 0), at every threshold, with and without the bucket order, next to the reference oracle"""
    ops = _ops()
    rs = np.random.RandomState(11)
    ncol = 8 if num_raw == 5 else 14
    base = _crowded_cloud(ncol)
    extra = []

    def cell_points(b, cx, cy, k):
        q = np.zeros((k, ncol), np.float32)
        q[:, 0] = b
        q[:, 1] = -51.2 + 0.2 * cx + rs.uniform(0.01, 0.19, k)
        q[:, 2] = -51.2 + 0.2 * cy + rs.uniform(0.01, 0.19, k)
        q[:, 3] = rs.uniform(-8, 0, k)
        q[:, 4:] = rs.uniform(0, 1, (k, ncol - 4))
        extra.append(q)
    for i, k in enumerate((5000, 1024, 257, 256, 255, 96, 65, 64, 63)):      # around every threshold used below, neighbours in the slot order
        cell_points(1, 40 + i, 40, k)
    cell_points(1, 0, 0, 300)                                                  # the very first pillar of a frame
    cell_points(3, 511, 511, 400)                                              # and the very last one of the cloud
    ring = synth.collate([synth.agent_cloud(5, 20000, 'car' if num_raw == 5 else 'lately', dist='ring')])
    ring[:, 0] = 2
    pts = np.concatenate([base] + extra + [ring[:, :ncol]], 0)
    pts = pts[rs.permutation(pts.shape[0])]
    st, (w0, b0, w1, b1) = _folded_vfe(num_raw)
    d = dev()
    pd = torch.from_numpy(pts).to(d)
    g = ops.make_grid(PC_RANGE, VOXEL, GRID, 4)
    outs = {}
    for thr in ('0', '64', '192', '100000'):
        lib_option('pfn_crowd', int(thr))
        vox = ops.pillarise_rows(pd, g, num_raw, want_coords=True, bucket_order=bucket_order)
        P = int(vox.counters[0])
        canvas = torch.full((4, 512, 512, 64), float('nan'), device=d)
        pf = torch.full((P, 64), float('nan'), device=d)
        ops.pfn_rows(vox, w0, b0, w1, b1, canvas=canvas, pillar_features=pf)
        torch.cuda.synchronize()
        assert not bool(torch.isnan(pf).any()) and not bool(torch.isnan(canvas).any())
        outs[thr] = (pf.clone(), canvas.clone(), vox.voxel_coords[:P].clone())
    for thr in ('64', '192', '100000'):
        assert torch.equal(outs[thr][0], outs['0'][0]) and torch.equal(outs[thr][1], outs['0'][1]) and torch.equal(outs[thr][2], outs['0'][2])
    arch = dict(num_raw=num_raw, pc_range=PC_RANGE, voxel_size=VOXEL, grid_size=GRID, vfe_filters=[64, 64])
    ref = opil.vfe_forward(pts, st, arch)
    np.testing.assert_allclose(outs['192'][0].cpu().numpy(), ref['pillar_features'], rtol=1e-4, atol=2e-5)


def test_pfn_rows_with_a_whole_cloud_in_one_cell(lib_option):
    """the degenerate end of the crowded-pillar path: 20 000 points in ONE cell (plus a handful elsewhere, before and behind it in the slot
    order), as the only frame of the batch and again as frame 1 of 2.  Bit for bit the wave tiles' result (option pfn_crowd = 0); against the
    oracle at the north star's 1e-3: its scatter_mean adds 20 000 float32 values in index order (the reference's own float32 atomics do no
    better), the kernel's sums are exact"""
    ops = _ops()
    rs = np.random.RandomState(3)
    st, (w0, b0, w1, b1) = _folded_vfe(5)
    arch = dict(num_raw=5, pc_range=PC_RANGE, voxel_size=VOXEL, grid_size=GRID, vfe_filters=[64, 64])
    d = dev()
    for B, frame in ((1, 0), (2, 1)):
        k = 20000
        q = np.zeros((k + 7, 8), np.float32)
        q[:, 0] = frame
        q[:k, 1] = -51.2 + 0.2 * 300 + rs.uniform(0.0, 0.2, k)
        q[:k, 2] = -51.2 + 0.2 * 17 + rs.uniform(0.0, 0.2, k)
        q[k:, 1] = rs.uniform(-50, 50, 7)
        q[k:, 2] = rs.uniform(-50, 50, 7)
        q[k + 5:, 1:3] = q[k + 4, 1:3]                            # a three-point pillar somewhere else
        q[:, 3] = rs.uniform(-8, 0, k + 7)
        q[:, 4:] = rs.uniform(0, 1, (k + 7, 4))
        pts = q[rs.permutation(k + 7)]
        ref = opil.vfe_forward(pts, st, arch)
        g = ops.make_grid(PC_RANGE, VOXEL, GRID, B)
        vox = ops.pillarise_rows(torch.from_numpy(pts).to(d), g, 5, want_coords=True)
        P = int(vox.counters[0])
        assert P == ref['pillar_features'].shape[0]
        canvas = torch.full((B, 512, 512, 64), float('nan'), device=d)
        pf = torch.full((P, 64), float('nan'), device=d)
        ops.pfn_rows(vox, w0, b0, w1, b1, canvas=canvas, pillar_features=pf)
        torch.cuda.synchronize()
        np.testing.assert_allclose(pf.cpu().numpy(), ref['pillar_features'], rtol=1e-3, atol=2e-4)
        assert not bool(torch.isnan(canvas).any()) and int((canvas != 0).any(-1).sum()) <= P
        vc = vox.voxel_coords[:P].long()
        assert torch.equal(canvas[vc[:, 0], vc[:, 2], vc[:, 3]], pf)
        lib_option('pfn_crowd', 0)
        vox0 = ops.pillarise_rows(torch.from_numpy(pts).to(d), g, 5)
        pf0 = torch.full((P, 64), float('nan'), device=d)
        ops.pfn_rows(vox0, w0, b0, w1, b1, canvas=None, pillar_features=pf0)
        torch.cuda.synchronize()
        lib_option('pfn_crowd', None)
        assert torch.equal(pf0, pf)


def test_pfn_rows_empty_cloud_and_single_point():
    ops = _ops()
    _st, (w0, b0, w1, b1) = _folded_vfe(5)
    d = dev()
    g = ops.make_grid([-6.4, -6.4, -8, 6.4, 6.4, 0], VOXEL, [64, 64, 1], 2)
    for pts in (np.zeros((0, 8), np.float32), np.array([[1, 0.05, 0.05, -1, 0.5, 0.1, 0, -1]], np.float32)):
        pd = torch.from_numpy(pts).to(d)
        vox = ops.pillarise_rows(pd, g, 5, want_coords=True)
        canvas = torch.full((2, 64, 64, 64), float('nan'), device=d)
        ops.pfn_rows(vox, w0, b0, w1, b1, canvas=canvas)
        torch.cuda.synchronize()
        P = int(vox.counters[0])
        assert P == pts.shape[0]
        assert not bool(torch.isnan(canvas).any())
        assert int((canvas != 0).any(-1).sum()) == P
        if P:
            assert bool((canvas[1, 32, 32] != 0).any())


def test_pfn_rows_after_select_transform_compact_equals_the_plain_chain():
    """the BEV makers' entry: compaction emits the cell ids, pcp_pillarise_rows(CELLS_READY) finishes; same pillar rows as pillarising the
    compacted cloud from scratch"""
    ops = _ops()
    d = dev()
    B = 2
    frames = [np.concatenate([synth.agent_cloud(a + 7 * b, 4000, 'disco') for a in range(3)], 0) for b in range(B)]
    for b in range(B):
        frames[b][:, -1] = np.repeat(np.arange(3), 4000)
    pts = torch.from_numpy(synth.collate(frames)).to(d)
    n, c = pts.shape
    agents = [0, 2]
    poses = np.tile(np.eye(4, dtype=np.float32)[:3].reshape(1, 1, 12), (2, B, 1))
    poses[1, :, 3] = 1.5
    present = np.ones((2, B), np.uint8)
    rows = 2 * B * 4000
    g = ops.make_grid(PC_RANGE, VOXEL, GRID, B * 2)
    ws = ops.rows_workspace(g, rows, 5, d)
    out = torch.empty((rows, c), device=d)
    ops.select_transform_compact(pts, c - 1, agents, poses, present, rows, out=out, vox_grid=g, vox_workspace=ws)
    vox = ops.pillarise_rows(out, g, 5, workspace=ws, cells_ready=True)
    _st, (w0, b0, w1, b1) = _folded_vfe(5)
    pf = torch.zeros((rows, 64), device=d)
    ops.pfn_rows(vox, w0, b0, w1, b1, pillar_features=pf)
    vox2 = ops.pillarise_rows(out, g, 5)
    pf2 = torch.zeros((rows, 64), device=d)
    ops.pfn_rows(vox2, w0, b0, w1, b1, pillar_features=pf2)
    torch.cuda.synchronize()
    assert torch.equal(vox.counters[:2], vox2.counters[:2]) and int(vox.counters[0]) > 1000
    assert torch.equal(pf, pf2)


# ---------------------------------------------------------------------------------------------------------------------
# a6 / a7 convolutions
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('cin,cout,h,w,stride,relu,batch', [
    (64, 64, 64, 64, 1, True, 1), (64, 64, 64, 64, 2, True, 2), (64, 128, 32, 48, 2, True, 1), (128, 128, 32, 32, 1, True, 2),
    (384, 64, 16, 16, 1, True, 1), (64, 9, 32, 32, 1, False, 2), (16, 32, 8, 8, 1, False, 1), (128, 256, 20, 12, 2, True, 1),
    (64, 64, 13, 21, 1, True, 1), (32, 96, 9, 9, 2, True, 1)])
def test_conv3x3_matches_torch_cpu(cin, cout, h, w, stride, relu, batch):
    ops = _ops()
    from pcp_amd import pack
    x = torch.from_numpy(_rand(1, (batch, cin, h, w)))
    wt = torch.from_numpy(_rand(2, (cout, cin, 3, 3), -0.05, 0.05))
    b = torch.from_numpy(_rand(3, (cout,), -0.2, 0.2))
    want = F.conv2d(x, wt, b, stride=stride, padding=1)
    if relu:
        want = F.relu(want)
    packed, bp, cpad = pack.pack_conv3x3(wt, b)
    d = dev()
    got = ops.conv3x3(ops.as_nhwc(x.to(d)), packed.to(d), bp.to(d), cin, cout, cpad, stride=stride, relu=relu)
    torch.cuda.synchronize()
    # exact-fp32 MFMA chain vs oneDNN blocking: relative 1e-5 of the accumulated magnitude
    np.testing.assert_allclose(got.permute(0, 3, 1, 2).cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4)


def test_conv3x3_channel_windows_and_linearity():
    """writes into a channel slice of a wider buffer (the 384-channel concat) and obeys conv(a x + y) = a conv(x) + conv(y)."""
    ops = _ops()
    from pcp_amd import pack
    d = dev()
    cin, cout = 64, 128
    wt = torch.from_numpy(_rand(7, (cout, cin, 3, 3), -0.05, 0.05))
    zero_b = torch.zeros(cout)
    packed, bp, cpad = pack.pack_conv3x3(wt, zero_b)
    x = torch.from_numpy(_rand(8, (1, 48, 40, 96))).to(d)          # NHWC buffer with ld 96; use channels 16..79
    y = torch.from_numpy(_rand(9, (1, 48, 40, 96))).to(d)
    out = torch.full((1, 48, 40, 384), 7.0, device=d)
    ops.conv3x3(x, packed.to(d), bp.to(d), cin, cout, cpad, relu=False, out=out, in_ch_off=16, out_ch_off=128)
    torch.cuda.synchronize()
    assert float((out[..., :128] - 7.0).abs().max()) == 0.0 and float((out[..., 256:] - 7.0).abs().max()) == 0.0
    want = F.conv2d(x[..., 16:80].permute(0, 3, 1, 2).cpu(), wt, None, padding=1)
    np.testing.assert_allclose(out[..., 128:256].permute(0, 3, 1, 2).cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4)
    cx = ops.conv3x3(x, packed.to(d), bp.to(d), cin, cout, cpad, relu=False, in_ch_off=16)
    cy = ops.conv3x3(y, packed.to(d), bp.to(d), cin, cout, cpad, relu=False, in_ch_off=16)
    cz = ops.conv3x3((2.0 * x + y).contiguous(), packed.to(d), bp.to(d), cin, cout, cpad, relu=False, in_ch_off=16)
    torch.cuda.synchronize()
    np.testing.assert_allclose(cz.cpu().numpy(), (2.0 * cx + cy).cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('mode,cin,cout,h,w', [('plain', 128, 128, 24, 24), ('plain', 256, 64, 16, 16), ('plain', 16, 1, 16, 16),
                                                ('s2d', 64, 128, 32, 32), ('d2s', 128, 128, 16, 16), ('d2s', 256, 128, 8, 12),
                                                ('plain', 384, 32, 1, 1000)])
def test_pointwise_matches_torch_cpu(mode, cin, cout, h, w):
    ops = _ops()
    from pcp_amd import lib, pack
    d = dev()
    B = 2
    x = torch.from_numpy(_rand(21, (B, cin, h, w)))
    b = torch.from_numpy(_rand(23, (cout,), -0.2, 0.2))
    if mode == 'plain':
        wt = torch.from_numpy(_rand(22, (cout, cin), -0.1, 0.1))
        want = F.relu(F.conv2d(x, wt[:, :, None, None], b))
        packed, bp, cpad = pack.pack_plain(wt, b)
        m = lib.PW_PLAIN
    elif mode == 's2d':
        wt = torch.from_numpy(_rand(22, (cout, cin, 2, 2), -0.1, 0.1))
        want = F.relu(F.conv2d(x, wt, b, stride=2))
        packed, bp, cpad = pack.pack_conv2x2_s2(wt, b)
        m = lib.PW_SPACE2DEPTH
    else:
        wt = torch.from_numpy(_rand(22, (cin, cout, 2, 2), -0.1, 0.1))
        want = F.relu(F.conv_transpose2d(x, wt, b, stride=2))
        packed, bp, cpad = pack.pack_convT2x2_s2(wt, b)
        m = lib.PW_DEPTH2SPACE
    got = ops.pointwise(ops.as_nhwc(x.to(d)), packed.to(d), bp.to(d), m, cin, cout, cpad, relu=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.permute(0, 3, 1, 2).cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------------------------------------------------------------
# a8 decode
# ---------------------------------------------------------------------------------------------------------------------
def _decode_kwargs(arch, ld=16):
    hd = arch['head']
    return dict(k=hd['max_obj'], num_class=1, ch_center=0, ch_z=2, ch_dim=3, ch_rot=6, ch_hm=8, stride=hd['stride'],
                voxel_x=float(np.float32(arch['voxel_size'][0])), voxel_y=float(np.float32(arch['voxel_size'][1])),
                min_x=float(np.float32(arch['pc_range'][0])), min_y=float(np.float32(arch['pc_range'][1])),
                limit=hd['limit_range'], score_thresh=hd['score_thresh'])


def _head_buffer(maps, ld=16):
    order = ['center', 'center_z', 'dim', 'rot', 'hm']
    cat = torch.cat([torch.from_numpy(np.asarray(maps[k])) for k in order], dim=1)     # (B, 9, H, W)
    B, C, H, W = cat.shape
    buf = torch.zeros((B, H, W, ld))
    buf[..., :C] = cat.permute(0, 2, 3, 1)
    return buf


@pytest.mark.parametrize('tag', ['car', 'ego'])
def test_decode_matches_oracle_on_golden_head_maps(tag):
    ops = _ops()
    g = load_golden('g1_%s.npz' % tag)
    arch = arch_of(g['meta'])
    maps = {k: g['head_' + k] for k in ('center', 'center_z', 'dim', 'rot', 'hm')}
    ref = obev.decode_boxes({k: torch.from_numpy(v) for k, v in maps.items()}, arch)
    boxes, scores, labels, cell, count = ops.centerhead_decode(_head_buffer(maps).to(dev()), _decode_kwargs(arch))
    torch.cuda.synchronize()
    for b in range(2):
        n = int(count[b])
        assert n == ref[b]['boxes'].shape[0]
        assert np.array_equal(cell[b, :n].cpu().numpy().astype(np.int64), ref[b]['cell'])      # same cells, same order
        np.testing.assert_allclose(scores[b, :n].cpu().numpy(), ref[b]['scores'], rtol=0, atol=1e-6)
        np.testing.assert_allclose(boxes[b, :n].cpu().numpy(), ref[b]['boxes'], rtol=1e-5, atol=1e-5)
        s = scores[b, :n]
        assert bool((s[1:] <= s[:-1]).all())


def test_decode_full_size_random_and_ties():
    ops = _ops()
    arch = dict(pc_range=PC_RANGE, voxel_size=VOXEL,
                head=dict(max_obj=500, stride=4, limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], score_thresh=0.1))
    B, H, W = 3, 128, 128
    maps = dict(center=_rand(31, (B, 2, H, W), 0, 1), center_z=_rand(32, (B, 1, H, W), -3, 0), dim=_rand(33, (B, 3, H, W), 0, 1.5),
                rot=_rand(34, (B, 2, H, W)), hm=_rand(35, (B, 1, H, W), -6, 2))
    maps['hm'][1] = -1.0                    # frame 1: every score equal -> ties resolved by the lowest cell index
    maps['hm'][2] = -9.0                    # frame 2: nothing above the score threshold
    ref = obev.decode_boxes({k: torch.from_numpy(v) for k, v in maps.items()}, arch)
    boxes, scores, labels, cell, count = ops.centerhead_decode(_head_buffer(maps).to(dev()), _decode_kwargs(arch))
    torch.cuda.synchronize()
    assert int(count[2]) == 0 and ref[2]['boxes'].shape[0] == 0
    n1 = int(count[1])
    assert n1 == ref[1]['boxes'].shape[0] == 500
    assert np.array_equal(cell[1, :n1].cpu().numpy(), np.arange(500))
    n0 = int(count[0])
    assert n0 == ref[0]['boxes'].shape[0]
    # random frame: identical cells except where two scores differ by less than the sigmoid ulp noise
    same = cell[0, :n0].cpu().numpy().astype(np.int64) == ref[0]['cell']
    assert same.mean() > 0.99
    assert set(cell[0, :n0].cpu().numpy().tolist()) == set(ref[0]['cell'].tolist())


# ---------------------------------------------------------------------------------------------------------------------
# a9 NMS
# ---------------------------------------------------------------------------------------------------------------------
def test_pairwise_iou_matches_reference_values():
    ops = _ops()
    g = load_golden('g3_nms.npz')
    order = g['order']
    b = torch.from_numpy(g['boxes'][order]).to(dev())
    iou = ops.boxes_bev_pairwise(b, b, 1).cpu().numpy()
    # sinf/cosf/atan2f differ by ulps between ocml and glibc: 1e-5 absolute on IoU in [0, 1]
    np.testing.assert_allclose(iou, g['iou_sorted'], rtol=0, atol=2e-5)


@pytest.mark.parametrize('thr,key', [(0.2, 'keep_02'), (0.3, 'keep_03')])
def test_nms_keep_list_matches_golden(thr, key):
    ops = _ops()
    g = load_golden('g3_nms.npz')
    boxes, scores = g['boxes'], g['scores']
    d = dev()
    keep, cnt = ops.nms_rotated(torch.from_numpy(boxes).to(d), torch.from_numpy(scores).to(d), thr, 1000, 500)
    torch.cuda.synchronize()
    n = int(cnt[0])
    got = keep[:n].cpu().numpy().astype(np.int64)
    # the keep list is a function of the boolean matrix (iou > thr); sinf/cosf/atan2f ulp differences can only matter
    # for pairs sitting on the threshold, so first check that no verdict flipped, then demand the golden list
    order = g['order']
    sb = torch.from_numpy(boxes[order]).to(d)
    giou = ops.boxes_bev_pairwise(sb, sb, 1).cpu().numpy()
    flips = (giou > thr) != (g['iou_sorted'] > thr)
    assert flips.sum() <= 2, 'verdict flips only expected for pairs within an ulp of the threshold'
    if flips.sum() == 0:
        assert np.array_equal(got, g[key])
    assert np.array_equal(got, order[onms.nms_from_iou(giou, thr)])
    keep83, cnt83 = ops.nms_rotated(torch.from_numpy(boxes).to(d), torch.from_numpy(scores).to(d), thr, 1000, 83)
    assert np.array_equal(keep83[:int(cnt83[0])].cpu().numpy().astype(np.int64), got[:83])


@pytest.mark.parametrize('n', [0, 1, 63, 64, 65, 500, 1000, 4096])
def test_nms_sizes_against_oracle(n):
    ops = _ops()
    s = 900 + n
    boxes = np.zeros((max(n, 1), 7), np.float32)[:n]
    if n:
        k = max(n // 6, 1)
        cx, cy = synth.uniform(s, 1, k, -40, 40), synth.uniform(s, 2, k, -40, 40)
        which = (synth.uniform01(s, 3, n) * k).astype(np.int64)
        boxes[:, 0] = cx[which] + synth.uniform(s, 4, n, -2, 2)
        boxes[:, 1] = cy[which] + synth.uniform(s, 5, n, -2, 2)
        boxes[:, 3] = synth.uniform(s, 7, n, 3.0, 5.5)
        boxes[:, 4] = synth.uniform(s, 8, n, 1.5, 2.5)
        boxes[:, 5] = 1.5
        boxes[:, 6] = synth.uniform(s, 10, n, -3.14, 3.14)
    scores = (np.argsort(np.argsort(synth.uniform01(s, 11, max(n, 1))[:n])).astype(np.float32) + 1) / np.float32(n + 1)
    d = dev()
    keep, cnt = ops.nms_rotated(torch.from_numpy(boxes).to(d).reshape(n, 7), torch.from_numpy(scores).to(d), 0.2, 10000, max(n, 1))
    torch.cuda.synchronize()
    got = keep[:int(cnt[0])].cpu().numpy().astype(np.int64)
    if n == 0:
        assert got.shape[0] == 0
        return
    order = np.argsort(-scores, kind='stable')
    iou = onms.iou_matrix(boxes[order], boxes[order])
    want = order[onms.nms_from_iou(iou, 0.2)]
    if np.any(np.abs(iou - 0.2) < 2e-5):      # ulp-sensitive pairs present: compare through the GPU's own IoU matrix
        giou = ops.boxes_bev_pairwise(torch.from_numpy(boxes[order]).to(d), torch.from_numpy(boxes[order]).to(d), 1).cpu().numpy()
        want = order[onms.nms_from_iou(giou, 0.2)]
    assert np.array_equal(got, want)


# ---------------------------------------------------------------------------------------------------------------------
# a11 / a12
# ---------------------------------------------------------------------------------------------------------------------
def test_warp_nearest_batch_is_the_job_by_job_warp():
    """pcp_warp_nearest_batch (every (agent, frame) pair of a DiscoNet forward in one launch, chunks of 32 jobs) against pcp_warp_nearest job by
    job: identical bits, with and without accumulation"""
    ops = _ops()
    d = dev()
    rng = np.random.RandomState(11)
    H, W, C = 40, 40, 8
    jobs, want = [], []
    for j in range(37):                                              # two launches: 32 + 5
        src = torch.from_numpy(rng.uniform(-1, 1, (H, W, C + 4)).astype(np.float32)).to(d)
        a = rng.uniform(-0.6, 0.6)
        theta = [float(np.cos(a)), float(-np.sin(a)), float(rng.uniform(-0.4, 0.4)), float(np.sin(a)), float(np.cos(a)), float(rng.uniform(-0.4, 0.4))]
        dst = torch.full((H, W, C), 0.25 * j, device=d)
        ref = dst.clone()
        jobs.append((src, dst, theta))
        want.append((src, ref, theta))
    for acc in (False, True):
        for src, ref, theta in want:
            ops.warp_nearest(src, ref, theta, C, accumulate=acc)
        ops.warp_nearest_batch(jobs, C, accumulate=acc)
        torch.cuda.synchronize()
        for (_s, got, _t), (_s2, ref, _t2) in zip(jobs, want):
            assert torch.equal(got, ref)
    ops.warp_nearest_batch([], C)                                    # nothing to do


def test_warp_nearest_matches_golden():
    ops = _ops()
    from pcp_amd import fusion_host
    g = load_golden('g4_warp.npz')
    d = dev()
    n_exact = n_total = 0
    for key in [str(k) for k in g['cases']]:
        H = int(key.split('_')[0][1:])
        pc_min, pix = [float(v) for v in g['H%d_params' % H]]
        img = torch.from_numpy(g['H%d_img' % H])                     # (3, H, W)
        src = torch.zeros((H, H, 4))
        src[..., :3] = img.permute(1, 2, 0)
        theta, ambiguous = fusion_host.warp_theta(torch.from_numpy(g[key + '_T']), H, H, pc_min, pix, return_ambiguous=True)
        dst = torch.full((H, H, 4), -5.0, device=d)
        ops.warp_nearest(src.to(d), dst, theta, 4)
        torch.cuda.synchronize()
        got = dst[..., :3].permute(2, 0, 1).cpu().numpy()
        want = g[key + '_out']
        ok = ~ambiguous.numpy()[None]                                # pixels whose source coordinate is not on a .5 tie
        assert np.array_equal(np.where(ok, got, 0), np.where(ok, want, 0)), key
        # on a .5 tie the pick depends on the last ulp of the affine evaluation (torch's CPU sgemm order is not ours):
        # there the kernel must return one of the two (four) neighbouring source pixels, or the zero pad
        th = np.asarray(theta, dtype=np.float64).reshape(2, 3)
        xs = (2.0 * np.arange(H) + 1.0) / H - 1.0
        yy, xx = np.meshgrid(xs, xs, indexing='ij')
        fx = ((xx * th[0, 0] + yy * th[0, 1] + th[0, 2] + 1.0) * H - 1.0) / 2.0
        fy = ((xx * th[1, 0] + yy * th[1, 1] + th[1, 2] + 1.0) * H - 1.0) / 2.0
        img_np = img.numpy()
        good = np.zeros((H, H), bool)
        for cx in (np.floor(fx), np.ceil(fx)):
            for cy in (np.floor(fy), np.ceil(fy)):
                inb = (cx >= 0) & (cx < H) & (cy >= 0) & (cy < H)
                cand = np.where(inb[None], img_np[:, np.clip(cy, 0, H - 1).astype(int), np.clip(cx, 0, H - 1).astype(int)], 0.0)
                good |= (cand == got).all(0)
        assert good[ambiguous.numpy()].all(), key
        n_exact += int(ok.sum())
        n_total += ok.size
    assert n_exact / n_total > 0.6


def test_softmax_fuse_matches_torch():
    ops = _ops()
    d = dev()
    B, H, W, C, A = 2, 16, 16, 128, 4
    maps = [torch.from_numpy(_rand(40 + a, (B, H, W, C))).to(d) for a in range(A)]
    wts = torch.from_numpy(_rand(50, (B, H, W, A), 0, 3)).to(d)
    out = torch.empty((B, H, W, C), device=d)
    ops.softmax_fuse(maps, wts, C, out)
    torch.cuda.synchronize()
    sm = torch.softmax(wts.cpu(), dim=-1)
    want = sum(maps[a].cpu() * sm[..., a:a + 1] for a in range(A))
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------
# a14
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('crowded', [False, True])
def test_hunter_point_ops_match_oracle(crowded):
    """crowded: LiDAR-like clouds -- the BEV cells under the sensor hold hundreds to thousands of rows (more than the 1 024 the scatter-mean
    kernel sorts): summed by row slots, tolerance of a float32 sum in another order"""
    ops = _ops()
    d = dev()
    B, H, W, C = 2, 32, 32, 64
    bev = torch.from_numpy(_rand(60, (B, C, H, W)))
    dist = 'ring' if crowded else 'uniform'
    pts_np = synth.collate([synth.agent_cloud(5, 30000 if crowded else 3000, 'car', xy_half=13.1, dist=dist),
                            synth.agent_cloud(6, 20000 if crowded else 2000, 'car', xy_half=13.1, dist=dist)])
    extra = np.zeros((4, 8), np.float32)
    extra[:, 1:3] = [[-12.8, 1.0], [1.0, -12.8], [12.799999, 0.3], [0.3, 12.799999]]
    pts_np = np.concatenate([pts_np, extra], 0)
    pts = torch.from_numpy(pts_np)
    rng = [-12.8, -12.8, -8.0, 12.8, 12.8, 0.0]
    want_feat, coord = obev.sample_point_features(bev, pts, rng, [0.8, 0.8])
    got = ops.bev_sample_bilinear(ops.as_nhwc(bev.to(d)), pts.to(d), rng[:2], [np.float32(0.2) * 4, np.float32(0.2) * 4])
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.cpu().numpy(), want_feat.numpy(), rtol=1e-5, atol=1e-6)
    want_img = obev.bev_scatter_mean(coord, pts[:, 0].long(), want_feat, (H, W), batch_size=B)
    got_img = ops.bev_scatter_mean(pts.to(d), got, B, H, W, rng[:2], [np.float32(0.2) * 4, np.float32(0.2) * 4])
    torch.cuda.synchronize()
    np.testing.assert_allclose(got_img.permute(0, 3, 1, 2).cpu().numpy(), want_img.numpy(), rtol=1e-5, atol=2e-5 if crowded else 1e-6)
    if crowded:
        cn = coord.numpy()
        m = (cn[:, 0] > 0) & (cn[:, 0] < W) & (cn[:, 1] > 0) & (cn[:, 1] < H)
        cnt = np.bincount(cn[m, 1].astype(np.int64) * W + cn[m, 0].astype(np.int64) + pts_np[m, 0].astype(np.int64) * H * W)
        assert cnt.max() > 1024 and (cnt > 32).sum() > 10                   # both the sorted and the unsorted long path ran


# ---------------------------------------------------------------------------------------------------------------------
# fused Winograd F(2x2, 3x3) variant of the stride-1 conv
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('cin,cout,h,w,relu,batch', [(64, 64, 32, 32, True, 1), (128, 128, 16, 48, True, 2), (8, 64, 16, 16, False, 1),
                                                      (384, 64, 16, 16, True, 1), (64, 320, 32, 32, True, 1), (64, 100, 20, 36, False, 2),
                                                      (24, 64, 7, 9, True, 1), (768, 768, 16, 16, True, 1)])
def test_conv3x3_winograd_matches_torch_cpu(cin, cout, h, w, relu, batch):
    ops = _ops()
    from pcp_amd import pack
    x = torch.from_numpy(_rand(71, (batch, cin, h, w)))
    wt = torch.from_numpy(_rand(72, (cout, cin, 3, 3), -0.05, 0.05))
    b = torch.from_numpy(_rand(73, (cout,), -0.2, 0.2))
    want = F.conv2d(x, wt, b, padding=1)
    if relu:
        want = F.relu(want)
    packed, bp, cpad = pack.pack_conv3x3_winograd(wt, b)
    d = dev()
    got = ops.conv3x3_winograd(ops.as_nhwc(x.to(d)), packed.to(d), bp.to(d), cin, cout, cpad, relu=relu)
    torch.cuda.synchronize()
    # Winograd transforms add a few fp32 roundings on top of the accumulation-order difference
    np.testing.assert_allclose(got.permute(0, 3, 1, 2).cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-4)


def test_conv3x3_winograd_channel_windows():
    ops = _ops()
    from pcp_amd import pack
    d = dev()
    cin, cout = 64, 128
    wt = torch.from_numpy(_rand(77, (cout, cin, 3, 3), -0.05, 0.05))
    packed, bp, cpad = pack.pack_conv3x3_winograd(wt, torch.zeros(cout))
    x = torch.from_numpy(_rand(78, (1, 48, 40, 96))).to(d)
    out = torch.full((1, 48, 40, 384), 7.0, device=d)
    ops.conv3x3_winograd(x, packed.to(d), bp.to(d), cin, cout, cpad, relu=False, out=out, in_ch_off=16, out_ch_off=128)
    torch.cuda.synchronize()
    assert float((out[..., :128] - 7.0).abs().max()) == 0.0 and float((out[..., 256:] - 7.0).abs().max()) == 0.0
    want = F.conv2d(x[..., 16:80].permute(0, 3, 1, 2).cpu(), wt, None, padding=1)
    np.testing.assert_allclose(out[..., 128:256].permute(0, 3, 1, 2).cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-4)


# ---------------------------------------------------------------------------------------------------------------------
# wave-stationary fused Winograd F(2x2, 3x3) (csrc/wino_ws.hip): both workgroup shapes, ragged image edges, padded cout, windows
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('cin,cout,h,w,relu,batch', [(64, 64, 32, 32, True, 1), (128, 128, 16, 48, True, 2), (32, 64, 16, 16, False, 1),
                                                      (384, 64, 16, 16, True, 1), (64, 320, 32, 32, True, 1), (64, 100, 20, 36, False, 2),
                                                      (96, 128, 7, 9, True, 1), (128, 384, 24, 40, True, 1), (128, 128, 64, 64, True, 3)])
def test_conv3x3_winograd_ws_matches_torch_cpu(cin, cout, h, w, relu, batch):
    ops = _ops()
    from pcp_amd import pack
    x = torch.from_numpy(_rand(171, (batch, cin, h, w)))
    wt = torch.from_numpy(_rand(172, (cout, cin, 3, 3), -0.05, 0.05))
    b = torch.from_numpy(_rand(173, (cout,), -0.2, 0.2))
    want = F.conv2d(x, wt, b, padding=1)
    if relu:
        want = F.relu(want)
    packed, bp, cpad = pack.pack_conv3x3_winograd_ws(wt, b)
    d = dev()
    got = ops.conv3x3_winograd_ws(ops.as_nhwc(x.to(d)), packed.to(d), bp.to(d), cin, cout, cpad, relu=relu)
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.permute(0, 3, 1, 2).cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-4)
    # deterministic: a second launch gives the same bits
    got2 = ops.conv3x3_winograd_ws(ops.as_nhwc(x.to(d)), packed.to(d), bp.to(d), cin, cout, cpad, relu=relu)
    assert torch.equal(got, got2)


def test_conv3x3_winograd_ws_channel_windows_and_bad_arguments():
    ops = _ops()
    from pcp_amd import lib, pack
    d = dev()
    cin, cout = 64, 128
    wt = torch.from_numpy(_rand(177, (cout, cin, 3, 3), -0.05, 0.05))
    packed, bp, cpad = pack.pack_conv3x3_winograd_ws(wt, torch.zeros(cout))
    x = torch.from_numpy(_rand(178, (1, 48, 40, 96))).to(d)
    out = torch.full((1, 48, 40, 384), 7.0, device=d)
    ops.conv3x3_winograd_ws(x, packed.to(d), bp.to(d), cin, cout, cpad, relu=False, out=out, in_ch_off=16, out_ch_off=128)
    torch.cuda.synchronize()
    assert float((out[..., :128] - 7.0).abs().max()) == 0.0 and float((out[..., 256:] - 7.0).abs().max()) == 0.0
    want = F.conv2d(x[..., 16:80].permute(0, 3, 1, 2).cpu(), wt, None, padding=1)
    np.testing.assert_allclose(out[..., 128:256].permute(0, 3, 1, 2).cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-4)
    with pytest.raises(lib.PcpError):                       # cin must be a multiple of the 32-channel staging chunk
        ops.conv3x3_winograd_ws(x, packed.to(d), bp.to(d), 40, cout, cpad, relu=False, out=out)
    with pytest.raises(lib.PcpError):                       # input window must start 16-byte aligned
        ops.conv3x3_winograd_ws(x, packed.to(d), bp.to(d), cin, cout, cpad, relu=False, out=out, in_ch_off=2)


# ---------------------------------------------------------------------------------------------------------------------
# FUSED Winograd F(4x4, 3x3) (csrc/wino4f.hip): ragged image edges (H, W not multiples of the 16 x 32 workgroup tile nor of 4), padded
# cout, odd slice counts, channel windows
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('cin,cout,h,w,relu,batch', [(64, 64, 32, 32, True, 1), (128, 128, 16, 64, True, 2), (8, 64, 16, 32, False, 1),
                                                      (384, 64, 16, 32, True, 1), (64, 320, 32, 32, True, 1), (64, 100, 20, 36, False, 2),
                                                      (24, 128, 7, 9, True, 1), (128, 384, 24, 40, True, 1), (128, 128, 64, 64, True, 3),
                                                      (256, 256, 64, 64, True, 1)])
@pytest.mark.parametrize('kernel', ['winograd4f', 'winograd4h', 'winograd4c'])
def test_conv3x3_winograd4f_matches_torch_cpu(cin, cout, h, w, relu, batch, kernel):
    """the fused F(4x4) kernels: k_wino4f (eight-wave workgroup, 16 x 32-pixel items), k_wino4h (two four-wave workgroups per CU,
    16 x 16-pixel items, csrc/wino4h.hip) and k_wino4c (the same items, waves split over output channels, csrc/wino4c.hip)"""
    ops = _ops()
    from pcp_amd import pack
    x = torch.from_numpy(_rand(271, (batch, cin, h, w)))
    wt = torch.from_numpy(_rand(272, (cout, cin, 3, 3), -0.05, 0.05))
    b = torch.from_numpy(_rand(273, (cout,), -0.2, 0.2))
    want = F.conv2d(x, wt, b, padding=1)
    if relu:
        want = F.relu(want)
    packed, bp, cpad = getattr(pack, 'pack_conv3x3_' + kernel)(wt, b)
    d = dev()
    run = getattr(ops, 'conv3x3_' + kernel)
    got = run(ops.as_nhwc(x.to(d)), packed.to(d), bp.to(d), cin, cout, cpad, relu=relu)
    torch.cuda.synchronize()
    # F(4x4) transforms round at ~1e-5 of the output scale (same bar as the through-memory F(4x4) path)
    np.testing.assert_allclose(got.permute(0, 3, 1, 2).cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-4 * max(1.0, float(want.abs().max())))
    got2 = run(ops.as_nhwc(x.to(d)), packed.to(d), bp.to(d), cin, cout, cpad, relu=relu)
    assert torch.equal(got, got2)                                            # deterministic


@pytest.mark.parametrize('kernel', ['winograd4f', 'winograd4h', 'winograd4c'])
def test_conv3x3_winograd4f_channel_windows_and_bad_arguments(kernel):
    ops = _ops()
    from pcp_amd import lib, pack
    d = dev()
    cin, cout = 64, 128
    wt = torch.from_numpy(_rand(277, (cout, cin, 3, 3), -0.05, 0.05))
    packed, bp, cpad = getattr(pack, 'pack_conv3x3_' + kernel)(wt, torch.zeros(cout))
    run = getattr(ops, 'conv3x3_' + kernel)
    x = torch.from_numpy(_rand(278, (1, 48, 40, 96))).to(d)
    out = torch.full((1, 48, 40, 384), 7.0, device=d)
    run(x, packed.to(d), bp.to(d), cin, cout, cpad, relu=False, out=out, in_ch_off=16, out_ch_off=128)
    torch.cuda.synchronize()
    assert float((out[..., :128] - 7.0).abs().max()) == 0.0 and float((out[..., 256:] - 7.0).abs().max()) == 0.0
    want = F.conv2d(x[..., 16:80].permute(0, 3, 1, 2).cpu(), wt, None, padding=1)
    np.testing.assert_allclose(out[..., 128:256].permute(0, 3, 1, 2).cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-4)
    with pytest.raises(lib.PcpError):
        run(x, packed.to(d), bp.to(d), 36, cout, cpad, relu=False, out=out)
    with pytest.raises(lib.PcpError):
        run(x, packed.to(d), bp.to(d), cin, cout, cpad, relu=False, out=out, in_ch_off=2)


@pytest.mark.parametrize('nw', ['4', '8'])
@pytest.mark.parametrize('cin,cout,h,w,batch', [(64, 64, 32, 32, 1), (128, 128, 48, 80, 2), (16, 52, 37, 50, 2), (384, 64, 16, 32, 1), (72, 132, 20, 100, 3),
                                                 (64, 384, 33, 47, 1), (8, 128, 16, 16, 1)])
def test_wino4c_gives_the_bits_of_wino4h(cin, cout, h, w, batch, nw, lib_option):
    """k_wino4c multiplies the same products in the same k order as k_wino4h and runs the same transform operations per lane: bitwise equal,
    in its 64-channel form (two four-wave workgroups per CU) and in its 128-channel form (one eight-wave workgroup, shared input transform;
    option wino4c_nw = 8 selects the 128-channel form wherever cout_pad is a multiple of 128)"""
    ops = _ops()
    from pcp_amd import lib, pack
    d = dev()
    lib_option('wino4c_nw', int(nw))
    x = ops.as_nhwc(torch.from_numpy(_rand(291, (batch, cin, h, w))).to(d))
    wt = torch.from_numpy(_rand(292, (cout, cin, 3, 3), -0.05, 0.05))
    b = torch.from_numpy(_rand(293, (cout,), -0.2, 0.2))
    uh, bh, cph = pack.pack_conv3x3_winograd4h(wt, b)
    uc, bc, cpc = pack.pack_conv3x3_winograd4c(wt, b)
    oh = ops.conv3x3_winograd4h(x, uh.to(d), bh.to(d), cin, cout, cph, relu=True)
    oc = ops.conv3x3_winograd4c(x, uc.to(d), bc.to(d), cin, cout, cpc, relu=True)
    torch.cuda.synchronize()
    assert torch.equal(oh, oc) and float(oh.abs().max()) > 0


def test_auto_dispatch_falls_back_when_the_fused_f4_kernel_refuses(monkeypatch):
    """a layer the fused F(4x4) kernel cannot take (over its 2 GiB input limit -- lowered here) still runs in auto mode, on the
    F(2x2) kernel, with the same result (every conv kernel of the library needs 16-byte aligned channel windows, so there is no fallback for
    those: the wrappers raise)"""
    import torch.nn as nn
    from pcdet.models import convnet
    ops = _ops()
    d = dev()
    torch.manual_seed(5)
    conv = nn.Conv2d(64, 64, 3, padding=1, bias=False)
    pc = convnet.pack_conv_module(conv, None, relu=True)
    for attr in ('w', 'b', 'wino', 'w4f'):
        v = getattr(pc, attr)
        setattr(pc, attr, tuple(t.to(d) if torch.is_tensor(t) else t for t in v) if isinstance(v, tuple) else v.to(d))
    x = torch.from_numpy(_rand(281, (8, 128, 128, 72))).to(d)                     # 8 x 8 x 4 = 256 workgroups: auto picks the fused kernel
    assert pc._use_winograd4f(x, None, 0, 4)
    assert pc._prefer_winograd4h(x)                                                # 64 input channels, 512 half-size items: k_wino4h
    want = pc.run(x, in_ch_off=4)                                                  # fused F(4x4), two workgroups per CU (k_wino4c)
    monkeypatch.setattr(convnet, 'WINOGRAD4H', '0')
    assert pc._use_winograd4f(x, None, 0, 4) and not pc._prefer_winograd4h(x)
    want8 = pc.run(x, in_ch_off=4)                                                 # fused F(4x4), one eight-wave workgroup per CU
    np.testing.assert_allclose(want8.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=2e-4)
    monkeypatch.setattr(convnet, 'WINOGRAD4H', 'auto')
    monkeypatch.setattr(convnet, 'WINOGRAD4F_MAX_INPUT_BYTES', 1 << 20)
    assert not pc._use_winograd4f(x, None, 0, 4)
    got = pc.run(x, in_ch_off=4)                                                   # F(2x2): no exception
    torch.cuda.synchronize()
    ref = F.relu(F.conv2d(x[..., 4:68].permute(0, 3, 1, 2).cpu(), conv.weight.detach(), None, padding=1))
    np.testing.assert_allclose(got.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(want.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy(), rtol=2e-4, atol=2e-4)


# ---------------------------------------------------------------------------------------------------------------------
# Winograd F(4x4, 3x3) through memory (input transform, 36 batched MFMA GEMMs, output transform): the wide layers
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('cin,cout,h,w,relu,batch', [(128, 256, 16, 16, True, 1), (384, 384, 32, 32, True, 2), (768, 768, 16, 16, True, 1),
                                                      (128, 384, 13, 18, False, 2), (256, 260, 7, 9, True, 1), (32, 4, 5, 6, False, 3),
                                                      (256, 256, 64, 64, True, 1)])
def test_conv3x3_winograd4_matches_torch_cpu(cin, cout, h, w, relu, batch):
    """covers: tiles not a multiple of the 128-row GEMM tile, H / W not multiples of 4 (zero-padded patches, clipped stores), cout below
    and beyond one 128-channel N tile.  F(4x4,3x3) rounding is ~2e-5 of the output scale (fp64 study in csrc/wino4.hip); bar 2e-4 of scale"""
    ops = _ops()
    from pcp_amd import pack
    x = torch.from_numpy(_rand(171, (batch, cin, h, w)))
    wt = torch.from_numpy(_rand(172, (cout, cin, 3, 3), -0.05, 0.05))
    b = torch.from_numpy(_rand(173, (cout,), -0.2, 0.2))
    want = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    if relu:
        want = F.relu(want)
    packed, bp, cpad = pack.pack_conv3x3_winograd4(wt, b)
    d = dev()
    got = ops.conv3x3_winograd4(ops.as_nhwc(x.to(d)), packed.to(d), bp.to(d), cin, cout, cpad, relu=relu)
    torch.cuda.synchronize()
    scale = float(want.abs().max())
    err = float((got.permute(0, 3, 1, 2).cpu().double() - want).abs().max())
    assert err <= 2e-4 * scale, 'max err %.3e vs scale %.3e' % (err, scale)
    # timed entry point: same result, three positive launch durations, the GEMM's flop count
    st = []
    got2 = ops.conv3x3_winograd4(ops.as_nhwc(x.to(d)), packed.to(d), bp.to(d), cin, cout, cpad, relu=relu, stage_times=st)
    assert torch.equal(got, got2)
    tiles = batch * ((h + 3) // 4) * ((w + 3) // 4)
    assert len(st) == 1 and min(st[0][:3]) > 0.0 and st[0][3] == 2.0 * 36 * ((tiles + 127) // 128 * 128) * cin * cpad


def test_conv3x3_winograd4_channel_windows_and_bad_arguments():
    ops = _ops()
    from pcp_amd import pack
    d = dev()
    cin, cout = 128, 256
    wt = torch.from_numpy(_rand(177, (cout, cin, 3, 3), -0.05, 0.05))
    packed, bp, cpad = pack.pack_conv3x3_winograd4(wt, torch.zeros(cout))
    x = torch.from_numpy(_rand(178, (1, 24, 20, 192))).to(d)
    out = torch.full((1, 24, 20, 512), 7.0, device=d)
    ops.conv3x3_winograd4(x, packed.to(d), bp.to(d), cin, cout, cpad, relu=False, out=out, in_ch_off=32, out_ch_off=128)
    torch.cuda.synchronize()
    assert float((out[..., :128] - 7.0).abs().max()) == 0.0 and float((out[..., 384:] - 7.0).abs().max()) == 0.0
    want = F.conv2d(x[..., 32:160].permute(0, 3, 1, 2).cpu().double(), wt.double(), None, padding=1)
    err = float((out[..., 128:384].permute(0, 3, 1, 2).cpu().double() - want).abs().max())
    assert err <= 2e-4 * float(want.abs().max())
    with pytest.raises(RuntimeError):       # cin not a multiple of the GEMM K slice
        ops.conv3x3_winograd4(x, packed.to(d), bp.to(d), 100, cout, cpad, relu=False, out=out)
    with pytest.raises(RuntimeError):       # window offset that breaks 16-byte alignment
        ops.conv3x3_winograd4(x, packed.to(d), bp.to(d), cin, cout, cpad, relu=False, out=out, in_ch_off=2)


def _random_boxes(n, seed, spread=12.0):
    rng = np.random.default_rng(seed)
    b = np.zeros((n, 7), np.float32)
    b[:, 0:2] = rng.uniform(-spread, spread, (n, 2))
    b[:, 2] = rng.uniform(-1, 1, n)
    b[:, 3:6] = rng.uniform(0.8, 5.0, (n, 3))
    b[:, 6] = rng.uniform(-3.14, 3.14, n)
    return b, rng.uniform(0.0, 1.0, n).astype(np.float32)


@pytest.mark.parametrize('n', [0, 1, 63, 64, 65, 700, 4096])
def test_nms_normal_gpu_matches_oracle(n):
    """iou3d_nms_utils.nms_normal_gpu (reference :102-117, nms_normal_kernel iou3d_nms_kernel.cu:314-372: axis-aligned IoU, heading
    ignored) against the oracle's float32 restatement: the keep list is bit exact"""
    from pcdet.ops.iou3d_nms import iou3d_nms_utils as U
    boxes, scores = _random_boxes(n, 900 + n)
    keep, _ = U.nms_normal_gpu(torch.from_numpy(boxes).to(dev()), torch.from_numpy(scores).to(dev()), 0.25)
    want = onms.nms_normal_gpu(boxes, scores, 0.25) if n else np.zeros(0, np.int64)
    assert keep.dtype == torch.int64 and keep.cpu().numpy().tolist() == want.tolist()
    if n >= 63:
        assert 0 < want.shape[0] < n                                                  # the threshold bites


def test_multi_classes_nms_cuts_to_pre_maxsize_for_the_axis_aligned_type_too():
    """the reference cuts every class to topk(NMS_PRE_MAXSIZE) before EITHER NMS type (model_nms_utils.py:50); nms_normal_gpu
    used to swallow pre_maxsize.  700 candidates of one class, PRE_MAXSIZE 200: the result is the reference's own line order --
    topk, NMS on the cut list -- restated with the oracle."""
    from pcdet.config import EasyDict
    from pcdet.models.model_utils import model_nms_utils as M
    boxes, scores = _random_boxes(700, 4321)
    cls = np.stack([scores, 0.5 * scores[::-1]], 1).astype(np.float32)
    for nms_type, oracle_fn in (('nms_normal_gpu', onms.nms_normal_gpu), ('nms_gpu', onms.nms_gpu)):
        cfg = EasyDict(NMS_TYPE=nms_type, NMS_THRESH=0.25, NMS_PRE_MAXSIZE=200, NMS_POST_MAXSIZE=50)
        ps, pl, pb = M.multi_classes_nms(torch.from_numpy(cls).to(dev()), torch.from_numpy(boxes).to(dev()), cfg, score_thresh=0.05)
        want_s, want_l = [], []
        for k in range(2):
            m = cls[:, k] >= 0.05
            sc, bx = cls[m, k], boxes[m]
            idx = np.argsort(-sc, kind='stable')[:200]                               # topk(PRE_MAXSIZE)
            keep = oracle_fn(bx[idx], sc[idx], 0.25)[:50]
            want_s.append(sc[idx][keep])
            want_l.append(np.full(len(keep), k))
        assert ps.cpu().numpy().tolist() == np.concatenate(want_s).tolist() and pl.cpu().numpy().tolist() == np.concatenate(want_l).tolist()
        assert pb.shape[0] == ps.shape[0] > 10


def test_boxes_bev_iou_cpu_takes_cpu_tensors_and_numpy():
    """iou3d_nms_utils.boxes_bev_iou_cpu (reference :12-29): CPU tensors / numpy in, the same kind out, values = the C oracle's (which is
    pinned bit for bit on the reference's own iou3d_cpu.cpp, tests/test_oracle_pins.py)"""
    from pcdet.ops.iou3d_nms import iou3d_nms_utils as U
    a, _ = _random_boxes(37, 5, spread=4.0)
    b, _ = _random_boxes(91, 6, spread=4.0)
    want = onms.iou_matrix(a, b)
    got_t = U.boxes_bev_iou_cpu(torch.from_numpy(a), torch.from_numpy(b))
    got_n = U.boxes_bev_iou_cpu(a, b)
    assert isinstance(got_t, torch.Tensor) and not got_t.is_cuda and isinstance(got_n, np.ndarray)
    np.testing.assert_allclose(got_t.numpy(), want, rtol=0, atol=2e-6)               # sinf / cosf / atan2f ulps (ocml vs glibc)
    assert np.array_equal(got_n, got_t.numpy())
    with pytest.raises(AssertionError):
        U.boxes_bev_iou_cpu(torch.from_numpy(a).to(dev()), torch.from_numpy(b))      # 'Only support CPU tensors', as the reference


def test_integration_b1_stub_runs_verbatim():
    """INTEGRATION.md B1 shows the ctypes module a maintainer drops in as pcdet/ops/iou3d_nms/iou3d_nms_cuda.py.  This test executes
    that block as written (only the library path is pointed at the in-tree build) under the REFERENCE's calling convention
    (iou3d_nms_utils.py:84-117: boxes sorted by the caller, `keep` a LongTensor on the CPU, the count returned) and checks the results
    against the oracle"""
    import re
    import types
    from pcp_amd import lib
    text = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'INTEGRATION.md')).read()
    sec = text[text.index('### B1.'):]
    block = re.search(r'```python\n(.*?)```', sec, re.S).group(1)
    assert 'ctypes.CDLL("libpcp_hip.so")' in block
    mod = types.ModuleType('iou3d_nms_cuda')
    exec(compile(block.replace('ctypes.CDLL("libpcp_hip.so")', 'ctypes.CDLL(%r)' % lib.LIB_PATH), 'INTEGRATION.md#B1', 'exec'), mod.__dict__)
    boxes, scores = _random_boxes(500, 77)
    order = np.argsort(-scores, kind='stable')
    sb = torch.from_numpy(boxes[order]).to(dev()).contiguous()
    for fn, oracle_keep in ((mod.nms_gpu, onms.nms_sorted(boxes[order], 0.2)),
                            (mod.nms_normal_gpu, onms.nms_from_iou(onms.iou_normal_matrix(boxes[order], boxes[order]), 0.2))):
        keep = torch.LongTensor(sb.size(0))
        num_out = fn(sb, keep, 0.2)
        assert not keep.is_cuda and num_out == oracle_keep.shape[0] and keep[:num_out].numpy().tolist() == oracle_keep.tolist()
    a, b = torch.from_numpy(boxes[:40]).to(dev()), torch.from_numpy(boxes[40:140]).to(dev())
    for fn, want in ((mod.boxes_iou_bev_gpu, onms.iou_matrix(boxes[:40], boxes[40:140])),
                     (mod.boxes_overlap_bev_gpu, onms.overlap_matrix(boxes[:40], boxes[40:140]))):
        out = torch.cuda.FloatTensor(torch.Size((40, 100))).zero_()
        fn(a.contiguous(), b.contiguous(), out)
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=2e-5)
    out = torch.zeros((40, 100))
    mod.boxes_iou_bev_cpu(torch.from_numpy(boxes[:40]), torch.from_numpy(boxes[40:140]), out)
    np.testing.assert_allclose(out.numpy(), onms.iou_matrix(boxes[:40], boxes[40:140]), rtol=0, atol=2e-6)


@pytest.mark.parametrize('fixture', ['g1_ego.npz', 'g1_car.npz'])
def test_decode_bbox_from_heatmap_adapter_equals_the_reference(fixture):
    """pcdet.models.model_utils.centernet_utils.decode_bbox_from_heatmap called the way center_head.py:312-333 calls it (heatmap =
    sigmoid scores, dim = exp'ed sizes, (B, C, H, W) tensors) on the reference's own head maps: the candidates it returns are the ones
    the reference handed to its NMS (fixture keys post_<b>_nms_*), bit for bit on the scores (the kernel reads the activated maps as they
    are: no inverse sigmoid) and to 1e-5 on the boxes (atan2f ulps)"""
    from helpers import assert_same_final_set
    from pcdet.models.model_utils import centernet_utils
    g = load_golden(fixture)
    pp = g['meta']['model']['DENSE_HEAD']['POST_PROCESSING']
    d = dev()
    hm = torch.from_numpy(g['head_hm']).to(d).sigmoid()
    dim = torch.from_numpy(g['head_dim']).to(d).exp()
    rot = torch.from_numpy(g['head_rot']).to(d)
    out = centernet_utils.decode_bbox_from_heatmap(
        heatmap=hm, rot_cos=rot[:, 0:1], rot_sin=rot[:, 1:2], center=torch.from_numpy(g['head_center']).to(d),
        center_z=torch.from_numpy(g['head_center_z']).to(d), dim=dim, point_cloud_range=g['meta']['pc_range'], voxel_size=g['meta']['voxel_size'],
        feature_map_stride=g['meta']['model']['DENSE_HEAD']['TARGET_ASSIGNER_CONFIG']['FEATURE_MAP_STRIDE'], K=pp['MAX_OBJ_PER_SAMPLE'],
        circle_nms=False, score_thresh=pp['SCORE_THRESH'], post_center_limit_range=torch.tensor(pp['POST_CENTER_LIMIT_RANGE']))
    for b in range(2):
        rb, rs = g['post_%d_nms_boxes' % b], g['post_%d_nms_scores' % b]
        gb, gs = out[b]['pred_boxes'].cpu().numpy(), out[b]['pred_scores'].cpu().numpy()
        assert rb.shape[0] > 100 and gb.shape == rb.shape
        # the adapter adds nothing to the caller's scores: every returned score is, bit for bit, one of the heat-map values passed in
        assert bool(np.isin(gs, hm[b].reshape(-1).cpu().numpy()).all())
        assert_same_final_set(rb, rs, gb, gs, tol=1e-5)
        assert bool((out[b]['pred_labels'] == 0).all())
    # the `vel` branch (reference :174-176, heads of the nuScenes CenterPoint configs): two more columns, gathered at the cell of each box --
    # with vel := (column, row) index maps they must reproduce the cell the box centre was decoded from
    Hh, Ww = hm.shape[2], hm.shape[3]
    vel = torch.stack([torch.arange(Ww, device=d).float().view(1, 1, Ww).expand(2, Hh, Ww), torch.arange(Hh, device=d).float().view(1, Hh, 1).expand(2, Hh, Ww)], 1)
    out9 = centernet_utils.decode_bbox_from_heatmap(
        heatmap=hm, rot_cos=rot[:, 0:1], rot_sin=rot[:, 1:2], center=torch.from_numpy(g['head_center']).to(d),
        center_z=torch.from_numpy(g['head_center_z']).to(d), dim=dim, point_cloud_range=g['meta']['pc_range'], voxel_size=g['meta']['voxel_size'],
        feature_map_stride=g['meta']['model']['DENSE_HEAD']['TARGET_ASSIGNER_CONFIG']['FEATURE_MAP_STRIDE'], K=pp['MAX_OBJ_PER_SAMPLE'], vel=vel,
        circle_nms=False, score_thresh=pp['SCORE_THRESH'], post_center_limit_range=torch.tensor(pp['POST_CENTER_LIMIT_RANGE']))
    stride, vs, pr = g['meta']['model']['DENSE_HEAD']['TARGET_ASSIGNER_CONFIG']['FEATURE_MAP_STRIDE'], g['meta']['voxel_size'], g['meta']['pc_range']
    ctr = torch.from_numpy(g['head_center']).to(d)
    for b in range(2):
        bx = out9[b]['pred_boxes']
        assert bx.shape[1] == 9 and torch.equal(bx[:, :7], out[b]['pred_boxes'])
        col, row = bx[:, 7].long(), bx[:, 8].long()
        x_back = (col.float() + ctr[b, 0, row, col]) * stride * vs[0] + pr[0]
        assert float((x_back - bx[:, 0]).abs().max()) < 1e-4
    with pytest.raises(AssertionError):                           # the reference's own `assert False, 'not checked yet'` (:158-160)
        centernet_utils.decode_bbox_from_heatmap(heatmap=hm, rot_cos=rot[:, 0:1], rot_sin=rot[:, 1:2], center=ctr, center_z=ctr[:, :1], dim=dim,
                                                 point_cloud_range=pr, voxel_size=vs, feature_map_stride=stride, K=10, circle_nms=True,
                                                 post_center_limit_range=torch.tensor(pp['POST_CENTER_LIMIT_RANGE']))


@pytest.mark.parametrize('pixels,n_maps', [(64, 1), (1000, 3), (4096 + 17, 6), (200, 8)])
def test_disco_weight_fuse_matches_torch_cpu(pixels, n_maps):
    """pcp_disco_weight_fuse (pixel weightor on cat[ego, map_a] for every map, softmax over the maps, weighted sum: one launch) against a
    plain torch fp32 CPU evaluation of v2x_fusion_disco.py:8-26,107-115 with BatchNorm already folded; ragged pixel counts (tiles of 64),
    1 .. 8 maps, a padded pixel stride.  fp32 MFMA products, fp32 accumulation: 2e-5 of the scale"""
    ops = _ops()
    d = dev()
    C, ld = 128, 132
    maps = [torch.from_numpy(_rand(300 + a, (pixels, ld), -1.0, 1.0)) for a in range(n_maps)]
    w1 = torch.from_numpy(_rand(311, (64, 2 * C), -0.2, 0.2))
    b1 = torch.from_numpy(_rand(312, (64,), -0.2, 0.2))
    w2 = torch.from_numpy(_rand(313, (16, 64), -0.3, 0.3))
    b2 = torch.from_numpy(_rand(314, (16,), -0.2, 0.2))
    w3 = torch.from_numpy(_rand(315, (16,), -0.5, 0.5))
    b3 = torch.from_numpy(_rand(316, (1,), 0.0, 0.3))
    logits = []
    for a in range(n_maps):
        x = torch.cat([maps[0][:, :C], maps[a][:, :C]], dim=1)
        h1 = F.relu(x @ w1.t() + b1)
        h2 = F.relu(h1 @ w2.t() + b2)
        logits.append(F.relu(h2 @ w3 + b3))
    logits = torch.stack(logits, dim=1)                                   # (pixels, n_maps)
    sm = torch.softmax(logits, dim=1)
    want = sum(sm[:, a:a + 1] * maps[a][:, :C] for a in range(n_maps))
    out = torch.full((pixels, C + 4), 7.0, device=d)
    lg = torch.full((pixels, 12), -1.0, device=d)
    ops.disco_weight_fuse([m.to(d) for m in maps], w1.to(d), b1.to(d), w2.to(d), b2.to(d), w3.to(d), b3.to(d), C, out, logits=lg)
    torch.cuda.synchronize()
    assert float((out[:, C:] - 7.0).abs().max()) == 0.0 and float((lg[:, n_maps:] + 1.0).abs().max()) == 0.0     # windows respected
    assert float(logits.max()) > 0.05                                      # the ReLU'd logits are not all clamped
    np.testing.assert_allclose(lg[:, :n_maps].cpu().numpy(), logits.numpy(), rtol=0, atol=2e-5 * max(1.0, float(logits.abs().max())))
    np.testing.assert_allclose(out[:, :C].cpu().numpy(), want.numpy(), rtol=0, atol=2e-5)
    out2 = torch.empty_like(out)
    ops.disco_weight_fuse([m.to(d) for m in maps], w1.to(d), b1.to(d), w2.to(d), b2.to(d), w3.to(d), b3.to(d), C, out2)
    assert torch.equal(out2[:, :C], out[:, :C])                            # deterministic, logits output optional


def test_column_ids_equals_torch_unique():
    """agent ids present in the cloud (bev_maker.py:153-156: torch.unique(points[:, -1].long())): presence-mask kernel == torch.unique
    of the truncated column; ids outside 0..63 raise (no torch fallback on the product path); the empty cloud gives no ids"""
    from pcp_amd import lib
    ops = _ops()
    d = dev()
    g = torch.Generator().manual_seed(3)
    pts = torch.rand((5000, 7), generator=g)
    pts[:, -1] = torch.tensor([0, 2, 5, 63])[torch.randint(0, 4, (5000,), generator=g)].float()
    assert ops.column_ids(pts.to(d), -1).tolist() == [0, 2, 5, 63]
    pts[17, -1] = 3.5                                           # .long() truncates: 3
    pts[18, -1] = -0.25                                         # -> 0
    assert ops.column_ids(pts.to(d), -1).tolist() == torch.unique(pts[:, -1].long()).tolist() == [0, 2, 3, 5, 63]
    pts[99, -1] = -1.0
    with pytest.raises(lib.PcpError):
        ops.column_ids(pts.to(d), -1)
    pts[99, -1] = 64.0
    with pytest.raises(lib.PcpError):
        ops.column_ids(pts.to(d), -1)
    assert ops.column_ids(torch.zeros((0, 7), device=d), -1).tolist() == []


def test_gather_detections_equals_the_reference_python_tail():
    """pcp_gather_detections vs the per-frame indexing of center_head.py:335-357 (boxes[keep], scores[keep],
    class_id_mapping[labels[keep]] + 1, cat over heads): two heads, ragged counts incl. 0 and the full keep_max, bit exact"""
    ops = _ops()
    d = dev()
    B, k, keep_max = 3, 50, (7, 12)
    g = torch.Generator().manual_seed(5)
    heads, want = [], [dict(b=[], s=[], l=[]) for _ in range(B)]
    cmaps = [torch.tensor([2, 0], dtype=torch.int32), torch.tensor([1], dtype=torch.int32)]
    counts = [[0, 7, 3], [12, 0, 5]]
    for hi in range(2):
        boxes = torch.rand((B, k, 7), generator=g)
        scores = torch.rand((B, k), generator=g)
        labels = torch.randint(0, len(cmaps[hi]), (B, k), generator=g, dtype=torch.int32)
        keep = torch.stack([torch.randperm(k, generator=g)[:keep_max[hi]] for _ in range(B)]).int()
        cnt = torch.tensor(counts[hi], dtype=torch.int32)
        heads.append(dict(boxes=boxes.to(d), scores=scores.to(d), labels=labels.to(d), keep=keep.to(d), keep_count=cnt.to(d),
                          class_map=cmaps[hi].to(d)))
        for b in range(B):
            sel = keep[b, :counts[hi][b]].long()
            want[b]['b'].append(boxes[b, sel])
            want[b]['s'].append(scores[b, sel])
            want[b]['l'].append(cmaps[hi][labels[b, sel].long()].long() + 1)
    ob, os_, ol, cnt = ops.gather_detections(heads, B)
    torch.cuda.synchronize()
    assert ol.dtype == torch.int64 and tuple(ob.shape) == (B, sum(keep_max), 7)
    for b in range(B):
        n = counts[0][b] + counts[1][b]
        assert int(cnt[b]) == n
        assert torch.equal(ob[b, :n].cpu(), torch.cat(want[b]['b'])) and torch.equal(os_[b, :n].cpu(), torch.cat(want[b]['s']))
        assert torch.equal(ol[b, :n].cpu(), torch.cat(want[b]['l']))
        assert float(ob[b, n:].abs().max()) == 0.0 if n < sum(keep_max) else True


@pytest.mark.parametrize('n_points,batch,xy_half,grid_n,cout', [(3000, 2, 13.1, 128, 64), (200, 1, 13.1, 128, 64), (40000, 2, 13.1, 128, 64),
                                                                   (2500, 3, 6.6, 66, 32), (0, 1, 13.1, 128, 64)])
def test_sparse_first_layer_equals_dense_conv_on_the_canvas(n_points, batch, xy_half, grid_n, cout):
    """pcp_sparse_conv3x3_s2 (pillar list + cell -> rank table) vs the dense path it replaces: PFN canvas -> ZeroPad2d(1) + 3x3 stride-2
    conv + bias + ReLU (torch CPU on the canvas the PFN wrote).  Cases: 18 % occupancy, nearly empty tiles, crowded cells (> 32 occupied
    rows per tap and tile -> several chunks), a grid that is not a multiple of the 8 x 16 tile (66 -> 33 outputs), cout 32, empty cloud."""
    ops = _ops()
    from pcp_amd import pack
    d = dev()
    half = 0.1 * grid_n
    rng = [-half, -half, -8.0, half, half, 0.0]
    clouds = [synth.agent_cloud(31 + f, max(n_points, 1), 'car', xy_half=xy_half) for f in range(batch)]
    pts_np = synth.collate(clouds)
    if n_points == 0:
        pts_np = pts_np[:0]
    pts = torch.from_numpy(pts_np).to(d)
    grid = ops.make_grid(rng, [0.2, 0.2, 8.0], [grid_n, grid_n, 1], batch)
    g = torch.Generator().manual_seed(11)
    w0 = ((torch.rand(32, 11, generator=g) - 0.5) * 0.6).to(d)
    b0 = ((torch.rand(32, generator=g) - 0.5) * 0.2).to(d)
    w1 = ((torch.rand(64, 64, generator=g) - 0.5) * 0.3).to(d)
    b1 = ((torch.rand(64, generator=g) - 0.5) * 0.2).to(d)
    canvas = torch.zeros((batch, grid_n, grid_n, 64), device=d)
    vox = ops.voxelize(pts, grid, want_inverse=False, want_counts=False)
    pf = torch.zeros((max(pts.shape[0], 1), 64), device=d)
    if pts.shape[0]:
        ops.pfn_scatter(pts, vox, 5, w0, b0, w1, b1, canvas=canvas, pillar_features=pf)
    wc = (torch.rand(cout, 64, 3, 3, generator=g) - 0.5) * 0.1
    bc = (torch.rand(cout, generator=g) - 0.5) * 0.2
    wp, bp = pack.pack_conv3x3_sparse_s2(wc, bc)
    got = ops.sparse_conv3x3_s2(pf, vox, wp.to(d), bp.to(d), cout, relu=True)
    torch.cuda.synchronize()
    want = F.relu(F.conv2d(canvas.permute(0, 3, 1, 2).cpu().double(), wc.double(), bc.double(), stride=2, padding=1))
    assert tuple(got.shape) == (batch, (grid_n - 1) // 2 + 1, (grid_n - 1) // 2 + 1, cout)
    err = float((got.permute(0, 3, 1, 2).cpu().double() - want).abs().max())
    assert err <= 2e-5 * max(float(want.abs().max()), 1.0), err
    if pts.shape[0]:
        assert int(vox.counters[0]) > 0 and float(want.abs().max()) > 0.1


def test_grouped_small_head_conv_matches_torch():
    ops = _ops()
    d = dev()
    B, H, W = 2, 20, 28
    ks = [2, 1, 3, 2, 1]
    offs = [0, 2, 3, 6, 8, 9]
    x = torch.from_numpy(_rand(91, (B, 320, H, W)))
    ws = [torch.from_numpy(_rand(92 + i, (k, 64, 3, 3), -0.1, 0.1)) for i, k in enumerate(ks)]
    bs = torch.from_numpy(_rand(99, (9,), -0.5, 0.5))
    want = torch.cat([F.conv2d(x[:, 64 * i:64 * i + 64], ws[i], bs[offs[i]:offs[i + 1]], padding=1) for i in range(5)], 1)
    wg = torch.cat(ws, 0).permute(0, 2, 3, 1).reshape(9, 9, 64).contiguous()
    out = torch.full((B, H, W, 16), -3.0, device=d)
    ops.conv3x3_grouped_small(ops.as_nhwc(x.to(d)), wg.to(d), bs.to(d), offs, out)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out[..., :9].permute(0, 3, 1, 2).cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4)
    assert float((out[..., 9:] + 3.0).abs().max()) == 0.0


def test_fused_point_head_matches_unfused_ops_and_torch():
    ops = _ops()
    d = dev()
    B, H, W, C = 2, 32, 32, 384
    bev = torch.from_numpy(_rand(101, (B, C, H, W)))
    pts_np = synth.collate([synth.agent_cloud(5, 1500, 'car', xy_half=13.1), synth.agent_cloud(6, 777, 'car', xy_half=13.1)])
    pts = torch.from_numpy(pts_np)
    rng = [-12.8, -12.8, -8.0, 12.8, 12.8, 0.0]
    w1, b1 = torch.from_numpy(_rand(102, (32, C), -0.05, 0.05)), torch.from_numpy(_rand(103, (32,), -0.1, 0.1))
    w2, b2 = torch.from_numpy(_rand(104, (C, 32), -0.1, 0.1)), torch.from_numpy(_rand(105, (C,), -0.1, 0.1))
    wh, bh = torch.from_numpy(_rand(106, (8, C), -0.05, 0.05)), torch.from_numpy(_rand(107, (8,), -0.1, 0.1))
    want_pf, _ = obev.sample_point_features(bev, pts, rng, [0.8, 0.8])
    h1 = torch.relu(want_pf @ w1.t() + b1)
    final = torch.relu(h1 @ w2.t() + b2) + want_pf
    want_head = final @ wh.t() + bh
    pix = [np.float32(0.2) * 4, np.float32(0.2) * 4]
    pf, head = ops.hunter_point_head(ops.as_nhwc(bev.to(d)), pts.to(d), rng[:2], pix, w1.to(d), b1.to(d), w2.to(d), b2.to(d), wh.to(d),
                                     bh.to(d), channels=C)
    torch.cuda.synchronize()
    np.testing.assert_allclose(pf.cpu().numpy(), want_pf.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(head.cpu().numpy(), want_head.numpy(), rtol=1e-4, atol=1e-4)
    # the sampled rows are bitwise those of the stand-alone sampler
    pf2 = ops.bev_sample_bilinear(ops.as_nhwc(bev.to(d)), pts.to(d), rng[:2], pix)
    torch.cuda.synchronize()
    assert torch.equal(pf, pf2)
    # extended form: sorted visiting order + fused flow correction / re-sampling == the unfused three-op sequence, bit for bit
    bev_d = ops.as_nhwc(bev.to(d))
    wts = [t.to(d) for t in (w1, b1, w2, b2, wh, bh)]
    wts[5] = wts[5].clone()
    hd = head.float()
    wts[5][2] += float((torch.maximum(hd[:, 0], hd[:, 1]) - hd[:, 2]).median())      # about half of the rows become dynamic foreground
    p_ref = pts.to(d).clone()
    pf_a, head_a = ops.hunter_point_head(bev_d, p_ref, rng[:2], pix, *wts, channels=C)
    dyn_a = ops.hunter_apply_flow(p_ref, head_a, 0.3)
    ops.bev_sample_bilinear(bev_d, p_ref, rng[:2], pix, out=pf_a, row_mask=dyn_a, channels=C)
    grid = ops.make_grid(rng, [0.2, 0.2, 8.0], [128, 128, 1], B)
    p_fused = pts.to(d).clone()
    order = ops.voxelize_row_order(ops.voxelize(p_fused, grid, want_inverse=False, want_counts=False))
    assert sorted(order.cpu().tolist()) == list(range(p_fused.shape[0]))                 # a permutation of ALL rows
    pf_b, head_b, dyn_b = ops.hunter_point_head(bev_d, p_fused, rng[:2], pix, *wts, channels=C, order=order, flow_thresh=0.3)
    torch.cuda.synchronize()
    assert 50 < int(dyn_a.sum()) < p_ref.shape[0] - 50
    assert torch.equal(dyn_a, dyn_b) and torch.equal(head_a, head_b) and torch.equal(p_ref, p_fused) and torch.equal(pf_a, pf_b)


def test_small_n_conv_over_many_channels_matches_torch():
    """the 768 -> 2 weight conv of HunterJr through the grouped small-N kernel (one group, 12 chunks of 64 channels)"""
    ops = _ops()
    d = dev()
    B, H, W, C = 1, 17, 23, 768
    x = torch.from_numpy(_rand(111, (B, C, H, W)))
    wt = torch.from_numpy(_rand(112, (2, C, 3, 3), -0.05, 0.05))
    bs = torch.from_numpy(_rand(113, (2,), -0.5, 0.5))
    want = F.conv2d(x, wt, bs, padding=1)
    out = torch.empty((B, H, W, 2), device=d)
    ops.conv3x3_grouped_small(ops.as_nhwc(x.to(d)), wt.permute(0, 2, 3, 1).reshape(2, 9, C).contiguous().to(d), bs.to(d), [0, 2], out)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.permute(0, 3, 1, 2).cpu().numpy(), want.numpy(), rtol=1e-4, atol=2e-4)


# ---- exchange producer / consumer ---------------------------------------------------------------------

def test_points_in_boxes_and_modar_ingest_match_oracle_and_golden():
    ops = _ops()
    from helpers import load_golden
    from oracle import exchange as oex
    from pcdet.ops.roiaware_pool3d import roiaware_pool3d_utils
    g = load_golden('g8_exchange.npz')
    fg = torch.from_numpy(g['foreground']).cuda()
    modar = torch.from_numpy(g['modar']).cuda()
    idx = roiaware_pool3d_utils.points_in_boxes_gpu(fg[None, :, :3].contiguous(), modar[None, :, :7].contiguous())[0].cpu().numpy()
    want = oex.points_in_boxes(g['foreground'][:, :3], g['modar'][:, :7])
    # a point within float rounding of a box face may flip (cosf/sinf ulps): none here
    assert np.array_equal(idx, want) and np.array_equal(idx, g['box_idx'])
    # two frames with different box sets in one call
    pts2 = torch.stack([fg[:1500, :3], fg[1500:3000, :3]]).contiguous()
    bx2 = torch.stack([modar[:20, :7], modar[20:40, :7]]).contiguous()
    idx2 = ops.points_in_boxes(pts2, bx2).cpu().numpy()
    assert np.array_equal(idx2[0], oex.points_in_boxes(g['foreground'][:1500, :3], g['modar'][:20, :7]))
    assert np.array_equal(idx2[1], oex.points_in_boxes(g['foreground'][1500:, :3], g['modar'][20:, :7]))
    rows = ops.modar_ingest(modar, fg, g['pose'], float(g['max_sweep_idx'])).cpu().numpy()
    np.testing.assert_allclose(rows, g['rows'], rtol=0, atol=5e-6)
    rows2 = ops.modar_ingest(modar, None, g['pose'], float(g['max_sweep_idx'])).cpu().numpy()
    np.testing.assert_allclose(rows2, g['rows_no_foreground'], rtol=0, atol=5e-6)
    assert ops.modar_ingest(modar[:0], fg, g['pose'], 10.0).shape == (0, 13)


def test_hunter_foreground_rows_compaction():
    ops = _ops()
    from oracle import exchange as oex
    n = 70001
    pts = np.zeros((n, 8), dtype=np.float32)
    pts[:, 0] = np.floor(synth.uniform(5, 1, n, 0, 3.999))
    pts[:, 1:] = synth.uniform(5, 2, n * 7, -50, 50).reshape(n, 7)
    head = synth.uniform(5, 3, n * 8, -4, 4).reshape(n, 8).astype(np.float32)
    want_rows, want_b = oex.foreground_rows(pts, head[:, :3], head[:, 3:6])
    rows, rb = ops.hunter_foreground_rows(torch.from_numpy(pts).cuda(), torch.from_numpy(head).cuda(), 0.3)
    assert rows.shape == want_rows.shape and np.array_equal(rb.cpu().numpy(), want_b)       # same rows, same ORDER
    np.testing.assert_allclose(rows.cpu().numpy(), want_rows, rtol=0, atol=1e-6)
    none, nb = ops.hunter_foreground_rows(torch.from_numpy(pts).cuda(), torch.full((n, 8), 9.0).cuda(), 0.3)
    assert none.shape[0] == 0 and nb.shape[0] == 0


def test_topk_boxes_selection_ties_and_masks():
    """pcp_topk_boxes: k largest keys, descending, ties to the LOWER index; count = min(k, number of valid keys)"""
    ops = _ops()
    B, N = 3, 50000
    sc = synth.uniform(9, 1, B * N, 0.0, 1.0).reshape(B, N).astype(np.float32)
    sc[0, 100:140] = 0.987654                                   # a plateau that straddles the k-th place
    sc[0, 40000:40040] = 0.987654
    sc[1] = np.where(sc[1] > 0.999, sc[1], 0.0)                 # frame 1: only ~50 valid candidates (< k)
    sc[2] = 0.0                                                  # frame 2: nothing passes the score mask
    keys = np.where(sc > 0, sc.view(np.uint32) + 1, 0).astype(np.uint32)
    boxes = synth.uniform(9, 2, B * N * 7, -1, 1).reshape(B, N, 7).astype(np.float32)
    labels = (np.arange(B * N) % 3).reshape(B, N).astype(np.int32)
    k = 4096
    ob, os_, ol, oi, cnt = ops.topk_boxes(torch.from_numpy(keys.view(np.int32)).cuda(), torch.from_numpy(labels).cuda(),
                                          torch.from_numpy(boxes).cuda(), k)
    cnt = cnt.cpu().numpy()
    for b in range(B):
        valid = np.nonzero(sc[b] > 0)[0]
        order = valid[np.lexsort((valid, -sc[b, valid].astype(np.float64)))][:k]
        assert cnt[b] == order.shape[0]
        got = oi[b, :cnt[b]].cpu().numpy()
        assert np.array_equal(got, order), b
        assert np.array_equal(os_[b, :cnt[b]].cpu().numpy(), sc[b, order])
        assert np.array_equal(ob[b, :cnt[b]].cpu().numpy(), boxes[b, order])
        assert np.array_equal(ol[b, :cnt[b]].cpu().numpy(), labels[b, order])
    assert cnt[2] == 0 and 0 < cnt[1] < k and cnt[0] == k


@pytest.mark.parametrize('cin,cout,h,w,stride,relu,batch', [(64, 64, 40, 56, 1, True, 2), (128, 128, 33, 17, 1, False, 1), (384, 64, 32, 32, 1, True, 2),
                                                            (64, 128, 64, 48, 2, True, 2), (16, 200, 21, 35, 1, True, 1)])
def test_conv3x3_bf16x3_optin_matches_torch_cpu(cin, cout, h, w, stride, relu, batch):
    """opt-in split-bf16 arithmetic (three bf16 MFMAs per product, fp32 accumulate): measured error ~1e-5 relative to the output scale,
    asserted at 1e-4 (10x inside the path's 1e-3 bar; the fp32 kernels are asserted at the same 1e-4)."""
    ops = _ops()
    d = dev()
    from pcp_amd import pack
    x = torch.from_numpy(_rand(300 + cin, (batch, cin, h, w)))
    wt = torch.from_numpy(_rand(301 + cout, (cout, cin, 3, 3), -0.08, 0.08))
    bs = torch.from_numpy(_rand(302, (cout,), -0.3, 0.3))
    want = F.conv2d(F.pad(x, (1, 1, 1, 1)), wt, bs, stride=stride)
    if relu:
        want = torch.relu(want)
    p3, b3, cp3 = pack.pack_conv3x3_bf16x3(wt, bs)
    out = ops.conv3x3_bf16x3(ops.as_nhwc(x.to(d)), p3.to(d), b3.to(d), cin, cout, cp3, stride=stride, relu=relu)
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2).cpu()
    err = float((got - want).abs().max()) / float(want.abs().max())
    assert err < 1e-4, err


# ---------------------------------------------------------------------------------------------------------------------
# a13: ego -> agent point transform, bit for bit the reference's (bev_maker.py:172-179) at BASELINE's full size
# ---------------------------------------------------------------------------------------------------------------------
def test_select_transform_points_bit_equal_to_the_reference():
    """tests/golden/g2_disco_full.npz holds, per remote agent, the SHA-256 of the xyz the reference's frozen chain received after
    `agent_points[mask, 1:4] @ R^T + t` (torch CPU: an FMA chain then a rounded add).  pcp_select_transform_points must produce the
    same bits for all 60 000 rows of every agent -- which is what lets the DiscoNet map tests demand 1e-3 on every pixel."""
    import hashlib
    from pcp_amd import synth
    ops = _ops()
    g = load_golden('g2_disco_full.npz')
    agents = (0, 1, 2, 3, 4, 5)
    clouds = []
    for a in agents:
        c = synth.agent_cloud(agent=a, n_points=60000, layout='disco')
        c[:, -1] = float(a)
        clouds.append(c)
    pts = torch.from_numpy(synth.collate([np.concatenate(clouds, axis=0)])).to(dev())
    for a in agents:
        if a == 1:
            continue
        T = torch.from_numpy(g['pose_%d' % a]).float().numpy()                  # the reference casts the float64 pose to float32 first
        pose = np.concatenate([T[:3, :3], T[:3, 3:4]], axis=1).reshape(1, 12)
        out = ops.select_transform_points(pts, -1, a, pose, np.array([True]))
        rows = out[out[:, 0] >= 0].cpu().numpy()
        assert rows.shape[0] == int(g['car_agent_%d_rows' % a]) == 60000
        assert np.array_equal(rows[:8, 1:4], g['car_agent_%d_xyz_head' % a])
        assert hashlib.sha256(np.ascontiguousarray(rows[:, 1:4]).tobytes()).hexdigest() == str(g['car_agent_%d_xyz_sha' % a])


def _compact_case(n, stride=7, batch=3, agents_in_cloud=(0, 1, 2, 4, 5), seed=11):
    g = torch.Generator().manual_seed(seed)
    pts = (torch.rand((n, stride), generator=g) * 80.0 - 40.0)
    if n:
        pts[:, 0] = torch.randint(0, batch, (n,), generator=g).float()
        pts[:, -1] = torch.tensor(agents_in_cloud)[torch.randint(0, len(agents_in_cloud), (n,), generator=g)].float()
    return pts


@pytest.mark.parametrize('n', [0, 1, 1000, 1024, 5000, 262144 + 77])
def test_select_transform_compact_is_the_cat_of_the_masked_selections(n):
    """pcp_select_transform_compact against the round-2 kernel it replaces (pcp_select_transform_points, itself bit-equal to the
    reference's `points[mask] @ R^T + t`): slot s's segment == the kept rows of the per-agent masked copy, same order, same bits; the tail
    carries frame index -1; slot_start / total are the exact counts; an agent absent from a frame loses that frame's rows (bev_maker.py:172-179);
    the row counts of pcp_column_id_counts equal torch's"""
    ops = _ops()
    d = dev()
    batch = 3
    pts = _compact_case(n, batch=batch)
    agents = [0, 2, 4, 5, 3]                                               # 3 has no rows; 1 (the ego) is not selected
    rng = np.random.default_rng(5)
    poses = rng.standard_normal((len(agents), batch, 12)).astype(np.float32)
    present = np.ones((len(agents), batch), np.uint8)
    present[1, 2] = 0                                                      # agent 2 absent from frame 2
    present[3, 0] = 0
    dp = pts.to(d)
    ids, rows = ops.column_id_counts(dp, -1)
    want_ids = torch.unique(pts[:, -1].long()).tolist() if n else []
    assert ids.tolist() == want_ids
    for a in want_ids:
        assert rows[a] == int((pts[:, -1].long() == a).sum())
    cap = sum(rows.get(a, 0) for a in agents)
    slot_start = torch.full((len(agents) + 1,), -7, dtype=torch.int32, device=d)
    out = ops.select_transform_compact(dp, -1, agents, poses, present, cap, slot_start=slot_start)
    torch.cuda.synchronize()
    out = out[:cap].cpu().numpy()
    starts = slot_start.cpu().numpy()
    off = 0
    for s, a in enumerate(agents):
        ref = ops.select_transform_points(dp, -1, float(a), poses[s], present[s], batch_offset=s * batch).cpu().numpy() if n else np.zeros((0, 7), np.float32)
        ref = ref[ref[:, 0] >= 0]
        assert starts[s] == off
        assert np.array_equal(out[off:off + ref.shape[0]], ref), (s, a)
        off += ref.shape[0]
    assert starts[-1] == off <= cap
    assert np.all(out[off:, 0] == -1.0) and np.all(out[off:, 1:] == 0.0)


def test_select_transform_compact_rejects_bad_arguments():
    from pcp_amd import lib
    ops = _ops()
    d = dev()
    pts = _compact_case(100).to(d)
    with pytest.raises(lib.PcpError):                                      # more than 8 slots
        ops.select_transform_compact(pts, -1, list(range(9)), np.zeros((9, 3, 12), np.float32), np.ones((9, 3), np.uint8), 100)
    with pytest.raises(lib.PcpError):                                      # pose table beyond the kernel-argument budget (slots x frames > 64)
        ops.select_transform_compact(pts, -1, list(range(8)), np.zeros((8, 9, 12), np.float32), np.ones((8, 9), np.uint8), 100)


@pytest.mark.parametrize('n', [3000, 200000])
def test_voxelize_cells_ready_equals_the_full_pillariser(n):
    """the compaction emits the cell id + histogram of every row it writes; pcp_voxelize_cells_ready (the pillariser minus its first pass)
    must then leave the same pillar list as pcp_voxelize on the compacted cloud: voxel_coords, counters, and the per-pillar features the PFN
    computes from the buckets (order inside a bucket is free: means are fixed point, maxima commute), bit for bit"""
    ops = _ops()
    d = dev()
    batch = 2
    pts = _compact_case(n, batch=batch, seed=23)
    pts[:, 1:3] = pts[:, 1:3] * 0.2                                       # inside an 80 x 80 cell grid around the origin (+ some rows outside)
    agents = [0, 2, 5]
    rng = np.random.default_rng(9)
    poses = np.zeros((len(agents), batch, 12), np.float32)
    for s in range(len(agents)):
        for b in range(batch):
            yaw = rng.uniform(-1, 1)
            R = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]], np.float32)
            poses[s, b] = np.concatenate([R, rng.uniform(-3, 3, (3, 1)).astype(np.float32)], 1).reshape(-1)
    present = np.ones((len(agents), batch), np.uint8)
    dp = pts.to(d)
    _ids, rows = ops.column_id_counts(dp, -1)
    cap = sum(rows[a] for a in agents)
    grid = ops.make_grid([-8.0, -8.0, -3.0, 8.0, 8.0, 1.0], [0.2, 0.2, 4.0], [80, 80, 1], batch * len(agents))
    ws = ops.voxelize_workspace(grid, cap, d)
    out = ops.select_transform_compact(dp, -1, agents, poses, present, cap, vox_grid=grid, vox_workspace=ws)
    fast = ops.voxelize(out[:cap], grid, want_inverse=False, want_counts=True, workspace=ws, cells_ready=True)
    full = ops.voxelize(out[:cap].clone(), grid, want_inverse=False, want_counts=True)
    torch.cuda.synchronize()
    cf, cu = fast.counters.cpu().numpy(), full.counters.cpu().numpy()
    assert np.array_equal(cf, cu) and cf[0] > 0
    P = int(cf[0])
    assert torch.equal(fast.voxel_coords[:P], full.voxel_coords[:P]) and torch.equal(fast.unq_cnt[:P], full.unq_cnt[:P])
    w0 = torch.from_numpy(_rand(31, (32, 11), -0.3, 0.3)).to(d)
    b0 = torch.from_numpy(_rand(32, (32,), -0.1, 0.1)).to(d)
    w1 = torch.from_numpy(_rand(33, (64, 64), -0.2, 0.2)).to(d)
    b1 = torch.from_numpy(_rand(34, (64,), -0.1, 0.1)).to(d)
    pf = []
    for vox in (fast, full):
        f = torch.zeros((cap, 64), device=d)
        ops.pfn_scatter(out[:cap], vox, 5, w0, b0, w1, b1, canvas=None, pillar_features=f)
        pf.append(f[:P])
    torch.cuda.synchronize()
    assert torch.equal(pf[0], pf[1])


def test_modar_ingest_batched_equals_the_per_pair_ingestion():
    """pcp_modar_ingest_batched (all (frame, remote agent) groups in one device-driven call, padded detections, device counts) against
    the per-pair oracle on the reference's MoDAR + foreground rows of tests/golden/g10_lately_chain.npz; ragged groups, an empty group and
    a group without foreground rows included"""
    ops = _ops()
    from oracle import exchange as oex
    g = load_golden('g10_lately_chain.npz')
    meta = g['meta']
    keys = ['%d_%d' % (f, s) for f in range(meta['frames']) for s in range(len(meta['remote_agents']))]
    G, M = len(keys), 83
    counts = [g['modar_' + k].shape[0] for k in keys]
    counts[2], counts[5] = 40, 0                                  # ragged + empty group
    fg_keep = [True] * G
    fg_keep[7] = False                                            # a group that sent no foreground rows
    boxes = np.zeros((G, M, 7), np.float32)
    scores = np.zeros((G, M), np.float32)
    labels = np.zeros((G, M), np.int64)
    fg_rows, fg_group = [], []
    for i, k in enumerate(keys):
        m = g['modar_' + k][:counts[i]]
        boxes[i, :counts[i]], scores[i, :counts[i]], labels[i, :counts[i]] = m[:, :7], m[:, 7], m[:, 8].astype(np.int64)
        if fg_keep[i]:
            fg_rows.append(g['foreground_' + k])
            fg_group.append(np.full(g['foreground_' + k].shape[0], i, np.int32))
    fg_rows, fg_group = np.concatenate(fg_rows, 0), np.concatenate(fg_group, 0)
    cap = fg_rows.shape[0] + 100                                  # capacity beyond the count: the tail must be ignored
    fg_buf = np.full((cap, fg_rows.shape[1]), 1e9, np.float32)
    fg_buf[:fg_rows.shape[0]] = fg_rows
    grp_buf = np.full(cap, 3, np.int32)
    grp_buf[:fg_group.shape[0]] = fg_group
    poses = np.stack([np.asarray(g['target_se3_lidar_' + k], np.float64)[:3, :4].reshape(-1) for k in keys], 0)
    sweeps = np.array([float(g['max_sweep_idx_%d' % int(k[0])]) for k in keys], np.float32)
    frames = np.array([int(k[0]) for k in keys], np.int32)
    d = dev()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)
    out = ops.modar_ingest_batched((t(boxes), t(scores), t(labels), t(np.array(counts, np.int32))), t(fg_buf), t(grp_buf),
                                   t(np.array([fg_rows.shape[0]], np.int32)), t(poses), t(sweeps), t(frames)).cpu().numpy().reshape(G, M, 14)
    for i, k in enumerate(keys):
        n = counts[i]
        assert bool((out[i, n:, 0] == -1).all()) and bool((out[i, :n, 0] == frames[i]).all())
        if n == 0:
            continue
        want = oex.modar_ingest(g['modar_' + k][:n], g['foreground_' + k] if fg_keep[i] else None, g['target_se3_lidar_' + k], float(sweeps[i]))
        np.testing.assert_allclose(out[i, :n, 1:], want, rtol=0, atol=5e-6)
        if n == g['modar_' + k].shape[0] and fg_keep[i]:
            np.testing.assert_allclose(out[i, :n, 1:], g['ingest_rows_' + k], rtol=0, atol=5e-6)      # = the reference's own rows
    # no foreground at all
    out2 = ops.modar_ingest_batched((t(boxes), t(scores), t(labels), t(np.array(counts, np.int32))), None, None, None, t(poses), t(sweeps),
                                    t(frames)).cpu().numpy().reshape(G, M, 14)
    want = oex.modar_ingest(g['modar_' + keys[0]], None, g['target_se3_lidar_' + keys[0]], float(sweeps[0]))
    np.testing.assert_allclose(out2[0, :counts[0], 1:], want, rtol=0, atol=5e-6)


def test_voxelize_sort_pillar_rows_makes_the_bucket_order_a_function_of_the_input():
    """the pillariser hands the points of a pillar out in atomic order; pcp_voxelize_sort_pillar_rows (training path) sorts every pillar's run
    by point index: the bucket order stays grouped by pillar (pillars ascending), holds every kept point once, ascends inside each pillar and
    is the same for two runs"""
    ops = _ops()
    from pcp_amd import synth
    d = dev()
    rng = np.random.RandomState(7)
    n = 40000
    pts = np.zeros((n, 6), dtype=np.float32)
    pts[:, 0] = rng.randint(0, 2, n)
    pts[:, 1:3] = rng.uniform(-6.0, 6.0, (n, 2))                    # 60 x 60 cells of 0.2 m: ~5 points per pillar and frame
    pts[:, 3] = rng.uniform(-7.0, -1.0, n)
    pts[rng.rand(n) < 0.05, 1] = 500.0                              # out of range rows
    # crowded cells (a LiDAR-like cloud has them next to the sensor): runs longer than one thread sorts alone are ranked by the workgroup --
    # around its limits: 33 (> SORT_SHORT), 300, 2 049 (> one LDS tile), 5 000 (several passes of 1 024 elements)
    at = 0
    for k, (cxk, cyk) in zip((5000, 2049, 300, 33, 32), ((10, 10), (11, 10), (12, 10), (13, 10), (14, 10))):
        pts[at:at + k, 0] = 1
        pts[at:at + k, 1] = -51.2 + 0.2 * cxk + rng.uniform(0.01, 0.19, k)
        pts[at:at + k, 2] = -51.2 + 0.2 * cyk + rng.uniform(0.01, 0.19, k)
        at += k
    pts = pts[rng.permutation(n)]
    points = torch.from_numpy(pts).to(d)
    grid = ops.make_grid([-51.2, -51.2, -8.0, 51.2, 51.2, 0.0], [0.2, 0.2, 8.0], [512, 512, 1], 2)
    orders = []
    for _rep in range(2):
        vox = ops.voxelize(points, grid, want_inverse=True, want_counts=True)
        ops.voxelize_sort_pillar_rows(vox)
        order = ops.voxelize_row_order(vox)
        torch.cuda.synchronize()
        kept = int(vox.counters[1].item())
        orders.append(order[:kept].cpu().numpy().copy())
    o = orders[0]
    assert np.array_equal(o, orders[1])
    inside = np.abs(pts[:, 1]) < 51.2
    assert kept == int(inside.sum()) and np.array_equal(np.sort(o), np.nonzero(inside)[0])
    # pillar id of every point from the CPU: (frame, x cell, y cell) in the reference's merged order
    cx = np.floor((pts[:, 1] + 51.2) / 0.2).astype(np.int64)
    cy = np.floor((pts[:, 2] + 51.2) / 0.2).astype(np.int64)
    merged = pts[:, 0].astype(np.int64) * 512 * 512 + cx * 512 + cy
    m = merged[o]
    assert (np.diff(m) >= 0).all()                                   # grouped by pillar, pillars ascending
    same = np.diff(m) == 0
    assert (np.diff(o)[same] > 0).all() and same.sum() > 10000       # ascending point index inside every pillar


@pytest.mark.parametrize('case', ['car_1x60k', 'batch3_ragged', 'empty', 'all_masked', 'edges', 'ring_crowded'])
@pytest.mark.parametrize('aggregate', [0, 1])
def test_pillar_index_export_rebuilds_what_the_pillariser_would_have_written(case, aggregate, lib_option):
    """pcp_pillar_index_export on the workspace of a pcp_pillarise_rows call WITHOUT index outputs (the pipeline mode) and of a pcp_voxelize call
    (num_raw = 0: no records) == the voxel_coords / unq_inv the same pillariser writes when asked, == the oracle; with the histogram pass's
    LDS pre-aggregation on and off (option vox_aggregate: the pillar lists are bit-identical)."""
    ops = _ops()
    lib_option('vox_aggregate', aggregate)
    B = 1
    if case == 'car_1x60k':
        pts = synth.collate([synth.agent_cloud(0, 60000, 'car')])
    elif case == 'batch3_ragged':
        B = 3
        pts = synth.collate([synth.agent_cloud(1, 5000, 'car'), synth.agent_cloud(2, 17, 'car'), synth.agent_cloud(3, 30001, 'car', dist='ring')])
    elif case == 'ring_crowded':
        B = 2
        pts = synth.collate([np.concatenate([synth.agent_cloud(a, 40000, 'car', dist='ring') for a in range(3)], 0),
                             synth.agent_cloud(9, 60000, 'car', dist='ring')])
    elif case == 'empty':
        pts = np.zeros((0, 8), np.float32)
    elif case == 'all_masked':
        pts = synth.collate([synth.agent_cloud(0, 1000, 'car')])
        pts[:, 1] += 500.0
    else:
        xs = np.array([-51.2, -51.200001, 51.2, 51.199997, 0.0, 0.2, 0.19999999, -0.2, 0.6000000238, 0.6, np.nan, np.inf, -np.inf, 10.0], np.float32)
        pts = np.zeros((xs.shape[0] * 2, 8), np.float32)
        pts[:xs.shape[0], 1], pts[:xs.shape[0], 2] = xs, 1.0
        pts[xs.shape[0]:, 1], pts[xs.shape[0]:, 2] = 1.0, xs
        pts[:, 3] = -1.0
    g = ops.make_grid(PC_RANGE, VOXEL, GRID, B)
    pd = torch.from_numpy(pts).to(dev())
    full = ops.pillarise_rows(pd, g, 5, want_inverse=True, want_coords=True)
    P, kept = int(full.counters[0]), int(full.counters[1])
    bare = ops.pillarise_rows(pd, g, 5)                                   # no index outputs: what bench.py / --fast run
    coords, inv, cnt, slot_rank, slot_row = ops.pillar_index_export(bare, want_records=True)
    assert int(cnt[0]) == P and int(cnt[1]) == kept
    assert torch.equal(coords, full.voxel_coords[:P]) and torch.equal(inv, full.unq_inv[:kept])
    assert torch.equal(torch.bincount(slot_rank.long(), minlength=P), torch.bincount(inv, minlength=P)) if kept else slot_rank.numel() == 0
    if P:
        want_row = (coords[:, 0].long() * GRID[1] + coords[:, 2].long()) * GRID[0] + coords[:, 3].long()
        assert torch.equal(slot_row.long(), want_row[slot_rank.long()])
    vox = ops.voxelize(pd, g, want_inverse=False, want_counts=False)      # the rounds 1 - 4 pillariser: same front tables, no records
    c2, i2, n2 = ops.pillar_index_export(vox)
    assert int(n2[0]) == P and torch.equal(c2, coords) and torch.equal(i2, inv)
    if pts.shape[0]:
        ref = opil.voxelize(pts, 5, PC_RANGE, VOXEL, GRID)
        assert np.array_equal(coords.cpu().numpy(), ref['coords']) and np.array_equal(inv.cpu().numpy(), ref['inv'])
