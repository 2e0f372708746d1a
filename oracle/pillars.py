"""ORACLE (test infrastructure only) -- dynamic pillarisation, PillarFeatureNet, scatter-to-BEV in numpy.

Follows /root/reference/pcdet/models/backbones_3d/vfe/dynamic_pillar_vfe.py:94-147 (DynamicPillarVFE.forward),
:35-46 (PFNLayerV2.forward) and /root/reference/pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py:14-37.
torch_scatter (third party, not vendored, unpinned -- README.md:68-71) is restated from its published semantics:
scatter_mean = sequential fp32 sum in index order / count, scatter_max = per-row maximum.
"""
import numpy as np

F32 = np.float32


def voxelize(points, num_raw, pc_range, voxel_size, grid_size):
    """dynamic_pillar_vfe.py:96-108, 137-147.

    points: (N, 1+C) float32 with column 0 = batch index.  Returns a dict with
      keep      (N,) bool      rows surviving the x/y mask (order preserved)
      cell_xy   (N', 2) int32  [cx, cy] of the kept rows
      unq       (P,) int32     sorted unique merged ids  b*nx*ny + cx*ny + cy
      inv       (N',) int64    pillar rank of every kept row
      cnt       (P,) int64     rows per pillar
      coords    (P, 4) int32   [b, 0, y, x]
    """
    pts = np.ascontiguousarray(points[:, :1 + num_raw], dtype=F32)
    rmin = np.asarray(pc_range[:2], dtype=F32)
    vs = np.asarray(voxel_size[:2], dtype=F32)
    g = np.asarray(grid_size[:2], dtype=np.int32)
    with np.errstate(invalid='ignore', over='ignore'):
        c = np.floor((pts[:, 1:3] - rmin[None, :]) / vs[None, :])
        c = np.nan_to_num(c, nan=-1.0, posinf=-1.0, neginf=-1.0)
        c = np.clip(c, -2.0, 2.0 ** 30).astype(np.int32)
    keep = ((c >= 0) & (c < g[None, :])).all(axis=1)
    cell = c[keep]
    b = pts[keep, 0].astype(np.int32)
    scale_xy = np.int32(int(grid_size[0]) * int(grid_size[1]))
    scale_y = np.int32(int(grid_size[1]))
    merged = b * scale_xy + cell[:, 0] * scale_y + cell[:, 1]
    unq, inv, cnt = np.unique(merged, return_inverse=True, return_counts=True)
    unq = unq.astype(np.int32)
    coords = np.stack([unq // scale_xy, np.zeros_like(unq), unq % scale_y, (unq % scale_xy) // scale_y],
                      axis=1).astype(np.int32)
    return dict(keep=keep, cell_xy=cell.astype(np.int32), unq=unq, inv=inv.astype(np.int64).reshape(-1),
                cnt=cnt.astype(np.int64), coords=coords)


def pillar_mean(xyz, inv, num_pillars):
    """torch_scatter.scatter_mean(points_xyz, unq_inv, dim=0) (dynamic_pillar_vfe.py:110): fp32, index order."""
    acc = np.zeros((num_pillars, xyz.shape[1]), dtype=F32)
    np.add.at(acc, inv, xyz.astype(F32))
    cnt = np.bincount(inv, minlength=num_pillars).astype(F32)
    return acc / np.maximum(cnt, F32(1.0))[:, None]


def point_features(points, num_raw, vox, pc_range, voxel_size, use_absolute_xyz=True, with_distance=False):
    """dynamic_pillar_vfe.py:110-126 -> (N', F) float32 = [raw (x,y,z,...) or raw[3:] (:117-120), f_cluster(3), f_center(3),
    |xyz| if with_distance (:122-124)]."""
    pts = np.ascontiguousarray(points[:, :1 + num_raw], dtype=F32)[vox['keep']]
    xyz = pts[:, 1:4]
    mean = pillar_mean(xyz, vox['inv'], vox['unq'].shape[0])
    f_cluster = xyz - mean[vox['inv']]
    vx, vy, vz = (F32(v) for v in voxel_size)
    # offsets exactly as the constructor computes them (:80-82): python float / 2 + numpy float32 range
    x_off = F32(F32(voxel_size[0] / 2) + F32(pc_range[0]))
    y_off = F32(F32(voxel_size[1] / 2) + F32(pc_range[1]))
    z_off = F32(F32(voxel_size[2] / 2) + F32(pc_range[2]))
    f_center = np.empty_like(xyz)
    f_center[:, 0] = xyz[:, 0] - (vox['cell_xy'][:, 0].astype(F32) * vx + x_off)
    f_center[:, 1] = xyz[:, 1] - (vox['cell_xy'][:, 1].astype(F32) * vy + y_off)
    f_center[:, 2] = xyz[:, 2] - z_off
    parts = [pts[:, 1:] if use_absolute_xyz else pts[:, 4:], f_cluster, f_center]
    if with_distance:
        parts.append(np.sqrt((xyz * xyz).sum(axis=1, dtype=F32), dtype=F32)[:, None])
    return np.concatenate(parts, axis=1).astype(F32), mean


def _bn_eval(x, st, prefix, eps):
    g = st[prefix + '.weight'].astype(F32)
    b = st[prefix + '.bias'].astype(F32)
    m = st[prefix + '.running_mean'].astype(F32)
    v = st[prefix + '.running_var'].astype(F32)
    return (x - m[None, :]) / np.sqrt(v[None, :] + F32(eps)) * g[None, :] + b[None, :]


def pfn(features, inv, num_pillars, st, prefix='vfe', num_layers=2, eps=1e-3):
    """PFNLayerV2 x num_layers (dynamic_pillar_vfe.py:35-46): Linear(no bias) -> BN1d(eval, eps 1e-3) [or Linear with bias] -> ReLU ->
    per-pillar max; non-last layers concatenate [x, x_max[inv]]."""
    x = features
    for li in range(num_layers):
        w = st['%s.pfn_layers.%d.linear.weight' % (prefix, li)].astype(F32)
        y = x @ w.T
        if '%s.pfn_layers.%d.norm.weight' % (prefix, li) in st:
            y = _bn_eval(y, st, '%s.pfn_layers.%d.norm' % (prefix, li), eps)
        else:                                       # USE_NORM False: Linear keeps its bias (:26-32)
            y = y + st['%s.pfn_layers.%d.linear.bias' % (prefix, li)].astype(F32)[None, :]
        y = np.maximum(y, F32(0))
        ymax = np.full((num_pillars, y.shape[1]), -np.inf, dtype=F32)
        np.maximum.at(ymax, inv, y)
        if li == num_layers - 1:
            return ymax
        x = np.concatenate([y, ymax[inv]], axis=1)
    return x


def scatter_to_bev(pillar_features, coords, batch_size, nx, ny):
    """pointpillar_scatter.py:14-37 -> (B, C, ny, nx) float32; index = z + y*nx + x."""
    c = pillar_features.shape[1]
    canvas = np.zeros((batch_size, c, ny * nx), dtype=F32)
    idx = coords[:, 1].astype(np.int64) + coords[:, 2].astype(np.int64) * nx + coords[:, 3].astype(np.int64)
    canvas[coords[:, 0], :, idx] = pillar_features
    return canvas.reshape(batch_size, c, ny, nx)


def vfe_forward(points, st, arch, prefix='vfe'):
    """a1-a5 in one call.  Returns dict(vox, features, pillar_features, spatial_features)."""
    nr = arch['num_raw']
    vox = voxelize(points, nr, arch['pc_range'], arch['voxel_size'], arch['grid_size'])
    feats, mean = point_features(points, nr, vox, arch['pc_range'], arch['voxel_size'], arch.get('use_absolute_xyz', True),
                                 arch.get('with_distance', False))
    pf = pfn(feats, vox['inv'], vox['unq'].shape[0], st, prefix=prefix, num_layers=len(arch['vfe_filters']))
    bs = int(vox['coords'][:, 0].max()) + 1 if vox['coords'].shape[0] else 1
    canvas = scatter_to_bev(pf, vox['coords'], bs, int(arch['grid_size'][0]), int(arch['grid_size'][1]))
    return dict(vox=vox, point_features=feats, pillar_mean=mean, pillar_features=pf, spatial_features=canvas)
