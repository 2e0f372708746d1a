"""Counter-based synthetic inputs and deterministic weights.

Everything here is a pure function of (seed, stream, index) through splitmix64, so the
golden-vector generator (which runs the reference in the CPU container), the parity
tests and bench.py on the GPU box regenerate bit-identical float32 inputs without
depending on any numpy/torch RNG implementation.

Layouts follow the reference's collate contract (pcdet/datasets/dataset.py:224-229: a
batch-index column is prepended to every point row) and the per-config column lists of
SURVEY.md section 8(d).
"""
import math

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

SEED_BASE = 20260000


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def uniform01(seed, stream, n, offset=0):
    """n float32 values in [0, 1): top 24 bits of splitmix64(key(seed, stream) + i)."""
    with np.errstate(over='ignore'):
        key = _mix(np.uint64(seed) * _GOLDEN + np.uint64(stream) * _M2 + np.uint64(1))
        idx = np.arange(offset, offset + n, dtype=np.uint64)
        z = _mix(key + (idx + np.uint64(1)) * _GOLDEN)
    return ((z >> np.uint64(40)).astype(np.float32)) * np.float32(1.0 / 16777216.0)


def uniform(seed, stream, n, lo, hi):
    u = uniform01(seed, stream, n).astype(np.float64)
    return (lo + (hi - lo) * u).astype(np.float32)


def stream_id(agent, column):
    return agent * 65536 + column


def agent_cloud(agent, n_points=60000, layout='car', seed=SEED_BASE, xy_half=52.0, dist='uniform',
                z_range=(-8.0, 0.0)):
    """One agent's cloud WITHOUT the batch column.

    layout:
      'car' / 'early' -> [x, y, z, intensity, time, sweep_idx, inst_idx]                     (7 cols)
      'lately'        -> [x, y, z, i, t, dx, dy, dz, heading, score, label, sweep, inst]      (13 cols)
      'disco'         -> [x, y, z, intensity, time, agent_idx]                                (6 cols)
    dist: 'uniform' (x, y ~ U(-xy_half, xy_half), about 3 % out of range at 52.0)
          'ring'    (LiDAR-like: r = 70 u^2, uniform azimuth, z = -2 + 0.3 * n)
    """
    s = lambda c: stream_id(agent, c)
    if dist == 'uniform':
        x = uniform(seed, s(0), n_points, -xy_half, xy_half)
        y = uniform(seed, s(1), n_points, -xy_half, xy_half)
        z = uniform(seed, s(2), n_points, z_range[0], z_range[1])
    elif dist == 'ring':
        u = uniform01(seed, s(0), n_points).astype(np.float64)
        az = uniform01(seed, s(1), n_points).astype(np.float64) * (2.0 * math.pi)
        r = 70.0 * u * u * (xy_half / 52.0)
        x = (r * np.cos(az)).astype(np.float32)
        y = (r * np.sin(az)).astype(np.float32)
        g = uniform01(seed, s(2), n_points).astype(np.float64) + uniform01(seed, s(9), n_points) - 1.0
        z = (-2.0 + 0.3 * 2.449 * g).astype(np.float32)
    else:
        raise ValueError(dist)
    inten = uniform01(seed, s(3), n_points)
    tstep = np.floor(uniform01(seed, s(4), n_points).astype(np.float64) * 11.0)
    tstep = np.minimum(tstep, 10.0)
    time = (tstep * 0.1).astype(np.float32)
    sweep = (10.0 - tstep).astype(np.float32)
    inst = np.full(n_points, -1.0, dtype=np.float32)
    if layout in ('car', 'early'):
        cols = [x, y, z, inten, time, sweep, inst]
    elif layout == 'disco':
        cols = [x, y, z, inten, time, np.full(n_points, float(agent), dtype=np.float32)]
    elif layout == 'lately':
        n_modar = min(300, n_points // 4)
        dx = np.zeros(n_points, np.float32)
        dy = np.zeros(n_points, np.float32)
        dz = np.zeros(n_points, np.float32)
        hd = np.zeros(n_points, np.float32)
        sc = np.zeros(n_points, np.float32)
        lb = np.zeros(n_points, np.float32)
        m = slice(n_points - n_modar, n_points)
        dx[m] = uniform(seed, s(5), n_modar, 1.5, 5.0)
        dy[m] = uniform(seed, s(6), n_modar, 1.5, 5.0)
        dz[m] = uniform(seed, s(7), n_modar, 1.5, 5.0)
        hd[m] = uniform(seed, s(8), n_modar, -math.pi, math.pi)
        sc[m] = uniform(seed, s(10), n_modar, 0.1, 1.0)
        lb[m] = 1.0
        inten = inten.copy()
        time = time.copy()
        inten[m] = 0.0
        time[m] = 0.0
        cols = [x, y, z, inten, time, dx, dy, dz, hd, sc, lb, sweep, inst]
    else:
        raise ValueError(layout)
    return np.stack(cols, axis=1).astype(np.float32)


def mask_outside_range(points_xyz_first, pc_range):
    """Host pre-mask applied by the dataset in configs 1-4 (reference: data_processor.py:78-92 ->
    common_utils.py:64-68): keep rows with min <= p < max on x, y and z.  The synthetic benchmark clouds are
    NOT pre-masked (about 3 % of the rows stay out of range so the VFE's own x/y mask is exercised)."""
    p = points_xyz_first
    keep = np.ones(p.shape[0], dtype=bool)
    for a in range(3):
        keep &= (p[:, a] >= pc_range[a]) & (p[:, a] < pc_range[a + 3])
    return p[keep]


def collate(clouds):
    """clouds: list (one per batch element) of (n_i, C) arrays -> (sum n_i, 1 + C) with batch-index column."""
    rows = []
    for b, c in enumerate(clouds):
        rows.append(np.concatenate([np.full((c.shape[0], 1), float(b), np.float32), c], axis=1))
    return np.ascontiguousarray(np.concatenate(rows, axis=0), dtype=np.float32)


def agent_pose(agent):
    """se3_from_ego[agent] (4x4 float64): yaw 0.3*a rad, translation (3a, -2a, 0) m  (SURVEY 8(d))."""
    yaw = 0.3 * agent
    c, s = math.cos(yaw), math.sin(yaw)
    T = np.eye(4, dtype=np.float64)
    T[:3, :3] = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    T[:3, 3] = [3.0 * agent, -2.0 * agent, 0.0]
    return T


# ---------------------------------------------------------------------------------------------------
# deterministic weights keyed by state-dict name
# ---------------------------------------------------------------------------------------------------

def fill_state_dict(shapes, seed=SEED_BASE + 7, scheme='survey'):
    """shapes: ordered mapping name -> shape (tuple) [+ dtype inferred from the name].
    scheme 'he': conv / linear weights U(-k, k) with k = sqrt(6 / fan_in) (variance preserving through ReLU) -- with the 'survey' bound
    1/sqrt(fan_in) the signal decays by ~0.4 per layer, the head maps end up almost constant (every candidate within 5e-4 of
    sigmoid(-2.19), ~1e-6 apart) and the final box SET becomes a function of float noise; 'he' keeps an O(1) spatial signal so that
    detections are separated by margins far above any fp32 implementation's rounding (the well-conditioned fixtures g13).
    Returns name -> np.ndarray following SURVEY 8(d):
      conv / linear weights U(-k, k), k = 1/sqrt(fan_in); biases U(-0.1, 0.1) except the final 'hm' bias = -2.19;
      BN weight U(0.5, 1.5), bias U(-0.1, 0.1), running_mean U(-0.1, 0.1), running_var U(0.5, 1.5);
      num_batches_tracked / global_step = 0.
    The stream id is the rank of the name in sorted order so the result is independent of module build order."""
    names = sorted(shapes.keys())
    out = {}
    for rank, name in enumerate(names):
        shape = tuple(int(v) for v in shapes[name])
        n = int(np.prod(shape)) if len(shape) else 1
        leaf = name.split('.')[-1]
        if leaf in ('num_batches_tracked', 'global_step'):
            out[name] = np.zeros(shape, dtype=np.int64)
            continue
        st = 1000003 + rank
        if leaf == 'running_mean':
            v = uniform(seed, st, n, -0.1, 0.1)
        elif leaf == 'running_var':
            v = uniform(seed, st, n, 0.5, 1.5)
        elif len(shape) == 1 and leaf == 'weight':          # BatchNorm gamma
            v = uniform(seed, st, n, 0.5, 1.5)
        elif leaf == 'bias':
            v = uniform(seed, st, n, -0.1, 0.1)
            if '.hm.' in name:
                # only the LAST conv of the hm branch carries the -2.19 prior (center_head.py:32)
                parent_idx = name.split('.')[-2]
                if parent_idx.isdigit() and _is_last_hm_conv(name, names):
                    v = np.full(n, -2.19, dtype=np.float32)
        else:                                               # conv / linear / deconv weight
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            k = _weight_gain(scheme) / math.sqrt(max(fan_in, 1))
            v = uniform(seed, st, n, -k, k)
        out[name] = v.reshape(shape).astype(np.float32)
    return out


def _weight_gain(scheme):
    if scheme == 'survey':
        return 1.0
    if scheme == 'he':
        return math.sqrt(6.0)
    if scheme.startswith('gain:'):
        return float(scheme[5:])
    raise ValueError('unknown weight scheme %r' % (scheme,))


def _is_last_hm_conv(name, names):
    prefix = name[:name.index('.hm.') + 4]
    idxs = []
    for other in names:
        if other.startswith(prefix) and other.endswith('.bias'):
            tok = other[len(prefix):].split('.')[0]
            if tok.isdigit():
                idxs.append(int(tok))
    tok = name[len(prefix):].split('.')[0]
    return tok.isdigit() and int(tok) == max(idxs)


def instance_foreground(index, gt, n_sweeps=11, per_local=12, max_instances=12):
    """Foreground points for HunterJr training: points inside the first `max_instances` boxes of gt (n, 8) at three sweeps, rows
    [x, y, z, intensity, time, sweep, instance], and instances_tf (n, n_sweeps, 3, 4): odd instances drive along their heading (the rigid
    motion takes a sweep-s point to the newest sweep), even instances stand still."""
    n = gt.shape[0]
    tf = np.zeros((n, n_sweeps, 3, 4), dtype=np.float32)
    tf[..., :3, :3] = np.eye(3, dtype=np.float32)
    rows = []
    s = SEED_BASE + 6000 + index
    for i in range(min(n, max_instances)):
        c, dims, yaw = gt[i, 0:3].astype(np.float64), gt[i, 3:6].astype(np.float64), float(gt[i, 6])
        moving = i % 2 == 1
        speed = 3.0 + 0.5 * i
        R = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
        for sw in range(n_sweeps):
            if moving:
                tf[i, sw, :3, 3] = speed * (10 - sw) * 0.1 * np.array([np.cos(yaw), np.sin(yaw), 0.0])
        for k, sw in enumerate((10, 7, 3)):
            dt = (10 - sw) * 0.1
            u = uniform(s, 20 * i + k, per_local * 3, -0.5, 0.5).reshape(per_local, 3).astype(np.float64)
            p = (u * dims) @ R.T + c - (tf[i, sw, :3, 3].astype(np.float64) if moving else 0.0)
            r = np.zeros((per_local, 7), dtype=np.float32)
            r[:, 0:3], r[:, 3], r[:, 4], r[:, 5], r[:, 6] = p, 0.5, dt, sw, i
            rows.append(r)
    return np.concatenate(rows, 0), tf
