"""Host-side parameter preparation for the DiscoNet warp: the 2x3 affine `theta` exactly as the reference builds it
(pcdet/models/bev_layers/v2x_fusion_disco.py:32-35), in float32 torch-CPU arithmetic.  Six floats per (agent, frame);
the per-pixel work happens in pcp_warp_nearest.
"""
import numpy as np
import torch


def ego_se3_agent(se3_from_ego_agent):
    """float32(inv(float64 se3_from_ego[agent]))  (v2x_fusion_disco.py:93)"""
    return torch.from_numpy(np.linalg.inv(np.asarray(se3_from_ego_agent, dtype=np.float64))).float()


def warp_theta(dst_se3_src, h, w, pc_range_min, pix_size, return_ambiguous=False):
    """dst_se3_src: (4, 4) float32 torch tensor.  Returns 6 python floats (row-major 2x3).
    NB the reference normalises BOTH translation components by H (bev_in_src.shape[1])."""
    T = dst_se3_src.detach().cpu().float()
    rot = T[:2, :2]
    t = T[:2, [-1]]
    t_pix_norm = 2.0 * ((t - pc_range_min) / pix_size) / h - 1.0
    theta = torch.cat([rot.T, -torch.matmul(rot.T, t_pix_norm)], dim=1)      # (2, 3) float32
    vals = [float(v) for v in theta.reshape(-1)]
    if not return_ambiguous:
        return vals
    # pixels whose source coordinate lies within 1e-3 of a .5 tie (rounding there depends on the last ulp of the
    # affine evaluation order): used only by tests to exclude them from exact comparison
    th = theta.double()
    xs = (2.0 * torch.arange(w, dtype=torch.float64) + 1.0) / w - 1.0
    ys = (2.0 * torch.arange(h, dtype=torch.float64) + 1.0) / h - 1.0
    yy, xx = torch.meshgrid(ys, xs, indexing='ij')
    gx = xx * th[0, 0] + yy * th[0, 1] + th[0, 2]
    gy = xx * th[1, 0] + yy * th[1, 1] + th[1, 2]
    fx = ((gx + 1.0) * w - 1.0) / 2.0
    fy = ((gy + 1.0) * h - 1.0) / 2.0
    amb = ((fx - torch.floor(fx) - 0.5).abs() < 1e-3) | ((fy - torch.floor(fy) - 0.5).abs() < 1e-3)
    return vals, amb


# ---------------------------------------------------------------------------------------------------------------------------------------
# All warps of a forward at once.  warp_theta above is ~20 torch-CPU dispatches per (agent, frame) pair (77 us each: 1.5 ms of host time
# per DiscoNet step at 20 pairs, 120 scalar reads); here the same float32 arithmetic runs vectorised in numpy over every pair of the forward.
# Rounding order reproduced:
#   t_pix_norm = 2.0 * ((t - pc_min) / pix) / h - 1.0        elementwise IEEE float32 ops in this order
#   R^T . t_pix_norm                                          torch's (2, 2) @ (2, 1) matmul -- whose rounding depends on the HOST: on the
#                                                             build container (MKL with FMA kernels) it is the fused chain fma(a1, b1, rn(a0 * b0)),
#                                                             on the MI355X boxes (MKL on an AMD EPYC: no FMA path) it is rn(a0 * b0) + rn(a1 * b1)
#                                                             (2 000 random poses each, 0 mismatches for the matching form, ~25 % for the others).
# _form() finds out which form THIS machine's torch uses (256 random poses against warp_theta, once per map geometry); only if none of the
# known forms matches are the pairs evaluated through torch one at a time.
# ---------------------------------------------------------------------------------------------------------------------------------------
_CAL = {}
_FORMS = ('fma_k1', 'mul_add', 'fma_k0', 'exact')


def _thetas_numpy(se3_list, h, pc_range_min, pix_size, form='fma_k1'):
    T32 = np.stack([np.linalg.inv(np.asarray(T, dtype=np.float64)) for T in se3_list]).astype(np.float32)      # (n, 4, 4), as ego_se3_agent
    rot = T32[:, :2, :2]
    t = T32[:, :2, 3]
    f32 = np.float32
    tp = f32(2.0) * ((t - f32(pc_range_min)) / f32(pix_size)) / f32(h) - f32(1.0)                            # (n, 2)
    rt = np.transpose(rot, (0, 2, 1))                                                                        # R^T
    a0, a1 = rt[:, :, 0].astype(np.float64), rt[:, :, 1].astype(np.float64)                                  # products of two float32 are exact in float64
    b0, b1 = tp[:, 0:1].astype(np.float64), tp[:, 1:2].astype(np.float64)
    if form == 'fma_k1':
        y = ((a0 * b0).astype(np.float32).astype(np.float64) + a1 * b1).astype(np.float32)
    elif form == 'mul_add':
        y = ((a0 * b0).astype(np.float32) + (a1 * b1).astype(np.float32)).astype(np.float32)
    elif form == 'fma_k0':
        y = ((a1 * b1).astype(np.float32).astype(np.float64) + a0 * b0).astype(np.float32)
    else:
        y = (a0 * b0 + a1 * b1).astype(np.float32)
    theta = np.concatenate([rt, -y[:, :, None]], axis=2)                                                     # (n, 2, 3)
    return np.ascontiguousarray(theta.reshape(len(se3_list), 6), dtype=np.float32)


def _form(h, pc_range_min, pix_size):
    """the rounding form that reproduces warp_theta bit for bit on 256 random rigid poses with this map geometry on THIS machine, or None"""
    key = (int(h), float(pc_range_min), float(pix_size))
    if key not in _CAL:
        rng = np.random.RandomState(97)
        poses = []
        for _ in range(256):
            a = rng.uniform(-np.pi, np.pi)
            T = np.eye(4)
            T[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
            T[:3, 3] = rng.uniform(-60, 60, 3)
            poses.append(T)
        slow = np.array([warp_theta(ego_se3_agent(T), h, h, pc_range_min, pix_size) for T in poses], dtype=np.float32)
        _CAL[key] = None
        for form in _FORMS:
            if np.array_equal(_thetas_numpy(poses, h, pc_range_min, pix_size, form), slow):
                _CAL[key] = form
                break
    return _CAL[key]


def _calibrated(h, pc_range_min, pix_size):
    return _form(h, pc_range_min, pix_size) is not None


def warp_thetas(se3_from_ego_list, h, w, pc_range_min, pix_size):
    """[6 floats] for every se3_from_ego pose of the list (the reference's transform_bev_img arithmetic, v2x_fusion_disco.py:32-35 with
    dst_se3_src = float32(inv(se3_from_ego)), :93)"""
    if not se3_from_ego_list:
        return []
    form = _form(h, pc_range_min, pix_size)
    if form is not None:
        return [[float(v) for v in row] for row in _thetas_numpy(se3_from_ego_list, h, pc_range_min, pix_size, form)]
    return [warp_theta(ego_se3_agent(T), h, w, pc_range_min, pix_size) for T in se3_from_ego_list]
