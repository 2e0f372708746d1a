"""ORACLE (test infrastructure only) -- the exchange producer / consumer steps of lately fusion, numpy.

  points_in_boxes   /root/reference/pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:16-36, 313-336 (CUDA only in the reference:
                    restated from source, PARITY UNPINNED for this function -- no CPU build of it exists to run here)
  foreground_rows   /root/reference/pcdet/models/bev_layers/hunter_jr.py:377-397
  modar_ingest      /root/reference/pcdet/datasets/v2x_sim/v2x_sim_dataset_ego.py:196-232 with apply_se3_
                    (/root/reference/pcdet/datasets/nuscenes/nuscenes_temporal_utils.py:28-29, 66-70); pinned by tests/golden/g8_exchange.npz,
                    which runs the reference's own apply_se3_ and the torch_scatter shim on the same inputs.
"""
import numpy as np

F32 = np.float32


def points_in_boxes(points, boxes, margin=1e-5):
    """points (M, 3), boxes (T, 7) float32 -> (M,) int32: first containing box or -1.

    Follows check_pt_in_box3d of /root/reference/pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:23-36 (MARGIN 1e-5, the kernel
    points_in_boxes_gpu runs, :313-336: the first box that contains the point) and, with margin=1e-2, check_pt_in_box3d_cpu of
    roiaware_pool3d.cpp:123-137 -- the same test with another constant -- which IS compiled from the reference (oracle/_ref) and pins this
    restatement.  C semantics kept: the z test and the two face tests compare in double (dz / 2.0, dx / 2.0 + MARGIN with MARGIN a float
    constant), the rotation is float arithmetic on cos / sin evaluated in double and rounded to float."""
    pts = np.asarray(points, dtype=F32)
    bx = np.asarray(boxes, dtype=F32)
    out = np.full(pts.shape[0], -1, dtype=np.int32)
    mg = np.float64(F32(margin))
    for k in range(bx.shape[0] - 1, -1, -1):                       # reverse order so the FIRST containing box wins
        cx, cy, cz, dx, dy, dz, rz = (bx[k, i] for i in range(7))
        inz = ~(np.abs(pts[:, 2] - cz).astype(np.float64) > np.float64(dz) / 2.0)
        cosa, sina = F32(np.cos(np.float64(-rz))), F32(np.sin(np.float64(-rz)))
        sx, sy = pts[:, 0] - cx, pts[:, 1] - cy
        lx = sx * cosa + sy * (-sina)
        ly = sx * sina + sy * cosa
        inside = inz & (np.abs(lx).astype(np.float64) < np.float64(dx) / 2.0 + mg) & (np.abs(ly).astype(np.float64) < np.float64(dy) / 2.0 + mg)
        out[inside] = k
    return out


def foreground_rows(points, cls_logit, flow, thresh=0.3):
    """points (N, 1+F), cls_logit (N, 3), flow (N, 3) -> rows (n, F+6), batch index (n,)"""
    prob = (1.0 / (1.0 + np.exp(-cls_logit.astype(np.float64)))).astype(F32)
    mask = prob[:, 0] < F32(thresh)
    rows = np.concatenate([points[mask, 1:], prob[mask], flow[mask]], axis=1).astype(F32)
    return rows, points[mask, 0].astype(np.int32)


def modar_ingest(modar, foreground, target_se3_lidar, max_sweep_idx):
    """modar (n, 9) float32, foreground (m, 13) float32 or None, target_se3_lidar (4, 4) float64 -> rows (n, 13) float32"""
    modar = np.array(modar, dtype=F32)
    if foreground is not None and foreground.shape[0] > 0:
        idx = points_in_boxes(foreground[:, :3], modar[:, :7])
        valid = idx > -1
        fg, idx = foreground[valid], idx[valid]
        for k in np.unique(idx):
            sel = fg[idx == k, -3:].astype(F32)
            acc = np.zeros(3, dtype=F32)
            for r in sel:                                          # scatter(reduce='mean'): sequential float32 sum / count
                acc = acc + r
            modar[k, :3] += (acc / F32(sel.shape[0])) * F32(2.0)
    T = np.asarray(target_se3_lidar, dtype=np.float64)
    out = modar.copy()
    out[:, :3] = modar[:, :3] @ T[:3, :3].T + T[:3, -1]            # float64 product rounded into the float32 array
    out[:, 6] = out[:, 6] + np.arctan2(T[1, 0], T[0, 0])
    out[:, 6] = np.arctan2(np.sin(out[:, 6]), np.cos(out[:, 6]))
    rows = np.zeros((modar.shape[0], 13), dtype=np.float64)
    rows[:, :3] = out[:, :3]
    rows[:, 5:11] = out[:, 3:]
    rows[:, -2] = max_sweep_idx
    rows[:, -1] = -1
    return rows.astype(F32)
