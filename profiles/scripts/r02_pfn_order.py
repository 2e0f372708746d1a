"""PFN time on the same 1.44 M-point cloud in random row order (the benchmark's) and pre-sorted by pillar (streaming gather)."""
import os, sys, numpy as np, torch
R = os.environ.get('GRAFT_REPO_ROOT', '.')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'practical-collab-perception_amd'))
import bench
from pcp_amd import ops
pts_np, _ = bench.make_points(bench.CONFIGS['early'], 4, 0, 'uniform')
rng = [-51.2, -51.2, -8.0, 51.2, 51.2, 0.0]
cx = np.floor((pts_np[:, 1] + 51.2) / 0.2).astype(np.int64); cy = np.floor((pts_np[:, 2] + 51.2) / 0.2).astype(np.int64)
order = np.lexsort((cy, cx, pts_np[:, 0].astype(np.int64)))
w0 = torch.randn(32, 11, device='cuda') * 0.1; b0 = torch.zeros(32, device='cuda')
w1 = torch.randn(64, 64, device='cuda') * 0.1; b1 = torch.zeros(64, device='cuda')
def run(p_np, tag):
    p = torch.from_numpy(np.ascontiguousarray(p_np)).cuda()
    grid = ops.make_grid(rng, [0.2, 0.2, 8.0], [512, 512, 1], 4)
    vox = ops.voxelize(p, grid, want_inverse=False, want_counts=False)
    canvas = torch.zeros((4, 512, 512, 64), device='cuda')
    f = lambda: ops.pfn_scatter(p, vox, 5, w0, b0, w1, b1, canvas=canvas, pillar_features=None)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    v0, v1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    v0.record()
    for _ in range(20): ops.voxelize(p, grid, want_inverse=False, want_counts=False, workspace=vox.workspace)
    v1.record(); torch.cuda.synchronize()
    print('%-28s pfn %.1f us   voxelize %.1f us   (%d points)' % (tag, e0.elapsed_time(e1) / 20 * 1e3, v0.elapsed_time(v1) / 20 * 1e3, p.shape[0]))
run(pts_np, 'random row order')
run(pts_np[order], 'rows sorted by pillar')
