"""In-kernel s_memtime stamps of one k_pfn workgroup (build: csrc/build_variant.sh pfnstamp "-DPFN_STAMP=<workgroup>"): cycles per phase
of the fused PFN on the 1.44 M-point early-fusion cloud.  Diagnostic tool, not part of the product path."""
import os, sys, ctypes, numpy as np, torch
R = os.environ.get('GRAFT_REPO_ROOT', '.')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'practical-collab-perception_amd'))
os.environ['PCP_HIP_LIB'] = os.path.join(R, 'practical-collab-perception_amd', 'lib', 'variants', 'libpcp_hip_pfnstamp.so')
import bench
from pcp_amd import ops, lib
pts_np, _ = bench.make_points(bench.CONFIGS['early'], 4, 0, 'uniform')
p = torch.from_numpy(pts_np).cuda()
grid = ops.make_grid([-51.2, -51.2, -8.0, 51.2, 51.2, 0.0], [0.2, 0.2, 8.0], [512, 512, 1], 4)
w0 = torch.randn(32, 11, device='cuda') * 0.1; b0 = torch.zeros(32, device='cuda'); w1 = torch.randn(64, 64, device='cuda') * 0.1; b1 = torch.zeros(64, device='cuda')
vox = ops.voxelize(p, grid, want_inverse=False, want_counts=False)
canvas = torch.zeros((4, 512, 512, 64), device='cuda')
L = lib.load()
L._LIB if hasattr(L, '_LIB') else None
raw = ctypes.CDLL(os.environ['PCP_HIP_LIB'])
names = ['start', 'P+s1 known', 'LDS set-up + barriers', 'sweep1 + mean', 'chunk0: rows ready', 'layer0 + x tile', 'barrier', 'MFMA', 'dmax atomics', 'barrier', 'loop end', 'epilogue']
for rep in range(3):
    ops.pfn_scatter(p, vox, 5, w0, b0, w1, b1, canvas=canvas, pillar_features=None)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    raw.pcp_debug_read_pfn(buf, 32 * 8)
    t = [buf[i] for i in range(12)]
    print('   setup: init+loads issued %d, first barrier %d, pl_of+2nd barrier %d | epilogue: mfma %d, dmax+stores %d' % (buf[12] - t[1], buf[13] - buf[12], t[2] - buf[13], buf[14] - t[10], t[11] - buf[14]))
    print('rep', rep, ' total %d cycles:' % (t[11] - t[0]), '  '.join('%s %d' % (names[i], t[i] - t[i - 1]) for i in range(1, 12)))
