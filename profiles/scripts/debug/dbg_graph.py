import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'practical-collab-perception_amd'); sys.path.insert(0, 'tests')
from helpers import load_golden
from pcp_amd import synth, ops
which = sys.argv[1]
g = load_golden('g1_car.npz')
pts = torch.from_numpy(g['points']).cuda()
rng = [-12.8, -12.8, -8.0, 12.8, 12.8, 0.0]
grid = ops.make_grid(rng, [0.2, 0.2, 8.0], [128, 128, 1], 2)
w0 = torch.randn(32, 11).cuda(); b0 = torch.randn(32).cuda(); w1 = torch.randn(64, 64).cuda(); b1 = torch.randn(64).cuda()
canvas = torch.zeros((2, 128, 128, 64), device='cuda')
state = {'ws': None, 'prev': None}
def run():
    if 'c' in which and state['prev'] is not None:
        ops.canvas_clear(state['prev'], canvas)
    vox = ops.voxelize(pts, grid, want_inverse=False, want_counts=False, workspace=state['ws'])
    state['ws'] = vox.workspace
    if 'p' in which:
        ops.pfn_scatter(pts, vox, 5, w0, b0, w1, b1, canvas=canvas, pillar_features=None)
    state['prev'] = vox
    return vox
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): v = run()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
print('warm ok P', int(v.counters[0]), flush=True)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    v = run()
print('captured', flush=True)
for i in range(4):
    gr.replay(); torch.cuda.synchronize(); print('replay', i, 'ok P', int(v.counters[0]), float(canvas.abs().sum()), flush=True)
