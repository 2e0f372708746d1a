"""how long the HOST needs to queue one bf16 training iteration (no device wait in the timed part except the loop's own host reads) against the
iteration's wall time: is the loop GPU-bound or launch-bound?"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd')); sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd', 'tools'))
os.environ['PCP_CONV_ALGO'] = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
import torch
import bench
from train_utils.optimization import build_optimizer, build_scheduler
conf = bench.CONFIGS['disco']
cfg = bench.load_cfg(conf['yaml'])
batch = int(cfg.OPTIMIZATION.BATCH_SIZE_PER_GPU)
model, _s, _d = bench.build_model(cfg)
dev = torch.device('cuda:0')
model = model.to(dev)
model.overlap_makers = True
opt = build_optimizer(model, cfg.OPTIMIZATION)
sched, _ = build_scheduler(opt, 1000, 1, -1, cfg.OPTIMIZATION)
pts_np, metas = bench.make_points(conf, batch, 0)
import numpy as np
from pcp_amd import synth
gtn = np.zeros((batch, 40, 8), dtype=np.float32)
for f in range(batch):
    n = 40 - 3 * f
    for col, (lo, hi) in enumerate([(-50.0, 50.0), (-50.0, 50.0), (-3.0, -1.0), (3.0, 5.5), (1.5, 2.5), (1.4, 2.0), (-3.14159, 3.14159)]):
        gtn[f, :n, col] = synth.uniform(77 + f, col + 1, n, lo, hi)
    gtn[f, :n, 7] = 1.0
gt = torch.from_numpy(gtn).to(dev)
pts = torch.from_numpy(pts_np).to(dev)


def one(it):
    sched.step(it)
    model.train()
    opt.zero_grad()
    bd = {'points': pts, 'batch_size': batch, 'metadata': metas}
    if gt is not None:
        bd['gt_boxes'] = gt
    ret, tb, _ = model(bd)
    model.update_global_step()
    ret['loss'].backward()
    opt.clip_grad_norm(cfg.OPTIMIZATION.GRAD_NORM_CLIP)
    opt.step()


for it in range(5):
    one(it)
torch.cuda.synchronize()
host, wall = [], []
for it in range(5, 25):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    one(it)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    wall.append((t2 - t0) * 1e3)
m = lambda v: sorted(v)[len(v) // 2]
print('%s: host returns after %.2f ms (median), iteration done after %.2f ms; device idle at the end of queueing for %.2f ms' % (
    os.environ['PCP_CONV_ALGO'], m(host), m(wall), m(wall) - m(host)))
