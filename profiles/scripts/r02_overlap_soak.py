import sys, os, numpy as np, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'practical-collab-perception_amd'))
import bench
cfg = bench.load_cfg('v2x_pointpillar_disco.yaml')
model, state, ds = bench.build_model(cfg)
model = model.cuda().eval()
for m in model.modules():
    if hasattr(m, 'materialize_pillars'):
        m.materialize_pillars, m.reuse_buffers, m.sparse_first_layer = False, True, True
model.overlap_makers = True
pts_np, metas = bench.make_points(bench.CONFIGS['disco'], 4, 0, 'uniform')
pts = torch.from_numpy(pts_np).cuda()
ref = None
bad = 0
for it in range(300):
    bd = {'points': pts, 'batch_size': 4, 'metadata': metas}
    with torch.no_grad():
        pred, _ = model(bd)
    sig = (bd['spatial_features_2d'].double().sum().item(), float(bd['spatial_features_2d'].abs().max()),
           tuple(int(p['pred_boxes'].shape[0]) for p in pred), float(sum(p['pred_scores'].double().sum() for p in pred)))
    if ref is None: ref = sig
    elif sig != ref:
        bad += 1
        print('iteration', it, 'differs', sig, ref)
print('soak done: 300 forwards, mismatches', bad, ref)
