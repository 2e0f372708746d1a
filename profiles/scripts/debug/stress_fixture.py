import os, sys, torch
ROOT = '/root/repo' if os.path.isdir('/root/repo') else os.environ['GRAFT_REPO_ROOT']
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'practical-collab-perception_amd'))
import test_gpu_e2e as T
from pcdet.models.pipelined import PipelinedDetector
case = sys.argv[1]; n = int(sys.argv[2]); replicas = int(sys.argv[3])
g = T.load_golden('g13_conditioned.npz')
model = T._g13_model(g, case)
for m in model.modules():
    if hasattr(m, 'materialize_pillars'):
        m.materialize_pillars, m.reuse_buffers, m.sparse_first_layer = False, True, True
if hasattr(model, 'overlap_makers') and case.startswith('disco'):
    model.overlap_makers = True
pts, B = T._g13_points(case)
metadata = [{'se3_from_ego': {a: g['disco_full_pose_%d' % a] for a in (0, 2, 3, 4, 5)}}] if case == 'disco_full' else [{} for _ in range(B)]
base = torch.from_numpy(pts.copy()).cuda()
variants = []
for k in range(4):
    v = base.clone(); v[:, 1:3] += 0.011 * k; variants.append(v)
want = []
for v in variants:
    with torch.no_grad():
        pred, _ = model({'points': v.clone(), 'batch_size': B, 'metadata': metadata})
    torch.cuda.synchronize()
    want.append([{k: t.clone() for k, t in p.items()} for p in pred])
# is the batch-by-batch result itself reproducible?
rep_bad = 0
for it in range(40):
    v = variants[it % 4]
    with torch.no_grad():
        pred, _ = model({'points': v.clone(), 'batch_size': B, 'metadata': metadata})
    torch.cuda.synchronize()
    for pa, pb in zip(pred, want[it % 4]):
        for k in ('pred_boxes', 'pred_scores', 'pred_labels'):
            if pa[k].shape != pb[k].shape or not torch.equal(pa[k], pb[k]):
                rep_bad += 1
pipe = PipelinedDetector(model, replicas=replicas)
bufs = [torch.empty_like(base), torch.empty_like(base)]
bad = []
def check(i, preds):
    for b, (pa, pb) in enumerate(zip(preds, want[i % 4])):
        for k in ('pred_boxes', 'pred_scores', 'pred_labels'):
            if pa[k].shape != pb[k].shape or not torch.equal(pa[k], pb[k]):
                bad.append((i, b, k, tuple(pa[k].shape), tuple(pb[k].shape)))
for i in range(n):
    out = pipe.submit(bufs[i & 1], B, metadata, copy_from=variants[i % 4])
    if out is not None: check(i - 1, out)
check(n - 1, pipe.flush())
print('%s replicas %d early %s: sequential repeat mismatches %d; pipelined %d batches, %d mismatching tensors %s' % (case, replicas, os.environ.get('PCP_PIPELINE_EARLY_MAKERS', '1'), rep_bad, n, len(bad), bad[:5]))
