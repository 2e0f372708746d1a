"""Reproduces the in-process sequence [disco mini, replicas 2] -> [disco_full, replicas 1] of tests/test_gpu_e2e.py and tells which side differs."""
import os, sys, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'practical-collab-perception_amd'))
import test_gpu_e2e as T
from pcdet.models.pipelined import PipelinedDetector

def run(case, n, replicas, share=True):
    g = T.load_golden('g13_conditioned.npz')
    model = T._g13_model(g, case)
    for m in model.modules():
        if hasattr(m, 'materialize_pillars'):
            m.materialize_pillars, m.reuse_buffers, m.sparse_first_layer = False, True, True
    model.overlap_makers = os.environ.get('REPRO_OVERLAP', '1') == '1'
    model.share_voxelization = share
    pts, B = T._g13_points(case)
    if case == 'disco':
        metadata = [{'se3_from_ego': {0: g['disco_pose_0'], 2: g['disco_pose_2']}}, {'se3_from_ego': {0: g['disco_pose_0']}}]
    else:
        metadata = [{'se3_from_ego': {a: g['disco_full_pose_%d' % a] for a in (0, 2, 3, 4, 5)}}]
    base = torch.from_numpy(pts.copy()).cuda()
    variants = []
    for k in range(4):
        v = base.clone(); v[:, 1:3] += 0.011 * k; variants.append(v)
    def seq():
        out = []
        for v in variants:
            with torch.no_grad():
                pred, _ = model({'points': v.clone(), 'batch_size': B, 'metadata': metadata})
            torch.cuda.synchronize()
            out.append([{k: t.clone() for k, t in p.items()} for p in pred])
        return out
    want = seq()
    pipe = PipelinedDetector(model, replicas=replicas)
    bufs = [torch.empty_like(base), torch.empty_like(base)]
    got = []
    for i in range(n):
        o = pipe.submit(bufs[i & 1], B, metadata, copy_from=variants[i % 4])
        if o is not None: got.append(o)
    got.append(pipe.flush())
    again = seq()
    def same(a, b):
        return all(pa[k].shape == pb[k].shape and torch.equal(pa[k], pb[k]) for pa, pb in zip(a, b) for k in ('pred_boxes', 'pred_scores', 'pred_labels'))
    bad = [i for i in range(n) if not same(got[i], want[i % 4])]
    print('%s replicas %d: pipelined != first sequential at batches %s; second sequential == first: %s; boxes (want/again) %s' % (
        case, replicas, bad[:8], [same(a, b) for a, b in zip(want, again)], [(w[0]['pred_boxes'].shape[0], a[0]['pred_boxes'].shape[0]) for w, a in zip(want, again)]))

run('disco', 40, 1)
run('disco', 40, 2)
run('disco_full', 10, 1, share=os.environ.get('REPRO_SHARE', '1') == '1')
