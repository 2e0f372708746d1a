"""Generates tests/golden/*.npz by running the REFERENCE'S OWN modules (imported read-only from /root/reference)
on seeded synthetic inputs with deterministic weights.  Run only in the CPU container:

    python tests/golden/make_golden.py [g1 g2 g3 g4 g7 g8 g9]

Fixtures are data (inputs, expected outputs, the config dict that produced them); no reference source is stored.
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd'))

import ref_harness as rh  # noqa: E402
from pcp_amd import synth  # noqa: E402

MINI_RANGE = [-12.8, -12.8, -8.0, 12.8, 12.8, 0.0]


WEIGHT_SCHEME = 'survey'          # 'he' while the well-conditioned fixtures (g13) are generated


def fill_weights(model):
    sd = model.state_dict()
    shapes = {k: [int(x) for x in v.shape] for k, v in sd.items()}
    filled = synth.fill_state_dict(shapes, scheme=WEIGHT_SCHEME)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    return shapes


def run_modules(model, batch_dict):
    snaps = {}
    for name, mod in zip(_module_names(model), model.module_list):
        batch_dict = mod(batch_dict)
        snaps[name] = {k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in batch_dict.items()
                       if k in ('spatial_features_2d',)}
    return batch_dict, snaps


def _module_names(model):
    names = []
    for m in model.module_list:
        for n in model.module_topology:
            if getattr(model, n, None) is m:
                names.append(n)
    return names


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def mini_points(layout, batch, n_per, seed_shift=0, xy_half=13.1):
    clouds = [synth.agent_cloud(agent=10 + b, n_points=n_per, layout=layout, seed=synth.SEED_BASE + seed_shift, xy_half=xy_half)
              for b in range(batch)]
    return clouds


class PostCapture:
    """Records, while the reference's CenterHead post-processing runs, what decides the FINAL box set and how close each decision was:
      * per frame the candidates handed to class_agnostic_nms (boxes, scores in the reference's order),
      * `near_score`: decoded top-K candidates whose score is within 1e-5 of SCORE_THRESH,
      * `topk_gap`: score[K-1] - score[K] of the heat map (the K cut; Q7: tie order is implementation defined),
      * `tie_gap`: smallest gap between consecutive candidate scores,
      * `near_iou`: candidate index pairs whose rotated BEV IoU (reference iou3d_cpu.cpp) is within 1e-4 of NMS_THRESH.
    tests/test_gpu_e2e.py demands an EXACT final set (count and one-to-one match at 1e-3) whenever these lists are empty and the gaps
    are above float noise, and only otherwise falls back to a tolerance sized by the listed cases."""

    def __init__(self, model):
        self.model = model
        self.frames = {}

    def __enter__(self):
        from pcdet.models.model_utils import centernet_utils, model_nms_utils
        from pcdet.ops.iou3d_nms import iou3d_nms_cuda
        self._cu, self._mu = centernet_utils, model_nms_utils
        self._dec, self._nms = centernet_utils.decode_bbox_from_heatmap, model_nms_utils.class_agnostic_nms
        cap = self
        cap._next = 0

        def decode(heatmap, *a, **kw):
            K, thr = kw['K'], kw['score_thresh']
            B = heatmap.shape[0]
            cap._base = cap._next
            for k in range(B):
                flat = heatmap[k].reshape(-1)
                top = torch.topk(flat, min(K + 1, flat.numel()))[0]
                f = cap.frames.setdefault(cap._base + k, {})
                f['topk_gap'] = float(top[K - 1] - top[K]) if flat.numel() > K else float('inf')
                sc = top[:K]
                f['near_score'] = sc[(sc - thr).abs() < 1e-5].numpy() if thr is not None else np.zeros(0, np.float32)
                f['score_thresh'] = float(thr) if thr is not None else -1.0
            cap._k = 0
            return cap._dec(heatmap, *a, **kw)

        def nms(box_scores, box_preds, nms_config, score_thresh=None):
            f = cap.frames.setdefault(cap._base + cap._k, {})
            cap._k += 1
            cap._next = max(cap._next, cap._base + cap._k)
            n = box_scores.shape[0]
            f['nms_scores'] = box_scores.numpy().copy()
            f['nms_boxes'] = box_preds.numpy().copy()
            f['nms_thresh'] = float(nms_config.NMS_THRESH)
            if n > 1:
                srt = torch.sort(box_scores, descending=True)[0]
                f['tie_gap'] = float((srt[:-1] - srt[1:]).min())
                iou = torch.zeros(n, n)
                iou3d_nms_cuda.boxes_iou_bev_cpu(box_preds[:, :7].contiguous(), box_preds[:, :7].contiguous(), iou)
                near = ((iou - nms_config.NMS_THRESH).abs() < 1e-4) & torch.triu(torch.ones(n, n, dtype=torch.bool), 1)
                f['near_iou'] = near.nonzero().numpy().astype(np.int32).reshape(-1, 2)
            else:
                f['tie_gap'] = float('inf')
                f['near_iou'] = np.zeros((0, 2), np.int32)
            return cap._nms(box_scores, box_preds, nms_config, score_thresh)
        centernet_utils.decode_bbox_from_heatmap = decode
        model_nms_utils.class_agnostic_nms = nms
        return self

    def __exit__(self, *exc):
        self._cu.decode_bbox_from_heatmap = self._dec
        self._mu.class_agnostic_nms = self._nms

    def dump(self, out, prefix='post'):
        for b, f in sorted(self.frames.items()):
            for k in ('nms_scores', 'nms_boxes', 'near_score', 'near_iou'):
                out['%s_%d_%s' % (prefix, b, k)] = np.asarray(f.get(k, np.zeros(0, np.float32)))
            out['%s_%d_gaps' % (prefix, b)] = np.array([f.get('topk_gap', np.inf), f.get('tie_gap', np.inf), f.get('score_thresh', -1.0),
                                                        f.get('nms_thresh', -1.0)], dtype=np.float64)     # topk, tie, SCORE_THRESH, NMS_THRESH


def capture_common(model, bd, out):
    out['voxel_coords'] = bd['voxel_coords'].numpy()
    out['pillar_features'] = bd['pillar_features'].numpy()
    out['spatial_features_2d'] = bd['spatial_features_2d'].numpy()
    pd = model.dense_head.forward_ret_dict['pred_dicts'][0]
    for k, v in pd.items():
        out['head_' + k] = v.detach().numpy()
    for b, d in enumerate(bd['final_box_dicts']):
        out['final_boxes_%d' % b] = d['pred_boxes'].numpy()
        out['final_scores_%d' % b] = d['pred_scores'].numpy()
        out['final_labels_%d' % b] = d['pred_labels'].numpy()


def g1_single(tag, yaml_name, layout, extra_points=None, score_thresh=None):
    ov = {'DATA_CONFIG.POINT_CLOUD_RANGE': MINI_RANGE}
    if score_thresh is not None:
        ov['MODEL.DENSE_HEAD.POST_PROCESSING.SCORE_THRESH'] = score_thresh
    cfg = rh.load_cfg(yaml_name, ov)
    model, ds = rh.build_model(cfg)
    shapes = fill_weights(model)
    clouds = mini_points(layout, 2, 3000)
    if extra_points is not None:
        clouds[0] = np.concatenate([clouds[0], extra_points(clouds[0].shape[1])], axis=0)
    pts = synth.collate(clouds)
    bd = {'points': torch.from_numpy(pts.copy()), 'batch_size': 2, 'metadata': [{}, {}]}
    unq_inv_holder = {}
    # capture unq_inv through the torch_scatter shim (it is not stored in batch_dict by the reference)
    import torch_scatter
    orig_mean = torch_scatter.scatter_mean

    def spy_mean(src, index, dim=0, dim_size=None):
        if src.shape[1] == 3 and 'inv' not in unq_inv_holder:
            unq_inv_holder['inv'] = index.clone()
        return orig_mean(src, index, dim, dim_size)
    torch_scatter.scatter_mean = spy_mean
    try:
        with torch.no_grad(), PostCapture(model) as post:
            bd, snaps = run_modules(model, bd)
    finally:
        torch_scatter.scatter_mean = orig_mean
    out = {'points': pts, 'unq_inv': unq_inv_holder['inv'].numpy()}
    capture_common(model, bd, out)
    post.dump(out)
    if 'backbone_2d' in snaps and cfg.MODEL.get('CORRECTOR', None) is not None:
        out['backbone_out'] = snaps['backbone_2d']['spatial_features_2d'].numpy()
    if cfg.MODEL.get('CORRECTOR', None) is not None:
        out['points_after'] = bd['points'].numpy()      # HunterJr mutates xyz in place (quirk Q8)
    out['meta_json'] = np.array(json.dumps(dict(model=rh.to_plain(cfg.MODEL), pc_range=MINI_RANGE,
                                                 voxel_size=[0.2, 0.2, 8.0], class_names=list(cfg.CLASS_NAMES),
                                                 yaml=yaml_name, layout=layout, state_shapes=shapes)))
    np.savez_compressed(os.path.join(HERE, 'g1_%s.npz' % tag), **out)
    print('g1', tag, 'P =', out['voxel_coords'].shape[0], 'boxes', [out['final_boxes_%d' % b].shape[0] for b in range(2)])


def hunter_edge_points(ncols):
    """rows whose BEV coordinate is exactly 0.0 / just below the upper edge (pins the strict float mask of
    hunter_toolbox.py:80-81) and rows outside the x/y range."""
    rows = np.zeros((6, ncols), dtype=np.float32)
    rows[:, 2] = -3.0
    rows[:, 3] = 0.5
    rows[0, 0:2] = [-12.8, 1.0]            # bev x == 0.0 exactly -> dropped by bev_scatter, kept by the VFE
    rows[1, 0:2] = [1.0, -12.8]            # bev y == 0.0
    rows[2, 0:2] = [12.799999, 0.3]       # just below the upper edge
    rows[3, 0:2] = [0.3, 12.799999]
    rows[4, 0:2] = [12.8, 0.0]            # x == max -> cell 128 -> masked by the VFE
    rows[5, 0:2] = [-12.9, 0.0]            # negative cell
    rows[:, -1] = -1.0
    return rows


def g1_disco():
    tmp = tempfile.mkdtemp()
    empty = os.path.join(tmp, 'empty.pth')
    torch.save({'model_state': {}}, empty)
    ov = {'DATA_CONFIG.POINT_CLOUD_RANGE': MINI_RANGE, 'MODEL.BEV_MAKER_RSU.CKPT': empty, 'MODEL.BEV_MAKER_CAR.CKPT': empty,
          'MODEL.BEV_MAKER_EARLY.CKPT': empty, 'MODEL.V2X_MID_FUSION.PC_RANGE_MIN': MINI_RANGE[0],
          'MODEL.DENSE_HEAD.POST_PROCESSING.SCORE_THRESH': 0.02}
    cfg = rh.load_cfg('v2x_pointpillar_disco.yaml', ov)
    model, ds = rh.build_model(cfg)
    shapes = fill_weights(model)
    # 3 agents (0 = rsu, 1 = ego, 2 = car); batch of 2; agent 2 absent from batch element 1.
    poses = {0: synth.agent_pose(0), 2: synth.agent_pose(2)}
    poses[0][:3, 3] = [0.8, -0.4, 0.0]
    poses[0][:3, :3] = synth.agent_pose(1)[:3, :3]        # yaw 0.3
    poses[2][:3, 3] = [-1.6, 2.4, 0.0]                    # exact multiples of the 0.8 m pixel -> half-pixel cases
    metadata = [{'se3_from_ego': {0: poses[0], 2: poses[2]}}, {'se3_from_ego': {0: poses[0]}}]
    clouds = []
    for b in range(2):
        per_agent = []
        for a in (0, 1, 2):
            if b == 1 and a == 2:
                continue
            c = synth.agent_cloud(agent=20 + 3 * b + a, n_points=1500, layout='disco', xy_half=13.1)
            c[:, -1] = float(a)
            per_agent.append(c)
        clouds.append(np.concatenate(per_agent, axis=0))
    pts = synth.collate(clouds)
    bd = {'points': torch.from_numpy(pts.copy()), 'batch_size': 2, 'metadata': metadata}
    with torch.no_grad(), PostCapture(model) as post:
        bd, snaps = run_modules(model, bd)
    out = {'points': pts}
    capture_common(model, bd, out)
    post.dump(out)
    out['backbone_out'] = snaps['backbone_2d']['spatial_features_2d'].numpy()
    for aid, m in bd['bev_img'].items():
        out['bev_img_%d' % aid] = m.numpy()
    out['bev_img_early_probe'] = bd['bev_img_early'].numpy()[:, ::4].copy()    # channels 0,4,8,...
    out['pose_0'] = poses[0]
    out['pose_2'] = poses[2]
    out['meta_json'] = np.array(json.dumps(dict(model=rh.to_plain(cfg.MODEL), pc_range=MINI_RANGE, voxel_size=[0.2, 0.2, 8.0],
                                                 class_names=list(cfg.CLASS_NAMES), yaml='v2x_pointpillar_disco.yaml',
                                                 layout='disco', absent=[[], [2]], state_shapes=shapes)))
    np.savez_compressed(os.path.join(HERE, 'g1_disco.npz'), **out)
    print('g1 disco', {k: v.shape for k, v in out.items() if k.startswith('bev_img')})


G2_SCORE_THRESH = {'ego': 0.02, 'early': 0.02}


def g2_full():
    out = {}
    # with the synthetic weights the ego / early models score below the YAML's SCORE_THRESH 0.1 everywhere (0 final boxes: decode + NMS
    # would be vacuous at full size), so those two run with the threshold the mini fixtures use; the test builds its model with the same
    # override, recorded here as <tag>_score_thresh
    for tag, yaml_name, layout, n_agents, thr in (('car', 'v2x_pointpillar_basic_car.yaml', 'car', 1, None),
                                                  ('ego', 'v2x_pointpillar_basic_ego.yaml', 'lately', 1, G2_SCORE_THRESH['ego']),
                                                  ('early', 'v2x_pointpillar_basic_ego_early.yaml', 'early', 6, G2_SCORE_THRESH['early'])):
        cfg = rh.load_cfg(yaml_name, {} if thr is None else {'MODEL.DENSE_HEAD.POST_PROCESSING.SCORE_THRESH': thr})
        model, ds = rh.build_model(cfg)
        fill_weights(model)
        cloud = np.concatenate([synth.agent_cloud(agent=a, n_points=60000, layout=layout) for a in range(n_agents)], axis=0)
        pts = synth.collate([cloud])
        bd = {'points': torch.from_numpy(pts.copy()), 'batch_size': 1, 'metadata': [{}]}
        import torch_scatter
        holder = {}
        orig_mean = torch_scatter.scatter_mean

        def spy_mean(src, index, dim=0, dim_size=None):
            if src.shape[1] == 3 and 'inv' not in holder:
                holder['inv'] = index.clone()
            return orig_mean(src, index, dim, dim_size)
        torch_scatter.scatter_mean = spy_mean
        try:
            with torch.no_grad(), PostCapture(model) as post:
                bd, snaps = run_modules(model, bd)
        finally:
            torch_scatter.scatter_mean = orig_mean
        post.dump(out, prefix=tag + '_post')
        out[tag + '_score_thresh'] = np.array(float(cfg.MODEL.DENSE_HEAD.POST_PROCESSING.SCORE_THRESH))
        out[tag + '_hm_max'] = np.array(float(model.dense_head.forward_ret_dict['pred_dicts'][0]['hm'].sigmoid().max()))
        vc = bd['voxel_coords'].numpy()
        pf = bd['pillar_features'].numpy()
        sf = bd['spatial_features_2d'].numpy()
        inv = holder['inv'].numpy()
        out[tag + '_P'] = np.array(vc.shape[0])
        out[tag + '_N'] = np.array(pts.shape[0])
        out[tag + '_coords_sha'] = np.array(sha(vc.astype(np.int32)))
        out[tag + '_inv_sha'] = np.array(sha(inv.astype(np.int64)))
        out[tag + '_cnt_hist'] = np.bincount(np.bincount(inv), minlength=16)[:64]
        out[tag + '_pf_sum'] = pf.astype(np.float64).sum(0)
        out[tag + '_pf_abs'] = np.abs(pf.astype(np.float64)).sum(0)
        out[tag + '_pf_max'] = pf.max(0)
        out[tag + '_sf2d_sum'] = sf.astype(np.float64).sum((0, 2, 3))
        out[tag + '_sf2d_abs'] = np.abs(sf.astype(np.float64)).sum((0, 2, 3))
        out[tag + '_sf2d_max'] = sf.max(axis=(0, 2, 3))
        out[tag + '_sf2d_probe'] = sf[0, :, ::16, ::16].copy()
        pd = model.dense_head.forward_ret_dict['pred_dicts'][0]
        out[tag + '_hm_probe'] = pd['hm'].numpy()[0, 0, ::4, ::4].copy()
        for name in ('center', 'center_z', 'dim', 'rot', 'hm'):          # the complete head maps: decode + NMS are pinned EXACTLY on them
            out[tag + '_head_' + name] = pd[name].numpy().copy()
        out[tag + '_boxes'] = bd['final_box_dicts'][0]['pred_boxes'].numpy()
        out[tag + '_scores'] = bd['final_box_dicts'][0]['pred_scores'].numpy()
        print('g2', tag, 'N', pts.shape[0], 'P', vc.shape[0], 'final', out[tag + '_boxes'].shape[0])
    np.savez_compressed(os.path.join(HERE, 'g2_full.npz'), **out)


def g2_disco_full():
    """Config 5 at BASELINE's full size: 6 agents x 60 000 points, 512 x 512 grid, one frame, the poses of SURVEY 8(d); digests of what
    the reference's DiscoNet forward produces (per-agent BEV maps, fused map, detections)."""
    tmp = tempfile.mkdtemp()
    empty = os.path.join(tmp, 'empty.pth')
    torch.save({'model_state': {}}, empty)
    ov = {'MODEL.BEV_MAKER_RSU.CKPT': empty, 'MODEL.BEV_MAKER_CAR.CKPT': empty, 'MODEL.BEV_MAKER_EARLY.CKPT': empty}
    cfg = rh.load_cfg('v2x_pointpillar_disco.yaml', ov)
    model, ds = rh.build_model(cfg)
    fill_weights(model)
    agents = (0, 1, 2, 3, 4, 5)
    poses = {a: synth.agent_pose(a) for a in agents if a != 1}
    clouds = []
    for a in agents:
        c = synth.agent_cloud(agent=a, n_points=60000, layout='disco')
        c[:, -1] = float(a)
        clouds.append(c)
    pts = synth.collate([np.concatenate(clouds, axis=0)])
    bd = {'points': torch.from_numpy(pts.copy()), 'batch_size': 1, 'metadata': [{'se3_from_ego': poses}]}
    # the ego -> agent point transform of bev_maker.py:172-179 as the frozen chain receives it: per agent, in call order
    seen = []
    hook = model.bev_maker_car.module_list[0].register_forward_pre_hook(lambda m, args: seen.append(args[0]['points'].detach().clone()))
    with torch.no_grad(), PostCapture(model) as post:
        bd, snaps = run_modules(model, bd)
    hook.remove()
    out = {'N': np.array(pts.shape[0])}
    post.dump(out)
    car_agents = [a for a in agents if a != 1]
    assert len(seen) == len(car_agents)
    for a, ap in zip(car_agents, seen):
        ap = ap.numpy()
        assert np.all(ap[:, -1] == a)
        out['car_agent_%d_rows' % a] = np.array(ap.shape[0])
        out['car_agent_%d_xyz_sha' % a] = np.array(sha(ap[:, 1:4].astype(np.float32)))
        out['car_agent_%d_xyz_head' % a] = ap[:8, 1:4].copy()
    pd = model.dense_head.forward_ret_dict['pred_dicts'][0]
    for name in ('center', 'center_z', 'dim', 'rot', 'hm'):
        out['head_' + name] = pd[name].numpy().copy()
    for a in poses:
        out['pose_%d' % a] = poses[a]
    for aid, m in bd['bev_img'].items():
        a = m.numpy()
        out['bev_%d_probe' % aid] = a[0, ::8, ::8, ::8].copy()
        out['bev_%d_sum' % aid] = a.astype(np.float64).sum((0, 2, 3))
        out['bev_%d_max' % aid] = a.max(axis=(0, 2, 3))
    sf = bd['spatial_features_2d'].numpy()
    out['sf2d_probe'] = sf[0, :, ::16, ::16].copy()
    out['sf2d_sum'] = sf.astype(np.float64).sum((0, 2, 3))
    out['sf2d_max'] = sf.max(axis=(0, 2, 3))
    out['voxel_P'] = np.array(bd['voxel_coords'].shape[0])
    out['coords_sha'] = np.array(sha(bd['voxel_coords'].numpy().astype(np.int32)))
    out['boxes'] = bd['final_box_dicts'][0]['pred_boxes'].numpy()
    out['scores'] = bd['final_box_dicts'][0]['pred_scores'].numpy()
    print('g2 disco full: N', pts.shape[0], 'P', int(out['voxel_P']), 'agents', sorted(bd['bev_img'].keys()), 'final', out['boxes'].shape[0])
    np.savez_compressed(os.path.join(HERE, 'g2_disco_full.npz'), **out)


def g3_nms():
    rh.install()
    nmsmod = sys.modules['pcdet.ops.iou3d_nms.iou3d_nms_cuda']
    from pcdet.ops.iou3d_nms import iou3d_nms_utils
    n = 500
    s = synth.SEED_BASE + 33
    # clustered boxes so that many pairs overlap: 60 cluster centres, jitter 1.5 m
    cx = synth.uniform(s, 1, 60, -45, 45)
    cy = synth.uniform(s, 2, 60, -45, 45)
    which = (synth.uniform01(s, 3, n) * 60).astype(np.int64)
    boxes = np.zeros((n, 7), dtype=np.float32)
    boxes[:, 0] = cx[which] + synth.uniform(s, 4, n, -1.5, 1.5)
    boxes[:, 1] = cy[which] + synth.uniform(s, 5, n, -1.5, 1.5)
    boxes[:, 2] = synth.uniform(s, 6, n, -3, -1)
    boxes[:, 3] = synth.uniform(s, 7, n, 3.0, 5.5)
    boxes[:, 4] = synth.uniform(s, 8, n, 1.5, 2.5)
    boxes[:, 5] = synth.uniform(s, 9, n, 1.4, 2.0)
    boxes[:, 6] = synth.uniform(s, 10, n, -3.14159, 3.14159)
    boxes[7, 6] = 0.0                                   # axis-aligned cases
    boxes[8] = boxes[7]
    boxes[8, 0] += 0.5
    boxes[9] = boxes[7]                                  # exact duplicate
    scores = synth.uniform(s, 11, n, 0.1, 1.0)
    scores = (np.argsort(np.argsort(scores)).astype(np.float32) + 1.0) / np.float32(n + 1)   # distinct scores
    out = {'boxes': boxes, 'scores': scores}
    tb, tsc = torch.from_numpy(boxes), torch.from_numpy(scores)
    order = torch.sort(tsc, descending=True)[1]
    iou = torch.zeros(n, n)
    nmsmod.ref.boxes_iou_bev_cpu(tb[order].contiguous(), tb[order].contiguous(), iou)
    out['iou_sorted'] = iou.numpy()
    out['order'] = order.numpy()
    for thr in (0.2, 0.3):
        keep, _ = iou3d_nms_utils.nms_gpu(tb, tsc, thr)
        out['keep_%02d' % int(thr * 10)] = keep.numpy()
        print('g3 thr', thr, 'kept', keep.shape[0])
    np.savez_compressed(os.path.join(HERE, 'g3_nms.npz'), **out)


def g4_warp():
    rh.install()
    from pcdet.models.bev_layers.v2x_fusion_disco import transform_bev_img
    out = {}
    cases = []
    for (H, pc_min, pix) in ((32, -12.8, 0.8), (128, -51.2, 0.8)):
        yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='ij')
        img = np.stack([xx + 1.0, yy + 1.0, 1000.0 + yy * H + xx], axis=0).astype(np.float32)
        tfs = [(0.0, 0.0, 0.0), (0.0, 0.8, 0.0), (0.0, 0.4, -0.4), (0.0, -1.2, 2.0), (np.pi / 2, 0.0, 0.0), (0.3, 3.0, -2.0),
               (-1.1, 5.1, 0.77), (np.pi, 0.4, 0.4)]
        for i, (yaw, tx, ty) in enumerate(tfs):
            T = np.eye(4, dtype=np.float64)
            c, s = np.cos(yaw), np.sin(yaw)
            T[:2, :2] = [[c, -s], [s, c]]
            T[0, 3], T[1, 3] = tx, ty
            Tt = torch.from_numpy(T).float()
            res = transform_bev_img(Tt, torch.from_numpy(img), pc_min, pix).numpy()
            key = 'H%d_case%d' % (H, i)
            out[key + '_T'] = Tt.numpy()
            out[key + '_out'] = res
            cases.append(key)
        out['H%d_img' % H] = img
        out['H%d_params' % H] = np.array([pc_min, pix], dtype=np.float64)
    out['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, 'g4_warp.npz'), **out)
    print('g4', len(cases), 'cases')


def train_gt_boxes(batch, m_max, seed_shift):
    """(B, M, 8) [x, y, z, dx, dy, dz, heading, class]; rows past the per-frame count are zero (collate_batch padding,
    dataset.py:260-266); one box with its centre outside the range (clamped by center_head.py:131-132)."""
    s = synth.SEED_BASE + 700 + seed_shift
    gt = np.zeros((batch, m_max, 8), dtype=np.float32)
    for b in range(batch):
        n = m_max - 2 * b
        gt[b, :n, 0] = synth.uniform(s, 10 * b + 1, n, -12.0, 12.0)
        gt[b, :n, 1] = synth.uniform(s, 10 * b + 2, n, -12.0, 12.0)
        gt[b, :n, 2] = synth.uniform(s, 10 * b + 3, n, -3.0, -1.0)
        gt[b, :n, 3] = synth.uniform(s, 10 * b + 4, n, 3.0, 5.5)
        gt[b, :n, 4] = synth.uniform(s, 10 * b + 5, n, 1.5, 2.5)
        gt[b, :n, 5] = synth.uniform(s, 10 * b + 6, n, 1.4, 2.0)
        gt[b, :n, 6] = synth.uniform(s, 10 * b + 7, n, -3.14159, 3.14159)
        gt[b, :n, 7] = 1.0
    gt[0, 0, 0:2] = [12.9, -13.5]          # centre outside the map -> clamped to the border cell
    gt[0, 1, 0:2] = gt[0, 2, 0:2] + 0.3    # two boxes in neighbouring / the same cell (gaussian max blending)
    return gt


def _digest(t):
    a = t.detach().double().reshape(-1)
    return np.array([float(a.norm()), float(a.sum()), float(a.abs().max())], dtype=np.float64)


def _sample(t, cap=4096):
    a = t.detach().reshape(-1)
    if a.numel() <= cap:
        return a.numpy().copy()
    step = a.numel() // 1024
    return a[::step][:1024].numpy().copy()


def g7_train():
    """Config 5 (disco) training contract on the mini geometry: two iterations of the reference's own train step
    (tools/train_utils/train_utils.py:39-65) with its own optimizer / scheduler builders."""
    tmp = tempfile.mkdtemp()
    empty = os.path.join(tmp, 'empty.pth')
    torch.save({'model_state': {}}, empty)
    ov = {'DATA_CONFIG.POINT_CLOUD_RANGE': MINI_RANGE, 'MODEL.BEV_MAKER_RSU.CKPT': empty, 'MODEL.BEV_MAKER_CAR.CKPT': empty,
          'MODEL.BEV_MAKER_EARLY.CKPT': empty, 'MODEL.V2X_MID_FUSION.PC_RANGE_MIN': MINI_RANGE[0]}
    cfg = rh.load_cfg('v2x_pointpillar_disco.yaml', ov)
    model, ds = rh.build_model(cfg)
    shapes = fill_weights(model)
    sys.path.insert(0, os.path.join(rh.REF_ROOT, 'tools'))
    from train_utils.optimization import build_optimizer, build_scheduler
    from torch.nn.utils import clip_grad_norm_
    poses = {0: synth.agent_pose(0), 2: synth.agent_pose(2)}
    poses[0][:3, 3] = [0.8, -0.4, 0.0]
    poses[0][:3, :3] = synth.agent_pose(1)[:3, :3]
    poses[2][:3, 3] = [-1.6, 2.4, 0.0]
    metadata = [{'se3_from_ego': {0: poses[0], 2: poses[2]}}, {'se3_from_ego': {0: poses[0]}}]
    clouds = []
    for b in range(2):
        per_agent = []
        for a in (0, 1, 2):
            if b == 1 and a == 2:
                continue
            c = synth.agent_cloud(agent=40 + 3 * b + a, n_points=1500, layout='disco', xy_half=13.1)
            c[:, -1] = float(a)
            per_agent.append(c)
        clouds.append(np.concatenate(per_agent, axis=0))
    pts = synth.collate(clouds)
    gt = train_gt_boxes(2, 7, 0)
    total_it_each_epoch, epochs = 5, cfg.OPTIMIZATION.NUM_EPOCHS
    optimizer = build_optimizer(model, cfg.OPTIMIZATION)
    lr_scheduler, _ = build_scheduler(optimizer, total_iters_each_epoch=total_it_each_epoch, total_epochs=epochs,
                                      last_epoch=-1, optim_cfg=cfg.OPTIMIZATION)
    out = {'points': pts, 'gt_boxes': gt, 'pose_0': poses[0], 'pose_2': poses[2]}
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    out['trainable'] = np.array(names)
    for it in range(2):
        lr_scheduler.step(it)
        out['it%d_lr' % it] = np.array(float(optimizer.lr))
        out['it%d_mom' % it] = np.array(float(optimizer.mom))
        model.train()
        optimizer.zero_grad()
        bd = {'points': torch.from_numpy(pts.copy()), 'batch_size': 2, 'metadata': metadata, 'gt_boxes': torch.from_numpy(gt.copy())}
        ret, tb, _disp = model(bd)
        loss = ret['loss']
        model.update_global_step()
        loss.backward()
        out['it%d_loss' % it] = np.array(float(loss))
        out['it%d_tb_json' % it] = np.array(json.dumps({k: float(v) for k, v in tb.items()}))
        if it == 0:
            td = model.dense_head.forward_ret_dict['target_dicts']
            out['tgt_heatmap'] = td['heatmaps'][0].numpy()
            out['tgt_boxes'] = td['target_boxes'][0].numpy()
            out['tgt_inds'] = td['inds'][0].numpy()
            out['tgt_mask'] = td['masks'][0].numpy()
            out['fused_probe'] = bd['spatial_features_2d'].detach().numpy()[:, ::8].copy()
        params = dict(model.named_parameters())
        out['it%d_grad_digest' % it] = np.stack([_digest(params[n].grad) for n in names])
        if it == 0:
            for n in names:
                out['g0/' + n] = _sample(params[n].grad)
        norm = clip_grad_norm_(model.parameters(), cfg.OPTIMIZATION.GRAD_NORM_CLIP)
        out['it%d_grad_norm' % it] = np.array(float(norm))
        optimizer.step()
        out['it%d_param_digest' % it] = np.stack([_digest(params[n]) for n in names])
        if it == 0:
            for n in names:
                out['p1/' + n] = _sample(params[n])
            sd = model.state_dict()
            bn_keys = [k for k in sd if ('running_' in k) and not k.startswith('bev_maker')]
            out['bn_keys'] = np.array(bn_keys)
            out['it0_bn_digest'] = np.stack([_digest(sd[k]) for k in bn_keys])
        print('g7 it', it, 'loss', float(loss), 'norm', float(norm), tb)
    out['meta_json'] = np.array(json.dumps(dict(model=rh.to_plain(cfg.MODEL), optimization=rh.to_plain(cfg.OPTIMIZATION),
                                                 pc_range=MINI_RANGE, voxel_size=[0.2, 0.2, 8.0],
                                                 class_names=list(cfg.CLASS_NAMES), yaml='v2x_pointpillar_disco.yaml',
                                                 layout='disco', absent=[[], [2]], state_shapes=shapes,
                                                 total_it_each_epoch=total_it_each_epoch)))
    np.savez_compressed(os.path.join(HERE, 'g7_train.npz'), **out)
    print('g7 saved', os.path.getsize(os.path.join(HERE, 'g7_train.npz')) // 1024, 'KiB')


def g7_train_full(b4=False):
    """Config 5 training at BASELINE's full size (6 agents x 60 000 points, 512 x 512 grid, one frame, 12 GT boxes): ONE iteration of the
    reference's train step; loss terms, gradient norm and per-tensor gradient digests.
    b4=True (round 6): the batch bench.py --train times -- bench.make_points(CONFIGS['disco'], 4, 0) with bench.make_gt_boxes(4, 0) -- into
    g7_train_full_b4.npz."""
    tmp = tempfile.mkdtemp()
    empty = os.path.join(tmp, 'empty.pth')
    torch.save({'model_state': {}}, empty)
    ov = {'MODEL.BEV_MAKER_RSU.CKPT': empty, 'MODEL.BEV_MAKER_CAR.CKPT': empty, 'MODEL.BEV_MAKER_EARLY.CKPT': empty}
    cfg = rh.load_cfg('v2x_pointpillar_disco.yaml', ov)
    model, ds = rh.build_model(cfg)
    fill_weights(model)
    sys.path.insert(0, os.path.join(rh.REF_ROOT, 'tools'))
    from train_utils.optimization import build_optimizer, build_scheduler
    from torch.nn.utils import clip_grad_norm_
    agents = (0, 1, 2, 3, 4, 5)
    poses = {a: synth.agent_pose(a) for a in agents if a != 1}
    clouds = []
    for a in agents:
        c = synth.agent_cloud(agent=a, n_points=60000, layout='disco')
        c[:, -1] = float(a)
        clouds.append(c)
    pts = synth.collate([np.concatenate(clouds, axis=0)])
    s0 = synth.SEED_BASE + 950
    n = 12
    gt = np.zeros((1, n, 8), dtype=np.float32)
    gt[0, :, 0] = synth.uniform(s0, 1, n, -48.0, 48.0)
    gt[0, :, 1] = synth.uniform(s0, 2, n, -48.0, 48.0)
    gt[0, :, 2] = synth.uniform(s0, 3, n, -3.0, -1.0)
    gt[0, :, 3] = synth.uniform(s0, 4, n, 3.0, 5.5)
    gt[0, :, 4] = synth.uniform(s0, 5, n, 1.5, 2.5)
    gt[0, :, 5] = synth.uniform(s0, 6, n, 1.4, 2.0)
    gt[0, :, 6] = synth.uniform(s0, 7, n, -3.14159, 3.14159)
    gt[0, :, 7] = 1.0
    optimizer = build_optimizer(model, cfg.OPTIMIZATION)
    lr_scheduler, _ = build_scheduler(optimizer, total_iters_each_epoch=5, total_epochs=cfg.OPTIMIZATION.NUM_EPOCHS, last_epoch=-1,
                                      optim_cfg=cfg.OPTIMIZATION)
    names = [n_ for n_, p_ in model.named_parameters() if p_.requires_grad]
    lr_scheduler.step(0)
    model.train()
    optimizer.zero_grad()
    B = 1
    metas = [{'se3_from_ego': poses}]
    if b4:
        sys.path.insert(0, REPO)
        import bench
        B = 4
        pts, metas = bench.make_points(bench.CONFIGS['disco'], 4, 0)
        gt = bench.make_gt_boxes(4, 0)
    bd = {'points': torch.from_numpy(pts.copy()), 'batch_size': B, 'metadata': metas, 'gt_boxes': torch.from_numpy(gt.copy())}
    ret, tb, _disp = model(bd)
    loss = ret['loss']
    loss.backward()
    params = dict(model.named_parameters())
    out = {'gt_boxes': gt, 'N': np.array(pts.shape[0]), 'trainable': np.array(names), 'loss': np.array(float(loss)),
           'tb_json': np.array(json.dumps({k: float(v) for k, v in tb.items()})),
           'grad_digest': np.stack([_digest(params[n_].grad) for n_ in names])}
    for a in poses:
        out['pose_%d' % a] = poses[a]
    norm = clip_grad_norm_(model.parameters(), cfg.OPTIMIZATION.GRAD_NORM_CLIP)
    out['grad_norm'] = np.array(float(norm))
    out['optimization_json'] = np.array(json.dumps(rh.to_plain(cfg.OPTIMIZATION)))
    print('g7 full: loss', float(loss), 'norm', float(norm), tb)
    out['batch'] = np.array(B)
    np.savez_compressed(os.path.join(HERE, 'g7_train_full_b4.npz' if b4 else 'g7_train_full.npz'), **out)


def g7b_train_single(tag, yaml_name, layout):
    """Training contract of a single-model config (VFE -> scatter -> backbone -> CenterHead, no fusion: configs 3 / 4) on the mini
    geometry: two iterations of the reference's own train step with its own optimizer / scheduler builders."""
    cfg = rh.load_cfg(yaml_name, {'DATA_CONFIG.POINT_CLOUD_RANGE': MINI_RANGE})
    model, ds = rh.build_model(cfg)
    shapes = fill_weights(model)
    sys.path.insert(0, os.path.join(rh.REF_ROOT, 'tools'))
    from train_utils.optimization import build_optimizer, build_scheduler
    from torch.nn.utils import clip_grad_norm_
    pts = synth.collate(mini_points(layout, 2, 2500))
    gt = train_gt_boxes(2, 6, 1)
    total_it_each_epoch, epochs = 5, cfg.OPTIMIZATION.NUM_EPOCHS
    optimizer = build_optimizer(model, cfg.OPTIMIZATION)
    lr_scheduler, _ = build_scheduler(optimizer, total_iters_each_epoch=total_it_each_epoch, total_epochs=epochs,
                                      last_epoch=-1, optim_cfg=cfg.OPTIMIZATION)
    out = {'points': pts, 'gt_boxes': gt}
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    out['trainable'] = np.array(names)
    for it in range(2):
        lr_scheduler.step(it)
        out['it%d_lr' % it] = np.array(float(optimizer.lr))
        out['it%d_mom' % it] = np.array(float(optimizer.mom))
        model.train()
        optimizer.zero_grad()
        bd = {'points': torch.from_numpy(pts.copy()), 'batch_size': 2, 'metadata': [{}, {}], 'gt_boxes': torch.from_numpy(gt.copy())}
        ret, tb, _disp = model(bd)
        loss = ret['loss']
        model.update_global_step()
        loss.backward()
        out['it%d_loss' % it] = np.array(float(loss))
        out['it%d_tb_json' % it] = np.array(json.dumps({k: float(v) for k, v in tb.items()}))
        params = dict(model.named_parameters())
        out['it%d_grad_digest' % it] = np.stack([_digest(params[n].grad) for n in names])
        if it == 0:
            out['map_probe'] = bd['spatial_features_2d'].detach().numpy()[:, ::8].copy()
            for n in names:
                out['g0/' + n] = _sample(params[n].grad)
        norm = clip_grad_norm_(model.parameters(), cfg.OPTIMIZATION.GRAD_NORM_CLIP)
        out['it%d_grad_norm' % it] = np.array(float(norm))
        optimizer.step()
        if it == 0:
            for n in names:
                out['p1/' + n] = _sample(params[n])
            sd = model.state_dict()
            bn_keys = [k for k in sd if 'running_' in k]
            out['bn_keys'] = np.array(bn_keys)
            out['it0_bn_digest'] = np.stack([_digest(sd[k]) for k in bn_keys])
        print('g7b', tag, 'it', it, 'loss', float(loss), 'norm', float(norm), tb)
    out['meta_json'] = np.array(json.dumps(dict(model=rh.to_plain(cfg.MODEL), optimization=rh.to_plain(cfg.OPTIMIZATION),
                                                 pc_range=MINI_RANGE, voxel_size=[0.2, 0.2, 8.0], class_names=list(cfg.CLASS_NAMES),
                                                 yaml=yaml_name, layout=layout, state_shapes=shapes,
                                                 num_point_features=int(ds.point_feature_encoder.num_point_features),
                                                 total_it_each_epoch=total_it_each_epoch)))
    path = os.path.join(HERE, 'g7b_train_%s.npz' % tag)
    np.savez_compressed(path, **out)
    print('g7b saved', tag, os.path.getsize(path) // 1024, 'KiB')


PFN_VARIANTS = {
    # tag: (num_raw, USE_ABSLOTE_XYZ, WITH_DISTANCE, NUM_FILTERS, USE_NORM)
    'dist': (5, True, True, [64, 64], True),
    'rel': (5, False, False, [64, 64], True),
    'one': (4, True, False, [64], True),
    'three': (5, False, True, [32, 64, 128], True),
    'wide_nonorm': (7, True, True, [48, 96], False),
}


def g16_pfn_variants():
    """the reference's DynamicPillarVFE (+ PointPillarScatter) alone, with the compositions none of its configs use"""
    rh.install()
    from pcdet.models.backbones_3d.vfe.dynamic_pillar_vfe import DynamicPillarVFE
    from pcdet.models.backbones_2d.map_to_bev.pointpillar_scatter import PointPillarScatter
    assert DynamicPillarVFE.__module__.startswith('pcdet') and '/root/reference' in sys.modules[DynamicPillarVFE.__module__].__file__
    out = {}
    voxel = [0.2, 0.2, 8.0]
    rng_pc = np.asarray(MINI_RANGE, dtype=np.float32)
    grid = np.round((rng_pc[3:] - rng_pc[:3]) / np.asarray(voxel, dtype=np.float32)).astype(np.int64)
    base = synth.collate(mini_points('lately', 2, 700, seed_shift=16))          # 11 raw columns
    meta = {}
    for tag, (nr, use_abs, with_dist, filters, use_norm) in PFN_VARIANTS.items():
        mc = rh.AttrDict(NAME='DynPillarVFE', WITH_DISTANCE=with_dist, USE_ABSLOTE_XYZ=use_abs, USE_NORM=use_norm, NUM_FILTERS=filters)
        vfe = DynamicPillarVFE(model_cfg=mc, num_point_features=nr, voxel_size=voxel, grid_size=grid, point_cloud_range=rng_pc).eval()
        shapes = {'vfe.' + k: [int(x) for x in v.shape] for k, v in vfe.state_dict().items()}
        filled = synth.fill_state_dict(shapes, scheme='he')
        vfe.load_state_dict({k[len('vfe.'):]: torch.from_numpy(v) for k, v in filled.items()})
        pts = np.ascontiguousarray(base[:, :1 + nr])
        sc = PointPillarScatter(model_cfg=rh.AttrDict(NUM_BEV_FEATURES=filters[-1]), grid_size=grid)
        with torch.no_grad():
            bd = sc(vfe({'points': torch.from_numpy(pts.copy()), 'batch_size': 2}))
        out[tag + '_points'] = pts
        out[tag + '_pillar_features'] = bd['pillar_features'].numpy()
        out[tag + '_voxel_coords'] = bd['voxel_coords'].numpy().astype(np.int32)
        out[tag + '_spatial_sha'] = np.array(sha(bd['spatial_features'].numpy()))
        meta[tag] = dict(num_raw=nr, use_absolute_xyz=use_abs, with_distance=with_dist, vfe_filters=filters, use_norm=use_norm,
                         state_shapes=shapes)
        print('g16', tag, 'P =', out[tag + '_voxel_coords'].shape[0], 'C =', out[tag + '_pillar_features'].shape[1])
    out['meta_json'] = np.array(json.dumps(dict(variants=meta, pc_range=MINI_RANGE, voxel_size=voxel, grid_size=[int(g) for g in grid],
                                                weight_scheme='he')))
    np.savez_compressed(os.path.join(HERE, 'g16_pfn_variants.npz'), **out)


def g8_exchange():
    """Lately-fusion exchange: runs the reference's own apply_se3_ and the lines of v2x_sim_dataset_ego.py:196-232
    (torch.unique + scatter(mean) through the shim) on seeded MoDAR boxes / foreground points.  points_in_boxes_gpu is CUDA-only in
    the reference; its indices come from oracle/exchange.py (restated from the kernel source) and are stored as data.  Round 5: the
    reference's own points_in_boxes_cpu (roiaware_pool3d.cpp:143-166, compiled from where it lies by oracle/build_ref.py) runs on the same
    boxes and points: its first-true index per point is stored as `box_idx_ref_cpu` (the CPU function tests with MARGIN 1e-2, the CUDA kernel
    with 1e-5: the two agree outside a 1 cm band around the box faces, `box_idx_margin_band` marks the points inside it)."""
    rh.install()
    sys.path.insert(0, REPO)
    from oracle import exchange as oex
    from pcdet.datasets.nuscenes.nuscenes_temporal_utils import apply_se3_
    from torch_scatter import scatter
    s = synth.SEED_BASE + 88
    n, m = 40, 3000
    modar = np.zeros((n, 9), dtype=np.float32)
    modar[:, 0] = synth.uniform(s, 1, n, -30, 30)
    modar[:, 1] = synth.uniform(s, 2, n, -30, 30)
    modar[:, 2] = synth.uniform(s, 3, n, -3, -1)
    modar[:, 3] = synth.uniform(s, 4, n, 3.0, 5.5)
    modar[:, 4] = synth.uniform(s, 5, n, 1.5, 2.5)
    modar[:, 5] = synth.uniform(s, 6, n, 1.4, 2.0)
    modar[:, 6] = synth.uniform(s, 7, n, -3.14159, 3.14159)
    modar[:, 7] = synth.uniform(s, 8, n, 0.1, 1.0)
    modar[:, 8] = 1.0
    modar[5, :7] = modar[4, :7]                                  # overlapping boxes: the first one wins the points
    modar[5, 0] += 0.4
    which = (synth.uniform01(s, 9, m) * (n + 8)).astype(np.int64)        # some points belong to no box
    fg = np.zeros((m, 13), dtype=np.float32)
    ctr = np.concatenate([modar[:, :3], np.full((8, 3), 80.0, dtype=np.float32)], 0)[which]
    fg[:, 0] = ctr[:, 0] + synth.uniform(s, 10, m, -2.5, 2.5)
    fg[:, 1] = ctr[:, 1] + synth.uniform(s, 11, m, -2.5, 2.5)
    fg[:, 2] = ctr[:, 2] + synth.uniform(s, 12, m, -1.0, 1.0)
    fg[:, 3:10] = synth.uniform(s, 13, m * 7, 0, 1).reshape(m, 7)
    fg[:, 10:13] = synth.uniform(s, 14, m * 3, -1.5, 1.5).reshape(m, 3)
    T = synth.agent_pose(3)
    max_sweep_idx = 10.0
    # ---- the reference's lines (v2x_sim_dataset_ego.py:203-232) ----
    modar_t, foregr = torch.from_numpy(modar.copy()), torch.from_numpy(fg.copy())
    box_idx = torch.from_numpy(oex.points_in_boxes(fg[:, :3], modar[:, :7])).long()
    mask_valid = box_idx > -1
    foregr = foregr[mask_valid]
    bi = box_idx[mask_valid]
    unq, inv = torch.unique(bi, return_inverse=True)
    boxes_offset = scatter(foregr[:, -3:], inv, dim=0, reduce='mean') * 2.
    modar_t[unq, :3] += boxes_offset
    mod = modar_t.numpy()
    mod[:, :7] = apply_se3_(T, boxes_=mod[:, :7], return_transformed=True)
    rows = np.zeros((mod.shape[0], 13))
    rows[:, :3] = mod[:, :3]
    rows[:, 4] = 0.
    rows[:, 5:11] = mod[:, 3:]
    rows[:, -2] = max_sweep_idx
    rows[:, -1] = -1
    out = dict(modar=modar, foreground=fg, pose=T, max_sweep_idx=np.array(max_sweep_idx), box_idx=box_idx.numpy().astype(np.int32),
               rows=rows.astype(np.float32))
    from oracle import build_ref
    ref_mask = build_ref.ref_points_in_boxes_cpu(fg[:, :3], modar[:, :7])                  # (boxes, points), the reference's compiled code
    assert ref_mask is not None, 'oracle/_ref/ref_roiaware_pool3d.so could not be built'
    ref_first = np.where(ref_mask.any(0), ref_mask.argmax(0), -1).astype(np.int32)
    band = ref_first != out['box_idx']
    assert np.array_equal(ref_first, oex.points_in_boxes(fg[:, :3], modar[:, :7], margin=1e-2)) and 0 < int(band.sum()) < 20
    out['box_idx_ref_cpu'] = ref_first
    out['box_idx_margin_band'] = band
    # no-foreground variant (path_foregr missing, :205)
    mod2 = modar.copy()
    mod2[:, :7] = apply_se3_(T, boxes_=mod2[:, :7], return_transformed=True)
    rows2 = np.zeros((n, 13))
    rows2[:, :3] = mod2[:, :3]
    rows2[:, 5:11] = mod2[:, 3:]
    rows2[:, -2] = max_sweep_idx
    rows2[:, -1] = -1
    out['rows_no_foreground'] = rows2.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'g8_exchange.npz'), **out)
    print('g8 boxes with points:', len(unq), 'points in boxes:', int(mask_valid.sum()))


ANCHOR_HEAD = {
    'NAME': 'AnchorHeadSingle', 'CLASS_AGNOSTIC': False, 'USE_DIRECTION_CLASSIFIER': True, 'DIR_OFFSET': 0.78539, 'DIR_LIMIT_OFFSET': 0.0,
    'NUM_DIR_BINS': 2,
    'ANCHOR_GENERATOR_CONFIG': [
        {'class_name': 'car', 'anchor_sizes': [[3.9, 1.6, 1.56]], 'anchor_rotations': [0, 1.57], 'anchor_bottom_heights': [-1.78],
         'align_center': False, 'feature_map_stride': 4, 'matched_threshold': 0.6, 'unmatched_threshold': 0.45},
        {'class_name': 'pedestrian', 'anchor_sizes': [[0.8, 0.6, 1.73]], 'anchor_rotations': [0, 1.57], 'anchor_bottom_heights': [-0.6],
         'align_center': False, 'feature_map_stride': 4, 'matched_threshold': 0.5, 'unmatched_threshold': 0.35},
        {'class_name': 'cyclist', 'anchor_sizes': [[1.76, 0.6, 1.73]], 'anchor_rotations': [0, 1.57], 'anchor_bottom_heights': [-0.6],
         'align_center': False, 'feature_map_stride': 4, 'matched_threshold': 0.5, 'unmatched_threshold': 0.35}],
    'TARGET_ASSIGNER_CONFIG': {'NAME': 'AxisAlignedTargetAssigner', 'POS_FRACTION': -1.0, 'SAMPLE_SIZE': 512, 'NORM_BY_NUM_EXAMPLES': False,
                               'MATCH_HEIGHT': False, 'BOX_CODER': 'ResidualCoder'},
    'LOSS_CONFIG': {'LOSS_WEIGHTS': {'cls_weight': 1.0, 'loc_weight': 2.0, 'dir_weight': 0.2,
                                     'code_weights': [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]}}}


def g9_anchor(tag, multi_class):
    """MODEL.NAME PointPillar with AnchorHeadSingle (no YAML of the V2X-Sim set uses it: the head block is the one
    of tools/cfgs/custom_models/second.yaml with three anchor classes at stride 4), trunk of v2x_pointpillar_basic_ego.yaml."""
    post = {'RECALL_THRESH_LIST': [0.3, 0.5, 0.7], 'SCORE_THRESH': 0.5, 'OUTPUT_RAW_SCORE': False, 'EVAL_METRIC': 'kitti',
            'NMS_CONFIG': {'MULTI_CLASSES_NMS': multi_class, 'NMS_TYPE': 'nms_gpu', 'NMS_THRESH': 0.1, 'NMS_PRE_MAXSIZE': 1024,
                           'NMS_POST_MAXSIZE': 100}}
    ov = {'DATA_CONFIG.POINT_CLOUD_RANGE': MINI_RANGE, 'CLASS_NAMES': ['car', 'pedestrian', 'cyclist'], 'MODEL.NAME': 'PointPillar',
          'MODEL.DENSE_HEAD': rh.AttrDict(ANCHOR_HEAD), 'MODEL.POST_PROCESSING': rh.AttrDict(post)}
    cfg = rh.load_cfg('v2x_pointpillar_basic_ego.yaml', ov)
    model, ds = rh.build_model(cfg)
    shapes = fill_weights(model)
    clouds = mini_points('lately', 2, 3000)
    pts = synth.collate(clouds)
    bd = {'points': torch.from_numpy(pts.copy()), 'batch_size': 2, 'metadata': [{}, {}]}
    with torch.no_grad():
        for mod in model.module_list:
            bd = mod(bd)
        pred_dicts, _ = model.post_processing(bd)
    out = {'points': pts, 'spatial_features_2d_probe': bd['spatial_features_2d'].numpy()[:, ::8].copy(), 'batch_cls_preds': bd['batch_cls_preds'].numpy(),
           'batch_box_preds': bd['batch_box_preds'].numpy(), 'anchors': torch.cat(model.dense_head.anchors, dim=-3).numpy()}
    for b, d in enumerate(pred_dicts):
        out['final_boxes_%d' % b] = d['pred_boxes'].numpy()
        out['final_scores_%d' % b] = d['pred_scores'].numpy()
        out['final_labels_%d' % b] = d['pred_labels'].numpy()
    out['meta_json'] = np.array(json.dumps(dict(model=rh.to_plain(cfg.MODEL), pc_range=MINI_RANGE, voxel_size=[0.2, 0.2, 8.0],
                                                 class_names=list(cfg.CLASS_NAMES), layout='lately', state_shapes=shapes)))
    np.savez_compressed(os.path.join(HERE, 'g9_anchor_%s.npz' % tag), **out)
    print('g9', tag, 'anchors', out['batch_box_preds'].shape, 'final', [out['final_boxes_%d' % b].shape[0] for b in range(2)],
          'labels', [np.bincount(out['final_labels_%d' % b], minlength=4).tolist() for b in range(2)])


def anchor_train_gt_boxes():
    """(2, 10, 8) boxes of three classes for the anchor-head training fixture: cars near / on anchor centres with axis-aligned headings (IoU
    over the matched threshold), cars and small objects that only get their best anchor (forced positives), an all-zero row in the MIDDLE
    of frame 1 (class 0 indexes CLASS_NAMES[-1] in axis_aligned_target_assigner.py:62-66) and zero padding at the end."""
    s = synth.SEED_BASE + 1100
    step = 25.6 / 31
    gt = np.zeros((2, 10, 8), dtype=np.float32)
    for b in range(2):
        n = 10 - 3 * b
        gt[b, :n, 0] = synth.uniform(s, 10 * b + 1, n, -11.0, 11.0)
        gt[b, :n, 1] = synth.uniform(s, 10 * b + 2, n, -11.0, 11.0)
        gt[b, :n, 2] = synth.uniform(s, 10 * b + 3, n, -2.0, -0.5)
        gt[b, :n, 6] = synth.uniform(s, 10 * b + 7, n, -3.14159, 3.14159)
        cls = np.array([1, 1, 1, 1, 2, 2, 2, 3, 3, 1])[:n]
        gt[b, :n, 7] = cls
        base = {1: (3.9, 1.6, 1.56), 2: (0.8, 0.6, 1.73), 3: (1.76, 0.6, 1.73)}
        jit = synth.uniform(s, 10 * b + 4, n * 3, 0.85, 1.2).reshape(n, 3)
        for i in range(n):
            gt[b, i, 3:6] = np.array(base[int(cls[i])], dtype=np.float32) * jit[i]
    # cars on / near anchor centres, headings near 0 and pi / 2: matched by threshold
    gt[0, 0, [0, 1, 6]] = [-12.8 + 9 * step, -12.8 + 20 * step, 0.0]
    gt[0, 0, 3:6] = [3.9, 1.6, 1.56]
    gt[0, 1, [0, 1, 6]] = [-12.8 + 22 * step + 0.1, -12.8 + 7 * step - 0.08, 1.55]
    gt[0, 1, 3:6] = [4.1, 1.7, 1.5]
    gt[1, 0, [0, 1, 6]] = [-12.8 + 15 * step + 0.15, -12.8 + 15 * step + 0.1, 3.1]
    gt[1, 0, 3:6] = [4.2, 1.65, 1.6]
    gt[0, 5, [0, 1]] = [-12.8 + 4 * step, -12.8 + 27 * step]          # a pedestrian exactly on an anchor centre
    gt[0, 5, 3:6] = [0.8, 0.6, 1.73]
    gt[0, 5, 6] = 0.0
    gt[1, 2] = 0.0                                                        # the zero row in the middle
    return gt


def g11_anchor_train():
    """Training contract of MODEL.NAME PointPillar + AnchorHeadSingle (three anchor classes, direction classifier) on the mini geometry:
    AxisAlignedTargetAssigner targets, focal / smooth-L1(sin difference) / direction losses and two iterations of the reference's own train
    step (anchor_head_template.py:89-216, axis_aligned_target_assigner.py:37-210, tools/train_utils/train_utils.py:39-65)."""
    post = {'RECALL_THRESH_LIST': [0.3, 0.5, 0.7], 'SCORE_THRESH': 0.5, 'OUTPUT_RAW_SCORE': False, 'EVAL_METRIC': 'kitti',
            'NMS_CONFIG': {'MULTI_CLASSES_NMS': False, 'NMS_TYPE': 'nms_gpu', 'NMS_THRESH': 0.1, 'NMS_PRE_MAXSIZE': 1024,
                           'NMS_POST_MAXSIZE': 100}}
    ov = {'DATA_CONFIG.POINT_CLOUD_RANGE': MINI_RANGE, 'CLASS_NAMES': ['car', 'pedestrian', 'cyclist'], 'MODEL.NAME': 'PointPillar',
          'MODEL.DENSE_HEAD': rh.AttrDict(ANCHOR_HEAD), 'MODEL.POST_PROCESSING': rh.AttrDict(post)}
    cfg = rh.load_cfg('v2x_pointpillar_basic_ego.yaml', ov)
    model, ds = rh.build_model(cfg)
    shapes = fill_weights(model)
    sys.path.insert(0, os.path.join(rh.REF_ROOT, 'tools'))
    from train_utils.optimization import build_optimizer, build_scheduler
    from torch.nn.utils import clip_grad_norm_
    pts = synth.collate(mini_points('lately', 2, 2500))
    gt = anchor_train_gt_boxes()
    total_it_each_epoch, epochs = 5, cfg.OPTIMIZATION.NUM_EPOCHS
    optimizer = build_optimizer(model, cfg.OPTIMIZATION)
    lr_scheduler, _ = build_scheduler(optimizer, total_iters_each_epoch=total_it_each_epoch, total_epochs=epochs,
                                      last_epoch=-1, optim_cfg=cfg.OPTIMIZATION)
    out = {'points': pts, 'gt_boxes': gt, 'anchors': torch.cat(model.dense_head.anchors, dim=-3).numpy()}
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    out['trainable'] = np.array(names)
    for it in range(2):
        lr_scheduler.step(it)
        out['it%d_lr' % it] = np.array(float(optimizer.lr))
        out['it%d_mom' % it] = np.array(float(optimizer.mom))
        model.train()
        optimizer.zero_grad()
        bd = {'points': torch.from_numpy(pts.copy()), 'batch_size': 2, 'metadata': [{}, {}], 'gt_boxes': torch.from_numpy(gt.copy())}
        ret, tb, _disp = model(bd)
        loss = ret['loss']
        model.update_global_step()
        loss.backward()
        out['it%d_loss' % it] = np.array(float(loss))
        out['it%d_tb_json' % it] = np.array(json.dumps({k: float(v) for k, v in tb.items()}))
        params = dict(model.named_parameters())
        out['it%d_grad_digest' % it] = np.stack([_digest(params[n].grad) for n in names])
        if it == 0:
            fr = model.dense_head.forward_ret_dict
            out['box_cls_labels'] = fr['box_cls_labels'].numpy().copy()
            out['box_reg_targets'] = fr['box_reg_targets'].numpy().copy()
            out['reg_weights'] = fr['reg_weights'].numpy().copy()
            out['cls_preds'] = fr['cls_preds'].detach().numpy().copy()
            out['box_preds'] = fr['box_preds'].detach().numpy().copy()
            out['dir_cls_preds'] = fr['dir_cls_preds'].detach().numpy().copy()
            out['map_probe'] = bd['spatial_features_2d'].detach().numpy()[:, ::8].copy()
            for n in names:
                out['g0/' + n] = _sample(params[n].grad)
        norm = clip_grad_norm_(model.parameters(), cfg.OPTIMIZATION.GRAD_NORM_CLIP)
        out['it%d_grad_norm' % it] = np.array(float(norm))
        optimizer.step()
        if it == 0:
            for n in names:
                out['p1/' + n] = _sample(params[n])
            sd = model.state_dict()
            bn_keys = [k for k in sd if 'running_' in k]
            out['bn_keys'] = np.array(bn_keys)
            out['it0_bn_digest'] = np.stack([_digest(sd[k]) for k in bn_keys])
        print('g11 it', it, 'loss', float(loss), 'norm', float(norm), tb)
    lab = out['box_cls_labels']
    print('g11 labels', [np.bincount(lab[b] + 1, minlength=5).tolist() for b in range(2)])
    out['meta_json'] = np.array(json.dumps(dict(model=rh.to_plain(cfg.MODEL), optimization=rh.to_plain(cfg.OPTIMIZATION),
                                                 pc_range=MINI_RANGE, voxel_size=[0.2, 0.2, 8.0], class_names=list(cfg.CLASS_NAMES),
                                                 layout='lately', state_shapes=shapes,
                                                 num_point_features=int(ds.point_feature_encoder.num_point_features),
                                                 total_it_each_epoch=total_it_each_epoch)))
    path = os.path.join(HERE, 'g11_anchor_train.npz')
    np.savez_compressed(path, **out)
    print('g11 saved', os.path.getsize(path) // 1024, 'KiB')


def anchor_full_gt():
    s0 = synth.SEED_BASE + 970
    n = 24
    gt = np.zeros((1, n, 8), dtype=np.float32)
    gt[0, :, 0] = synth.uniform(s0, 1, n, -48.0, 48.0)
    gt[0, :, 1] = synth.uniform(s0, 2, n, -48.0, 48.0)
    gt[0, :, 2] = synth.uniform(s0, 3, n, -2.0, -0.5)
    cls = 1 + (np.arange(n) % 3)
    base = np.array([[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]], dtype=np.float32)
    gt[0, :, 3:6] = base[cls - 1] * synth.uniform(s0, 4, n * 3, 0.85, 1.2).reshape(n, 3)
    gt[0, :, 6] = synth.uniform(s0, 7, n, -3.14159, 3.14159)
    gt[0, :, 7] = cls
    return gt


def g11f_anchor_train_full():
    """PointPillar + AnchorHeadSingle training at full geometry (60 000 points of 13-column rows, 512 x 512 grid, 128 x 128 x 6 anchors, 24
    boxes of three classes): ONE iteration of the reference's own train step; labels digest, loss terms, gradient norm, per-tensor digests."""
    post = {'RECALL_THRESH_LIST': [0.3, 0.5, 0.7], 'SCORE_THRESH': 0.5, 'OUTPUT_RAW_SCORE': False, 'EVAL_METRIC': 'kitti',
            'NMS_CONFIG': {'MULTI_CLASSES_NMS': False, 'NMS_TYPE': 'nms_gpu', 'NMS_THRESH': 0.1, 'NMS_PRE_MAXSIZE': 1024,
                           'NMS_POST_MAXSIZE': 100}}
    ov = {'CLASS_NAMES': ['car', 'pedestrian', 'cyclist'], 'MODEL.NAME': 'PointPillar', 'MODEL.DENSE_HEAD': rh.AttrDict(ANCHOR_HEAD),
          'MODEL.POST_PROCESSING': rh.AttrDict(post)}
    cfg = rh.load_cfg('v2x_pointpillar_basic_ego.yaml', ov)
    model, ds = rh.build_model(cfg)
    shapes = fill_weights(model)
    sys.path.insert(0, os.path.join(rh.REF_ROOT, 'tools'))
    from train_utils.optimization import build_optimizer, build_scheduler
    from torch.nn.utils import clip_grad_norm_
    pts = synth.collate([synth.agent_cloud(agent=1, n_points=60000, layout='lately')])
    gt = anchor_full_gt()
    optimizer = build_optimizer(model, cfg.OPTIMIZATION)
    lr_scheduler, _ = build_scheduler(optimizer, total_iters_each_epoch=5, total_epochs=cfg.OPTIMIZATION.NUM_EPOCHS, last_epoch=-1,
                                      optim_cfg=cfg.OPTIMIZATION)
    names = [n_ for n_, p_ in model.named_parameters() if p_.requires_grad]
    lr_scheduler.step(0)
    model.train()
    optimizer.zero_grad()
    bd = {'points': torch.from_numpy(pts.copy()), 'batch_size': 1, 'metadata': [{}], 'gt_boxes': torch.from_numpy(gt.copy())}
    ret, tb, _disp = model(bd)
    loss = ret['loss']
    loss.backward()
    params = dict(model.named_parameters())
    fr = model.dense_head.forward_ret_dict
    lab = fr['box_cls_labels'].numpy()
    out = {'N': np.array(pts.shape[0]), 'gt_boxes': gt, 'trainable': np.array(names), 'loss': np.array(float(loss)),
           'tb_json': np.array(json.dumps({k: float(v) for k, v in tb.items()})),
           'grad_digest': np.stack([_digest(params[n_].grad) for n_ in names]),
           'labels_sha': np.array(sha(lab.astype(np.int32))), 'label_hist': np.bincount(lab.reshape(-1) + 1, minlength=5),
           'reg_targets_digest': _digest(fr['box_reg_targets'])}
    norm = clip_grad_norm_(model.parameters(), cfg.OPTIMIZATION.GRAD_NORM_CLIP)
    out['grad_norm'] = np.array(float(norm))
    out['meta_json'] = np.array(json.dumps(dict(model=rh.to_plain(cfg.MODEL), optimization=rh.to_plain(cfg.OPTIMIZATION),
                                                 pc_range=[float(v) for v in cfg.DATA_CONFIG.POINT_CLOUD_RANGE], voxel_size=[0.2, 0.2, 8.0],
                                                 class_names=list(cfg.CLASS_NAMES), layout='lately', state_shapes=shapes)))
    print('g11 full: loss', float(loss), 'norm', float(norm), 'labels', out['label_hist'], tb)
    np.savez_compressed(os.path.join(HERE, 'g11_anchor_train_full.npz'), **out)


G12_SEG_BIAS_SHIFT = 0.0


def hunter_train_inputs():
    """basic_car training batch on the mini geometry: 2 frames x 2500 background points plus foreground points of 6 / 5 instances
    (instance index in the last column, sweep index before it), `instances_tf` (B, N_inst_max, 11, 3, 4) = the rigid motion that takes a
    sweep-s point of the instance to the newest sweep (identity for static instances; |t(sweep 0)| > 0.5 marks a moving one,
    hunter_jr.py:222), and gt_boxes with one centre outside the range (dropped by remove_gt_boxes_outside_range)."""
    s = synth.SEED_BASE + 1200
    gt = train_gt_boxes(2, 6, 3)
    gt[1, 4:] = 0.0                                     # frame 1: four instances, two padding rows
    B, M, S = 2, 6, 11
    tf = np.zeros((B, M, S, 3, 4), dtype=np.float32)
    tf[..., :3, :3] = np.eye(3, dtype=np.float32)
    clouds = mini_points('car', 2, 2500)
    out_clouds = []
    for b in range(B):
        rows = [clouds[b]]
        for i in range(M):
            if gt[b, i, 7] == 0 or (b == 0 and i == 0):    # padding rows; the out-of-range box has no points
                continue
            c, dims, yaw = gt[b, i, 0:3].astype(np.float64), gt[b, i, 3:6].astype(np.float64), float(gt[b, i, 6])
            moving = (i % 2 == 1)
            speed = 2.0 + 1.5 * i
            omega = 0.15 * (i - 2)
            sweeps = [10, 9, 7, 4, 0][:(3 + (i + b) % 3)]
            for k, sw in enumerate(sweeps):
                dt = (10 - sw) * 0.1
                n = 6 + 3 * ((i + k + b) % 4)
                u = synth.uniform(s, 100 * b + 10 * i + k, n * 3, -0.5, 0.5).reshape(n, 3).astype(np.float64)
                local = u * dims
                if moving:
                    yaw_s = yaw - omega * dt
                    c_s = c - speed * dt * np.array([np.cos(yaw), np.sin(yaw), 0.0])
                else:
                    yaw_s, c_s = yaw, c
                Rs = np.array([[np.cos(yaw_s), -np.sin(yaw_s), 0], [np.sin(yaw_s), np.cos(yaw_s), 0], [0, 0, 1]])
                p = local @ Rs.T + c_s
                if moving:
                    d = omega * dt
                    R = np.array([[np.cos(d), -np.sin(d), 0], [np.sin(d), np.cos(d), 0], [0, 0, 1]])
                    tf[b, i, sw, :3, :3] = R
                    tf[b, i, sw, :3, 3] = c - R @ c_s
                r = np.zeros((n, 7), dtype=np.float32)
                r[:, 0:3] = p
                r[:, 3] = 0.5
                r[:, 4] = dt
                r[:, 5] = sw
                r[:, 6] = i
                rows.append(r)
        if moving is not None:
            pass
        out_clouds.append(np.concatenate(rows, axis=0))
    # the motion flag is read from sweep 0 (hunter_jr.py:222) whether or not the instance has points there
    for b in range(B):
        for i in range(M):
            if gt[b, i, 7] != 0 and i % 2 == 1 and not (b == 0 and i == 0):
                c, yaw = gt[b, i, 0:3].astype(np.float64), float(gt[b, i, 6])
                speed, omega, dt = 2.0 + 1.5 * i, 0.15 * (i - 2), 1.0
                c_s = c - speed * dt * np.array([np.cos(yaw), np.sin(yaw), 0.0])
                d = omega * dt
                R = np.array([[np.cos(d), -np.sin(d), 0], [np.sin(d), np.cos(d), 0], [0, 0, 1]])
                tf[b, i, 0, :3, :3] = R
                tf[b, i, 0, :3, 3] = c - R @ c_s
    return synth.collate(out_clouds), gt, tf


def g12_hunter_train():
    """Training contract of configs 1 / 2 (VFE -> scatter -> backbone -> HunterJr -> CenterHead) on the mini geometry: two iterations of the
    reference's own train step; HunterJr's meta / targets / predictions / seven loss terms of iteration 0 are stored one by one."""
    cfg = rh.load_cfg('v2x_pointpillar_basic_car.yaml', {'DATA_CONFIG.POINT_CLOUD_RANGE': MINI_RANGE})
    model, ds = rh.build_model(cfg)
    shapes = fill_weights(model)
    if G12_SEG_BIAS_SHIFT:
        with torch.no_grad():
            model.corrector.point_head.seg[0].bias[2] += G12_SEG_BIAS_SHIFT
    sys.path.insert(0, os.path.join(rh.REF_ROOT, 'tools'))
    from train_utils.optimization import build_optimizer, build_scheduler
    from torch.nn.utils import clip_grad_norm_
    pts, gt, tf = hunter_train_inputs()
    total_it_each_epoch, epochs = 5, cfg.OPTIMIZATION.NUM_EPOCHS
    optimizer = build_optimizer(model, cfg.OPTIMIZATION)
    lr_scheduler, _ = build_scheduler(optimizer, total_iters_each_epoch=total_it_each_epoch, total_epochs=epochs,
                                      last_epoch=-1, optim_cfg=cfg.OPTIMIZATION)
    out = {'points': pts, 'gt_boxes': gt, 'instances_tf': tf}
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    out['trainable'] = np.array(names)
    for it in range(2):
        lr_scheduler.step(it)
        out['it%d_lr' % it] = np.array(float(optimizer.lr))
        out['it%d_mom' % it] = np.array(float(optimizer.mom))
        model.train()
        optimizer.zero_grad()
        bd = {'points': torch.from_numpy(pts.copy()), 'batch_size': 2, 'metadata': [{}, {}], 'gt_boxes': torch.from_numpy(gt.copy()),
              'instances_tf': torch.from_numpy(tf.copy())}
        ret, tb, _disp = model(bd)
        loss = ret['loss']
        model.update_global_step()
        loss.backward()
        out['it%d_loss' % it] = np.array(float(loss))
        out['it%d_tb_json' % it] = np.array(json.dumps({k: float(v) for k, v in tb.items()}))
        params = dict(model.named_parameters())
        out['it%d_grad_digest' % it] = np.stack([_digest(params[n].grad) if params[n].grad is not None else np.zeros(3) for n in names])
        if it == 0:
            fr = model.corrector.forward_return_dict
            meta, pred, tgt = fr['meta'], fr['prediction'], fr['target']
            for k in ('locals2fg', 'inst2locals', 'indices_locals_max_sweep', 'indices_locals_min_sweep', 'locals_bis', 'instance_bi',
                      'mask_fg'):
                out['meta/' + k] = meta[k].numpy().copy()
            for k in ('points_cls_logit', 'points_flow3d', 'points_embedding', 'locals_tf'):
                out['pred/' + k] = pred[k].detach().numpy().copy()
            for k in ('locals_tf', 'points_cls', 'fg_embedding', 'fg_offset'):
                out['tgt/' + k] = tgt[k].numpy().copy()
            out['tgt/mask_locals_mos'] = tgt['meta']['mask_locals_mos'].numpy().copy()
            out['points_after'] = bd['points'].detach().numpy().copy()
            out['gt_boxes_after'] = bd['gt_boxes'].numpy().copy()
            out['map_probe'] = bd['spatial_features_2d'].detach().numpy()[:, ::8].copy()
            out['no_grad'] = np.array([n for n in names if params[n].grad is None])
            for n in names:
                if params[n].grad is not None:
                    out['g0/' + n] = _sample(params[n].grad)
        norm = clip_grad_norm_(model.parameters(), cfg.OPTIMIZATION.GRAD_NORM_CLIP)
        out['it%d_grad_norm' % it] = np.array(float(norm))
        optimizer.step()
        if it == 0:
            for n in names:
                out['p1/' + n] = _sample(params[n])
            sd = model.state_dict()
            bn_keys = [k for k in sd if 'running_' in k]
            out['bn_keys'] = np.array(bn_keys)
            out['it0_bn_digest'] = np.stack([_digest(sd[k]) for k in bn_keys])
        print('g12 it', it, 'loss', float(loss), 'norm', float(norm), tb)
    print('g12 fg', int(out['meta/mask_fg'].sum()), 'locals', out['meta/locals_bis'].shape[0], 'inst', out['meta/instance_bi'].shape[0],
          'dyn points moved', int((np.abs(out['points_after'] - pts).max(1) > 0).sum()), 'moving locals', int(out['tgt/mask_locals_mos'].sum()),
          'no grad', list(out['no_grad']))
    out['meta_json'] = np.array(json.dumps(dict(model=rh.to_plain(cfg.MODEL), optimization=rh.to_plain(cfg.OPTIMIZATION),
                                                 pc_range=MINI_RANGE, voxel_size=[0.2, 0.2, 8.0], class_names=list(cfg.CLASS_NAMES),
                                                 yaml='v2x_pointpillar_basic_car.yaml', layout='car', state_shapes=shapes,
                                                 seg_bias_shift=G12_SEG_BIAS_SHIFT, total_it_each_epoch=total_it_each_epoch)))
    path = os.path.join(HERE, 'g12_hunter_train.npz')
    np.savez_compressed(path, **out)
    print('g12 saved', os.path.getsize(path) // 1024, 'KiB')


def hunter_full_inputs():
    """one full-size basic_car training frame: 60 000 background points (BASELINE's cloud) + the foreground of 12 instances"""
    s0 = synth.SEED_BASE + 960
    n = 12
    gt = np.zeros((1, n, 8), dtype=np.float32)
    gt[0, :, 0] = synth.uniform(s0, 1, n, -48.0, 48.0)
    gt[0, :, 1] = synth.uniform(s0, 2, n, -48.0, 48.0)
    gt[0, :, 2] = synth.uniform(s0, 3, n, -3.0, -1.0)
    gt[0, :, 3] = synth.uniform(s0, 4, n, 3.0, 5.5)
    gt[0, :, 4] = synth.uniform(s0, 5, n, 1.5, 2.5)
    gt[0, :, 5] = synth.uniform(s0, 6, n, 1.4, 2.0)
    gt[0, :, 6] = synth.uniform(s0, 7, n, -3.14159, 3.14159)
    gt[0, :, 7] = 1.0
    fg, tf = synth.instance_foreground(77, gt[0], per_local=40)
    cloud = np.concatenate([synth.agent_cloud(agent=0, n_points=60000, layout='car'), fg], axis=0)
    return synth.collate([cloud]), gt, tf[None]


def g12f_hunter_train_full():
    """Configs 1 / 2 training at BASELINE's full size (60 000 points, 512 x 512 grid, 128 x 128 BEV map, one frame): ONE iteration of the
    reference's own train step; all loss terms, gradient norm, per-tensor gradient digests."""
    cfg = rh.load_cfg('v2x_pointpillar_basic_car.yaml', {})
    model, ds = rh.build_model(cfg)
    fill_weights(model)
    sys.path.insert(0, os.path.join(rh.REF_ROOT, 'tools'))
    from train_utils.optimization import build_optimizer, build_scheduler
    from torch.nn.utils import clip_grad_norm_
    pts, gt, tf = hunter_full_inputs()
    optimizer = build_optimizer(model, cfg.OPTIMIZATION)
    lr_scheduler, _ = build_scheduler(optimizer, total_iters_each_epoch=5, total_epochs=cfg.OPTIMIZATION.NUM_EPOCHS, last_epoch=-1,
                                      optim_cfg=cfg.OPTIMIZATION)
    names = [n_ for n_, p_ in model.named_parameters() if p_.requires_grad]
    lr_scheduler.step(0)
    model.train()
    optimizer.zero_grad()
    bd = {'points': torch.from_numpy(pts.copy()), 'batch_size': 1, 'metadata': [{}], 'gt_boxes': torch.from_numpy(gt.copy()),
          'instances_tf': torch.from_numpy(tf.copy())}
    ret, tb, _disp = model(bd)
    loss = ret['loss']
    loss.backward()
    params = dict(model.named_parameters())
    meta = model.corrector.forward_return_dict['meta']
    out = {'N': np.array(pts.shape[0]), 'trainable': np.array(names), 'loss': np.array(float(loss)),
           'tb_json': np.array(json.dumps({k: float(v) for k, v in tb.items()})),
           'grad_digest': np.stack([_digest(params[n_].grad) for n_ in names]),
           'counts': np.array([int(meta['mask_fg'].sum()), meta['locals_bis'].shape[0], meta['instance_bi'].shape[0]]),
           'moved_rows': np.array(int((np.abs(bd['points'].detach().numpy() - pts).max(1) > 0).sum())),
           'map_probe': bd['spatial_features_2d'].detach().numpy()[0, ::8, ::16, ::16].copy()}
    norm = clip_grad_norm_(model.parameters(), cfg.OPTIMIZATION.GRAD_NORM_CLIP)
    out['grad_norm'] = np.array(float(norm))
    out['optimization_json'] = np.array(json.dumps(rh.to_plain(cfg.OPTIMIZATION)))
    print('g12 full: loss', float(loss), 'norm', float(norm), 'counts', out['counts'], 'moved', int(out['moved_rows']), tb)
    np.savez_compressed(os.path.join(HERE, 'g12_hunter_train_full.npz'), **out)


G10_SEG_BIAS_SHIFT = 1.0


def g10_lately_chain():
    """BASELINE config 3 end to end on the mini geometry, produced by chaining the REFERENCE'S OWN pieces: for every (frame, remote agent)
    pair the reference's basic_car model with RETURN_MODAR_POINTS / RETURN_SCENE_FLOW (center_head.py:409-427, hunter_jr.py:377-397), the
    ego-side ingestion lines of v2x_sim_dataset_ego.py:196-232 (torch.unique + scatter(mean) through the shim, its own apply_se3_;
    points_in_boxes_gpu is CUDA-only: indices from oracle/exchange.py, stored as data), then the reference's basic_ego model on the
    augmented cloud.  2 frames x 5 remote agents, 2 000 points per cloud."""
    rh.install()
    sys.path.insert(0, REPO)
    from oracle import exchange as oex
    from pcdet.datasets.nuscenes.nuscenes_temporal_utils import apply_se3_
    from torch_scatter import scatter
    car_cfg = rh.load_cfg('v2x_pointpillar_basic_car.yaml', {'DATA_CONFIG.POINT_CLOUD_RANGE': MINI_RANGE})
    car_cfg.MODEL.DENSE_HEAD.RETURN_MODAR_POINTS = True
    car_cfg.MODEL.CORRECTOR.RETURN_SCENE_FLOW = True
    car, _ = rh.build_model(car_cfg)
    car_shapes = fill_weights(car)
    # with the synthetic fill no point is ever foreground (P(background) ~ 0.5 everywhere) and the flow propagation would go untested:
    # lower the background logit's bias (the test applies the same shift, recorded in the meta)
    with torch.no_grad():
        car.corrector.point_head.seg[0].bias[0] -= G10_SEG_BIAS_SHIFT
    ego_cfg = rh.load_cfg('v2x_pointpillar_basic_ego.yaml', {'DATA_CONFIG.POINT_CLOUD_RANGE': MINI_RANGE,
                                                             'MODEL.DENSE_HEAD.POST_PROCESSING.SCORE_THRESH': 0.02})
    ego, _ = rh.build_model(ego_cfg)
    ego_shapes = fill_weights(ego)
    B, remote_agents, n_pts = 2, (0, 2, 3, 4, 5), 2000
    out = {}
    ego_rows = []
    for f in range(B):
        ego_cloud = synth.agent_cloud(agent=200 + 10 * f + 1, n_points=n_pts, layout='car', xy_half=13.1)
        out['ego_cloud_%d' % f] = ego_cloud
        max_sweep_idx = float(ego_cloud[:, -2].max())
        out['max_sweep_idx_%d' % f] = np.array(max_sweep_idx)
        pts13 = np.zeros((ego_cloud.shape[0], 13))
        pts13[:, :5] = ego_cloud[:, :5]
        pts13[:, -2:] = ego_cloud[:, -2:]
        for slot, a in enumerate(remote_agents):
            cloud = synth.agent_cloud(agent=200 + 10 * f + a, n_points=n_pts, layout='car', xy_half=13.1)
            pose = np.linalg.inv(synth.agent_pose(a))                       # remote lidar -> ego frame
            pose[:3, 3] *= 0.25                                             # keep the mapped boxes inside the mini range
            key = '%d_%d' % (f, slot)
            out['remote_cloud_' + key] = cloud
            out['target_se3_lidar_' + key] = pose
            bd = {'points': torch.from_numpy(synth.collate([cloud]).copy()), 'batch_size': 1,
                  'metadata': [{'sample_token': 'f%d' % f, 'lidar_id': a}]}
            with torch.no_grad():
                for mod in car.module_list:
                    bd = mod(bd)
            modar = bd['mo_pts'].numpy().copy() if 'mo_pts' in bd else np.zeros((0, 9), np.float32)
            fg = bd['scene_flow'].numpy().copy() if 'scene_flow' in bd else np.zeros((0, 13), np.float32)
            out['modar_' + key], out['foreground_' + key] = modar, fg
            # ---- the reference's ingestion lines (v2x_sim_dataset_ego.py:203-232) ----
            modar_t = torch.from_numpy(modar.copy())
            if fg.shape[0] > 0 and modar.shape[0] > 0:
                foregr = torch.from_numpy(fg.copy())
                box_idx = torch.from_numpy(oex.points_in_boxes(fg[:, :3], modar[:, :7])).long()
                out['box_idx_' + key] = box_idx.numpy().astype(np.int32)
                mask_valid = box_idx > -1
                foregr, bi = foregr[mask_valid], box_idx[mask_valid]
                if bi.numel() > 0:
                    unq_box_idx, inv = torch.unique(bi, return_inverse=True)
                    boxes_offset = scatter(foregr[:, -3:], inv, dim=0, reduce='mean') * 2.
                    modar_t[unq_box_idx, :3] += boxes_offset
            m = modar_t.numpy()
            if m.shape[0] > 0:
                m[:, :7] = apply_se3_(pose, boxes_=m[:, :7], return_transformed=True)
            modar_ = np.zeros((m.shape[0], 13))
            modar_[:, :3] = m[:, :3]
            modar_[:, 4] = 0.
            modar_[:, 5:11] = m[:, 3:]
            modar_[:, -2] = max_sweep_idx
            modar_[:, -1] = -1
            out['ingest_rows_' + key] = modar_.astype(np.float32)
            pts13 = np.concatenate((pts13, modar_))
        ego_rows.append(pts13.astype(np.float32))
        print('g10 frame', f, 'modar', [out['modar_%d_%d' % (f, s)].shape[0] for s in range(5)], 'foreground',
              [out['foreground_%d_%d' % (f, s)].shape[0] for s in range(5)])
    ego_pts = synth.collate(ego_rows)
    out['ego_points'] = ego_pts
    bd = {'points': torch.from_numpy(ego_pts.copy()), 'batch_size': B, 'metadata': [{} for _ in range(B)]}
    with torch.no_grad(), PostCapture(ego) as post:
        bd, _ = run_modules(ego, bd)
    capture_common(ego, bd, out)
    post.dump(out)
    out['meta_json'] = np.array(json.dumps(dict(
        car=dict(model=rh.to_plain(car_cfg.MODEL), pc_range=MINI_RANGE, voxel_size=[0.2, 0.2, 8.0], class_names=list(car_cfg.CLASS_NAMES),
                 yaml='v2x_pointpillar_basic_car.yaml', layout='car', state_shapes=car_shapes),
        ego=dict(model=rh.to_plain(ego_cfg.MODEL), pc_range=MINI_RANGE, voxel_size=[0.2, 0.2, 8.0], class_names=list(ego_cfg.CLASS_NAMES),
                 yaml='v2x_pointpillar_basic_ego.yaml', layout='lately', state_shapes=ego_shapes),
        frames=B, remote_agents=list(remote_agents), car_seg_bias_shift=G10_SEG_BIAS_SHIFT)))
    np.savez_compressed(os.path.join(HERE, 'g10_lately_chain.npz'), **out)
    print('g10 ego P', out['voxel_coords'].shape[0], 'final', [out['final_boxes_%d' % b].shape[0] for b in range(B)])


# ---------------------------------------------------------------------------------------------------------------------
# g13: WELL-CONDITIONED end-to-end fixtures ("what's weak" 1).  With the SURVEY 8(d) weight bound every candidate of a frame
# scores within 5e-4 of sigmoid(-2.19) and the final box SET is a function of float noise; these fixtures use He-scaled weights (an O(1)
# spatial signal survives the conv stack) and a SCORE_THRESH picked so that the reference's final set is INVARIANT under random
# perturbations of all five head maps 10x larger than any fp32 implementation's noise (certificate: `trials` perturbed runs of the
# reference's own generate_predicted_boxes give the same set).  The GPU tests then demand the exact final set (count + one-to-one match at
# 1e-3) through the whole HIP path, at mini size, at BASELINE's full sizes and for the config-3 chain.
# ---------------------------------------------------------------------------------------------------------------------
G13_NOISE = 1e-4
G13_TRIALS = 12
G13_COUNTS = [120, 100, 150, 80, 180, 60, 200, 240, 40, 300]      # candidates above SCORE_THRESH in frame 0 (tried in this order)
G13_GAINS = [1.6, 1.7, 1.8, 1.9, 2.0, 2.1, 2.2, 2.3, 2.4, 2.5, 2.6]   # conv / linear weights U(-k, k), k = gain / sqrt(fan_in).  The gain at which the signal
# neither dies nor explodes over the ~20 layers depends on the model and on the map size (zero padding damps small maps), so it is tuned per
# case: the smallest gain whose hm logits have a standard deviation in [0.15, 0.9] with box-size logits inside +-3.5 (metric boxes)


def _same_set(a_boxes, a_scores, b_boxes, b_scores, tol):
    """same detections (count + one-to-one match of centre, size and score within tol).  The heading is matched loosely (0.2 rad): it is
    atan2 of two head values that may both be small, so under the certificate's 1e-4 perturbation it moves by 1e-4 / |(cos, sin)| without
    any change of WHICH boxes survive -- the question this certificate answers"""
    if a_boxes.shape[0] != b_boxes.shape[0]:
        return False
    used = np.zeros(b_boxes.shape[0], bool)
    for i in range(a_boxes.shape[0]):
        d = np.abs(b_boxes - a_boxes[i])
        d[:, 6] = np.minimum(d[:, 6], np.abs(d[:, 6] - 2 * np.pi)) * (tol / 0.2)
        e = np.maximum(d.max(1), np.abs(b_scores - a_scores[i])) + used * 1e9
        j = int(np.argmin(e))
        if e[j] > tol:
            return False
        used[j] = True
    return True


def robust_threshold(head, batch_size, pred_dicts, what, more=()):
    """picks SCORE_THRESH for `head` (the reference's CenterHead, its raw head maps in pred_dicts) such that the final sets of all frames
    are invariant under G13_TRIALS random perturbations (uniform +-G13_NOISE on every value of the five maps); returns (thr, finals).
    more: further (batch_size, pred_dicts) evaluations of the same head that must be robust under the same threshold (the ten remote
    passes of the config-3 chain); finals then is the list over all evaluations."""
    cfg = head.model_cfg.POST_PROCESSING
    keep = cfg.SCORE_THRESH
    gen = torch.Generator().manual_seed(1234)
    K = int(cfg.MAX_OBJ_PER_SAMPLE)
    evals = [(batch_size, pred_dicts)] + list(more)
    tops = [torch.topk(pd[0]['hm'][b].sigmoid().reshape(-1), K)[0].double().numpy() for bs, pd in evals for b in range(bs)]
    # candidate thresholds: the middles of the WIDEST score gaps (over all frames together) among those that leave 30 .. 300 candidates in
    # every frame -- widest first, so the SCORE_THRESH cut has the largest margin the data offers
    lo = max(t[min(300, K - 1)] for t in tops)
    hi = min(t[30] for t in tops)
    union = np.sort(np.concatenate([t[(t >= lo) & (t <= hi)] for t in tops]))
    gaps = union[1:] - union[:-1]
    order = np.argsort(-gaps)[:24]
    try:
        for gi in order:
            thr = round(float(union[gi] + union[gi + 1]) / 2.0, 7)
            c = int((tops[0] > thr).sum())
            if gaps[gi] < 5e-5:
                break
            cfg.SCORE_THRESH = thr
            bases = []
            with torch.no_grad():
                for bs, pd in evals:
                    bases.append(head.generate_predicted_boxes(bs, pd))
            counts = [int(d['pred_boxes'].shape[0]) for base in bases for d in base]
            if min(counts) < 8:
                continue
            ok = True
            for _t in range(G13_TRIALS):
                for (bs, pd), base in zip(evals, bases):
                    noisy = [{k: v + (torch.rand(v.shape, generator=gen) * 2 - 1) * G13_NOISE for k, v in d.items()} for d in pd]
                    with torch.no_grad():
                        got = head.generate_predicted_boxes(bs, noisy)
                    for a, b in zip(base, got):
                        if not _same_set(a['pred_boxes'].numpy(), a['pred_scores'].numpy(), b['pred_boxes'].numpy(), b['pred_scores'].numpy(), 1e-3):
                            ok = False
                            break
                    if not ok:
                        break
                if not ok:
                    break
            print('   g13 %-12s thr %.6f (%d candidates in frame 0) finals %s -> %s' % (what, thr, c, counts, 'robust' if ok else 'order-sensitive'))
            if ok:
                return thr, (bases[0] if not more else bases)
    finally:
        cfg.SCORE_THRESH = keep
    raise RuntimeError('g13 %s: no threshold gives a perturbation-invariant final set' % what)


def _g13_store(out, tag, finals, thr):
    out[tag + '_score_thresh'] = np.array(float(thr))
    out[tag + '_frames'] = np.array(len(finals))
    for b, d in enumerate(finals):
        out['%s_boxes_%d' % (tag, b)] = d['pred_boxes'].numpy().copy()
        out['%s_scores_%d' % (tag, b)] = d['pred_scores'].numpy().copy()
        out['%s_labels_%d' % (tag, b)] = d['pred_labels'].numpy().copy()


def _g13_meta(cfg, yaml_name, layout, shapes, pc_range, extra=None):
    m = dict(model=rh.to_plain(cfg.MODEL), pc_range=pc_range, voxel_size=[0.2, 0.2, 8.0], class_names=list(cfg.CLASS_NAMES), yaml=yaml_name,
             layout=layout, state_shapes=shapes, weight_scheme=WEIGHT_SCHEME)
    m.update(extra or {})
    return m


def _g13_run(build, make_bd, tag, out):
    """build() -> (cfg, model) with WEIGHT_SCHEME applied; make_bd() -> a fresh batch dict.  Tunes the weight gain, runs the reference
    forward, picks the robust threshold on its own head maps and stores the final sets under that threshold."""
    global WEIGHT_SCHEME
    chosen = None
    for gain in G13_GAINS:
        WEIGHT_SCHEME = 'gain:%g' % gain
        cfg, model, shapes = build()
        bd = make_bd()
        with torch.no_grad():
            for mod in model.module_list:
                bd = mod(bd)
        pd0 = model.dense_head.forward_ret_dict['pred_dicts'][0]
        sd, dmax = float(pd0['hm'].std()), float(pd0['dim'].abs().max())
        print('   g13 %-12s gain %.1f: hm logits %.2f .. %.2f (std %.3f), |dim logits| <= %.2f' % (tag, gain, float(pd0['hm'].min()), float(pd0['hm'].max()), sd, dmax))
        if 0.15 <= sd <= 0.9 and dmax <= 3.5:
            chosen = gain
            break
    if chosen is None:
        raise RuntimeError('g13 %s: no gain gives a usable head map' % tag)
    head = model.dense_head
    pred_dicts = [{k: v.detach().clone() for k, v in pd.items()} for pd in head.forward_ret_dict['pred_dicts']]
    thr, finals = robust_threshold(head, bd['batch_size'], pred_dicts, tag)
    _g13_store(out, tag, finals, thr)
    out[tag + '_weight_scheme'] = np.array(WEIGHT_SCHEME)
    out[tag + '_hm_range'] = np.array([float(pred_dicts[0]['hm'].min()), float(pred_dicts[0]['hm'].max())])
    return thr, cfg, shapes


def g13_conditioned():
    global WEIGHT_SCHEME
    out, metas = {}, {}
    tmp = tempfile.mkdtemp()
    empty = os.path.join(tmp, 'empty.pth')
    torch.save({'model_state': {}}, empty)

    def builder(yaml_name, ov):
        def build():
            cfg = rh.load_cfg(yaml_name, ov)
            model, _ds = rh.build_model(cfg)
            shapes = fill_weights(model)
            return cfg, model, shapes
        return build
    try:
        # ---- mini geometry: the three single-model configs (car incl. HunterJr), 2 frames x 3 000 points -----------------------------
        for tag, yaml_name, layout in (('car', 'v2x_pointpillar_basic_car.yaml', 'car'), ('ego', 'v2x_pointpillar_basic_ego.yaml', 'lately'),
                                       ('early', 'v2x_pointpillar_basic_ego_early.yaml', 'early')):
            pts = synth.collate(mini_points(layout, 2, 3000))
            thr, cfg, shapes = _g13_run(builder(yaml_name, {'DATA_CONFIG.POINT_CLOUD_RANGE': MINI_RANGE}),
                                        lambda: {'points': torch.from_numpy(pts.copy()), 'batch_size': 2, 'metadata': [{}, {}]}, tag, out)
            cfg.MODEL.DENSE_HEAD.POST_PROCESSING.SCORE_THRESH = thr
            metas[tag] = _g13_meta(cfg, yaml_name, layout, shapes, MINI_RANGE)
        # ---- mini DiscoNet (the g1_disco scene: 3 agents, agent 2 absent from frame 1, half-pixel poses) ---------------------------------
        ov = {'DATA_CONFIG.POINT_CLOUD_RANGE': MINI_RANGE, 'MODEL.BEV_MAKER_RSU.CKPT': empty, 'MODEL.BEV_MAKER_CAR.CKPT': empty,
              'MODEL.BEV_MAKER_EARLY.CKPT': empty, 'MODEL.V2X_MID_FUSION.PC_RANGE_MIN': MINI_RANGE[0]}
        poses = {0: synth.agent_pose(0), 2: synth.agent_pose(2)}
        poses[0][:3, 3] = [0.8, -0.4, 0.0]
        poses[0][:3, :3] = synth.agent_pose(1)[:3, :3]
        poses[2][:3, 3] = [-1.6, 2.4, 0.0]
        metadata = [{'se3_from_ego': {0: poses[0], 2: poses[2]}}, {'se3_from_ego': {0: poses[0]}}]
        clouds = []
        for b in range(2):
            per_agent = []
            for a in (0, 1, 2):
                if b == 1 and a == 2:
                    continue
                c = synth.agent_cloud(agent=20 + 3 * b + a, n_points=1500, layout='disco', xy_half=13.1)
                c[:, -1] = float(a)
                per_agent.append(c)
            clouds.append(np.concatenate(per_agent, axis=0))
        dpts = synth.collate(clouds)
        thr, cfg, shapes = _g13_run(builder('v2x_pointpillar_disco.yaml', ov),
                                    lambda: {'points': torch.from_numpy(dpts.copy()), 'batch_size': 2, 'metadata': metadata}, 'disco', out)
        cfg.MODEL.DENSE_HEAD.POST_PROCESSING.SCORE_THRESH = thr
        metas['disco'] = _g13_meta(cfg, 'v2x_pointpillar_disco.yaml', 'disco', shapes, MINI_RANGE, dict(absent=[[], [2]]))
        out['disco_pose_0'], out['disco_pose_2'] = poses[0], poses[2]
        # ---- BASELINE's full sizes: 60 000 points per agent, 512 x 512 grid, one frame -----------------------------------------------------
        for tag, yaml_name, layout, n_agents in (('car_full', 'v2x_pointpillar_basic_car.yaml', 'car', 1),
                                                 ('ego_full', 'v2x_pointpillar_basic_ego.yaml', 'lately', 1),
                                                 ('early_full', 'v2x_pointpillar_basic_ego_early.yaml', 'early', 6)):
            cloud = np.concatenate([synth.agent_cloud(agent=a, n_points=60000, layout=layout) for a in range(n_agents)], axis=0)
            fpts = synth.collate([cloud])
            _g13_run(builder(yaml_name, {}), lambda: {'points': torch.from_numpy(fpts.copy()), 'batch_size': 1, 'metadata': [{}]}, tag, out)
        agents = (0, 1, 2, 3, 4, 5)
        fposes = {a: synth.agent_pose(a) for a in agents if a != 1}
        fclouds = []
        for a in agents:
            c = synth.agent_cloud(agent=a, n_points=60000, layout='disco')
            c[:, -1] = float(a)
            fclouds.append(c)
        dfpts = synth.collate([np.concatenate(fclouds, axis=0)])
        _g13_run(builder('v2x_pointpillar_disco.yaml', {'MODEL.BEV_MAKER_RSU.CKPT': empty, 'MODEL.BEV_MAKER_CAR.CKPT': empty,
                                                        'MODEL.BEV_MAKER_EARLY.CKPT': empty}),
                 lambda: {'points': torch.from_numpy(dfpts.copy()), 'batch_size': 1, 'metadata': [{'se3_from_ego': fposes}]}, 'disco_full', out)
        for a in fposes:
            out['disco_full_pose_%d' % a] = fposes[a]
        out['meta_json'] = np.array(json.dumps(dict(cases=metas, noise=G13_NOISE, trials=G13_TRIALS)))
        np.savez_compressed(os.path.join(HERE, 'g13_conditioned.npz'), **out)
        print('g13:', {k: (str(out[k]) if out[k].ndim == 0 else out[k].shape) for k in sorted(out) if k.endswith('_score_thresh') or k.endswith('_scheme') or '_boxes_' in k})
    finally:
        WEIGHT_SCHEME = 'survey'


def g13_chain(full=False):
    """BASELINE config 3 as a chain (the g10 scene: 2 frames x 5 remote agents x 2 000 points, mini geometry) on WELL-CONDITIONED weights:
    the remote detector's gain and SCORE_THRESH are tuned so that the final sets of all ten remote passes are perturbation-invariant, the
    ego detector's on the augmented cloud the reference's own ingestion lines build.  tests/golden/g13_chain.npz holds the clouds, the
    MoDAR rows of every pass and the final detections of the ego pass, which the GPU test demands EXACTLY (1e-3) of the whole device-side
    chain (pcdet/models/lately_chain.py).
    full=True: the same at BASELINE's FULL size -- one frame, 6 agents x 60 000 points, the YAMLs' own 102.4 m range --
    into tests/golden/g13_chain_full.npz.  The clouds are synth.agent_cloud streams (regenerated by the test from the recorded agent ids,
    not stored); the foreground rows are stored as a count + digest."""
    global WEIGHT_SCHEME
    rh.install()
    sys.path.insert(0, REPO)
    from oracle import exchange as oex
    from pcdet.datasets.nuscenes.nuscenes_temporal_utils import apply_se3_
    from torch_scatter import scatter
    # full='b4' (round 6): the batch bench.py --config lately6 times -- 4 frames, agent streams 10 f + a (rank 0), as bench.main builds them
    B, remote_agents, n_pts = ((4 if full == 'b4' else 1), (0, 2, 3, 4, 5), 60000) if full else (2, (0, 2, 3, 4, 5), 2000)
    ckw = {} if full else {'xy_half': 13.1}
    ov = {} if full else {'DATA_CONFIG.POINT_CLOUD_RANGE': MINI_RANGE}
    base_agent = (0 if full == 'b4' else 700) if full else 200
    out = {}
    try:
        clouds = {}
        for f in range(B):
            clouds[(f, 'ego')] = synth.agent_cloud(agent=base_agent + 10 * f + 1, n_points=n_pts, layout='car', **ckw)
            for slot, a in enumerate(remote_agents):
                clouds[(f, slot)] = synth.agent_cloud(agent=base_agent + 10 * f + a, n_points=n_pts, layout='car', **ckw)

        hunter_log = {}

        def car_passes(car):
            res = {}
            cap = {}
            hook = car.corrector.point_head.register_forward_hook(lambda m, a, o: cap.update(cls=o[1].detach().clone(), flow=o[2].detach().clone()))
            for f in range(B):
                for slot, a in enumerate(remote_agents):
                    bd = {'points': torch.from_numpy(synth.collate([clouds[(f, slot)]]).copy()), 'batch_size': 1,
                          'metadata': [{'sample_token': 'f%d' % f, 'lidar_id': a}]}
                    with torch.no_grad():
                        for mod in car.module_list:
                            bd = mod(bd)
                    res[(f, slot)] = (bd, [{k: v.detach().clone() for k, v in pd.items()} for pd in car.dense_head.forward_ret_dict['pred_dicts']])
                    hunter_log[(f, slot)] = (dict(cap), bd['points'].numpy().copy())
            hook.remove()
            return res

        # ---- remote detector: gain, then one SCORE_THRESH robust for all ten passes -----------------------------------------------------
        car = car_cfg = car_shapes = None
        for gain in G13_GAINS:
            WEIGHT_SCHEME = 'gain:%g' % gain
            car_cfg = rh.load_cfg('v2x_pointpillar_basic_car.yaml', dict(ov))
            car_cfg.MODEL.DENSE_HEAD.RETURN_MODAR_POINTS = True
            car_cfg.MODEL.CORRECTOR.RETURN_SCENE_FLOW = True
            car, _ = rh.build_model(car_cfg)
            car_shapes = fill_weights(car)
            with torch.no_grad():
                car.corrector.point_head.seg[0].bias[0] -= G10_SEG_BIAS_SHIFT
            res = car_passes(car)
            hm = torch.cat([pd[0]['hm'].reshape(-1) for _bd, pd in res.values()])
            dmax = max(float(pd[0]['dim'].abs().max()) for _bd, pd in res.values())
            print('   g13 chain car    gain %.1f: hm std %.3f, |dim logits| <= %.2f' % (gain, float(hm.std()), dmax))
            if 0.15 <= float(hm.std()) <= 0.9 and dmax <= 3.5:
                break
        else:
            raise RuntimeError('g13 chain: no gain for the remote detector')
        out['car_weight_scheme'] = np.array(WEIGHT_SCHEME)
        dyn_shift = 0.0
        if full == 'b4':
            # twenty passes x 60 000 rows: with every row flow-corrected (what the tuned gain gives) a few corrected points always land within
            # 1e-5 of a BEV pixel boundary and bev_scatter's .long() becomes an fp32 coin toss (see _g17_case).  Shift the dynamic-foreground
            # bias until ~0.1 % of the rows are corrected and none of them, in any pass, is near the verdict or a pixel boundary.
            pooled = torch.cat([torch.minimum(lg['cls'][:, 2] - torch.maximum(lg['cls'][:, 0], lg['cls'][:, 1]), lg['cls'][:, 2] - float(np.log(0.3 / 0.7)))
                                for lg, _after in hunter_log.values()])
            base_shift = float(torch.sort(pooled, descending=True)[0][max(1, pooled.shape[0] // 1000)])
            orig2 = float(car.corrector.point_head.seg[0].bias[2])
            for attempt in range(60):
                dyn_shift = round(base_shift + 0.011 * attempt, 4)
                with torch.no_grad():
                    car.corrector.point_head.seg[0].bias[2] = orig2 - dyn_shift
                res = car_passes(car)
                tot = near_n = edge_n = 0
                for lg, after in hunter_log.values():
                    mask, near, edge = _hunter_verdicts(car, lg, after)
                    tot, near_n, edge_n = tot + int(mask.sum()), near_n + int(near.sum()), edge_n + int(edge.sum())
                print('   g13 chain car    dyn bias shift %.4f: %d rows corrected over %d passes, %d near the verdict, %d near a BEV pixel boundary'
                      % (dyn_shift, tot, len(hunter_log), near_n, edge_n))
                if tot >= 10 * len(hunter_log) and near_n == 0 and edge_n == 0:
                    break
            else:
                raise RuntimeError('g13 chain b4: no dyn bias shift gives a well-conditioned HunterJr correction')
        out['car_seg_dyn_shift'] = np.array(dyn_shift, dtype=np.float64)
        keys = sorted(res.keys())
        thr_car, _ = robust_threshold(car.dense_head, 1, res[keys[0]][1], 'chain car', more=[(1, res[k][1]) for k in keys[1:]])
        car_cfg.MODEL.DENSE_HEAD.POST_PROCESSING.SCORE_THRESH = thr_car
        car.dense_head.model_cfg.POST_PROCESSING.SCORE_THRESH = thr_car
        out['car_score_thresh'] = np.array(float(thr_car))
        res = car_passes(car)                           # again, now with the threshold the MoDAR rows are made under
        # ---- the reference's ingestion lines (v2x_sim_dataset_ego.py:203-232), as in g10 -------------------------------------------------
        ego_rows = []
        for f in range(B):
            ego_cloud = clouds[(f, 'ego')]
            if not full:
                out['ego_cloud_%d' % f] = ego_cloud
            max_sweep_idx = float(ego_cloud[:, -2].max())
            out['max_sweep_idx_%d' % f] = np.array(max_sweep_idx)
            pts13 = np.zeros((ego_cloud.shape[0], 13))
            pts13[:, :5] = ego_cloud[:, :5]
            pts13[:, -2:] = ego_cloud[:, -2:]
            for slot, a in enumerate(remote_agents):
                pose = np.linalg.inv(synth.agent_pose(a))
                if not full:
                    pose[:3, 3] *= 0.25
                key = '%d_%d' % (f, slot)
                if not full:
                    out['remote_cloud_' + key] = clouds[(f, slot)]
                out['target_se3_lidar_' + key] = pose
                bd = res[(f, slot)][0]
                modar = bd['mo_pts'].numpy().copy() if 'mo_pts' in bd else np.zeros((0, 9), np.float32)
                fg = bd['scene_flow'].numpy().copy() if 'scene_flow' in bd else np.zeros((0, 13), np.float32)
                out['modar_' + key] = modar
                if full:
                    out['foreground_digest_' + key] = np.array([fg.shape[0], float(np.abs(fg.astype(np.float64)).sum())])
                else:
                    out['foreground_' + key] = fg
                modar_t = torch.from_numpy(modar.copy())
                if fg.shape[0] > 0 and modar.shape[0] > 0:
                    foregr = torch.from_numpy(fg.copy())
                    box_idx = torch.from_numpy(oex.points_in_boxes(fg[:, :3], modar[:, :7])).long()
                    mask_valid = box_idx > -1
                    foregr, bi = foregr[mask_valid], box_idx[mask_valid]
                    if bi.numel() > 0:
                        unq_box_idx, inv = torch.unique(bi, return_inverse=True)
                        modar_t[unq_box_idx, :3] += scatter(foregr[:, -3:], inv, dim=0, reduce='mean') * 2.
                m = modar_t.numpy()
                if m.shape[0] > 0:
                    m[:, :7] = apply_se3_(pose, boxes_=m[:, :7], return_transformed=True)
                modar_ = np.zeros((m.shape[0], 13))
                modar_[:, :3] = m[:, :3]
                modar_[:, 5:11] = m[:, 3:]
                modar_[:, -2] = max_sweep_idx
                modar_[:, -1] = -1
                pts13 = np.concatenate((pts13, modar_))
            ego_rows.append(pts13.astype(np.float32))
            print('   g13 chain frame', f, 'modar', [out['modar_%d_%d' % (f, sl)].shape[0] for sl in range(5)], 'foreground',
                  [int(out['foreground_digest_%d_%d' % (f, sl)][0]) if full else out['foreground_%d_%d' % (f, sl)].shape[0] for sl in range(5)])
        ego_pts = synth.collate(ego_rows)
        if not full:
            out['ego_points'] = ego_pts
        out['ego_points_rows'] = np.array(ego_pts.shape[0])
        # ---- ego detector on the augmented cloud ---------------------------------------------------------------------------------------------
        def build_ego():
            cfg = rh.load_cfg('v2x_pointpillar_basic_ego.yaml', dict(ov))
            model, _ds = rh.build_model(cfg)
            shapes = fill_weights(model)
            return cfg, model, shapes
        thr_ego, ego_cfg, ego_shapes = _g13_run(build_ego, lambda: {'points': torch.from_numpy(ego_pts.copy()), 'batch_size': B,
                                                                    'metadata': [{} for _ in range(B)]}, 'ego', out)
        ego_cfg.MODEL.DENSE_HEAD.POST_PROCESSING.SCORE_THRESH = thr_ego
        out['meta_json'] = np.array(json.dumps(dict(
            car=dict(model=rh.to_plain(car_cfg.MODEL), pc_range=[float(v) for v in car_cfg.DATA_CONFIG.POINT_CLOUD_RANGE], voxel_size=[0.2, 0.2, 8.0], class_names=list(car_cfg.CLASS_NAMES),
                     yaml='v2x_pointpillar_basic_car.yaml', layout='car', state_shapes=car_shapes),
            ego=dict(model=rh.to_plain(ego_cfg.MODEL), pc_range=[float(v) for v in ego_cfg.DATA_CONFIG.POINT_CLOUD_RANGE], voxel_size=[0.2, 0.2, 8.0], class_names=list(ego_cfg.CLASS_NAMES),
                     yaml='v2x_pointpillar_basic_ego.yaml', layout='lately', state_shapes=ego_shapes),
            frames=B, remote_agents=list(remote_agents), base_agent=base_agent, n_points=n_pts, full=bool(full), car_seg_bias_shift=G10_SEG_BIAS_SHIFT, noise=G13_NOISE, trials=G13_TRIALS)))
        np.savez_compressed(os.path.join(HERE, ('g13_chain_full_b4.npz' if full == 'b4' else 'g13_chain_full.npz') if full else 'g13_chain.npz'), **out)
        print('g13 chain: car', str(out['car_weight_scheme']), float(out['car_score_thresh']), 'ego', str(out['ego_weight_scheme']),
              float(out['ego_score_thresh']), 'final', [out['ego_boxes_%d' % b].shape[0] for b in range(B)])
    finally:
        WEIGHT_SCHEME = 'survey'


def g14_late_fusion():
    """V2XLateFusion (reference pcdet/models/detectors/v2x_late_fusion.py:13-54) driven with synthetic exchange boxes: the reference's own
    forward -- concatenation in dict order, SCORE_THRESH, class-agnostic rotated NMS (thr 0.3, pre 4096, post 500) -- and its 'ego_only'
    branch.  Scene: true objects seen by several agents with jittered boxes (NMS must merge them), some scores under the threshold, one
    agent without boxes, near-duplicate scores.  tests/golden/g14_late_fusion.npz holds the inputs and the reference's pred_dicts."""
    rh.install()
    cfg = rh.load_cfg('v2x_late_fusion.yaml')
    model, _ = rh.build_model(cfg)
    rng = np.random.RandomState(1404)
    out = {}
    frames = []
    for f, (n_obj, agents) in enumerate([(60, (0, 1, 2, 3, 4, 5)), (25, (1, 3)), (140, (0, 1, 2, 4))]):
        centers = np.concatenate([rng.uniform(-48, 48, (n_obj, 2)), rng.uniform(-2.5, -0.5, (n_obj, 1))], 1)
        dims = np.stack([rng.uniform(3.5, 5.2, n_obj), rng.uniform(1.6, 2.2, n_obj), rng.uniform(1.4, 1.9, n_obj)], 1)
        yaw = rng.uniform(-np.pi, np.pi, (n_obj, 1))
        ex = {}
        for a in agents:
            seen = rng.rand(n_obj) < 0.7
            k = int(seen.sum())
            b = np.concatenate([centers[seen] + rng.normal(0, 0.25, (k, 3)) * [1, 1, 0.2], dims[seen] * rng.uniform(0.93, 1.07, (k, 3)),
                                yaw[seen] + rng.normal(0, 0.08, (k, 1))], 1)
            sc = rng.uniform(0.05, 0.95, (k, 1))
            sc[rng.rand(k) < 0.15] = 0.08                                # under SCORE_THRESH
            if k > 4:
                sc[1] = sc[0]                                            # a tie
            fp = rng.randint(0, 6)                                       # a few false positives of this agent alone
            bf = np.concatenate([rng.uniform(-50, 50, (fp, 2)), rng.uniform(-2, -1, (fp, 1)), rng.uniform(3.8, 4.8, (fp, 1)),
                                 rng.uniform(1.7, 2.0, (fp, 1)), rng.uniform(1.5, 1.7, (fp, 1)), rng.uniform(-3, 3, (fp, 1))], 1)
            rows = np.concatenate([np.concatenate([b, sc, np.ones((k, 1))], 1),
                                   np.concatenate([bf, rng.uniform(0.11, 0.4, (fp, 1)), np.ones((fp, 1))], 1)], 0)
            ex[a] = rows.astype(np.float32)
        if f == 0:
            ex[2] = np.zeros((0, 9), np.float32)                         # an agent that detected nothing
        frames.append(ex)
        out['agents_%d' % f] = np.array(list(ex.keys()))
        for a, rows in ex.items():
            out['exchange_%d_%d' % (f, a)] = rows
    for method in ('nms', 'ego_only'):
        model.model_cfg.BOX_FUSION_METHOD = method
        use = frames if method == 'nms' else frames[:2]
        with torch.no_grad():
            pred, _rec = model({'metadata': [{'exchange_boxes': ex} for ex in use], 'batch_size': len(use)})
        out[method + '_frames'] = np.array(len(use))
        for b, d in enumerate(pred):
            out['%s_boxes_%d' % (method, b)] = d['pred_boxes'].numpy().copy()
            out['%s_scores_%d' % (method, b)] = d['pred_scores'].numpy().copy()
            out['%s_labels_%d' % (method, b)] = d['pred_labels'].numpy().copy()
        print('g14', method, [int(d['pred_boxes'].shape[0]) for d in pred], 'from', [sum(r.shape[0] for r in ex.values()) for ex in use])
    out['meta_json'] = np.array(json.dumps(dict(model=rh.to_plain(cfg.MODEL), class_names=list(cfg.CLASS_NAMES))))
    np.savez_compressed(os.path.join(HERE, 'g14_late_fusion.npz'), **out)


def g15_multi_classes_nms():
    """model_nms_utils.multi_classes_nms (reference pcdet/models/model_utils/model_nms_utils.py:28-66) on synthetic 3-class candidates
    (the g14 scene generator: several jittered views per object, scores under the threshold): the reference's own per-class loop ->
    tests/golden/g15_multi_classes_nms.npz"""
    rh.install()
    from pcdet.models.model_utils import model_nms_utils as ref_nms
    rng = np.random.RandomState(1505)
    n_obj, views = 70, 4
    centers = np.concatenate([rng.uniform(-48, 48, (n_obj, 2)), rng.uniform(-2.5, -0.5, (n_obj, 1))], 1)
    dims = np.stack([rng.uniform(0.6, 5.2, n_obj), rng.uniform(0.6, 2.2, n_obj), rng.uniform(1.2, 1.9, n_obj)], 1)
    yaw = rng.uniform(-np.pi, np.pi, (n_obj, 1))
    idx = np.repeat(np.arange(n_obj), views)
    n = idx.shape[0]
    boxes = np.concatenate([centers[idx] + rng.normal(0, 0.2, (n, 3)) * [1, 1, 0.2], dims[idx] * rng.uniform(0.95, 1.05, (n, 3)),
                            yaw[idx] + rng.normal(0, 0.06, (n, 1)), rng.uniform(-1, 1, (n, 2))], 1).astype(np.float32)      # (N, 7 + 2)
    cls = rng.uniform(0.0, 0.3, (n, 3)).astype(np.float32)
    true_cls = rng.randint(0, 3, n_obj)[idx]
    cls[np.arange(n), true_cls] = rng.uniform(0.05, 0.98, n).astype(np.float32)
    cfg = rh.AttrDict(dict(NMS_TYPE='nms_gpu', NMS_THRESH=0.2, NMS_PRE_MAXSIZE=100, NMS_POST_MAXSIZE=30, MULTI_CLASSES_NMS=True))
    out = {'boxes': boxes, 'cls_scores': cls}
    for tag, thr in (('thr', 0.1), ('nothr', None)):
        with torch.no_grad():
            sc, lb, bx = ref_nms.multi_classes_nms(torch.from_numpy(cls.copy()), torch.from_numpy(boxes.copy()), cfg, score_thresh=thr)
        out[tag + '_scores'], out[tag + '_labels'], out[tag + '_boxes'] = sc.numpy().copy(), lb.numpy().copy(), bx.numpy().copy()
        print('g15', tag, [int((lb == k).sum()) for k in range(3)])
    out['meta_json'] = np.array(json.dumps(dict(nms_config=dict(cfg), score_thresh=0.1)))
    np.savez_compressed(os.path.join(HERE, 'g15_multi_classes_nms.npz'), **out)



# ---------------------------------------------------------------------------------------------------------------------------------------
# g17: the two shapes no reference fixture covered -- the LiDAR-like "ring" cloud at BASELINE's full size (pillars of
# ~850 points under the sensor: torch.unique + scatter_mean over long runs, hunter_toolbox.bev_scatter over thousands of points per pixel) and
# the batch bench.py itself launches (make_points(disco, 4, rank 0): B = 4 frames x 6 agents x 60 000 points).  Well-conditioned weights (the
# g13 scheme: gain tuned until the head maps keep an O(1) signal, SCORE_THRESH in the widest score gap, final set invariant under 1e-4 noise),
# so the GPU test demands bit-exact pillar indices of EVERY VFE pass, 1e-3 on every map and the EXACT final set, in bench.py's own mode.
# ---------------------------------------------------------------------------------------------------------------------------------------
class VfeSpy:
    """every DynamicPillarVFE forward of the reference model in call order: module path, rows in, the agent the rows belong to (DiscoNet
    makers), P, N', voxel_coords and unq_inv (the index torch_scatter.scatter_mean receives at dynamic_pillar_vfe.py:110)"""

    def __init__(self, model):
        self.model = model
        self.calls = []
        self._inv = None
        self._hooks = []

    def __enter__(self):
        import torch_scatter
        self._ts, self._orig = torch_scatter, torch_scatter.scatter_mean
        spy = self

        def mean(src, index, dim=0, dim_size=None):
            if src.dim() == 2 and src.shape[1] == 3:
                spy._inv = index.detach().clone()
            return spy._orig(src, index, dim, dim_size)
        torch_scatter.scatter_mean = mean
        for name, mod in self.model.named_modules():
            if type(mod).__name__ == 'DynamicPillarVFE':
                def hook(m, args, out, name=name):
                    pts = args[0]['points'] if isinstance(args[0], dict) else out['points']
                    spy.calls.append(dict(name=name, n_in=int(pts.shape[0]), coords=out['voxel_coords'].detach().numpy().astype(np.int32).copy(),
                                          inv=spy._inv.numpy().astype(np.int64).copy(), last_col=pts[:, -1].detach().numpy().copy()))
                    spy._inv = None
                self._hooks.append(mod.register_forward_hook(hook))
        return self

    def __exit__(self, *exc):
        self._ts.scatter_mean = self._orig
        for h in self._hooks:
            h.remove()

    def dump(self, out, tag, disco):
        out[tag + '_vfe_calls'] = np.array(len(self.calls))
        names = []
        for i, c in enumerate(self.calls):
            k = '%s_vfe_%d_' % (tag, i)
            names.append(c['name'])
            cnt = np.bincount(c['inv'], minlength=c['coords'].shape[0])
            out[k + 'n_in'] = np.array(c['n_in'])
            out[k + 'P'] = np.array(c['coords'].shape[0])
            out[k + 'kept'] = np.array(c['inv'].shape[0])
            out[k + 'coords_sha'] = np.array(sha(c['coords']))
            out[k + 'inv_sha'] = np.array(sha(c['inv']))
            out[k + 'cnt_hist'] = np.bincount(np.minimum(cnt, 63), minlength=64)
            out[k + 'cnt_max'] = np.array(int(cnt.max()) if cnt.size else 0)
            out[k + 'frames'] = np.array(int(c['coords'][:, 0].max()) + 1 if c['coords'].shape[0] else 0)
            if disco:
                u = np.unique(c['last_col'])
                out[k + 'agent'] = np.array(int(u[0]) if u.size == 1 else -1)
        out[tag + '_vfe_names'] = np.array(json.dumps(names))


def _g17_forward(build, make_bd, disco, seg_shift=None):
    import time
    cfg, model, shapes = build()
    if seg_shift is not None:
        with torch.no_grad():
            model.corrector.point_head.seg[0].bias -= torch.tensor(seg_shift, dtype=torch.float32)
    bd = make_bd()
    before = bd['points'].clone()
    logits, hooks, seen, snaps = {}, [], [], {}
    if getattr(model, 'corrector', None) is not None:
        hooks.append(model.corrector.point_head.register_forward_hook(lambda m, a, o: logits.update(cls=o[1].detach().clone(), flow=o[2].detach().clone())))
    if disco:
        hooks.append(model.bev_maker_car.module_list[0].register_forward_pre_hook(lambda m, args: seen.append(args[0]['points'].detach().clone())))
    t0 = time.time()
    with torch.no_grad(), VfeSpy(model) as spy:
        for name, mod in zip(_module_names(model), model.module_list):
            bd = mod(bd)
            if name == 'backbone_2d':
                snaps['backbone_out'] = bd['spatial_features_2d'].detach().clone()
    for h in hooks:
        h.remove()
    return dict(cfg=cfg, model=model, shapes=shapes, bd=bd, before=before, logits=logits, seen=seen, snaps=snaps, spy=spy, secs=time.time() - t0)


def _hunter_verdicts(model, logits, after, seg_extra=0.0):
    """(mask of the rows HunterJr corrects, rows whose verdict could flip under 1e-4 of noise, corrected rows whose BEV coordinate lies within
    2e-4 pixel (0.16 mm; the corrected xyz of two fp32 implementations differ by ~0.03 mm) of a pixel boundary -- such a row may be scattered into the neighbouring pixel by any other fp32 implementation)"""
    cls = logits['cls'].clone()
    cls[:, 2] -= seg_extra
    p = torch.sigmoid(cls)
    top, idx = torch.max(p, dim=1)
    two = torch.topk(p, 2, dim=1)[0]
    thr_p = float(model.corrector.thresh_point_cls_prob)
    other = torch.maximum(p[:, 0], p[:, 1])
    near = ((idx == 2) & ((top - thr_p).abs() < 1e-4)) | (((p[:, 2] - other).abs() < 1e-4) & (top > thr_p - 1e-4))
    mask = (top > thr_p) & (idx == 2)
    edge = torch.zeros_like(mask)
    if after is not None:
        pr = model.corrector.point_cloud_range
        pix = float(model.corrector.voxel_size[0]) * float(model.corrector.bev_image_stride)
        c = (torch.from_numpy(after[:, 1:3].copy()) - torch.tensor([float(pr[0]), float(pr[1])])) / pix
        edge = mask & (((c - torch.round(c)).abs() < 2e-4).any(dim=1))
    return mask, near, edge


def _g17_case(tag, out, build, make_bd, disco=False, head_stride=1):
    """build() -> (cfg, model, shapes) under the current WEIGHT_SCHEME; make_bd() -> a fresh batch dict.  Tunes the gain as _g13_run does,
    runs the reference forward under the spies, stores digests of everything on the path and the robust final sets.
    With a HunterJr corrector the dynamic-foreground logit's bias is shifted (as g10 / g13_chain do) until about 1 % of the rows are
    corrected and none of them is an fp32 coin toss: with the tuned gain EVERY row would be corrected by a flow of metres, and 60 000
    moved points always include a few within 1e-5 of a BEV pixel boundary (bev_scatter's .long(), hunter_toolbox.py:84)."""
    global WEIGHT_SCHEME
    chosen = None
    for gain in G13_GAINS:
        WEIGHT_SCHEME = 'gain:%g' % gain
        r = _g17_forward(build, make_bd, disco)
        model = r['model']
        pd0 = model.dense_head.forward_ret_dict['pred_dicts'][0]
        sd, dmax = float(pd0['hm'].std()), float(pd0['dim'].abs().max())
        print('   g17 %-12s gain %.1f: hm logits %.2f .. %.2f (std %.3f), |dim logits| <= %.2f   [%.0f s]'
              % (tag, gain, float(pd0['hm'].min()), float(pd0['hm'].max()), sd, dmax, r['secs']))
        if 0.15 <= sd <= 0.9 and dmax <= 3.5:
            chosen = gain
            break
    if chosen is None:
        raise RuntimeError('g17 %s: no gain gives a usable head map' % tag)
    seg_shift = None
    if r['logits']:
        # the shift of the dynamic-foreground bias that leaves ~1 % of the rows corrected, then nudged until no corrected row is a coin toss
        cls = r['logits']['cls']
        margin = torch.minimum(cls[:, 2] - torch.maximum(cls[:, 0], cls[:, 1]), cls[:, 2] - float(np.log(0.3 / 0.7)))
        base = float(torch.sort(margin, descending=True)[0][max(1, margin.shape[0] // 100)])
        for attempt in range(40):
            seg_shift = [0.0, 0.0, round(base + 0.013 * attempt, 4)]
            r = _g17_forward(build, make_bd, disco, seg_shift)
            mask, near, edge = _hunter_verdicts(r['model'], r['logits'], r['bd']['points'].numpy())
            print('   g17 %-12s seg bias shift %.4f: %d rows corrected, %d near the verdict, %d near a BEV pixel boundary'
                  % (tag, seg_shift[2], int(mask.sum()), int(near.sum()), int(edge.sum())))
            if int(mask.sum()) >= 100 and int(near.sum()) == 0 and int(edge.sum()) == 0:
                break
        else:
            raise RuntimeError('g17 %s: no seg bias shift gives a well-conditioned HunterJr correction' % tag)
        out[tag + '_seg_bias_shift'] = np.array(seg_shift, dtype=np.float32)
    cfg, model, shapes, bd, before, logits, seen, snaps, spy = (r[k] for k in ('cfg', 'model', 'shapes', 'bd', 'before', 'logits', 'seen', 'snaps', 'spy'))
    B = int(bd['batch_size'])
    head = model.dense_head
    pred_dicts = [{k: v.detach().clone() for k, v in pd.items()} for pd in head.forward_ret_dict['pred_dicts']]
    thr, finals = robust_threshold(head, B, pred_dicts, tag)
    _g13_store(out, tag, finals, thr)
    out[tag + '_weight_scheme'] = np.array(WEIGHT_SCHEME)
    out[tag + '_N'] = np.array(int(before.shape[0]))
    out[tag + '_points_sha'] = np.array(sha(before.numpy()))
    spy.dump(out, tag, disco)
    for name in ('center', 'center_z', 'dim', 'rot', 'hm'):
        out['%s_head_%s' % (tag, name)] = pred_dicts[0][name].numpy()[:, :, ::head_stride, ::head_stride].copy()
    out[tag + '_head_stride'] = np.array(head_stride)
    sf = bd['spatial_features_2d'].numpy()
    out[tag + '_sf2d_probe'] = sf[:, :, ::16, ::16].copy()
    out[tag + '_sf2d_sum'] = sf.astype(np.float64).sum((0, 2, 3))
    out[tag + '_sf2d_max'] = sf.max(axis=(0, 2, 3))
    if 'backbone_out' in snaps:
        bo = snaps['backbone_out'].numpy()
        out[tag + '_backbone_probe'] = bo[:, :, ::16, ::16].copy()
        out[tag + '_backbone_sum'] = bo.astype(np.float64).sum((0, 2, 3))
    pf = bd['pillar_features'].numpy().astype(np.float64)            # the LAST VFE of the chain (the trainable branch)
    out[tag + '_pf_sum'], out[tag + '_pf_abs'], out[tag + '_pf_max'] = pf.sum(0), np.abs(pf).sum(0), pf.max(0).astype(np.float32)
    if disco:
        car_agents = sorted(int(a) for a in bd['bev_img'].keys())
        out[tag + '_bev_agents'] = np.array(car_agents)
        for aid in car_agents:
            a = bd['bev_img'][aid].numpy()
            out['%s_bev_%d_probe' % (tag, aid)] = a[:, ::8, ::8, ::8].copy()
            out['%s_bev_%d_sum' % (tag, aid)] = a.astype(np.float64).sum((0, 2, 3))
            out['%s_bev_%d_max' % (tag, aid)] = a.max(axis=(0, 2, 3))
        e = bd['bev_img_early'].numpy()
        out[tag + '_bev_early_probe'] = e[:, ::8, ::8, ::8].copy()
        out[tag + '_bev_early_sum'] = e.astype(np.float64).sum((0, 2, 3))
        for ap in seen:                                                # the ego -> agent transform as the frozen chain receives it, per agent
            ap = ap.numpy()
            a = int(ap[0, -1])
            out['%s_car_agent_%d_rows' % (tag, a)] = np.array(ap.shape[0])
            out['%s_car_agent_%d_xyz_sha' % (tag, a)] = np.array(sha(ap[:, 1:4].astype(np.float32)))
    if logits:
        # HunterJr (hunter_jr.py:251-264): the rows it corrects, their xyz afterwards (demanded to 1e-4), every other row bit for bit
        after = bd['points'].numpy()
        mask, near, edge = _hunter_verdicts(model, logits, after)
        assert int(near.sum()) == 0 and int(edge.sum()) == 0
        changed = np.nonzero(mask.numpy())[0].astype(np.int32)
        assert np.array_equal(np.nonzero((after != before.numpy()).any(1))[0], changed[(after[changed] != before.numpy()[changed]).any(1)])
        out[tag + '_hunter_dyn_rows'] = np.array(int(mask.sum()))
        out[tag + '_hunter_rows'] = changed
        out[tag + '_hunter_xyz_after'] = after[changed, 1:4].copy()
        print('   g17 %-12s HunterJr: %d of %d rows corrected, flow |max| %.2f m' % (tag, int(mask.sum()), after.shape[0], float(logits['flow'][mask].abs().max())))
    print('g17', tag, 'N', int(before.shape[0]), 'vfe calls', [(c['name'], c['coords'].shape[0]) for c in spy.calls], 'thr %.6f' % thr,
          'finals', [int(d['pred_boxes'].shape[0]) for d in finals])
    return thr, cfg, shapes


def _disco_builder():
    tmp = tempfile.mkdtemp()
    empty = os.path.join(tmp, 'empty.pth')
    torch.save({'model_state': {}}, empty)
    ov = {'MODEL.BEV_MAKER_RSU.CKPT': empty, 'MODEL.BEV_MAKER_CAR.CKPT': empty, 'MODEL.BEV_MAKER_EARLY.CKPT': empty}

    def build():
        cfg = rh.load_cfg('v2x_pointpillar_disco.yaml', ov)
        model, _ds = rh.build_model(cfg)
        shapes = fill_weights(model)
        return cfg, model, shapes
    return build


def g17_ring_full():
    """tests/golden/g2_ring_full.npz: basic_car (1 x 60 000, HunterJr incl. the corrected points) and DiscoNet (6 x 60 000, B = 1) on
    synth.agent_cloud(dist='ring')"""
    global WEIGHT_SCHEME
    out = {}
    try:
        def build_car():
            cfg = rh.load_cfg('v2x_pointpillar_basic_car.yaml', {})
            model, _ds = rh.build_model(cfg)
            return cfg, model, fill_weights(model)
        cpts = synth.collate([synth.agent_cloud(agent=0, n_points=60000, layout='car', dist='ring')])
        _g17_case('car', out, build_car, lambda: {'points': torch.from_numpy(cpts.copy()), 'batch_size': 1, 'metadata': [{}]})
        agents = (0, 1, 2, 3, 4, 5)
        poses = {a: synth.agent_pose(a) for a in agents if a != 1}
        clouds = []
        for a in agents:
            c = synth.agent_cloud(agent=a, n_points=60000, layout='disco', dist='ring')
            c[:, -1] = float(a)
            clouds.append(c)
        dpts = synth.collate([np.concatenate(clouds, axis=0)])
        _g17_case('disco', out, _disco_builder(), lambda: {'points': torch.from_numpy(dpts.copy()), 'batch_size': 1, 'metadata': [{'se3_from_ego': poses}]},
                  disco=True)
        for a in poses:
            out['disco_pose_%d' % a] = poses[a]
        out['meta_json'] = np.array(json.dumps(dict(noise=G13_NOISE, trials=G13_TRIALS, dist='ring', n_points=60000)))
        np.savez_compressed(os.path.join(HERE, 'g2_ring_full.npz'), **out)
    finally:
        WEIGHT_SCHEME = 'survey'


def g17_disco_b4():
    """tests/golden/g2_disco_full_b4.npz: the batch bench.py's headline times -- bench.make_points(CONFIGS['disco'], 4, rank 0), both
    distributions -- through the reference's DiscoNet forward"""
    global WEIGHT_SCHEME
    sys.path.insert(0, REPO)
    import bench
    out = {}
    try:
        for dist in ('uniform', 'ring'):
            pts, metas = bench.make_points(bench.CONFIGS['disco'], 4, 0, dist)
            _g17_case(dist, out, _disco_builder(), lambda: {'points': torch.from_numpy(pts.copy()), 'batch_size': 4,
                                                            'metadata': [{'se3_from_ego': dict(m['se3_from_ego'])} for m in metas]}, disco=True)
        out['meta_json'] = np.array(json.dumps(dict(noise=G13_NOISE, trials=G13_TRIALS, batch=4, source='bench.make_points(CONFIGS[disco], 4, 0, dist)')))
        np.savez_compressed(os.path.join(HERE, 'g2_disco_full_b4.npz'), **out)
    finally:
        WEIGHT_SCHEME = 'survey'


def g17_bench_b4(dist='uniform'):
    """tests/golden/g2_bench_b4.npz: the batches bench.py times for BASELINE configs 2 - 4 (`configs` of its line) -- bench.make_points(CONFIGS[c],
    4, rank 0) for c in car / ego / early -- through the reference's own forward; head maps stored on every second pixel.
    dist='ring': the same batches of `bench.py --dist ring` (car: HunterJr's bev_scatter with thousands of points per pixel in four frames; early:
    the merged 6-agent ring cloud) into g2_bench_b4_ring.npz"""
    global WEIGHT_SCHEME
    sys.path.insert(0, REPO)
    import bench
    out = {}
    try:
        for tag in (('car', 'ego', 'early') if dist == 'uniform' else ('car', 'early')):
            conf = bench.CONFIGS[tag]
            pts, _metas = bench.make_points(conf, 4, 0, dist)

            def build(yaml_name=conf['yaml']):
                cfg = rh.load_cfg(yaml_name, {})
                model, _ds = rh.build_model(cfg)
                return cfg, model, fill_weights(model)
            _g17_case(tag, out, build, lambda: {'points': torch.from_numpy(pts.copy()), 'batch_size': 4, 'metadata': [{} for _ in range(4)]}, head_stride=2)
        out['meta_json'] = np.array(json.dumps(dict(noise=G13_NOISE, trials=G13_TRIALS, batch=4, dist=dist, source='bench.make_points(CONFIGS[c], 4, 0, dist)')))
        np.savez_compressed(os.path.join(HERE, 'g2_bench_b4.npz' if dist == 'uniform' else 'g2_bench_b4_ring.npz'), **out)
    finally:
        WEIGHT_SCHEME = 'survey'


if __name__ == '__main__':
    todo = sys.argv[1:] or ['g1', 'g2', 'g3', 'g4']
    torch.set_num_threads(8)
    if 'g17r' in todo:
        g17_ring_full()
    if 'g17b' in todo:
        g17_disco_b4()
    if 'g17c' in todo:
        g17_bench_b4()
    if 'g17d' in todo:
        g17_bench_b4('ring')
    if 'g13' in todo:
        g13_conditioned()
    if 'g13c' in todo:
        g13_chain()
    if 'g13cf' in todo:
        g13_chain(full=True)
    if 'g13cb4' in todo:
        g13_chain(full='b4')
    if 'g14' in todo:
        g14_late_fusion()
    if 'g15' in todo:
        g15_multi_classes_nms()
    if 'g3' in todo:
        g3_nms()
    if 'g4' in todo:
        g4_warp()
    if 'g1' in todo:
        g1_single('car', 'v2x_pointpillar_basic_car.yaml', 'car', extra_points=hunter_edge_points)
        g1_single('rsu', 'v2x_pointpillar_basic_rsu.yaml', 'car', extra_points=hunter_edge_points)
        g1_single('ego', 'v2x_pointpillar_basic_ego.yaml', 'lately', score_thresh=0.02)
        g1_single('early', 'v2x_pointpillar_basic_ego_early.yaml', 'early', score_thresh=0.02)
        g1_disco()
    if 'g1rsu' in todo:
        g1_single('rsu', 'v2x_pointpillar_basic_rsu.yaml', 'car', extra_points=hunter_edge_points)
    if 'g2' in todo:
        g2_full()
    if 'g2d' in todo:
        g2_disco_full()
    if 'g7' in todo:
        g7_train()
    if 'g7f' in todo:
        g7_train_full()
    if 'g7fb4' in todo:
        g7_train_full(b4=True)
    if 'g7b' in todo:
        g7b_train_single('ego', 'v2x_pointpillar_basic_ego.yaml', 'lately')
    if 'g8' in todo:
        g8_exchange()
    if 'g16' in todo:
        g16_pfn_variants()
    if 'g10' in todo:
        g10_lately_chain()
    if 'g11' in todo:
        g11_anchor_train()
    if 'g11f' in todo:
        g11f_anchor_train_full()
    if 'g12' in todo:
        g12_hunter_train()
    if 'g12f' in todo:
        g12f_hunter_train_full()
    if 'g9' in todo:
        g9_anchor('agnostic', False)     # MULTI_CLASSES_NMS with a single (non multi-head) AnchorHeadSingle trips the reference's own
                                         # assertion (detector3d_template.py:283,295: arange(1, num_class) has num_class - 1 entries)
