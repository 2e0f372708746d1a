"""eager DiscoNet forward with the BEV makers' static (device-side) agent discovery against the normal eager forward: first tensor that differs"""
import os, sys
import numpy as np, torch
R = os.environ.get('GRAFT_REPO_ROOT', '.')
for p in (R, os.path.join(R, 'practical-collab-perception_amd'), os.path.join(R, 'tests')):
    sys.path.insert(0, p)
from helpers import load_golden
from pcdet.models import build_network_from_meta
from pcp_amd import synth
g = load_golden('g1_disco.npz')
model = build_network_from_meta(g['meta'])
st = synth.fill_state_dict(g['meta']['state_shapes'])
model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
model = model.cuda().eval()
for m in model.modules():
    if hasattr(m, 'materialize_pillars'):
        m.materialize_pillars, m.reuse_buffers = False, True
pts = torch.from_numpy(g['points']).cuda()
metadata = [{'se3_from_ego': {0: g['pose_0'], 2: g['pose_2']}}, {'se3_from_ego': {0: g['pose_0'], 2: g['pose_2']}}]
outs = []
for static in (False, True):
    bd = {'points': pts.clone(), 'batch_size': 2, 'metadata': metadata}
    if static:
        bd['_pcp_static_agents'] = True
    with torch.no_grad():
        for m in model.module_list:
            bd = m(bd)
            if type(m).__name__ == 'BEVMaker':
                print(static, m.maker_type, {k: (tuple(v.shape), float(v.float().abs().sum())) for k, v in bd.get('bev_img', {}).items()},
                      'early' if 'bev_img_early' in bd else '')
    torch.cuda.synchronize()
    outs.append(bd)
    if static:
        print('live', bd['_pcp_agent_live'].view(64, 2)[:4].tolist())
a, b = outs
for k in ('spatial_features_2d',):
    print(k, float((a[k].float() - b[k].float()).abs().max()))
