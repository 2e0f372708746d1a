"""20 training iterations of config 5 at full size (one frame, the fixture of tests/test_gpu_train_e2e.py) under three arithmetic modes:
fp32 (auto: Winograd / direct MFMA), fp32 with the direct kernels only (another summation order), and the bf16 loop."""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, 'practical-collab-perception_amd'), os.path.join(REPO, 'tests'), os.path.join(REPO, 'practical-collab-perception_amd', 'tools')):
    sys.path.insert(0, p)
from helpers import load_golden  # noqa: E402
from pcp_amd import synth  # noqa: E402
from pcdet.config import EasyDict, cfg_from_yaml_file  # noqa: E402
from pcdet.models import DatasetInfo, build_network  # noqa: E402
from train_utils.optimization import build_optimizer, build_scheduler  # noqa: E402

N_IT = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = load_golden('g7_train_full.npz')
out = {}
for algo in ('auto', 'direct', 'bf16'):
    os.environ['PCP_CONV_ALGO'] = algo
    cfg = cfg_from_yaml_file(os.path.join(REPO, 'practical-collab-perception_amd', 'tools', 'cfgs', 'v2x_sim_models', 'v2x_pointpillar_disco.yaml'), EasyDict())
    for key in ('BEV_MAKER_RSU', 'BEV_MAKER_CAR', 'BEV_MAKER_EARLY'):
        cfg.MODEL[key].CKPT = None
    vs = [p.VOXEL_SIZE for p in cfg.DATA_CONFIG.DATA_PROCESSOR if 'VOXEL_SIZE' in p][0]
    ds = DatasetInfo(cfg.CLASS_NAMES, cfg.DATA_CONFIG.POINT_CLOUD_RANGE, vs, len(cfg.DATA_CONFIG.POINT_FEATURE_ENCODING.used_feature_list))
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds)
    st = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    model = model.cuda()
    ocfg = EasyDict(json.loads(str(g['optimization_json'])))
    opt = build_optimizer(model, ocfg)
    sched, _ = build_scheduler(opt, N_IT, 1, -1, ocfg)
    clouds = []
    for a in range(6):
        c = synth.agent_cloud(agent=a, n_points=60000, layout='disco')
        c[:, -1] = float(a)
        clouds.append(c)
    pts = synth.collate([np.concatenate(clouds, axis=0)])
    poses = {a: g['pose_%d' % a] for a in range(6) if a != 1}
    losses = []
    for it in range(N_IT):
        sched.step(it)
        model.train()
        opt.zero_grad()
        ret, tb, _ = model({'points': torch.from_numpy(pts).cuda(), 'batch_size': 1, 'metadata': [{'se3_from_ego': poses}],
                            'gt_boxes': torch.from_numpy(g['gt_boxes']).cuda()})
        model.update_global_step()
        ret['loss'].backward()
        opt.clip_grad_norm(ocfg.GRAD_NORM_CLIP)
        opt.step()
        losses.append(round(float(ret['loss'].detach()), 5))
    out[algo] = losses
    print(algo, losses)
    del model, opt
    torch.cuda.empty_cache()
a, d, b = (np.array(out[k]) for k in ('auto', 'direct', 'bf16'))
print('max |direct - auto| / auto = %.4f   max |bf16 - auto| / auto = %.4f' % (float((np.abs(d - a) / a).max()), float((np.abs(b - a) / a).max())))
print('mean over the last 5 iterations: auto %.4f direct %.4f bf16 %.4f' % (a[-5:].mean(), d[-5:].mean(), b[-5:].mean()))
