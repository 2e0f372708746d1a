"""ORACLE (test infrastructure only) -- whole-detector CPU forward (the module chain of
/root/reference/pcdet/models/detectors/centerpoint.py:9-33 as built by detector3d_template.py:27-54) and the
config -> plain-dict "arch" translation used by tests and by bench.py's cpu_baseline leg.
"""
import numpy as np
import torch

from . import bev as obev
from . import pillars as opil


def arch_from_cfg(model_cfg, pc_range, voxel_size):
    """model_cfg: the MODEL section as nested plain dicts (yaml.safe_load output)."""
    g = lambda d, k, default=None: d.get(k, default) if d is not None else default
    grid = np.round((np.asarray(pc_range[3:6], dtype=np.float32) - np.asarray(pc_range[0:3], dtype=np.float32))
                    / np.asarray(voxel_size, dtype=np.float64)).astype(np.int64)

    def vfe_bb(sec):
        return dict(num_raw=sec['VFE']['NUM_RAW_POINT_FEATURES'], vfe_filters=list(sec['VFE']['NUM_FILTERS']),
                    backbone=dict(layer_nums=list(sec['BACKBONE_2D']['LAYER_NUMS']),
                                  strides=list(sec['BACKBONE_2D']['LAYER_STRIDES']),
                                  filters=list(sec['BACKBONE_2D']['NUM_FILTERS']),
                                  up_strides=list(sec['BACKBONE_2D']['UPSAMPLE_STRIDES']),
                                  up_filters=list(sec['BACKBONE_2D']['NUM_UPSAMPLE_FILTERS'])))

    arch = dict(pc_range=[float(v) for v in pc_range], voxel_size=[float(v) for v in voxel_size],
                grid_size=[int(v) for v in grid])
    arch.update(vfe_bb(model_cfg))
    hd = model_cfg['DENSE_HEAD']
    if hd['NAME'] == 'AnchorHeadSingle':                   # MODEL.NAME PointPillar: oracle/anchor.py reads the head block itself
        arch['head'] = dict(kind='anchor', cfg=hd)
    else:
        pp = hd['POST_PROCESSING']
        heads = [(n, hd['SEPARATE_HEAD_CFG']['HEAD_DICT'][n]['out_channels']) for n in hd['SEPARATE_HEAD_CFG']['HEAD_ORDER']]
        heads.append(('hm', len(hd['CLASS_NAMES_EACH_HEAD'][0])))
        arch['head'] = dict(shared=hd['SHARED_CONV_CHANNEL'], heads=heads, num_conv=hd['NUM_HM_CONV'],
                            stride=hd['TARGET_ASSIGNER_CONFIG']['FEATURE_MAP_STRIDE'], max_obj=pp['MAX_OBJ_PER_SAMPLE'],
                            score_thresh=pp['SCORE_THRESH'], limit_range=list(pp['POST_CENTER_LIMIT_RANGE']),
                            nms_thresh=pp['NMS_CONFIG']['NMS_THRESH'], nms_pre=pp['NMS_CONFIG']['NMS_PRE_MAXSIZE'],
                            nms_post=pp['NMS_CONFIG']['NMS_POST_MAXSIZE'])
    co = g(model_cfg, 'CORRECTOR')
    arch['corrector'] = None if co is None else dict(
        bev_stride=co['BEV_IMAGE_STRIDE'], point_hidden=list(co['POINT_HEAD_HIDDEN_CHANNELS']),
        thresh_cls=co.get('THRESHOLD_POINT_CLS_PROB', 0.3))
    fu = g(model_cfg, 'V2X_MID_FUSION')
    arch['fusion'] = None if fu is None else dict(compressed=fu['COMPRESSED_CHANNELS'], pc_min=fu.get('PC_RANGE_MIN', -51.2),
                                                  pix=fu.get('FINAL_BEV_PIXEL_SIZE', 0.2 * 4))
    arch['makers'] = {}
    for key, name in (('BEV_MAKER_RSU', 'bev_maker_rsu'), ('BEV_MAKER_CAR', 'bev_maker_car'), ('BEV_MAKER_EARLY', 'bev_maker_early')):
        sec = g(model_cfg, key)
        if sec is not None:
            a = dict(pc_range=arch['pc_range'], voxel_size=arch['voxel_size'], grid_size=arch['grid_size'],
                     maker_type=sec['MAKER_TYPE'])
            a.update(vfe_bb(sec))
            arch['makers'][name] = a
    return arch


def _np_state(st):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in st.items()}


def vfe_to_map(points_np, st_np, st_t, arch, prefix):
    """a1-a6 for one (sub)network whose keys start with `prefix` ('' for the main branch)."""
    pv = (prefix + '.' if prefix else '')
    v = opil.vfe_forward(points_np, st_np, arch, prefix=pv + 'vfe')
    x = torch.from_numpy(v['spatial_features'])
    m, feats = obev.backbone(x, st_t, arch, prefix=pv + 'backbone_2d')
    return v, m, feats


def bev_maker(points_np, metadata, st_np, st_t, march, name):
    """BEVMaker.forward (bev_maker.py:150-236).  Returns dict agent -> map (rsu/car) or the single 'early' map."""
    if march['maker_type'] == 'early':
        _, m, _ = vfe_to_map(points_np.copy(), st_np, st_t, march, name)
        return m
    out = {}
    agent_col = points_np[:, -1].astype(np.int64)
    for aid in np.unique(agent_col):
        if aid == 1 or (march['maker_type'] == 'rsu' and aid != 0):
            continue
        ap = torch.from_numpy(points_np[agent_col == aid].copy())
        bcol = ap[:, 0].long()
        for b, meta in enumerate(metadata):
            msk = bcol == b
            if not bool(msk.any()):
                continue
            T = torch.from_numpy(meta['se3_from_ego'][int(aid)]).float()
            ap[msk, 1:4] = ap[msk, 1:4] @ T[:3, :3].t() + T[:3, -1]
        _, m, _ = vfe_to_map(ap.numpy(), st_np, st_t, march, name)
        out[int(aid)] = m
    return out


def forward(points, state, arch, metadata=None, with_nms=True):
    """CenterPoint eval forward on CPU.  points: (N, 1+C) float32 numpy; state: name -> array/tensor.
    Returns dict with every boundary tensor of SURVEY 8(b)4."""
    st_np = _np_state(state)
    st_t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in st_np.items()}
    out = {}
    pts = np.ascontiguousarray(points, dtype=np.float32)
    bev_img, bev_early = None, None
    with torch.no_grad():
        for name in ('bev_maker_rsu', 'bev_maker_car', 'bev_maker_early'):
            if name in arch.get('makers', {}):
                r = bev_maker(pts, metadata, st_np, st_t, arch['makers'][name], name)
                if arch['makers'][name]['maker_type'] == 'early':
                    bev_early = r
                else:
                    bev_img = r          # a later maker REPLACES the dict (bev_maker.py:157)
        v, m, feats = vfe_to_map(pts, st_np, st_t, arch, '')
        out.update(voxel_coords=v['vox']['coords'], unq_inv=v['vox']['inv'], unq_cnt=v['vox']['cnt'],
                   pillar_features=v['pillar_features'], spatial_features=v['spatial_features'],
                   backbone_out=m.numpy().copy())
        if arch.get('corrector') is not None:
            m, hj = obev.hunter_jr(m, torch.from_numpy(pts), st_t, arch)
            out['hunter'] = hj
        if arch.get('fusion') is not None:
            se3 = [md['se3_from_ego'] for md in metadata]
            m, fz = obev.disco_fusion(m, bev_img, se3, st_t, arch)
            out['bev_img'] = bev_img
            out['bev_img_early'] = bev_early
            out['fusion'] = fz
        out['spatial_features_2d'] = m.numpy()
        maps = obev.center_head_maps(m, st_t, arch)
        out['head_maps'] = {k: t.numpy() for k, t in maps.items()}
        out['decoded'] = obev.decode_boxes(maps, arch)
        if with_nms:
            out['final_box_dicts'] = obev.head_postprocess(maps, arch)
    return out
