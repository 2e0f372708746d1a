"""build_network / load_data_to_gpu / model_fn_decorator with the reference's signatures (pcdet/models/__init__.py:16-50)."""
import types
from collections import namedtuple

import numpy as np
import torch

from .detectors import build_detector


def build_network(model_cfg, num_class, dataset):
    return build_detector(model_cfg=model_cfg, num_class=num_class, dataset=dataset)


def load_data_to_gpu(batch_dict):
    """numpy -> float32 CUDA for everything except bookkeeping keys (reference :23-34)."""
    for key, val in batch_dict.items():
        if not isinstance(val, np.ndarray) or key in ('frame_id', 'metadata', 'calib'):
            continue
        if key in ('image_shape',):
            batch_dict[key] = torch.from_numpy(val).int().cuda()
        else:
            batch_dict[key] = torch.from_numpy(val).float().cuda().contiguous()


def model_fn_decorator():
    ModelReturn = namedtuple('ModelReturn', ['loss', 'tb_dict', 'disp_dict'])

    def model_func(model, batch_dict):
        load_data_to_gpu(batch_dict)
        ret_dict, tb_dict, disp_dict = model(batch_dict)
        loss = ret_dict['loss']
        (model if hasattr(model, 'update_global_step') else model.module).update_global_step()
        return ModelReturn(loss, tb_dict, disp_dict)

    return model_func


class DatasetInfo:
    """The attributes Detector3DTemplate.build_networks reads from a dataset (detector3d_template.py:40-48 of the
    reference), derived from the config the way DatasetTemplate / DataProcessor derive them (dataset.py:25-46,
    data_processor.py:106-114).  Lets a model be built without any dataset on disk."""

    def __init__(self, class_names, point_cloud_range, voxel_size, num_point_features):
        self.class_names = list(class_names)
        self.point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        self.voxel_size = list(voxel_size)
        self.grid_size = np.round((self.point_cloud_range[3:6] - self.point_cloud_range[0:3]) / np.array(voxel_size)).astype(np.int64)
        self.point_feature_encoder = types.SimpleNamespace(num_point_features=int(num_point_features))
        self.depth_downsample_factor = None


def build_network_from_meta(meta):
    """meta: dict(model=<MODEL section as plain dicts>, pc_range, voxel_size, class_names) as stored with the golden
    fixtures -> an un-initialised detector on the CPU."""
    from ..config import EasyDict
    cfg = EasyDict(meta['model'])
    for key in ('BEV_MAKER_RSU', 'BEV_MAKER_CAR', 'BEV_MAKER_EARLY'):
        if cfg.get(key, None) is not None:
            cfg[key].CKPT = None
    n_feat = {'car': 7, 'early': 7, 'lately': 13, 'disco': 6}.get(meta.get('layout', 'car'), 7)
    ds = DatasetInfo(meta['class_names'], meta['pc_range'], meta['voxel_size'], n_feat)
    return build_network(cfg, len(meta['class_names']), ds)
