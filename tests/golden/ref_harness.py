"""Imports the reference (read-only, from /root/reference) in the CPU container so that golden vectors can be
generated from the reference's OWN modules.  Runs only where /root/reference exists; nothing here travels to the GPU
box except the .npz files it helped produce.

What is substituted:
  * packages the image lacks and the hot path never executes (spconv, nuscenes, kornia, ...) -> attribute-tolerant
    placeholder modules, only so that `import pcdet.models` succeeds;
  * easydict -> a 20-line dict-with-attributes;
  * torch_scatter (third party, not vendored, unpinned) -> torch-native scatter_{sum,mean,max,min}: THIS is the oracle
    definition of that arithmetic;
  * `.cuda()` -> identity, np.int -> int (reference quirks Q1, Q2);
  * the compiled op pcdet.ops.iou3d_nms.iou3d_nms_cuda -> nms_gpu built on oracle/_ref (the reference's own
    iou3d_cpu.cpp compiled where it lies) + the greedy loop of iou3d_nms.cpp:121-132.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

REF_ROOT = '/root/reference'
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))


class _Loose(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        obj = type(name, (torch.nn.Module,), {}) if name[0].isupper() else _Loose(self.__name__ + '.' + name)
        setattr(self, name, obj)
        return obj

    def __call__(self, *a, **k):
        raise RuntimeError('placeholder module called: ' + self.__name__)


class _LooseFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = {'nuscenes', 'torchvision', 'skimage', 'cv2', 'open3d', 'pyquaternion', 'spconv', 'kornia', 'numba',
             'shapely', 'av2', 'mayavi', 'cumm', 'tensorboardX', 'SharedArray', 'gitinfo', 'lovely_tensors'}

    def find_spec(self, name, path, target=None):
        if name.split('.')[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Loose(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


class AttrDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        elif isinstance(v, (list, tuple)):
            v = type(v)(AttrDict(x) if isinstance(x, dict) and not isinstance(x, AttrDict) else x for x in v)
        dict.__setitem__(self, k, v)
        object.__setattr__(self, k, v)

    __setitem__ = __setattr__

    def update(self, e=None, **f):
        for k, v in dict(e or {}, **f).items():
            setattr(self, k, v)


def _scatter_module():
    ts = types.ModuleType('torch_scatter')

    def _rows(index, dim_size):
        return int(index.max().item()) + 1 if dim_size is None else dim_size

    def scatter_sum(src, index, dim=0, dim_size=None):
        out = src.new_zeros((_rows(index, dim_size),) + tuple(src.shape[1:]))
        return out.index_add_(0, index, src)

    def scatter_mean(src, index, dim=0, dim_size=None):
        n = _rows(index, dim_size)
        s = scatter_sum(src, index, 0, n)
        c = torch.bincount(index, minlength=n).clamp_(min=1).to(src.dtype)
        return s / c.view(-1, *([1] * (src.dim() - 1)))

    def _extreme(src, index, n, red):
        idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
        out = src.new_zeros((n,) + tuple(src.shape[1:])).scatter_reduce(0, idx, src, red, include_self=False)
        pos = torch.arange(src.shape[0]).view(-1, *([1] * (src.dim() - 1))).expand_as(src)
        big = src.shape[0]
        arg = torch.full(out.shape, big, dtype=torch.long).scatter_reduce(
            0, idx, torch.where(src == out[index], pos, big), 'amin', include_self=True)
        return out, arg

    def scatter_max(src, index, dim=0, dim_size=None):
        return _extreme(src, index, _rows(index, dim_size), 'amax')

    def scatter_min(src, index, dim=0, dim_size=None):
        return _extreme(src, index, _rows(index, dim_size), 'amin')

    def scatter(src, index, dim=0, dim_size=None, reduce='sum'):
        return {'sum': scatter_sum, 'mean': scatter_mean}[reduce](src, index, dim, dim_size)

    for f in (scatter_sum, scatter_mean, scatter_max, scatter_min, scatter):
        setattr(ts, f.__name__, f)
    return ts


def _nms_module():
    sys.path.insert(0, REPO)
    from oracle import build_ref
    ref = build_ref.load_ref()
    m = types.ModuleType('pcdet.ops.iou3d_nms.iou3d_nms_cuda')

    def boxes_iou_bev_gpu(a, b, out):
        ref.boxes_iou_bev_cpu(a.contiguous(), b.contiguous(), out)
        return 1

    def nms_gpu(boxes, keep, thresh):
        n = boxes.shape[0]
        iou = torch.zeros(n, n)
        ref.boxes_iou_bev_cpu(boxes.contiguous(), boxes.contiguous(), iou)
        dead = np.zeros(n, dtype=bool)
        over = (iou > thresh).numpy()
        k = 0
        for i in range(n):                     # iou3d_nms.cpp:121-132
            if dead[i]:
                continue
            keep[k] = i
            k += 1
            dead[i + 1:] |= over[i, i + 1:]
        return k

    m.boxes_iou_bev_gpu = boxes_iou_bev_gpu
    m.boxes_iou_bev_cpu = ref.boxes_iou_bev_cpu
    m.nms_gpu = nms_gpu
    m.ref = ref
    return m


_READY = False


def install():
    global _READY
    if _READY:
        return
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError('the reference is not mounted here; golden vectors can only be regenerated in the CPU container')
    sys.dont_write_bytecode = True
    np.int = int
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.meta_path.append(_LooseFinder())
    import gitinfo
    gitinfo.get_git_info = lambda: {'commit': '0000000'}
    import lovely_tensors
    lovely_tensors.monkey_patch = lambda *a, **k: None
    ver = types.ModuleType('pcdet.version')
    ver.__version__ = '0.6.0+ref'
    sys.modules['pcdet.version'] = ver
    ed = types.ModuleType('easydict')
    ed.EasyDict = AttrDict
    sys.modules['easydict'] = ed
    sys.modules['torch_scatter'] = _scatter_module()
    sys.modules['pcdet.ops.iou3d_nms.iou3d_nms_cuda'] = _nms_module()
    for n in ('pcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda', 'pcdet.ops.roipoint_pool3d.roipoint_pool3d_cuda',
              'pcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda',
              'pcdet.ops.pointnet2.pointnet2_batch.pointnet2_batch_cuda'):
        mod = _Loose(n)
        mod.__path__ = []
        sys.modules[n] = mod
    sys.path.insert(0, REF_ROOT)
    os.chdir(os.path.join(REF_ROOT, 'tools'))          # _BASE_CONFIG_ paths are cwd-relative (quirk Q6)
    _READY = True


def load_cfg(yaml_name, overrides=None):
    """Returns a fresh AttrDict cfg for tools/cfgs/v2x_sim_models/<yaml_name> through the reference's own config.py."""
    install()
    from pcdet.config import cfg_from_yaml_file
    cfg = AttrDict()
    cfg_from_yaml_file(os.path.join(REF_ROOT, 'tools', 'cfgs', 'v2x_sim_models', yaml_name), cfg)
    for path, val in (overrides or {}).items():
        d = cfg
        keys = path.split('.')
        for k in keys[:-1]:
            d = d[k]
        d[keys[-1]] = val
    return cfg


class FakeDataset:
    """The six attributes Detector3DTemplate.build_networks reads (detector3d_template.py:40-48), computed the way
    DatasetTemplate.__init__ / DataProcessor do (dataset.py:25-46, data_processor.py:106-114)."""

    def __init__(self, data_cfg, class_names):
        self.class_names = class_names
        self.point_cloud_range = np.array(data_cfg.POINT_CLOUD_RANGE, dtype=np.float32)
        enc = data_cfg.POINT_FEATURE_ENCODING
        self.point_feature_encoder = types.SimpleNamespace(num_point_features=len(enc.used_feature_list))
        vs = None
        for p in data_cfg.DATA_PROCESSOR:
            if 'VOXEL_SIZE' in p:
                vs = p.VOXEL_SIZE
        self.voxel_size = vs
        self.grid_size = np.round((self.point_cloud_range[3:6] - self.point_cloud_range[0:3]) / np.array(vs)).astype(np.int64)
        self.depth_downsample_factor = None


def build_model(cfg):
    install()
    from pcdet.models import build_network
    ds = FakeDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES)
    model = build_network(model_cfg=cfg.MODEL, num_class=len(cfg.CLASS_NAMES), dataset=ds)
    model.eval()
    return model, ds


def to_plain(d):
    if isinstance(d, dict):
        return {k: to_plain(v) for k, v in d.items()}
    if isinstance(d, (list, tuple)):
        return [to_plain(v) for v in d]
    if isinstance(d, (np.integer,)):
        return int(d)
    if isinstance(d, (np.floating,)):
        return float(d)
    return d
