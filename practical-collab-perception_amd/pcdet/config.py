"""YAML config API, call-compatible with the reference's pcdet/config.py (:8-85): global `cfg`, `cfg_from_yaml_file`
(with recursive `_BASE_CONFIG_` merge), `cfg_from_list` (--set K V ...), `merge_new_config`, `log_config_to_file`.

Differences, on purpose: no dependency on the `easydict` package (a small attribute-dict is bundled), and a
`_BASE_CONFIG_` path that does not exist relative to the cwd is also tried relative to the including YAML and to this
package's tools/ directory (the reference only works when launched from tools/, quirk Q6).
"""
import os
from ast import literal_eval
from pathlib import Path

import yaml


class EasyDict(dict):
    """dict whose keys are also attributes; nested dicts (also inside lists/tuples) are converted on assignment."""

    def __init__(self, d=None, **kwargs):
        super().__init__()
        src = {} if d is None else dict(d)
        src.update(kwargs)
        for k, v in src.items():
            setattr(self, k, v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setattr__(self, name, value):
        value = self._wrap(value)
        super().__setattr__(name, value)
        super().__setitem__(name, value)

    __setitem__ = __setattr__

    def __delattr__(self, name):
        super().__delattr__(name)
        super().__delitem__(name)

    def update(self, other=None, **kwargs):
        src = {} if other is None else dict(other)
        src.update(kwargs)
        for k, v in src.items():
            setattr(self, k, v)

    def pop(self, k, *default):
        if hasattr(self, k) and k in self:
            super().__delattr__(k)
        return super().pop(k, *default)


def log_config_to_file(cfg, pre='cfg', logger=None):
    for key in cfg:
        val = cfg[key]
        if isinstance(val, EasyDict):
            logger.info('\n%s.%s = edict()' % (pre, key))
            log_config_to_file(val, pre='%s.%s' % (pre, key), logger=logger)
        else:
            logger.info('%s.%s: %s' % (pre, key, val))


def cfg_from_list(cfg_list, config):
    """`--set A.B.C value ...`: literal_eval the value, require the key to exist and the type to match (reference :16-48)."""
    assert len(cfg_list) % 2 == 0, 'expected KEY VALUE pairs'
    for dotted, raw in zip(cfg_list[0::2], cfg_list[1::2]):
        node = config
        parts = dotted.split('.')
        for sub in parts[:-1]:
            assert sub in node, 'NotFoundKey: %s' % sub
            node = node[sub]
        leaf = parts[-1]
        assert leaf in node, 'NotFoundKey: %s' % leaf
        try:
            value = literal_eval(raw)
        except Exception:
            value = raw
        current = node[leaf]
        if type(value) != type(current) and isinstance(current, EasyDict):
            for item in value.split(','):
                k, v = item.split(':')
                current[k] = type(current[k])(v)
        elif type(value) != type(current) and isinstance(current, list):
            node[leaf] = [type(current[0])(x) for x in value.split(',')]
        else:
            assert type(value) == type(current), 'type {} does not match original type {}'.format(type(value), type(current))
            node[leaf] = value


def _resolve_base(path, including_file=None):
    cands = [Path(path)]
    if including_file is not None:
        cands.append(Path(including_file).resolve().parent / path)
    tools_dir = Path(__file__).resolve().parent.parent / 'tools'
    cands.append(tools_dir / path)
    # '../tools/cfgs/x.yaml' style paths of the reference, relative to tools/
    cands.append(tools_dir / Path(*[p for p in Path(path).parts if p != '..'][1:])) if 'tools' in Path(path).parts else None
    for c in cands:
        if c is not None and c.is_file():
            return c
    raise FileNotFoundError('_BASE_CONFIG_ %s not found (tried %s)' % (path, [str(c) for c in cands if c is not None]))


def merge_new_config(config, new_config, _file=None):
    if '_BASE_CONFIG_' in new_config:
        base_path = _resolve_base(new_config['_BASE_CONFIG_'], _file)
        with open(base_path, 'r') as f:
            merge_new_config(config, yaml.safe_load(f), str(base_path))      # bases may chain (the reference allows one level)
    for key, val in new_config.items():
        if key == '_BASE_CONFIG_':
            config[key] = val
            continue
        if not isinstance(val, dict):
            config[key] = val
            continue
        if key not in config:
            config[key] = EasyDict()
        merge_new_config(config[key], val, _file)
    return config


def cfg_from_yaml_file(cfg_file, config):
    with open(cfg_file, 'r') as f:
        new_config = yaml.safe_load(f)
    merge_new_config(config=config, new_config=new_config, _file=cfg_file)
    return config


cfg = EasyDict()
cfg.ROOT_DIR = (Path(__file__).resolve().parent / '../').resolve()
cfg.LOCAL_RANK = 0
