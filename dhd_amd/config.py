"""Minimal loader for the reference's mmcv-style python configs, so that
projects/configs/DHD/*.py load unchanged: executes the file, resolves `_base_`
(base dicts merged recursively, child wins, `_delete_=True` replaces), supports `--cfg-options`
style overrides (tools/train.py:82-91,120-122).  The two base files the reference inherits from an
un-vendored mmdetection3d checkout are looked up by file name in dhd_amd/base_configs/ when the
relative path does not exist."""
import copy
import os
import types

_BASE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'base_configs')


class ConfigDict(dict):
    """dict with attribute access (mmcv ConfigDict behaviour: missing attribute -> AttributeError)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'ConfigDict' object has no attribute '{name}'") from None

    def __setattr__(self, name, value):
        self[name] = value


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_wrap(x) for x in v)
    return v


def _merge(base, child):
    out = copy.deepcopy(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != '_delete_'}
            out[k] = copy.deepcopy(v)
    return out


def _exec_file(path):
    ns = {'__file__': path, '__name__': '__dhd_cfg__'}
    with open(path) as f:
        exec(compile(f.read(), path, 'exec'), ns)
    return {k: v for k, v in ns.items()
            if not k.startswith('__') and not isinstance(v, (types.ModuleType, types.FunctionType, type))}


def _resolve_base(path, cfg_dir):
    cand = os.path.normpath(os.path.join(cfg_dir, path))
    if os.path.exists(cand):
        return cand
    bundled = os.path.join(_BASE_DIR, os.path.basename(path))
    if os.path.exists(bundled):
        return bundled
    raise FileNotFoundError(f'_base_ config {path!r} not found next to the config nor in {_BASE_DIR}')


def _load(path):
    path = os.path.abspath(path)
    cfg = _exec_file(path)
    bases = cfg.pop('_base_', [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        sub = _load(_resolve_base(b, os.path.dirname(path)))
        dup = set(merged) & set(sub)
        if dup:
            raise KeyError(f'duplicate keys in _base_ files: {sorted(dup)}')
        merged.update(sub)
    return _merge(merged, cfg)


class Config(ConfigDict):
    @classmethod
    def fromfile(cls, filename):
        c = cls(_wrap(_load(filename)))
        dict.__setitem__(c, 'filename', os.path.abspath(filename))
        return c

    def merge_from_dict(self, options):
        """`--cfg-options model.img_view_transformer.accelerate=True` style dotted overrides."""
        for key, val in options.items():
            node = self
            parts = key.split('.')
            for p in parts[:-1]:
                node = node[int(p)] if isinstance(node, list) else node.setdefault(p, ConfigDict())
            last = parts[-1]
            if isinstance(node, list):
                node[int(last)] = _wrap(val)
            else:
                node[last] = _wrap(val)
        return self
