"""Minimal stand-in for the mmcv/mmdet3d registries the reference's configs are written against
(`type='MGHS'` etc., projects/configs/DHD/DHD-S.py:42-155; registration by decorator as in
models/necks/lss_heightmap.py:12, mix.py:61).  Unknown `type` raises KeyError like mmcv."""
import copy


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, force=False):
        def deco(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._modules[key] = cls
            return cls
        return deco

    def get(self, key):
        return self._modules.get(key)

    def __contains__(self, key):
        return key in self._modules

    def build(self, cfg, **default_args):
        if cfg is None:
            return None
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise TypeError(f'cfg must be a dict with a "type" key, got {cfg!r}')
        args = copy.deepcopy(cfg)
        typ = args.pop('type')
        cls = self.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f'{typ} is not in the {self.name} registry')
        for k, v in default_args.items():
            args.setdefault(k, v)
        return cls(**args)


NECKS = Registry('neck')
BACKBONES = Registry('backbone')
HEADS = Registry('head')
DETECTORS = Registry('detector')
LOSSES = Registry('loss')
HOOKS = Registry('hook')


def build_neck(cfg):
    return NECKS.build(cfg)


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_detector(cfg):
    return DETECTORS.build(cfg)


def build_hook(cfg):
    """One entry of a config's `custom_hooks` list (DHD-S.py:272-278)."""
    return HOOKS.build(cfg)
