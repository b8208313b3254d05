"""Checkpoint ingestion for the preserved state-dict key names.

The reference's configs name weights to start from -- `load_from="ckpts/bevdet-r50-cbgs.pth"` (DHD-S.py:280),
`pretrained='torchvision://resnet50'` (DHD-S.py:53) -- which mmcv's runner / mmdet's backbones load with
`mmcv.runner.load_checkpoint` (un-vendored).  Its behaviour for a LOCAL file is restated here: the file is a `torch.save`d
dict, either the state dict itself or `{'meta': ..., 'state_dict': ..., 'optimizer': ...}`; a leading `module.` (a model saved
from inside (MM)DataParallel) is stripped; loading is non-strict by default and reports what did not match instead of
failing; size mismatches are reported, not loaded.  Schemes that need a network or an absent package (`torchvision://`,
`http(s)://`, `open-mmlab://`) raise: there is no egress on the target boxes.
"""
import os
import re
import warnings

import torch

__all__ = ['load_checkpoint', 'load_state_dict', 'save_checkpoint']


def _read(filename, map_location='cpu', trusted=False):
    if not isinstance(filename, (str, os.PathLike)):
        raise TypeError('checkpoint must be a path')
    name = str(filename)
    if re.match(r'^(torchvision|open-mmlab|openmmlab|mmcls|modelzoo|https?|s3|pavi)://', name):
        raise IOError(f'{name}: only local checkpoint files can be loaded here (no network, no torchvision); download the file and pass its path')
    if not os.path.isfile(name):
        raise IOError(f'{name} is not a checkpoint file')
    import pickle
    try:
        return torch.load(name, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as err:
        # checkpoints written by mmcv carry a `meta` dict that may hold python objects the safe unpickler refuses.  Running
        # the full unpickler executes whatever the file says, so it is the caller's explicit decision (`trusted=True`), never
        # a silent retry: `load_from` / `pretrained` files are third-party downloads.
        if not trusted:
            raise RuntimeError(f'{name}: refused by the weights-only unpickler ({err}); if the file comes from a source you trust, '
                               'pass trusted=True to load it with the full (code-executing) unpickler') from err
        warnings.warn(f'{name}: loading with the full unpickler (trusted=True)', stacklevel=3)
        return torch.load(name, map_location=map_location, weights_only=False)


def load_state_dict(module, state_dict, strict=False, revise_keys=((r'^module\.', ''),)):
    """Copy `state_dict` into `module`.  Returns (missing_keys, unexpected_keys, mismatched) where mismatched lists
    (key, checkpoint shape, model shape); with strict=True any of them raises."""
    sd = {}
    for k, v in state_dict.items():
        for pat, rep in revise_keys:
            k = re.sub(pat, rep, k)
        sd[k] = v
    own = module.state_dict()
    mismatched = [(k, tuple(v.shape), tuple(own[k].shape)) for k, v in sd.items() if k in own and hasattr(v, 'shape') and tuple(v.shape) != tuple(own[k].shape)]
    for k, _, _ in mismatched:
        sd.pop(k)
    res = module.load_state_dict(sd, strict=False)
    missing = [k for k in res.missing_keys if 'num_batches_tracked' not in k]   # absent from pre-0.4.1 checkpoints, as mmcv ignores them
    unexpected = list(res.unexpected_keys)
    if strict and (missing or unexpected or mismatched):
        raise RuntimeError(f'checkpoint does not match the model: missing {missing[:5]}..., unexpected {unexpected[:5]}..., '
                           f'size mismatch {mismatched[:5]}...')
    return missing, unexpected, mismatched


def load_checkpoint(model, filename, map_location='cpu', strict=False, revise_keys=((r'^module\.', ''),), prefix=None, quiet=False,
                    trusted=False):
    """mmcv.runner.load_checkpoint for a local file.  `prefix`: load only the entries under `prefix.` with the prefix removed
    (e.g. 'img_backbone' to initialise a backbone from a detector checkpoint).  `trusted`: allow the full unpickler for files
    the weights-only one refuses (see _read).  Returns the checkpoint dict; `ckpt['_load_report']` says what matched."""
    ckpt = _read(filename, map_location, trusted)
    if not isinstance(ckpt, dict):
        raise RuntimeError(f'No state_dict found in checkpoint file {filename}')
    state = ckpt['state_dict'] if 'state_dict' in ckpt else ckpt
    if prefix:
        p = prefix.rstrip('.') + '.'
        state = {k[len(p):]: v for k, v in ((re.sub(r'^module\.', '', k), v) for k, v in state.items()) if k.startswith(p)}
        if not state:
            raise RuntimeError(f'{filename} has no entries under {prefix!r}')
    target = model.module if hasattr(model, 'module') and isinstance(model.module, torch.nn.Module) else model
    missing, unexpected, mismatched = load_state_dict(target, state, strict, revise_keys)
    if not quiet and (missing or unexpected or mismatched):
        warnings.warn(f'load_checkpoint({filename}): {len(missing)} missing, {len(unexpected)} unexpected, {len(mismatched)} size-mismatched '
                      f'entries; missing e.g. {missing[:3]}, unexpected e.g. {unexpected[:3]}', stacklevel=2)
    ckpt['_load_report'] = dict(missing=missing, unexpected=unexpected, mismatched=mismatched,
                                loaded=len(state) - len(unexpected) - len(mismatched), model_entries=len(target.state_dict()))
    return ckpt


def save_checkpoint(model, filename, meta=None, optimizer=None):
    """The layout mmcv's CheckpointHook writes (and MEGVIIEMAHook for `epoch_N_ema.pth`, core/hook/ema.py:106-117):
    {'meta': ..., 'state_dict': ..., ['optimizer': ...]}, weights on the CPU."""
    target = model.module if hasattr(model, 'module') and isinstance(model.module, torch.nn.Module) else model
    ckpt = dict(meta=dict(meta or {}), state_dict={k: v.detach().cpu() for k, v in target.state_dict().items()})
    if optimizer is not None:
        ckpt['optimizer'] = optimizer.state_dict()
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    torch.save(ckpt, filename)
