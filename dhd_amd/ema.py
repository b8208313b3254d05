"""Weight EMA of the training loop and the reference's three runner hooks
(projects/mmdet3d_plugin/core/hook/ema.py, syncbncontrol.py, sequentialcontrol.py).

`ModelEMA.update` is the per-iteration cost: the reference touches every floating-point entry of
the state dict with two eager ops (ema.py:55-59, `v *= d; v += (1 - d) * m`), ~1000 launches per
iteration for DHD-S.  Here the float32 tensors on the GPU are covered by one chunk table and
updated by one launch of `dhd_ema_update` (csrc/ema.hip) with the same two roundings, so the result
is bit-identical to the reference's.  State on the CPU takes the reference's own torch expression
(host logic, used by the CPU tests); GPU state without the library raises."""
import math
import os
from copy import deepcopy

import numpy as np
import torch
from torch import nn

from . import _lib
from .registry import HOOKS

CHUNK = 1 << 16  # float32 elements per workgroup (256 KiB): ~2.3 k workgroups for DHD-S' 147 M values


def is_parallel(model):
    """core/hook/utils.py:7-13."""
    return isinstance(model, (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel))


def _inner(model):
    """The runner's model is an (MM)DataParallel-style wrapper around the detector, which may itself
    be wrapped once more (ema.py:39-40); a bare detector is accepted as well."""
    m = model.module if hasattr(model, 'module') and isinstance(getattr(model, 'module'), nn.Module) else model
    return m.module if is_parallel(m) else m


def _state_tensors(module):
    """The tensors of `module.state_dict()` in its order, without building the dict (the per-iteration host
    cost of ~800 entries): per module its parameters, then its persistent buffers, children after."""
    out = []
    for mod in module.modules():
        out.extend(p for p in mod._parameters.values() if p is not None)
        out.extend(b for n, b in mod._buffers.items() if b is not None and n not in mod._non_persistent_buffers_set)
    return out


class _ChunkTable:
    """Device-resident (ema address, model address, length) triples for all float32 GPU pairs."""

    def __init__(self, pairs, device):
        ea, ma, ln = [], [], []
        for e, m in pairs:
            n = e.numel()
            for off in range(0, n, CHUNK):
                ea.append(e.data_ptr() + 4 * off)
                ma.append(m.data_ptr() + 4 * off)
                ln.append(min(CHUNK, n - off))
        self.n = len(ln)
        self.ema_addr = torch.from_numpy(np.asarray(ea, dtype=np.uint64).view(np.int64)).to(device)
        self.model_addr = torch.from_numpy(np.asarray(ma, dtype=np.uint64).view(np.int64)).to(device)
        self.len = torch.tensor(ln, dtype=torch.int32, device=device)


class ModelEMA:
    """ema.py:19-59.  Keeps a moving average of everything in the model state dict (parameters and
    floating-point buffers); `decay` ramps up as decay * (1 - exp(-updates / 2000))."""

    events = None  # bench.py: a list that receives (start, end) HIP events right around the launch

    def __init__(self, model, decay=0.9999, updates=0):
        self.ema_model = deepcopy(model).eval()
        self.ema = _inner(self.ema_model)
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000))
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self._key, self._cpu, self._tables = None, [], {}
        self._decay_dev = {}     # device -> float32[2] {d, 1 - d}: read by launches captured into a HIP graph
        self.captured = False    # an update has been captured: advance() must run before every replay (GraphedStep does)

    def _set_decay_dev(self, d):
        pair = torch.tensor([d, 1.0 - d], dtype=torch.float32)   # 1 - d in double, both rounded to float32 (ema.py:58-59)
        for dev, t in self._decay_dev.items():
            t.copy_(pair, non_blocking=True)

    def advance(self):
        """Host side of ONE update whose kernel launch was captured into a HIP graph: the update count moves on and the decay of
        this update is placed where the captured launch reads it.  Call before every replay (dhd_amd.graph.GraphedStep does it
        for the ModelEMA objects it is given); without it a replay would repeat the decay and the count of the capture."""
        self.updates += 1
        self._set_decay_dev(self.decay(self.updates))

    def _plan(self, ours, theirs):
        """Sort the state into CPU pairs and per-device chunk tables (redone only when a tensor moved)."""
        cpu, by_dev = [], {}
        for v, m in zip(ours, theirs):
            if not v.dtype.is_floating_point or not v.numel():
                continue
            v, m = v.detach(), m.detach()
            if not v.is_cuda:
                cpu.append((v, m))
                continue
            # the kernel walks the two buffers linearly: any dense layout will do as long as both tensors share it
            # (contiguous, or e.g. both channels_last)
            dense = (v.is_contiguous() and m.is_contiguous()) or (
                v.stride() == m.stride() and v.is_contiguous(memory_format=torch.channels_last)
                and m.is_contiguous(memory_format=torch.channels_last)) if v.dim() == 4 else (v.is_contiguous() and m.is_contiguous())
            if v.dtype != torch.float32 or m.dtype != torch.float32 or not dense:
                raise _lib.DhdError('ModelEMA: GPU state must be dense float32 with one layout in the model and its EMA copy '
                                    '(the reference keeps the EMA in FP32)')
            if v.device != m.device or v.shape != m.shape:
                raise _lib.DhdError('ModelEMA: EMA and model state differ in device or shape')
            by_dev.setdefault(v.device, []).append((v, m))
        if by_dev:
            _lib.load()  # GPU state without the library: fail here, not silently on another path
        return cpu, {dev: _ChunkTable(plist, dev) for dev, plist in by_dev.items()}

    def update(self, trainer, model):
        with torch.no_grad():
            capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
            if capturing:
                # recorded, not executed: the count and the decay belong to the replays (advance()); kernel arguments would
                # be frozen at their capture-time values, so the launch reads {d, 1 - d} from device memory
                self.captured = True
                d = self.decay(self.updates + 1)
            else:
                self.updates += 1
                d = self.decay(self.updates)
            ours, theirs = _state_tensors(self.ema), _state_tensors(model.module if is_parallel(model) else model)
            if len(ours) != len(theirs):
                raise KeyError('ModelEMA: the model state no longer matches the EMA copy')
            key = [t.data_ptr() for t in ours] + [t.data_ptr() for t in theirs]
            if key != self._key:    # first call, or storage re-allocated (load_state_dict copies in place)
                self._cpu, self._tables = self._plan(ours, theirs)
                self._key = key
            for v, m in self._cpu:  # the reference's expression (ema.py:58-59)
                v *= d
                v += (1.0 - d) * m
            for dev, tab in self._tables.items():
                with torch.cuda.device(dev):
                    if self.events is not None:
                        marks = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                        marks[0].record()
                    if capturing:
                        if dev not in self._decay_dev:
                            raise _lib.DhdError('ModelEMA: run one eager update (a warm-up step) before capturing it into a graph')
                        _lib.check(_lib.load().dhd_ema_update_dev(_lib.ptr(tab.ema_addr), _lib.ptr(tab.model_addr), _lib.ptr(tab.len), tab.n,
                                                                  _lib.ptr(self._decay_dev[dev]), _lib.stream_ptr(dev)), 'dhd_ema_update_dev')
                    else:
                        if dev not in self._decay_dev:   # allocated outside any capture
                            self._decay_dev[dev] = torch.zeros(2, dtype=torch.float32, device=dev)
                        # Python computes 1 - d in double and torch rounds both scalars to float32 (ema.py:58-59)
                        _lib.check(_lib.load().dhd_ema_update(_lib.ptr(tab.ema_addr), _lib.ptr(tab.model_addr), _lib.ptr(tab.len), tab.n,
                                                              float(d), float(1.0 - d), _lib.stream_ptr(dev)), 'dhd_ema_update')
                    if self.events is not None:
                        marks[1].record()
                        self.events.append(marks)


class Hook:
    """The subset of mmcv.runner.Hook's stages the three hooks use."""

    def before_run(self, runner):
        pass

    def before_train_epoch(self, runner):
        pass

    def after_train_iter(self, runner):
        pass

    def after_train_epoch(self, runner):
        pass


@HOOKS.register_module()
class MEGVIIEMAHook(Hook):
    """ema.py:62-117 (DHD-S.py:272-278: init_updates=10560)."""

    def __init__(self, init_updates=0, decay=0.9990, resume=None, priority='NORMAL'):
        self.init_updates = init_updates
        self.resume = resume
        self.decay = decay
        self.priority = priority

    def before_run(self, runner):
        # process groups cannot be deep-copied: detach them from the SyncBatchNorm layers for the copy
        held = [(m, m.process_group) for m in runner.model.modules() if isinstance(m, nn.SyncBatchNorm)]
        for m, _ in held:
            m.process_group = None
        runner.ema_model = ModelEMA(runner.model, self.decay)
        for m, group in held:
            m.process_group = group
        runner.ema_model.updates = self.init_updates
        if self.resume is not None:
            runner.logger.info(f'resume ema checkpoint from {self.resume}')
            cpt = torch.load(self.resume, map_location='cpu')
            runner.ema_model.ema.load_state_dict(cpt['state_dict'], strict=False)
            runner.ema_model.updates = cpt['updates']

    def after_train_iter(self, runner):
        runner.ema_model.update(runner, runner.model.module)

    def after_train_epoch(self, runner):
        self.save_checkpoint(runner)

    def save_checkpoint(self, runner):
        if getattr(runner, 'rank', 0) != 0:  # @master_only
            return
        path = os.path.join(runner.work_dir, f'epoch_{runner.epoch + 1}_ema.pth')
        torch.save({'epoch': runner.epoch, 'state_dict': runner.ema_model.ema.state_dict(),
                    'updates': runner.ema_model.updates}, path)
        runner.logger.info(f'Saving ema checkpoint at {path}')


@HOOKS.register_module()
class SyncbnControlHook(Hook):
    """syncbncontrol.py:9-33 (DHD-L.py): BatchNorm -> SyncBatchNorm from `syncbn_start_epoch` on."""

    def __init__(self, syncbn_start_epoch=1, priority='NORMAL'):
        self.is_syncbn = False
        self.syncbn_start_epoch = syncbn_start_epoch
        self.priority = priority

    def cvt_syncbn(self, runner):
        holder = runner.model.module if is_parallel(runner.model.module) else runner.model
        holder.module = nn.SyncBatchNorm.convert_sync_batchnorm(holder.module, process_group=None)

    def before_train_epoch(self, runner):
        if runner.epoch >= self.syncbn_start_epoch and not self.is_syncbn:
            print('start use syncbn')
            self.cvt_syncbn(runner)
            self.is_syncbn = True


@HOOKS.register_module()
class SequentialControlHook(Hook):
    """sequentialcontrol.py:8-27: the temporal branch (`with_prev`) is switched on after
    `temporal_start_epoch`."""

    def __init__(self, temporal_start_epoch=1, priority='NORMAL'):
        self.temporal_start_epoch = temporal_start_epoch
        self.priority = priority

    def set_temporal_flag(self, runner, flag):
        _inner(runner.model).with_prev = flag

    def before_run(self, runner):
        self.set_temporal_flag(runner, False)

    def before_train_epoch(self, runner):
        if runner.epoch > self.temporal_start_epoch:
            self.set_temporal_flag(runner, True)
