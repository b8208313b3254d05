"""Persistent half copies of the convolution / linear weights for autocast training.

Under `torch.autocast` every float32 weight is cast to the half type each time its layer runs: one 4-5 us launch per weight and
forward -- and once more in the recomputation of an activation-checkpointed stack (`with_cp=True` in the DHD configs), whose
autocast context is new.  In a DHD-S fp16 step that is ~240 launches, ~1.1 ms of 49 (profiles/r5-r6: 268 `float16_copy` launches
per step).  The casts all produce the same thing: the rounded image of a parameter that only changes in `optimizer.step()`.

`HalfWeightCache(model, dtype)` keeps ONE half copy per weight, refreshed by a multi-tensor copy after the optimizer step
(`refresh()`: two launches for the whole model, capturable), and routes the layers' forwards through it:

    y = conv(x, UseHalf(weight_fp32, weight_half), bias)

`UseHalf` returns the half copy (no kernel) and hands its gradient back to the float32 parameter as float32 -- the same node
autocast's own cast creates, so `.grad`, gradient clipping, `GradScaler`, DDP's bucket hooks and the optimizer see what they
see without the cache.  A routed layer's output is bit for bit the plain layer's under autocast (`tests/test_detector.py`).

Only plain `nn.Conv2d` / `nn.ConvTranspose2d` / `nn.Linear` forwards are routed (a subclass with its own `forward` keeps
autocast's path), only while autocast with the cache's dtype is active and the input is a GPU tensor, and only while the copy is
CURRENT -- it remembers the parameter tensor and version it was refreshed from, so an optimizer step without a refresh, a
`load_state_dict`, or a `deepcopy` of the model (EMA copies) fall back to autocast's own cast instead of using a stale or foreign
copy.  Parameters, buffers and state-dict keys are untouched: the copies are plain attributes.
The reference has no counterpart (mmcv's fp16 hook casts per step as autocast does); this is an execution detail of the training
loop, like `dhd_amd.graph.GraphedStep`."""
import torch
from torch import nn
import torch.nn.functional as F


class _UseHalf(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w32, w16):
        return w16.detach()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.float32), None


_PLAIN = {nn.Conv2d: nn.Conv2d.forward, nn.ConvTranspose2d: nn.ConvTranspose2d.forward, nn.Linear: nn.Linear.forward}


def _half_of(m, x):
    """The module's half weight if it is current, else None (-> the module's ordinary forward, i.e. autocast's own cast).
    Current = autocast is on with the copy's dtype, x is a GPU tensor, and the copy was refreshed from THIS parameter tensor in its
    present version: an optimizer step, load_state_dict or any other in-place update since the last refresh(), and a deepcopy of the
    module (another parameter tensor), all fall back to the ordinary path instead of computing with a stale or foreign copy."""
    h = m.__dict__.get('_dhd_w16')
    if h is None or not x.is_cuda or not torch.is_autocast_enabled() or torch.get_autocast_dtype('cuda') != h.dtype:
        return None
    w = m.weight
    if m.__dict__.get('_dhd_w16_key') != (w.data_ptr(), w._version):
        return None
    return _UseHalf.apply(w, h)


def _conv2d_forward(self, x):
    h = _half_of(self, x)
    return nn.Conv2d.forward(self, x) if h is None else self._conv_forward(x, h, self.bias)


def _conv_transpose2d_forward(self, x, output_size=None):
    h = None if (output_size is not None or self.padding_mode != 'zeros') else _half_of(self, x)
    if h is None:
        return nn.ConvTranspose2d.forward(self, x, output_size)
    return F.conv_transpose2d(x, h, self.bias, self.stride, self.padding, self.output_padding, self.groups, self.dilation)


def _linear_forward(self, x):
    h = _half_of(self, x)
    return nn.Linear.forward(self, x) if h is None else F.linear(x, h, self.bias)


_ROUTED = {nn.Conv2d: _conv2d_forward, nn.ConvTranspose2d: _conv_transpose2d_forward, nn.Linear: _linear_forward}


class HalfWeightCache:
    def __init__(self, model, dtype=torch.float16):
        import types
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError('HalfWeightCache: float16 or bfloat16')
        self.dtype = dtype
        self.modules = []
        seen = set()
        for m in model.modules():
            # exactly these classes, or subclasses that did not override forward; a subclass with its own forward keeps autocast's path
            base = next((b for b in _PLAIN if isinstance(m, b)), None)
            if base is None or type(m).forward is not _PLAIN[base] or 'forward' in m.__dict__:
                continue
            w = m.weight
            if w is None or w.dtype != torch.float32 or not w.is_cuda or id(w) in seen:
                continue
            seen.add(id(w))
            self.modules.append(m)
            m._dhd_w16 = torch.empty_like(w, dtype=dtype)      # preserve_format: channels_last weights stay so
            m._dhd_w16_key = None
            m.forward = types.MethodType(_ROUTED[base], m)     # a bound method: deepcopy re-binds it to the copy
        self.params = [m.weight for m in self.modules]
        self.halves = [m._dhd_w16 for m in self.modules]
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        """half copies <- parameters (after optimizer.step(), load_state_dict ...): one multi-tensor copy for the whole model."""
        if self.halves:
            torch._foreach_copy_(self.halves, self.params)
            for m in self.modules:
                w = m.weight
                m._dhd_w16_key = (w.data_ptr(), w._version)

    def remove(self):
        for m in self.modules:
            for k in ('forward', '_dhd_w16', '_dhd_w16_key'):
                m.__dict__.pop(k, None)
        self.modules, self.params, self.halves = [], [], []

    def __len__(self):
        return len(self.modules)
