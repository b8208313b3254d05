"""Training-mode BatchNorm2d of the dense callers as a HIP operator (csrc/batchnorm.hip).

`BatchNorm2d` is `torch.nn.BatchNorm2d` (same parameters, buffers and state-dict keys) whose training
forward/backward on a GPU run as three streaming launches each instead of MIOpen's kernels (which move these
activations at 0.7-3.6 TB/s: 11 % of a DHD-S training step).  Everything else -- eval mode, CPU tensors, shapes
the kernels do not cover (H*W not a multiple of 4 / 8 elements) -- takes the parent's path unchanged."""
import os

import torch
from torch import nn

from . import _lib

_DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


class _BNTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, factor, eps):
        n, c = x.shape[:2]
        hw = x[0, 0].numel()
        lib = _lib.load()
        dev = x.device
        with torch.cuda.device(dev):
            y = torch.empty_like(x)
            mean = torch.empty(c, dtype=torch.float32, device=dev)
            rstd = torch.empty_like(mean)
            ws = torch.empty(lib.dhd_bn_workspace_bytes(n, c, hw), dtype=torch.uint8, device=dev)
            _lib.check(lib.dhd_bn_train_forward(_lib.ptr(x), _DTYPES[x.dtype], n, c, hw, _lib.ptr(weight), _lib.ptr(bias),
                                                _lib.ptr(running_mean), _lib.ptr(running_var), factor, eps, _lib.ptr(y), _lib.ptr(mean),
                                                _lib.ptr(rstd), _lib.ptr(ws), _lib.stream_ptr(dev)), 'dhd_bn_train_forward')
        ctx.save_for_backward(x, weight, mean, rstd)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, mean, rstd = ctx.saved_tensors
        n, c = x.shape[:2]
        hw = x[0, 0].numel()
        lib = _lib.load()
        dev = x.device
        gy = gy.contiguous()
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        with torch.cuda.device(dev):
            gx = torch.empty_like(x)
            dgamma = torch.empty(c, dtype=torch.float32, device=dev)
            dbeta = torch.empty_like(dgamma)
            ws = torch.empty(lib.dhd_bn_workspace_bytes(n, c, hw), dtype=torch.uint8, device=dev)
            _lib.check(lib.dhd_bn_train_backward(_lib.ptr(x), _lib.ptr(gy), _DTYPES[x.dtype], n, c, hw, _lib.ptr(weight), _lib.ptr(mean),
                                                 _lib.ptr(rstd), _lib.ptr(gx), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ws),
                                                 _lib.stream_ptr(dev)), 'dhd_bn_train_backward')
        gw = dgamma.to(weight.dtype) if weight is not None and ctx.needs_input_grad[1] else None
        gb = dbeta if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb, None, None, None, None


BN_RELU, BN_ADD = 1, 2   # include/dhd_amd.h DHD_BN_RELU / DHD_BN_ADD


def _is_nhwc(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()


class _BNTrainNHWC(torch.autograd.Function):
    """Training BatchNorm of a channels_last tensor with the ReLU (and the residual add) that follows it fused in
    (csrc/batchnorm.hip, second half; include/dhd_amd.h section 9b)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, factor, eps, flags, residual):
        n, c, h, w = x.shape
        rows = n * h * w
        lib = _lib.load()
        dev = x.device
        with torch.cuda.device(dev):
            y = torch.empty_like(x)     # keeps the channels_last strides
            mean = torch.empty(c, dtype=torch.float32, device=dev)
            rstd = torch.empty_like(mean)
            affine = torch.empty(2 * c, dtype=torch.float32, device=dev)
            ws = torch.empty(lib.dhd_bn_nhwc_workspace_bytes(rows, c), dtype=torch.uint8, device=dev)
            _lib.check(lib.dhd_bn_nhwc_train_forward(_lib.ptr(x), _lib.ptr(residual), _DTYPES[x.dtype], rows, c, flags, _lib.ptr(weight),
                                                     _lib.ptr(bias), _lib.ptr(running_mean), _lib.ptr(running_var), factor, eps, _lib.ptr(y),
                                                     _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(affine), _lib.ptr(ws), _lib.stream_ptr(dev)),
                       'dhd_bn_nhwc_train_forward')
        ctx.save_for_backward(x, weight, mean, rstd, affine, y if flags & BN_ADD else None)
        ctx.has_bias, ctx.flags = bias is not None, flags
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, mean, rstd, affine, y = ctx.saved_tensors
        n, c, h, w = x.shape
        rows = n * h * w
        lib = _lib.load()
        dev = x.device
        gy = gy.contiguous(memory_format=torch.channels_last)
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        want_res = bool(ctx.flags & BN_ADD) and ctx.needs_input_grad[8]
        with torch.cuda.device(dev):
            gx = torch.empty_like(x)
            gres = torch.empty_like(x) if want_res else None
            dgamma = torch.empty(c, dtype=torch.float32, device=dev)
            dbeta = torch.empty_like(dgamma)
            ws = torch.empty(lib.dhd_bn_nhwc_workspace_bytes(rows, c), dtype=torch.uint8, device=dev)
            _lib.check(lib.dhd_bn_nhwc_train_backward(_lib.ptr(x), _lib.ptr(y), _lib.ptr(gy), _DTYPES[x.dtype], rows, c, ctx.flags,
                                                      _lib.ptr(weight), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(affine), _lib.ptr(gx),
                                                      _lib.ptr(gres), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ws), _lib.stream_ptr(dev)),
                       'dhd_bn_nhwc_train_backward')
        gw = dgamma.to(weight.dtype) if weight is not None and ctx.needs_input_grad[1] else None
        gb = dbeta if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb, None, None, None, None, None, gres


class BatchNorm2d(nn.BatchNorm2d):
    """Drop-in for nn.BatchNorm2d; `use_hip = False` switches the operator off.

    Measured on MI355X (experiments/bn_native_vs_miopen.py, forward + backward): MIOpen parallelises over channels
    and collapses on few-channel, large-plane tensors -- (24, 64, 128, 352) float16: 922 us against 247 us here --
    while for many-channel tensors it is on par and this Python-level operator costs more host time per call
    (~0.2 ms).  Hence the operator only takes tensors of at least `min_numel` elements with at most
    `max_channels` channels, or of at least `big_numel` elements (`DHD_BN_ROUTING=min,max_c,big` overrides; lower
    thresholds measured slower end to end: 78.1 / 78.9 / 80.4 ms per DHD-S step for 16M,128,64M / 4M,128,32M /
    1M,256,16M)."""

    use_hip = True
    # (round 5: re-measured UNDER THE HIP GRAPH, where the operator's host time does not count -- experiments/e2e_bn_routing_ab.sh,
    # DHD-S fp16 step, two alternating rounds: 16M,128,64M 67.85 / 67.69 ms, 4M,128,32M 67.06 / 67.06, 1M,256,16M 67.31 / 67.35,
    # 256k,4096,4M 68.55 / 68.57, everything 68.49 / 68.50)
    _DEFAULT_ROUTING = (1 << 22, 128, 1 << 25)
    _routing = None
    # `num_batches_tracked += 1` is one 5 us launch per layer and step (191 of them in a DHD-S step: 1.3 % of it).  With a fixed
    # momentum the counter is pure bookkeeping, so a caller that owns the step (detector.DHD) may set defer_counter on its layers:
    # forward then only counts its calls on the host and `flush_counters(model)` adds them in ONE multi-tensor launch per step.
    defer_counter = False
    _pending = 0

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        # `_pending` is host state: whoever reads the counter through state_dict() (checkpoints, EMA copies, load_state_dict of a
        # deepcopy) must see it flushed, also when the owner of the step did not call flush_counters (a sub-module trained on its
        # own, forward_train called directly, a checkpointed recompute that counted after the step's flush) -- ADVICE r5
        self.register_state_dict_pre_hook(BatchNorm2d._flush_own)

    @staticmethod
    def _flush_own(module, prefix, keep_vars):
        t = module.num_batches_tracked
        if module._pending and t is not None and not (t.is_cuda and torch.cuda.is_current_stream_capturing()):
            module.num_batches_tracked.add_(module._pending)
            module._pending = 0

    @classmethod
    def routing(cls):
        """(min_numel, max_channels, big_numel); DHD_BN_ROUTING is read at first use, a malformed value falls back
        to the defaults with a warning instead of breaking `import dhd_amd`."""
        if cls._routing is None:
            cls._routing = cls._DEFAULT_ROUTING
            env = os.environ.get('DHD_BN_ROUTING')
            if env:
                try:
                    vals = tuple(int(v) for v in env.split(','))
                    if len(vals) != 3:
                        raise ValueError(env)
                    cls._routing = vals
                except ValueError:
                    import warnings
                    warnings.warn(f'DHD_BN_ROUTING={env!r} is not "min_numel,max_channels,big_numel"; using the defaults')
        return cls._routing

    def _hip_ok(self, x, force=False):
        if not (self.use_hip and self.training and x.is_cuda and x.dim() == 4 and x.dtype in _DTYPES and x.numel() > 0):
            return False
        if not force and not x.is_contiguous():   # channels_last activations stay with the layout-preserving library path
            return False
        if x.data_ptr() % 16:                     # the kernels move 16-byte vectors: a view at an odd storage offset goes to torch
            return False
        min_numel, max_channels, big_numel = self.routing()
        if not force and not ((x.numel() >= min_numel and x.shape[1] <= max_channels) or x.numel() >= big_numel):
            return False
        if self.weight is not None and (self.weight.dtype != torch.float32 or (self.bias is not None and self.bias.dtype != torch.float32)):
            return False
        if self.track_running_stats and self.running_mean.dtype != torch.float32:
            return False
        n, c = x.shape[:2]
        return bool(_lib.load().dhd_bn_supported(_DTYPES[x.dtype], n, c, x[0, 0].numel()))

    use_nhwc = not os.environ.get('DHD_BN_NO_NHWC')   # A/B switch: channels_last tensors go to the library's kernels

    def _nhwc_ok(self, x):
        if not (self.use_hip and self.use_nhwc and self.training and x.is_cuda and x.dtype in _DTYPES and x.numel() > 0 and _is_nhwc(x)
                and x.data_ptr() % 16 == 0):      # 16-byte vector loads / stores
            return False
        if self.weight is not None and (self.weight.dtype != torch.float32 or (self.bias is not None and self.bias.dtype != torch.float32)):
            return False
        if self.track_running_stats and self.running_mean.dtype != torch.float32:
            return False
        n, c, h, w = x.shape
        return bool(_lib.load().dhd_bn_nhwc_supported(_DTYPES[x.dtype], n * h * w, c))

    def forward(self, x, relu=False, residual=None):
        """`relu` / `residual`: the caller's `relu(bn(x))` or `relu(bn(x) + residual)` (resnet.py:282-300, mmcv's ConvModule) in
        one call; on channels_last GPU tensors in training they are fused into the normalisation's kernels, everywhere else they
        are applied with torch operators after it."""
        defer = (self.defer_counter and self.training and self.track_running_stats and self.momentum is not None
                 and self.num_batches_tracked is not None)
        nhwc = self._nhwc_ok(x)
        if not nhwc and not self._hip_ok(x):
            if defer:   # nn.BatchNorm2d.forward in training mode with running statistics, minus the counter launch
                self._check_input_dim(x)
                self._pending += 1
                y = nn.functional.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, True, self.momentum, self.eps)
            else:
                y = super().forward(x)
            return self._tail(y, relu, residual)
        self._check_input_dim(x)
        factor = 0.0 if self.momentum is None else self.momentum
        rm = rv = None
        if self.track_running_stats:
            rm, rv = self.running_mean, self.running_var
            if defer:
                self._pending += 1
            elif self.num_batches_tracked is not None:
                self.num_batches_tracked.add_(1)
                if self.momentum is None:
                    factor = 1.0 / float(self.num_batches_tracked)
        if nhwc:
            fuse_add = (residual is not None and residual.shape == x.shape and residual.dtype == x.dtype and _is_nhwc(residual)
                        and residual.data_ptr() % 16 == 0)
            flags = BN_ADD if fuse_add else (BN_RELU if relu and residual is None else 0)
            y = _BNTrainNHWC.apply(x, self.weight, self.bias, rm, rv, float(factor), float(self.eps), flags, residual if fuse_add else None)
            return y if flags or not (relu or residual is not None) else self._tail(y, relu, residual)
        return self._tail(_BNTrain.apply(x.contiguous(), self.weight, self.bias, rm, rv, float(factor), float(self.eps)), relu, residual)

    @staticmethod
    def _tail(y, relu, residual):
        if residual is not None:
            y = y + residual
        return torch.relu_(y) if (relu or residual is not None) else y


def bn_act(bn, x, relu=False, residual=None):
    """`relu(bn(x))` / `relu(bn(x) + residual)` / `bn(x)` for ANY normalisation module: this file's BatchNorm2d takes the
    element-wise tail along (fused on channels_last GPU tensors), every other module -- nn.SyncBatchNorm after
    `convert_sync_batchnorm` (SyncbnControlHook, DHD-L.py:308-311), a plain nn.BatchNorm2d, GroupNorm -- is called as it is and
    the tail applied with torch operators."""
    if isinstance(bn, BatchNorm2d):
        return bn(x, relu=relu, residual=residual)
    return BatchNorm2d._tail(bn(x), relu, residual)


def defer_counters(model, on=True):
    """Switch the per-layer `num_batches_tracked` launches of every dhd_amd BatchNorm2d under `model` to host-side counting
    (see BatchNorm2d.defer_counter); the owner of the step then calls flush_counters(model) once per step."""
    for m in model.modules():
        if isinstance(m, BatchNorm2d):
            m.defer_counter = bool(on)


def flush_counters(model):
    """num_batches_tracked += (calls since the last flush) for every deferring layer, in one multi-tensor launch.  Inside a HIP
    graph the captured increments are those of the captured step -- the same every replay."""
    ts, ns = [], []
    for m in model.modules():
        if isinstance(m, BatchNorm2d) and m._pending:
            ts.append(m.num_batches_tracked)
            ns.append(m._pending)
            m._pending = 0
    if ts:
        torch._foreach_add_(ts, ns)
