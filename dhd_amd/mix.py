"""SFA (dual-feature aggregation) with the reference's registry name, constructor and state-dict
keys (projects/mmdet3d_plugin/models/necks/mix.py: channel_spatial_stage :8-59, SFA :61-90).

The attention stage is memory-bound: in eager PyTorch it is one reduction plus ~6 element-wise
passes over (B,256..512,200,200) tensors.  Here the reduction and the two convex blends are HIP
kernels (csrc/sfa.hip) driven by ONE autograd node, so forward touches x three times and backward
writes dL/dx into a single buffer instead of summing three full-size partial gradients.  The tiny
fc and the two 1x1 conv + BatchNorm layers in between stay ordinary PyTorch modules (dense, MFMA
library path); their parameter gradients are produced by replaying their own sub-graphs inside the
node's backward.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .batchnorm import BatchNorm2d
from .registry import NECKS
from .trace import traced


def _call(fn_name, *args):
    lib = _lib.load()
    _lib.check(getattr(lib, fn_name)(*args), fn_name)


class _AttentionStage(torch.autograd.Function):
    """x (B,2C,H,W) -> x_fuse (B,C,H,W); `stage` owns fc / spacial_leanring."""

    @staticmethod
    @traced('dhd.sfa.blend_stage.forward')
    def forward(ctx, x, stage, *params):
        x = _lib.require_gpu_tensor(x.contiguous(), torch.float32, 'SFA input')  # cast happens outside the node
        if x.data_ptr() % 16:
            x = x.clone()
        b, c2, h, w = x.shape
        c, hw = c2 // 2, h * w
        dev = x.device
        st = _lib.stream_ptr(dev)
        # Function.forward runs with grad mode off; the inner modules are recorded explicitly
        build_graph = any(ctx.needs_input_grad)
        with torch.cuda.device(dev):
            s = torch.empty((b, c2), dtype=torch.float32, device=dev)
            _call('dhd_sfa_channel_mean', _lib.ptr(x), _lib.ptr(s), b, c2, hw, st)
            with torch.set_grad_enabled(build_graph):
                s_in = s.requires_grad_(build_graph)
                a1 = stage.fc(s_in)  # (B, C), post-sigmoid (mix.py:43)
            a1d = a1.detach().float().contiguous()  # the inner modules may run under autocast (bf16/fp16)
            u = torch.empty((b, c, h, w), dtype=torch.float32, device=dev)
            _call('dhd_sfa_blend1', _lib.ptr(x), _lib.ptr(a1d), _lib.ptr(u), b, c, hw, st)
            with torch.set_grad_enabled(build_graph):
                u_in = u.requires_grad_(build_graph)
                s2 = stage.spacial_leanring(u_in)  # pre-sigmoid attention_2 (mix.py:51)
            s2d = s2.detach().float().contiguous()
            out = torch.empty((b, c, h, w), dtype=torch.float32, device=dev)
            _call('dhd_sfa_blend2', _lib.ptr(x), _lib.ptr(a1d), _lib.ptr(s2d), _lib.ptr(out), b, c, hw, st)
        if build_graph:
            ctx.inner = (s_in, a1, u_in, s2)
            ctx.fc_params = [p for p in stage.fc.parameters() if p.requires_grad]
            ctx.sp_params = [p for p in stage.spacial_leanring.parameters() if p.requires_grad]
            ctx.save_for_backward(x, a1d, s2d)
            ctx.dims = (b, c, hw)
        return out

    @staticmethod
    @traced('dhd.sfa.blend_stage.backward')
    def backward(ctx, go):
        x, a1d, s2d = ctx.saved_tensors
        s_in, a1, u_in, s2 = ctx.inner
        b, c, hw = ctx.dims
        dev = x.device
        st = _lib.stream_ptr(dev)
        go = go.float().contiguous()
        with torch.cuda.device(dev):
            gx = torch.empty_like(x)
            gs2 = torch.empty_like(s2d)
            ga1 = torch.empty((b, c), dtype=torch.float32, device=dev)
            _call('dhd_sfa_blend2_backward', _lib.ptr(x), _lib.ptr(a1d), _lib.ptr(s2d), _lib.ptr(go), _lib.ptr(gx),
                  _lib.ptr(gs2), _lib.ptr(ga1), b, c, hw, st)
            # 1x1 conv / BN branch: parameter grads accumulate into .grad, dL/du comes back in u_in.grad
            torch.autograd.backward([s2], [gs2.to(s2.dtype)], inputs=[u_in] + ctx.sp_params)
            gu = u_in.grad.float().contiguous()
            u_in.grad = None
            _call('dhd_sfa_blend1_backward', _lib.ptr(x), _lib.ptr(a1d), _lib.ptr(gu), _lib.ptr(gx), _lib.ptr(ga1),
                  b, c, hw, st)
            torch.autograd.backward([a1], [ga1.to(a1.dtype)], inputs=[s_in] + ctx.fc_params)
            gs = s_in.grad.float().contiguous()
            s_in.grad = None
            _call('dhd_sfa_mean_backward', _lib.ptr(gs), _lib.ptr(gx), b, 2 * c, hw, st)
        ctx.inner = None
        return (gx, None) + (None,) * len(ctx.needs_input_grad[2:])


def _stage_scratch(dev, nbytes):
    """Reusable scratch of the fused stage: grow-only, one buffer per (device, stream); outgrown buffers stay alive
    (a captured HIP graph may hold their address) -- mghs_op.scratch_pool."""
    from .mghs_op import scratch_pool
    return scratch_pool.get(dev, nbytes, 'sfa')


_GEMM_ENV = {'0': 'f32', '1': 'bf16x6', '3': 'bf16x3'}   # DHD_SFA_GEMM_MODE also accepts round 2's numbers


def default_gemm():
    """GEMM precision of stages built without an explicit `gemm`: the environment variable DHD_SFA_GEMM_MODE
    (bf16x3 | bf16x6 | f32), else the library default (bf16x3: include/dhd_amd.h, dhd_sfa_weights.gemm)."""
    import os
    e = os.environ.get('DHD_SFA_GEMM_MODE', '').strip().lower()
    e = _GEMM_ENV.get(e, e)
    return e if e in _lib.SFA_GEMM else 'default'


def _bn_momentum(bn, training):
    """(update factor nn.BatchNorm2d would use for this call, device address of num_batches_tracked for the operator to
    increment or None) -- torch/nn/modules/batchnorm.py: forward.  With a fixed momentum the counter is pure bookkeeping
    and the stage operator increments it itself; momentum=None (cumulative average) needs its value on the host."""
    if not (training and bn.track_running_stats):
        return 0.0, None
    if bn.momentum is None:
        bn.num_batches_tracked.add_(1)
        return 1.0 / float(bn.num_batches_tracked), None
    nbt = bn.num_batches_tracked
    if nbt is not None and nbt.is_cuda and nbt.dtype == torch.int64:
        return float(bn.momentum), nbt.data_ptr()
    if nbt is not None:
        nbt.add_(1)
    return float(bn.momentum), None


_STAGE_PARAMS = ('fc1_w', 'fc1_b', 'fc2_w', 'fc2_b', 'conv1_w', 'conv1_b', 'bn1_w', 'bn1_b',
                 'conv2_w', 'conv2_b', 'bn2_w', 'bn2_b')


def _stage_params(stage):
    sp = stage.spacial_leanring
    return (stage.fc[0].weight, stage.fc[0].bias, stage.fc[2].weight, stage.fc[2].bias,
            sp[0].weight, sp[0].bias, sp[1].weight, sp[1].bias, sp[3].weight, sp[3].bias, sp[4].weight, sp[4].bias)


class _FusedStage(torch.autograd.Function):
    """The whole stage in libdhd_amd.so (dhd_sfa_stage_forward/backward): the 1x1 convolutions on
    the f32 MFMA with the blends, BatchNorm and ReLU fused into their operand paths."""

    @staticmethod
    @traced('dhd.sfa.stage.forward')
    def forward(ctx, x, stage, *params):
        # A half x (a caller inside an autocast region: the concatenated encoder outputs) is widened once for the operator, whose
        # arithmetic, saved tensors and parameters are float32; the stage's result then leaves in x's dtype (dhd_sfa_weights.
        # io_dtype: rounded to nearest even, what the next convolution's cast would make of a float32 result), the gradient comes
        # back in that dtype and the input gradient is returned in it -- no float32 round trips of (B,C,H,W) / (B,2C,H,W) tensors
        io_dtype = x.dtype if x.dtype in (torch.float16, torch.bfloat16) else torch.float32
        b, c2, h, w = x.shape
        c, hw = c2 // 2, h * w
        # HALF STORAGE (dhd_sfa_weights.storage_dtype, ABI 4): a half x is read as it is and every (B,C,H,W) tensor the operator
        # keeps or passes between its kernels stays in that type -- the reference's own formulation under autocast (mix.py:37-59
        # with DHD-S.py:281); float32 arithmetic, statistics, parameters and parameter gradients.  C == 128 / 256, hw % 8 == 0;
        # otherwise (or with stage.half_storage = False) x is widened once and only the edges are half (ABI 3 behaviour)
        half_storage = (io_dtype != torch.float32 and stage.half_storage
                        and bool(_lib.load().dhd_sfa_stage_half_storage_supported(c, hw)))
        if half_storage:
            x = _lib.require_gpu_tensor(x.contiguous(), io_dtype, 'SFA input')
        else:
            x = _lib.require_gpu_tensor(x.float().contiguous(), torch.float32, 'SFA input')
        if x.data_ptr() % 16:
            x = x.clone()
        dev = x.device
        lib = _lib.load()
        bn1, bn2 = stage.spacial_leanring[1], stage.spacial_leanring[4]
        ps = [_lib.require_gpu_tensor(p.detach().contiguous(), torch.float32, 'SFA ' + n) for n, p in zip(_STAGE_PARAMS, params)]
        hidden = ps[0].shape[0]
        wts = _lib.SfaWeights()
        for n, p in zip(_STAGE_PARAMS, ps):
            setattr(wts, n, p.data_ptr())
        # batch statistics iff nn.BatchNorm2d would use them (train mode, or no running buffers)
        training = int(bn1.training or bn1.running_mean is None)
        if training != int(bn2.training or bn2.running_mean is None):
            raise _lib.DhdError('SFA: the two BatchNorm layers of the stage must be in the same mode')
        for tag, bn in (('bn1', bn1), ('bn2', bn2)):
            track = bn.running_mean is not None
            setattr(wts, tag + '_mean', bn.running_mean.data_ptr() if track else None)
            setattr(wts, tag + '_var', bn.running_var.data_ptr() if track else None)
        wts.hidden, wts.training = hidden, training
        wts.gemm = _lib.SFA_GEMM[stage.gemm or default_gemm()]   # per call; backward reuses this struct
        wts.io_dtype = _lib.dtype_code(io_dtype)
        wts.storage_dtype = wts.io_dtype if half_storage else 0
        wts.eps1, wts.eps2 = bn1.eps, bn2.eps
        (wts.momentum1, wts.bn1_batches), (wts.momentum2, wts.bn2_batches) = _bn_momentum(bn1, training), _bn_momentum(bn2, training)
        if training and not bn1.training:
            wts.bn1_mean = wts.bn1_var = wts.bn2_mean = wts.bn2_var = None
        group = _sync_group(stage)   # None, or the process group whose ranks share BatchNorm statistics (nn.SyncBatchNorm)
        with torch.cuda.device(dev):
            nsaved, nscratch = C.c_size_t(), C.c_size_t()
            _lib.check(lib.dhd_sfa_stage_workspace_bytes(b, c, hw, hidden, wts.storage_dtype, C.byref(nsaved), C.byref(nscratch)),
                       'dhd_sfa_stage_workspace_bytes')
            saved = torch.empty(nsaved.value, dtype=torch.uint8, device=dev)
            scratch = _stage_scratch(dev, nscratch.value)
            out = torch.empty((b, c, h, w), dtype=io_dtype, device=dev)
            if group is None:
                _lib.check(lib.dhd_sfa_stage_forward(_lib.ptr(x), C.byref(wts), _lib.ptr(out), _lib.ptr(saved), _lib.ptr(scratch),
                                                     b, c, hw, _lib.stream_ptr(dev)), 'dhd_sfa_stage_forward')
            else:
                # the operator cut at its two statistics points; (2C + 1) float64 sums are all-reduced in between
                sums = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
                for phase in range(3):
                    _lib.check(lib.dhd_sfa_stage_forward_phase(_lib.ptr(x), C.byref(wts), _lib.ptr(out), _lib.ptr(saved), _lib.ptr(scratch),
                                                               b, c, hw, phase, _lib.ptr(sums), _lib.stream_ptr(dev)),
                               'dhd_sfa_stage_forward_phase')
                    if phase < 2:
                        _all_reduce_sum(sums, group)
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(x, saved, *ps)
            ctx.wts = wts
            ctx.dims = (b, c, hw)
            ctx.group = group
            ctx.io_dtype = io_dtype
            ctx.nscratch = nscratch.value
        return out

    @staticmethod
    @traced('dhd.sfa.stage.backward')
    def backward(ctx, go):
        x, saved = ctx.saved_tensors[:2]
        ps = ctx.saved_tensors[2:]
        b, c, hw = ctx.dims
        dev = x.device
        lib = _lib.load()
        go = go.to(ctx.io_dtype).contiguous()
        with torch.cuda.device(dev):
            gx = torch.empty(x.shape, dtype=ctx.io_dtype, device=dev)
            gps = [torch.empty_like(p) for p in ps]
            grads = _lib.SfaGrads()
            for n, g in zip(_STAGE_PARAMS, gps):
                setattr(grads, n, g.data_ptr())
            scratch = _stage_scratch(dev, ctx.nscratch)
            if ctx.group is None:
                _lib.check(lib.dhd_sfa_stage_backward(_lib.ptr(x), C.byref(ctx.wts), _lib.ptr(saved), _lib.ptr(go), _lib.ptr(gx),
                                                      C.byref(grads), _lib.ptr(scratch), b, c, hw, _lib.stream_ptr(dev)),
                           'dhd_sfa_stage_backward')
            else:
                sums = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
                for phase in range(3):
                    _lib.check(lib.dhd_sfa_stage_backward_phase(_lib.ptr(x), C.byref(ctx.wts), _lib.ptr(saved), _lib.ptr(go), _lib.ptr(gx),
                                                                C.byref(grads), _lib.ptr(scratch), b, c, hw, phase, _lib.ptr(sums),
                                                                _lib.stream_ptr(dev)), 'dhd_sfa_stage_backward_phase')
                    if phase < 2:
                        _all_reduce_sum(sums, ctx.group)
        need = ctx.needs_input_grad
        return (gx if need[0] else None, None) + tuple(g if n else None for g, n in zip(gps, need[2:]))


def needs_cross_rank_statistics(stage):
    """True when a BatchNorm of the stage was converted to nn.SyncBatchNorm (SyncbnControlHook, DHD-L.py:308-311) and is
    training in a process group of more than one rank: its statistics are sums over all ranks' batches."""
    return _sync_group(stage) is not None


_WORLD = object()   # marker: the default process group


def _sync_group(stage):
    """The process group over which the stage's BatchNorm statistics are shared, or None (plain BatchNorm, eval mode, or a
    world of one rank).  nn.SyncBatchNorm.process_group is None for the default group."""
    sp = stage.spacial_leanring
    bns = [bn for bn in (sp[1], sp[4]) if isinstance(bn, nn.SyncBatchNorm) and bn.training]
    if not bns:
        return None
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if len(bns) != 2 or bns[0].process_group is not bns[1].process_group:
        raise _lib.DhdError('SFA: the two BatchNorm layers of the stage must both be SyncBatchNorm over the same process group')
    pg = bns[0].process_group
    if dist.get_world_size(pg) <= 1:
        return None
    return _WORLD if pg is None else pg


def _all_reduce_sum(t, group):
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=None if group is _WORLD else group)


def fused_stage_supported(stage, x):
    """True when libdhd_amd.so runs the whole stage itself (float32 parameters, C == 128 or C % 256 == 0)."""
    if not x.is_cuda or x.dim() != 4:
        return False
    params = _stage_params(stage)
    if any(p is None or p.dtype != torch.float32 for p in params):
        return False
    sp = stage.spacial_leanring
    if sp[0].weight.shape != (stage.channels, stage.channels, 1, 1) or sp[1].weight is None or sp[4].weight is None:
        return False
    if needs_cross_rank_statistics(stage) and (stage.gemm or default_gemm()) == 'f32':
        return False     # the phased operator (cross-rank statistics) exists for the bf16 GEMM precisions
    return bool(_lib.load().dhd_sfa_stage_supported(stage.channels, x.shape[2] * x.shape[3]))


class channel_spatial_stage(nn.Module):
    def __init__(self, features):
        """features: channels of cat[x_bev, x_voxel] (mix.py:9-35)."""
        super().__init__()
        reduction = 16
        self.channels = features // 2
        self.fc = nn.Sequential(nn.Linear(features, features // reduction), nn.ReLU(inplace=False),
                                nn.Linear(features // reduction, self.channels), nn.Sigmoid())
        self.spacial_leanring = nn.Sequential(
            nn.Conv2d(self.channels, self.channels, kernel_size=1, stride=1, padding=0),
            BatchNorm2d(self.channels), nn.ReLU(inplace=True),
            nn.Conv2d(self.channels, self.channels, kernel_size=1, stride=1, padding=0),
            BatchNorm2d(self.channels))
        self.sigmoid = nn.Sigmoid()

    fused = True  # set False to force the generic path (library convolutions between the blend kernels)
    # a half x (autocast region): keep every tensor of the fused stage in that type (see _FusedStage.forward); DHD_SFA_HALF_STORAGE=0
    # in the environment makes round 4's form (half edges, float32 inside) the default, for A/B runs
    half_storage = __import__('os').environ.get('DHD_SFA_HALF_STORAGE', '1') != '0'
    gemm = None   # 'bf16x3' | 'bf16x6' | 'f32': precision of the fused stage's C x C GEMMs; None = default_gemm()

    def forward(self, x):
        if self.fused and fused_stage_supported(self, x):
            return _FusedStage.apply(x, self, *_stage_params(self))
        params = list(self.fc.parameters()) + list(self.spacial_leanring.parameters())
        out = _AttentionStage.apply(x.float(), self, *params)
        # the same dtype as the fused path returns for a half x (and as the reference returns under autocast)
        return out.to(x.dtype) if x.dtype in (torch.float16, torch.bfloat16) else out


@NECKS.register_module()
class SFA(nn.Module):
    def __init__(self, in_channels, out_channels, stride=1):
        super().__init__()
        self.mysk_7 = channel_spatial_stage(features=in_channels)
        self.mix_channels = in_channels
        self.out_channels = out_channels
        self.mix_residual = nn.Sequential(
            nn.Conv2d(in_channels // 2, out_channels, kernel_size=3, stride=stride, padding=1, bias=False),
            BatchNorm2d(out_channels), nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1, bias=False),
            BatchNorm2d(out_channels))
        self.mix_shortcut = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, stride=stride, kernel_size=1, bias=False),
            BatchNorm2d(out_channels))
        self.relu = nn.ReLU(inplace=True)

    def forward(self, inputs):
        fused = self.mysk_7(inputs)
        return self.relu(self.mix_residual(fused) + self.mix_shortcut(inputs))
