"""`bev_pool_v2` operator, same Python surface as the reference's
projects/mmdet3d_plugin/ops/bev_pool_v2/bev_pool.py (QuickCumsumCuda :11-83, bev_pool_v2 :86-106),
backed by the gfx950 kernels in csrc/bev_pool_v2.hip through the C ABI.

This is the operator-level seam (the indices are supplied by the caller).  The training hot
path does not go through here: MGHS uses the fused entry points (csrc/mghs_prepare.hip, csrc/mghs_pool.hip), which also
remove the argsort the reference repeats in every backward (bev_pool.py:47-57).
"""
import torch

from . import _lib
from .trace import traced

__all__ = ['bev_pool_v2', 'QuickCumsumCuda', 'fused_supported', 'clear_caches']


def _as(t, dtype):
    """`t.contiguous().to(dtype)` of the reference wrapper (bev_pool.py:20-25,59-69) without the Python cost of two no-op calls
    when the tensor already is what the kernel needs (the operator's time through this surface is mostly host time)."""
    return t if (t.dtype == dtype and t.is_contiguous()) else t.contiguous().to(dtype)


class _on:
    """`with torch.cuda.device(dev)` only when `dev` is not the current device (the context manager costs ~10 us per use)."""

    def __init__(self, dev):
        self.ctx = None if dev.index == torch.cuda.current_device() else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


class QuickCumsumCuda(torch.autograd.Function):
    """Name kept from the reference (bev_pool.py:11)."""

    @staticmethod
    @traced('dhd.bev_pool_v2.forward')
    def forward(ctx, depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts,
                interval_lengths):
        lib = _lib.load()
        if not depth.is_cuda:
            raise _lib.DhdError('bev_pool_v2 runs only on the GPU (no CPU path, as in the reference)')
        i32, f32 = torch.int32, torch.float32
        ranks_bev = _as(ranks_bev, i32)
        depth = _as(depth, f32)
        feat = _as(feat, f32)
        ranks_depth = _as(ranks_depth, i32)
        ranks_feat = _as(ranks_feat, i32)
        interval_lengths = _as(interval_lengths, i32)
        interval_starts = _as(interval_starts, i32)
        out = feat.new_zeros(bev_feat_shape)  # (B, Dz, Dy, Dx, C)
        with _on(depth.device):
            rc = lib.dhd_bev_pool_v2_forward(
                depth.data_ptr(), feat.data_ptr(), out.data_ptr(), ranks_depth.data_ptr(), ranks_feat.data_ptr(),
                ranks_bev.data_ptr(), interval_lengths.data_ptr(), interval_starts.data_ptr(),
                feat.shape[-1], interval_lengths.numel(), torch.cuda.current_stream(depth.device).cuda_stream)
        if rc:
            _lib.check(rc, 'dhd_bev_pool_v2_forward')
        ctx.save_for_backward(ranks_bev, depth, feat, ranks_feat, ranks_depth)
        return out

    @staticmethod
    @traced('dhd.bev_pool_v2.backward')
    def backward(ctx, out_grad):
        lib = _lib.load()
        ranks_bev, depth, feat, ranks_feat, ranks_depth = ctx.saved_tensors
        # The backward kernel walks one feature pixel per wave (or lane group), so the points are regrouped by pixel --
        # what bev_pool.py:47-57 does with an argsort, three gathers and a run-length scan on EVERY call.  Here: a device
        # counting sort (dhd_bev_pool_v2_regroup, no host synchronisation), kept while the same rank tensors come back
        # unmodified (index lists of a static rig; voxel_pooling_v2 rebuilds them per call in training).
        c = feat.shape[-1]
        rd, rf, rb, starts, lengths = _regrouped(lib, ranks_depth, ranks_feat, ranks_bev, feat.numel() // c)
        depth_grad = torch.zeros_like(depth)
        feat_grad = torch.zeros_like(feat)
        out_grad = _as(out_grad, torch.float32)
        with _on(depth.device):
            rc = lib.dhd_bev_pool_v2_backward(
                out_grad.data_ptr(), depth_grad.data_ptr(), feat_grad.data_ptr(), depth.data_ptr(), feat.data_ptr(),
                rd.data_ptr(), rf.data_ptr(), rb.data_ptr(), lengths.data_ptr(), starts.data_ptr(), c, lengths.numel(),
                torch.cuda.current_stream(depth.device).cuda_stream)
        if rc:
            _lib.check(rc, 'dhd_bev_pool_v2_backward')
        return depth_grad, feat_grad, None, None, None, None, None, None, None


_regroup_cache = {}   # device index -> (stamp, tensors kept alive, result): the last regrouping per device


def _regrouped(lib, ranks_depth, ranks_feat, ranks_bev, n_pixels):
    """(ranks_depth, ranks_feat, ranks_bev) ordered by feature pixel + one (start, length) interval per pixel."""
    dev = ranks_feat.device
    srcs = (ranks_depth, ranks_feat, ranks_bev)
    stamp = tuple((t.data_ptr(), t._version, t.numel()) for t in srcs) + (n_pixels,)
    hit = _regroup_cache.get(dev.index)
    if hit is not None and hit[0] == stamp:
        return hit[2]
    n = int(ranks_feat.numel())
    with torch.cuda.device(dev):
        out = [torch.empty(max(n, 1), dtype=torch.int32, device=dev) for _ in range(3)]
        starts = torch.empty(n_pixels, dtype=torch.int32, device=dev)
        lengths = torch.empty(n_pixels, dtype=torch.int32, device=dev)
        nbytes = int(lib.dhd_bev_pool_v2_regroup_scratch_bytes(n, n_pixels))
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        rc = lib.dhd_bev_pool_v2_regroup(_lib.ptr(ranks_depth), _lib.ptr(ranks_feat), _lib.ptr(ranks_bev), n, n_pixels,
                                         _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(out[2]), _lib.ptr(starts), _lib.ptr(lengths),
                                         _lib.ptr(scratch), nbytes, _lib.stream_ptr(dev))
    _lib.check(rc, 'dhd_bev_pool_v2_regroup')
    res = (out[0], out[1], out[2], starts, lengths)
    _regroup_cache[dev.index] = (stamp, srcs, res)   # srcs kept alive: their addresses cannot be recycled for other lists
    return res


_fused_scratch = {}   # (device index, stream) -> uint8 scratch of the fused operator (reusable between calls on a stream)


def fused_supported(bev_feat_shape, n_intervals=1):
    """Whether dhd_bev_pool_v2_fused_* takes this shape.  The library is asked (dhd_bev_pool_v2_fused_workspace_bytes returns
    DHD_EUNSUPPORTED otherwise: C == 64, Dy % 4 == 0, Dx % 4 == 0, Dx <= 256, B*Dz*Dy*Dx <= 2^30 today), so the gate cannot
    drift from the C side."""
    import ctypes as C
    b, dz, dy, dx, c = (int(v) for v in bev_feat_shape)
    sb, cb = C.c_size_t(), C.c_size_t()
    return _lib.load().dhd_bev_pool_v2_fused_workspace_bytes(c, b, dz, dy, dx, max(int(n_intervals), 1), C.byref(sb), C.byref(cb)) == 0


class _FusedPool(torch.autograd.Function):
    """bev_pool_v2 + `permute(0, 4, 1, 2, 3).contiguous()` (bev_pool.py:86-106) as one node: the (B, C, Dz, Dy, Dx) tensor is
    written once, zeros included, and its gradient is read once in that layout (dhd_bev_pool_v2_fused_forward / _backward)."""

    @staticmethod
    @traced('dhd.bev_pool_v2.fused.forward')
    def forward(ctx, depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts, interval_lengths, cache=True):
        # `cache` is a per-call argument (not class state): a checkpointed recompute or another thread sees its own caller's choice
        lib = _lib.load()
        if not depth.is_cuda:
            raise _lib.DhdError('bev_pool_v2 runs only on the GPU (no CPU path, as in the reference)')
        i32, f32 = torch.int32, torch.float32
        ranks_bev = _as(ranks_bev, i32)
        depth = _as(depth, f32)
        feat = _as(feat, f32)
        ranks_depth = _as(ranks_depth, i32)
        ranks_feat = _as(ranks_feat, i32)
        interval_lengths = _as(interval_lengths, i32)
        interval_starts = _as(interval_starts, i32)
        b, dz, dy, dx, c = (int(v) for v in bev_feat_shape)
        n_iv = interval_lengths.numel()
        dev = depth.device
        sizes = _fused_sizes(lib, c, b, dz, dy, dx, n_iv)
        # the voxel -> row map depends on the index lists only: kept while the same tensors come back unmodified (a static rig), as
        # the backward's regrouping is
        srcs = (ranks_bev, interval_starts, interval_lengths)
        with _on(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            stamp = tuple((t.data_ptr(), t._version, t.numel()) for t in srcs) + (b, dz, dy, dx, stream)   # filled on this stream
            hit = _state_cache.get(dev.index) if cache else None
            valid = hit is not None and hit[0] == stamp
            out = torch.empty((b, c, dz, dy, dx), dtype=f32, device=dev)
            state = hit[2] if valid else torch.empty(sizes[0], dtype=torch.uint8, device=dev)
            scratch = _scratch_for(dev, stream, sizes[1])
            rc = lib.dhd_bev_pool_v2_fused_forward(
                depth.data_ptr(), feat.data_ptr(), out.data_ptr(), ranks_depth.data_ptr(), ranks_feat.data_ptr(),
                ranks_bev.data_ptr(), interval_lengths.data_ptr(), interval_starts.data_ptr(), c, n_iv, b, dz, dy, dx,
                state.data_ptr(), sizes[0], 1 if valid else 0, scratch.data_ptr(), scratch.numel(), stream)
        if rc:
            _lib.check(rc, 'dhd_bev_pool_v2_fused_forward')
        if not valid and cache:
            _state_cache[dev.index] = (stamp, srcs, state)   # srcs kept alive: their addresses cannot be recycled for other lists
        ctx.save_for_backward(ranks_bev, depth, feat, ranks_feat, ranks_depth, state)
        ctx.dims = (b, dz, dy, dx, c, n_iv, sizes)
        return out

    @staticmethod
    @traced('dhd.bev_pool_v2.fused.backward')
    def backward(ctx, out_grad):
        lib = _lib.load()
        ranks_bev, depth, feat, ranks_feat, ranks_depth, state = ctx.saved_tensors
        b, dz, dy, dx, c, n_iv, sizes = ctx.dims
        rd, rf, rb, starts, lengths = _regrouped(lib, ranks_depth, ranks_feat, ranks_bev, feat.numel() // c)
        dev = depth.device
        out_grad = _as(out_grad, torch.float32)
        with _on(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            # one zero-fill for both gradients (bev_pool.py:67-68); feat's part starts on a 16-byte boundary
            nd = (depth.numel() + 3) // 4 * 4
            both = torch.zeros(nd + feat.numel(), dtype=torch.float32, device=dev)
            depth_grad, feat_grad = both[:depth.numel()].view(depth.shape), both[nd:].view(feat.shape)
            scratch = _scratch_for(dev, stream, sizes[1])
            rc = lib.dhd_bev_pool_v2_fused_backward(
                out_grad.data_ptr(), depth_grad.data_ptr(), feat_grad.data_ptr(), depth.data_ptr(), feat.data_ptr(),
                rd.data_ptr(), rf.data_ptr(), rb.data_ptr(), lengths.data_ptr(), starts.data_ptr(), c, lengths.numel(), n_iv,
                b, dz, dy, dx, state.data_ptr(), sizes[0], scratch.data_ptr(), scratch.numel(), stream)
        if rc:
            _lib.check(rc, 'dhd_bev_pool_v2_fused_backward')
        return depth_grad, feat_grad, None, None, None, None, None, None, None


_fused_size_cache = {}
_state_cache = {}        # device index -> (stamp, index tensors kept alive, state): the last voxel -> row map per device


def _fused_sizes(lib, c, b, dz, dy, dx, n_iv):
    key = (c, b, dz, dy, dx, n_iv)
    hit = _fused_size_cache.get(key)
    if hit is None:
        import ctypes as C
        sb, cb = C.c_size_t(), C.c_size_t()
        _lib.check(lib.dhd_bev_pool_v2_fused_workspace_bytes(c, b, dz, dy, dx, n_iv, C.byref(sb), C.byref(cb)),
                   'dhd_bev_pool_v2_fused_workspace_bytes')
        if len(_fused_size_cache) > 64:
            _fused_size_cache.clear()
        hit = _fused_size_cache[key] = (int(sb.value), int(cb.value))
    return hit


def _scratch_for(dev, stream, nbytes):
    key = (dev.index, stream)
    buf = _fused_scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _fused_scratch[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return buf


def clear_caches():
    """Drop what the wrapper keeps between calls per device: the last regrouping of the backward, the last voxel -> row map of
    the fused forward (each holds its index tensors alive) and the fused operator's scratch buffers."""
    _regroup_cache.clear()
    _state_cache.clear()
    _fused_scratch.clear()


def bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts,
                interval_lengths, fused=False, cache=True):
    """
    Args (identical to the reference, bev_pool.py:86-106):
        depth: (B, N, D, fH, fW)
        feat:  (B, N, fH, fW, C)
        ranks_depth, ranks_feat, ranks_bev: (N_points,)
        bev_feat_shape: (B, D_Z, D_Y, D_X, C)
        interval_starts, interval_lengths: (N_pillar,)
        fused (not in the reference): write the returned (B, C, Dz, Dy, Dx) tensor directly, once, zeros included, instead
            of zero-fill + kernel + permute copy; same values bit for bit.  Shapes the library's fused entry points do not
            take (`fused_supported`, which asks the library) run the reference's three steps.
        cache (fused only): keep the voxel -> row map of the last call per device and reuse it while the SAME index tensors
            come back unmodified -- recognised by (data_ptr, torch's version counter, numel) of ranks_bev / interval_starts /
            interval_lengths.  CONTRACT: while cached, those tensors must only be modified through torch operations (which bump
            the version counter); writes through `.data`, raw-pointer kernels or other libraries are invisible and would reuse
            a stale map.  The cache keeps the three tensors and the map alive until `clear_caches()`.  cache=False rebuilds the
            map on every call (+ ~10 us) and pins nothing.
    Returns:
        bev feature (B, C, Dz, Dy, Dx)
    """
    if fused and fused_supported(bev_feat_shape, interval_lengths.numel()):
        return _FusedPool.apply(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts,
                                interval_lengths, bool(cache))
    x = QuickCumsumCuda.apply(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                              interval_starts, interval_lengths)
    return x.permute(0, 4, 1, 2, 3).contiguous()
