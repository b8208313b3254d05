"""`bev_pool_v2` operator, same Python surface as the reference's
projects/mmdet3d_plugin/ops/bev_pool_v2/bev_pool.py (QuickCumsumCuda :11-83, bev_pool_v2 :86-106),
backed by the gfx950 kernels in csrc/bev_pool_v2.hip through the C ABI.

This is the operator-level seam (the indices are supplied by the caller).  The training hot
path does not go through here: MGHS uses the fused entry points (csrc/mghs_prepare.hip, csrc/mghs_pool.hip), which also
remove the argsort the reference repeats in every backward (bev_pool.py:47-57).
"""
import torch

from . import _lib

__all__ = ['bev_pool_v2', 'QuickCumsumCuda']


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
        return depth_grad, feat_grad, None, None, None, None, None, None


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


def bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts,
                interval_lengths):
    """
    Args (identical to the reference, bev_pool.py:86-106):
        depth: (B, N, D, fH, fW)
        feat:  (B, N, fH, fW, C)
        ranks_depth, ranks_feat, ranks_bev: (N_points,)
        bev_feat_shape: (B, D_Z, D_Y, D_X, C)
        interval_starts, interval_lengths: (N_pillar,)
    Returns:
        bev feature (B, C, Dz, Dy, Dx)
    """
    x = QuickCumsumCuda.apply(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                              interval_starts, interval_lengths)
    return x.permute(0, 4, 1, 2, 3).contiguous()
