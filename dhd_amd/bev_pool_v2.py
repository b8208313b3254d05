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


class QuickCumsumCuda(torch.autograd.Function):
    """Name kept from the reference (bev_pool.py:11)."""

    @staticmethod
    def forward(ctx, depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts,
                interval_lengths):
        lib = _lib.load()
        if not depth.is_cuda:
            raise _lib.DhdError('bev_pool_v2 runs only on the GPU (no CPU path, as in the reference)')
        ranks_bev = ranks_bev.int().contiguous()
        depth = depth.contiguous().float()
        feat = feat.contiguous().float()
        ranks_depth = ranks_depth.contiguous().int()
        ranks_feat = ranks_feat.contiguous().int()
        interval_lengths = interval_lengths.contiguous().int()
        interval_starts = interval_starts.contiguous().int()
        out = feat.new_zeros(bev_feat_shape)  # (B, Dz, Dy, Dx, C)
        with torch.cuda.device(depth.device):
            rc = lib.dhd_bev_pool_v2_forward(
                _lib.ptr(depth), _lib.ptr(feat), _lib.ptr(out), _lib.ptr(ranks_depth), _lib.ptr(ranks_feat),
                _lib.ptr(ranks_bev), _lib.ptr(interval_lengths), _lib.ptr(interval_starts),
                int(feat.shape[-1]), int(interval_lengths.numel()), _lib.stream_ptr(depth.device))
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
        rd, rf, rb, starts, lengths = _regrouped(lib, ranks_depth, ranks_feat, ranks_bev, feat.numel() // int(feat.shape[-1]))
        depth_grad = depth.new_zeros(depth.shape)
        feat_grad = feat.new_zeros(feat.shape)
        out_grad = out_grad.contiguous().float()
        with torch.cuda.device(depth.device):
            rc = lib.dhd_bev_pool_v2_backward(
                _lib.ptr(out_grad), _lib.ptr(depth_grad), _lib.ptr(feat_grad), _lib.ptr(depth), _lib.ptr(feat),
                _lib.ptr(rd), _lib.ptr(rf), _lib.ptr(rb), _lib.ptr(lengths),
                _lib.ptr(starts), int(feat.shape[-1]), int(lengths.numel()), _lib.stream_ptr(depth.device))
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
