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
        # The backward kernel walks one feature pixel per wave, so regroup the points by pixel
        # (what bev_pool.py:47-57 does with an argsort + run-length scan).
        order = torch.argsort(ranks_feat, stable=True)
        ranks_feat = ranks_feat[order].contiguous()
        ranks_depth = ranks_depth[order].contiguous()
        ranks_bev = ranks_bev[order].contiguous()
        _, lengths = torch.unique_consecutive(ranks_feat, return_counts=True)
        starts = (torch.cumsum(lengths, 0) - lengths).int()
        lengths = lengths.int()
        depth_grad = depth.new_zeros(depth.shape)
        feat_grad = feat.new_zeros(feat.shape)
        out_grad = out_grad.contiguous().float()
        with torch.cuda.device(depth.device):
            rc = lib.dhd_bev_pool_v2_backward(
                _lib.ptr(out_grad), _lib.ptr(depth_grad), _lib.ptr(feat_grad), _lib.ptr(depth), _lib.ptr(feat),
                _lib.ptr(ranks_depth), _lib.ptr(ranks_feat), _lib.ptr(ranks_bev), _lib.ptr(lengths),
                _lib.ptr(starts), int(feat.shape[-1]), int(lengths.numel()), _lib.stream_ptr(depth.device))
        _lib.check(rc, 'dhd_bev_pool_v2_backward')
        return depth_grad, feat_grad, None, None, None, None, None, None


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
