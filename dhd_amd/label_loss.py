"""Height / depth supervision of the view transformer as HIP operators (csrc/label_loss.hip):
get_downsampled_gt_depth / _height (lss_heightmap.py:625-701) -> bin indices, and the foreground
binary cross entropy of get_height_loss (:595-622) / get_depth_and_height_loss (:859-897)."""
import torch

from . import _lib


def bin_labels(gt_depth, gt_height, downsample, depth_cfg, n_depth, height_offset, height_step, n_height, sid=False):
    """(B,N,H,W) sparse maps -> (depth_bin, height_bin), each (B*N*fH*fW) int16; 0 = no label.  sid: spacing-increasing
    depth bins (lss_heightmap.py:655-660)."""
    gt_depth = _lib.require_gpu_tensor(gt_depth.contiguous(), torch.float32, 'gt_depth')
    gt_height = _lib.require_gpu_tensor(gt_height.contiguous(), torch.float32, 'gt_height')
    b, n, h, w = gt_depth.shape
    if gt_height.shape != gt_depth.shape or h % downsample or w % downsample:
        raise _lib.DhdError('gt maps must have equal shapes divisible by the downsample factor')
    fh, fw = h // downsample, w // downsample
    dev = gt_depth.device
    with torch.cuda.device(dev):
        dbin = torch.empty(b * n * fh * fw, dtype=torch.int16, device=dev)
        hbin = torch.empty_like(dbin)
        if sid:
            # the two scalars exactly as the reference forms them: float32 tensors through torch.log (:656-658)
            log_d0 = float(torch.log(torch.tensor(depth_cfg[0]).float()))
            log_ratio = float(torch.log(torch.tensor(depth_cfg[1] - 1.).float() / depth_cfg[0]))
            _lib.check(_lib.load().dhd_sparse_bin_labels_sid(_lib.ptr(gt_depth), _lib.ptr(gt_height), b * n, fh, fw, downsample,
                                                             log_d0, log_ratio, n_depth, float(height_offset), float(height_step),
                                                             n_height, _lib.ptr(dbin), _lib.ptr(hbin), _lib.stream_ptr(dev)),
                       'dhd_sparse_bin_labels_sid')
            return dbin, hbin
        # the reference subtracts the Python double (d0 - dstep) from a float32 tensor: the scalar is rounded to float32
        _lib.check(_lib.load().dhd_sparse_bin_labels(_lib.ptr(gt_depth), _lib.ptr(gt_height), b * n, fh, fw, downsample,
                                                     float(depth_cfg[0] - depth_cfg[2]), float(depth_cfg[2]), n_depth,
                                                     float(height_offset), float(height_step), n_height,
                                                     _lib.ptr(dbin), _lib.ptr(hbin), _lib.stream_ptr(dev)), 'dhd_sparse_bin_labels')
    return dbin, hbin


class _BinBCE(torch.autograd.Function):
    """pred (BN, C, fH, fW) probabilities, labels as bin indices -> weight * sum_fg BCE / max(1, n_fg)."""

    @staticmethod
    def forward(ctx, pred, bin_idx, fg_bin, weight):
        pred = _lib.require_gpu_tensor(pred.contiguous(), torch.float32, 'prediction map')
        bn, c = pred.shape[:2]
        hw = pred[0, 0].numel()
        lib = _lib.load()
        dev = pred.device
        with torch.cuda.device(dev):
            ws = torch.empty(lib.dhd_bin_bce_workspace_bytes(), dtype=torch.uint8, device=dev)
            loss = torch.empty(1, dtype=torch.float32, device=dev)
            _lib.check(lib.dhd_bin_bce_forward(_lib.ptr(pred), _lib.ptr(bin_idx), _lib.ptr(fg_bin), bn, c, hw, weight, _lib.ptr(loss),
                                               _lib.ptr(ws), _lib.stream_ptr(dev)), 'dhd_bin_bce_forward')
        ctx.save_for_backward(pred, bin_idx, fg_bin, ws)
        ctx.args = (bn, c, hw, weight)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        pred, bin_idx, fg_bin, ws = ctx.saved_tensors
        bn, c, hw, weight = ctx.args
        dev = pred.device
        g = g.float().reshape(1).contiguous()
        with torch.cuda.device(dev):
            grad = torch.empty_like(pred)
            _lib.check(_lib.load().dhd_bin_bce_backward(_lib.ptr(pred), _lib.ptr(bin_idx), _lib.ptr(fg_bin), bn, c, hw, weight, _lib.ptr(g),
                                                        _lib.ptr(ws), _lib.ptr(grad), _lib.stream_ptr(dev)), 'dhd_bin_bce_backward')
        return grad, None, None, None


def fg_bce(pred, bin_idx, fg_bin, weight):
    return _BinBCE.apply(pred.float(), bin_idx, fg_bin, float(weight))


def points_to_maps(points, height, width, downsample=1, depth_range=(1.0, 45.0)):
    """PointToMultiViewDepthandHeight.points2depthmap / points2heightmap (loading_new.py:35-99) for all cameras at once:
    points (n_cams, n_points, 4) = (u, v, d, h) -> (gt_depth, gt_height), each (n_cams, H/ds, W/ds)."""
    points = _lib.require_gpu_tensor(points.contiguous(), torch.float32, 'projected points')
    if points.dim() != 3 or points.shape[2] != 4:
        raise _lib.DhdError('points must be (n_cams, n_points, 4)')
    n_cams, n_pts = points.shape[:2]
    h, w = height // downsample, width // downsample
    dev = points.device
    with torch.cuda.device(dev):
        dm = torch.empty((n_cams, h, w), dtype=torch.float32, device=dev)
        hm = torch.empty_like(dm)
        zbuf = torch.empty(n_cams * h * w, dtype=torch.int64, device=dev)
        _lib.check(_lib.load().dhd_points_to_maps(_lib.ptr(points), n_cams, n_pts, height, width, downsample, float(depth_range[0]),
                                                  float(depth_range[1]), _lib.ptr(dm), _lib.ptr(hm), _lib.ptr(zbuf), _lib.stream_ptr(dev)),
                   'dhd_points_to_maps')
    return dm, hm
