"""predictor.loss (models/dense_heads/occ_head.py:102-139) as one HIP operator: class-balanced,
camera-masked cross entropy + sem_scal + geo_scal losses of the (M, 18) occupancy logits in two
streaming passes (csrc/occ_loss.hip), no (M, 18) temporaries, no device->host round trips."""
import ctypes as C

import torch

from . import _lib

NUM_CLASSES = 18


class _OccLosses(torch.autograd.Function):
    """logits (M,18) f32, labels (M) u8, mask (M) u8, class_weight (18) f32 -> losses (3,) f32."""

    events = None  # bench.py: a list that receives (start, end) HIP events around the gradient kernel launch

    @staticmethod
    def forward(ctx, logits, labels, mask, class_weight, ignore_index, non_empty_idx):
        logits = _lib.require_gpu_tensor(logits.contiguous(), torch.float32, 'occupancy logits')
        labels = _lib.require_gpu_tensor(labels.contiguous(), torch.uint8, 'voxel labels')
        mask = _lib.require_gpu_tensor(mask.contiguous(), torch.uint8, 'camera mask')
        cw = _lib.require_gpu_tensor(class_weight.contiguous(), torch.float32, 'class weights')
        m, k = logits.shape
        if labels.numel() != m or mask.numel() != m or cw.numel() != k:
            raise _lib.DhdError('occupancy losses: inconsistent shapes')
        if logits.data_ptr() % 16:
            logits = logits.clone()
        lib = _lib.load()
        dev = logits.device
        with torch.cuda.device(dev):
            ws = torch.empty(lib.dhd_occ_loss_workspace_bytes(), dtype=torch.uint8, device=dev)
            losses = torch.empty(3, dtype=torch.float32, device=dev)
            _lib.check(lib.dhd_occ_loss_forward(_lib.ptr(logits), _lib.ptr(labels), _lib.ptr(mask), _lib.ptr(cw), m, k, ignore_index,
                                                non_empty_idx, _lib.ptr(losses), _lib.ptr(ws), _lib.stream_ptr(dev)),
                       'dhd_occ_loss_forward')
        ctx.save_for_backward(logits, labels, mask, cw, ws)
        ctx.args = (m, k, ignore_index, non_empty_idx)
        return losses

    @staticmethod
    def backward(ctx, g):
        logits, labels, mask, cw, ws = ctx.saved_tensors
        m, k, ignore_index, non_empty_idx = ctx.args
        lib = _lib.load()
        dev = logits.device
        g = g.float().contiguous()
        with torch.cuda.device(dev):
            grad = torch.empty_like(logits)
            ev = _OccLosses.events
            if ev is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            _lib.check(lib.dhd_occ_loss_backward(_lib.ptr(logits), _lib.ptr(labels), _lib.ptr(mask), _lib.ptr(cw), m, k, ignore_index,
                                                 non_empty_idx, _lib.ptr(g), _lib.ptr(ws), _lib.ptr(grad), _lib.stream_ptr(dev)),
                       'dhd_occ_loss_backward')
            if ev is not None:
                e1.record()
                ev.append((e0, e1))
        return grad, None, None, None, None, None


def supported(logits):
    return logits.is_cuda and logits.dim() == 2 and logits.shape[1] == NUM_CLASSES


def occ_losses(logits, labels, mask_camera, class_weight, ignore_index=255, non_empty_idx=17):
    """(loss_occ, loss_voxel_sem_scal, loss_voxel_geo_scal) before the head's weight_* factors.
    labels / mask_camera may be any integer or bool dtype (converted to uint8 once)."""
    labels = labels.reshape(-1).to(torch.uint8)
    mask = mask_camera.reshape(-1).to(torch.uint8)
    out = _OccLosses.apply(logits.float(), labels, mask, class_weight.to(device=logits.device, dtype=torch.float32),
                           int(ignore_index), int(non_empty_idx))
    return out[0], out[1], out[2]


def occ_argmax_hist(logits, labels=None, mask_camera=None, hist=None):
    """predictor.get_occ + Metric_mIoU.hist_info in one pass: returns (pred (M,) uint8, hist (18,18) int64).
    `hist` (device int64) is accumulated into when given; rows = ground truth, columns = prediction."""
    logits = _lib.require_gpu_tensor(logits.reshape(-1, NUM_CLASSES).contiguous(), torch.float32, 'occupancy logits')
    if logits.data_ptr() % 16:
        logits = logits.clone()
    m = logits.shape[0]
    dev = logits.device
    with torch.cuda.device(dev):
        pred = torch.empty(m, dtype=torch.uint8, device=dev)
        lab = msk = None
        if labels is not None:
            lab = labels.reshape(-1).to(torch.uint8).contiguous()
            msk = None if mask_camera is None else mask_camera.reshape(-1).to(torch.uint8).contiguous()
            if hist is None:
                hist = torch.zeros(NUM_CLASSES, NUM_CLASSES, dtype=torch.int64, device=dev)
        _lib.check(_lib.load().dhd_occ_argmax_hist(_lib.ptr(logits), _lib.ptr(lab), _lib.ptr(msk), m, NUM_CLASSES, _lib.ptr(pred),
                                                   _lib.ptr(hist) if labels is not None else None, _lib.stream_ptr(dev)),
                   'dhd_occ_argmax_hist')
    return pred, hist


def miou_from_hist(hist):
    """Metric_mIoU.per_class_iu / count_miou (occ_metrics.py:106-108,159-167): mean IoU of the 17 semantic classes, in %."""
    h = hist.double()
    iu = torch.diag(h) / (h.sum(1) + h.sum(0) - torch.diag(h))
    return float(torch.nanmean(iu[:NUM_CLASSES - 1]) * 100), iu
