"""NCHW <-> channels_last conversion of 4-D activations through the library's tiled transpose (csrc/layout.hip,
include/dhd_amd.h section 11), as an autograd node whose gradient travels back in the layout of the INPUT: the custom operators
(MGHS, the SFA stage) take and return NCHW, the dense stacks around them may run in channels_last (detector.use_channels_last).
CPU tensors, other ranks, dtypes that are not 2 or 4 bytes wide and tensors that are in neither format take torch's
`.contiguous(memory_format=...)`."""
import os

import torch

from . import _lib

_TORCH_ONLY = bool(os.environ.get('DHD_TORCH_LAYOUT'))   # A/B switch: torch's strided copy

_NCHW, _NHWC = torch.contiguous_format, torch.channels_last


def _format_of(t):
    """torch.contiguous_format / torch.channels_last if `t` is dense in exactly that format (a tensor that is both -- one channel
    or one pixel -- counts as whatever is asked for), else None."""
    if t.dim() != 4:
        return None
    a, b = t.is_contiguous(), t.is_contiguous(memory_format=_NHWC)
    return 'both' if (a and b) else (_NCHW if a else (_NHWC if b else None))


def _convert(t, fmt):
    src = _format_of(t)
    if src == 'both' or src == fmt:
        return t
    if _TORCH_ONLY or src is None or not t.is_cuda or t.element_size() not in (2, 4) or t.numel() == 0 or t.data_ptr() % 16:
        # (the kernel moves 16-byte vectors / 4-byte pairs: a dense view at an odd storage offset takes torch's strided copy)
        return t.contiguous(memory_format=fmt)
    n, c, h, w = t.shape
    out = torch.empty((n, c, h, w), dtype=t.dtype, device=t.device, memory_format=fmt)
    rows, cols = (c, h * w) if fmt == _NHWC else (h * w, c)      # the matrix the SOURCE is, per image
    with torch.cuda.device(t.device):
        _lib.check(_lib.load().dhd_transpose_batched(_lib.ptr(t), _lib.ptr(out), t.element_size(), n, rows, cols, _lib.stream_ptr(t.device)),
                   'dhd_transpose_batched')
    return out


class _ToLayout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, fmt):
        src = _format_of(t)
        ctx.back = _NCHW if src in (_NCHW, 'both', None) else _NHWC
        return _convert(t, fmt)

    @staticmethod
    def backward(ctx, g):
        return _convert(g, ctx.back), None


def to_layout(t, fmt):
    """`t` dense in memory format `fmt` (torch.contiguous_format or torch.channels_last); `t` itself if it already is."""
    src = _format_of(t)
    if src == 'both' or src == fmt:
        return t
    if t.requires_grad and torch.is_grad_enabled():
        return _ToLayout.apply(t, fmt)
    return _convert(t, fmt)
