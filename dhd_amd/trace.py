"""Named ranges around the HIP operators for rocprofv3 --marker-trace (roctx through torch.cuda.nvtx, which is ROCTX on ROCm).

The reference has no profiler hooks at all (wall-clock FPS scripts only).  Off by default: a range costs a
few microseconds of host time per call, which is what the operator seam is short of.  DHD_AMD_TRACE=1 or dhd_amd.trace.enable().
"""
import functools
import os

_on = os.environ.get('DHD_AMD_TRACE', '') not in ('', '0')


def enable(flag=True):
    global _on
    _on = bool(flag)


def enabled():
    return _on


def traced(name):
    """Decorator: run the function inside a named range when tracing is on (put it UNDER @staticmethod)."""
    def deco(fn):
        @functools.wraps(fn)
        def wrap(*args, **kwargs):
            if not _on:
                return fn(*args, **kwargs)
            import torch
            torch.cuda.nvtx.range_push(name)
            try:
                return fn(*args, **kwargs)
            finally:
                torch.cuda.nvtx.range_pop()
        return wrap
    return deco
