"""Whole-step HIP graph for the training loop around the hot path.

An end-to-end DHD-S step is ~3 000 kernel launches of 5-50 us each; issued eagerly from Python the GPU idles between
them (measured: busy 77 % of a steady-state fp16 step, profiles/r1).  Every operator of libdhd_amd.so launches on the
caller's stream, allocates nothing and never synchronises, so the whole step -- forward_train, backward, gradient
clipping, the fused optimizer, the weight EMA -- can be captured once into a HIP graph and replayed (`hipGraphLaunch`),
which is how launch-bound loops are meant to run on this hardware.  Inputs live in static device buffers that
`load()` overwrites before each replay.  (The reference runs eager PyTorch under mmcv's runner; this is an execution
detail below the module boundary: parameters, gradients and optimizer state are the ordinary tensors.)"""
import torch


class GraphedStep:
    """Capture `step_fn()` (no arguments; it must read its inputs from tensors that stay alive and in place) after
    `warmup` eager runs on a side stream, then replay it.  `step_fn` returns a tensor (e.g. the loss) or None.

    Host-side scalars are frozen at capture: a kernel argument computed in Python (a float learning rate set by a
    scheduler, a decay factor) keeps its capture-time value in every replay, and host counters do not advance.  What must
    change per step has to live in device memory and be refreshed before the replay: `before_replay` callables run before
    every `graph.replay()`.  `emas`: ModelEMA objects whose update is part of the step -- their ramping decay and update
    count (ema.py:29,55; the count is written into the EMA checkpoints) are advanced per replay (ModelEMA.advance).  For
    the optimizer use a capturable one with a tensor learning rate if a schedule has to act inside the graph."""

    def __init__(self, step_fn, warmup=3, before_replay=(), emas=()):
        self.step_fn = step_fn
        self.before_replay = list(before_replay) + [e.advance for e in emas]
        self._emas = list(emas)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step_fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.result = step_fn()
        for e in self._emas:
            if not e.captured:
                raise RuntimeError('GraphedStep: a ModelEMA was passed in `emas` but its update is not part of the captured step')

    def __call__(self):
        for fn in self.before_replay:
            fn()
        self.graph.replay()
        return self.result


def load(static, fresh):
    """Copy a (nested list / dict of) fresh input tensors into the static buffers the captured step reads."""
    if isinstance(static, torch.Tensor):
        static.copy_(fresh, non_blocking=True)
    elif isinstance(static, dict):
        for k in static:
            load(static[k], fresh[k])
    elif isinstance(static, (list, tuple)):
        for a, b in zip(static, fresh):
            load(a, b)
