"""dhd_stereo_cost_volume at the DHD-L stereo size (12 cameras, 128 x 352, D = 88) under three sampling grids: every hypothesis at
one position (no tap load after the first: the kernel's arithmetic / latency floor), a smooth epipolar-like walk of `step` pixels per
hypothesis, and random positions (every tap a fresh 1-KB gather).  usage: cost_volume_bench.py [channels]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dhd_amd import _lib, mghs_op
dev = torch.device('cuda:0')
bn, c, h, w, d = 12, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 128, 352, 88
torch.manual_seed(0)
prev = mghs_op._nchw_to_nhwc(torch.randn(bn, c, h, w, device=dev))
curr = mghs_op._nchw_to_nhwc(torch.randn(bn, c, h, w, device=dev))
out = torch.empty(bn, d, h, w, device=dev)
ys, xs = torch.meshgrid(torch.linspace(-1, 1, h, device=dev), torch.linspace(-1, 1, w, device=dev), indexing='ij')

def walk(step):      # hypothesis k sits k * step pixels to the right / below of the pixel itself
    k = torch.arange(d, device=dev, dtype=torch.float32).view(1, d, 1, 1)
    gx = xs.view(1, 1, h, w) + k * step * 2 / (w - 1)
    gy = ys.view(1, 1, h, w) + k * step * 0.3 * 2 / (h - 1)
    return torch.stack([gx.expand(bn, d, h, w), gy.expand(bn, d, h, w)], -1).contiguous()

grids = {'one position': walk(0.0), 'walk 0.1 px': walk(0.1), 'walk 0.5 px': walk(0.5), 'walk 2 px': walk(2.0),
         'random': (torch.rand(bn, d, h, w, 2, device=dev) * 2 - 1).contiguous()}
lib = _lib.load()
for name, g in grids.items():
    def run():
        _lib.check(lib.dhd_stereo_cost_volume(_lib.ptr(prev), _lib.ptr(curr), _lib.ptr(g), bn, c, h, w, d, 10.0, c - 4, _lib.ptr(out),
                                              _lib.stream_ptr(dev)), 'cv')
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f'{name:14s} {e0.elapsed_time(e1) / 5:8.3f} ms', flush=True)
