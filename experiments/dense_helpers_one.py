"""One dense-caller helper kernel family, ten launches, for rocprofv3 (kernel durations / PMC counters): see pmc_dense_helpers.sh.
usage: dense_helpers_one.py bn|transpose|upsample|window|cost_volume"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device('cuda:0')
op = sys.argv[1]
torch.manual_seed(0)
cl = torch.channels_last
if op == 'bn':            # the image encoder's largest activation: (24, 64, 128, 352) half, BatchNorm + ReLU forward + backward
    from dhd_amd.batchnorm import BatchNorm2d
    x = torch.randn(24, 64, 128, 352, device=dev, dtype=torch.float16).contiguous(memory_format=cl).requires_grad_()
    g = torch.randn_like(x)
    bn = BatchNorm2d(64).to(dev).train()
    def run():
        bn(x, relu=True).backward(g); x.grad = None
elif op == 'transpose':   # (4, 512, 200, 200) half: the SFA stage's input, channels_last -> NCHW and back
    from dhd_amd.layout import to_layout
    x = torch.randn(4, 512, 200, 200, device=dev, dtype=torch.float16)
    def run():
        to_layout(to_layout(x, cl), torch.contiguous_format)
elif op == 'upsample':    # FPN_LSS.up2: (4, 256, 100, 100) -> 200 x 200, channels_last half, forward + backward
    from dhd_amd.detector import Upsample
    up = Upsample(scale_factor=2, mode='bilinear', align_corners=True)
    x = torch.randn(4, 256, 100, 100, device=dev, dtype=torch.float16).contiguous(memory_format=cl).requires_grad_()
    g = torch.randn(4, 256, 200, 200, device=dev, dtype=torch.float16).contiguous(memory_format=cl)
    def run():
        up(x).backward(g); x.grad = None
elif op == 'window':      # Swin stage 0 of DHD-L: 36 images of 128 x 352 tokens, 128 channels, float32 -> bfloat16 windows and back
    from dhd_amd.swin import _WindowRows
    x = torch.randn(36, 128, 352, 128, device=dev)
    def run():
        w = _WindowRows.apply(x, 128, 352, 7, 3, False, torch.bfloat16)
        _WindowRows.apply(w, 128, 352, 7, 3, True, torch.bfloat16)
else:                     # stereo cost volume at the DHD-L size, half-pixel walk
    from dhd_amd import _lib, mghs_op
    bn_, c, h, w, d = 12, 128, 128, 352, 88
    prev = mghs_op._nchw_to_nhwc(torch.randn(bn_, c, h, w, device=dev)); curr = mghs_op._nchw_to_nhwc(torch.randn(bn_, c, h, w, device=dev))
    out = torch.empty(bn_, d, h, w, device=dev)
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, h, device=dev), torch.linspace(-1, 1, w, device=dev), indexing='ij')
    k = torch.arange(d, device=dev, dtype=torch.float32).view(1, d, 1, 1)
    grid = torch.stack([(xs.view(1, 1, h, w) + k * 0.5 * 2 / (w - 1)).expand(bn_, d, h, w), (ys.view(1, 1, h, w) + k * 0.15 * 2 / (h - 1)).expand(bn_, d, h, w)], -1).contiguous()
    lib = _lib.load()
    def run():
        _lib.check(lib.dhd_stereo_cost_volume(_lib.ptr(prev), _lib.ptr(curr), _lib.ptr(grid), bn_, c, h, w, d, 10.0, c - 4, _lib.ptr(out), _lib.stream_ptr(dev)), 'cv')
for _ in range(10):
    run()
torch.cuda.synchronize()
