"""channels_last BatchNorm (+ReLU / + residual + ReLU), forward and backward: the library's kernels against torch (MIOpen NHWC BatchNorm +
element-wise kernels) on the DHD-S image encoder's activation shapes (24 images, float16).  Prints microseconds and effective GB/s
(algorithmic bytes: forward 3 passes / 4 with a residual, backward 5 / 7)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dhd_amd.batchnorm import BatchNorm2d
dev = torch.device('cuda:0')
shapes = [(24, 64, 128, 352), (24, 64, 64, 176), (24, 256, 64, 176), (24, 128, 32, 88), (24, 512, 32, 88), (24, 256, 16, 44),
          (24, 1024, 16, 44), (24, 512, 8, 22), (24, 2048, 8, 22), (4, 64, 200, 200), (4, 128, 100, 100), (4, 512, 25, 25)]
modes = sys.argv[1:] or ['relu', 'add']

def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for shape in shapes:
    for mode in modes:
        x = torch.randn(shape, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last).requires_grad_()
        res = torch.randn_like(x).requires_grad_() if mode == 'add' else None
        g = torch.randn_like(x)
        nbytes = x.numel() * 2
        out = []
        for hip in (True, False):
            bn = BatchNorm2d(shape[1]).to(dev).train()
            bn.use_nhwc = hip
            y = [None]
            def fwd():
                y[0] = bn(x, relu=mode == 'relu', residual=res)
            def bwd():
                y[0].backward(g, retain_graph=True)
                x.grad = None
                if res is not None:
                    res.grad = None
            tf = timed(fwd)
            tb = timed(bwd)
            out.append((tf, tb))
        (hf, hb), (tf, tb) = out
        pf, pb = (4, 7) if mode == 'add' else (3, 5)
        print(f'{str(shape):22s} {mode:5s} fwd hip {hf:7.1f} us ({pf * nbytes / hf / 1e3:6.0f} GB/s)  torch {tf:7.1f} us | bwd hip {hb:7.1f} us ({pb * nbytes / hb / 1e3:6.0f} GB/s)  torch {tb:7.1f} us',
              flush=True)
