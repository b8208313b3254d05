"""Where HeightNet's 0.17 ms reduce comes from: the bias gradient of a 1x1 convolution with few output channels (65 / 108) is
aten::sum over (N, H, W) of the output gradient; timed here for a dense NCHW and a channels_last gradient, half and float, next to
the whole convolution backward in both layouts."""
import torch
import torch.nn.functional as F

dev = torch.device('cuda')


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for co in (65, 108, 18, 256):
    for dt in (torch.float16, torch.float32):
        g = torch.randn(24, co, 16, 44, device=dev, dtype=dt)
        gcl = g.contiguous(memory_format=torch.channels_last)
        print(f'C_out={co:4d} {str(dt):14s} sum(0,2,3): NCHW {t(lambda: g.sum((0, 2, 3))):7.1f} us   channels_last {t(lambda: gcl.sum((0, 2, 3))):7.1f} us', flush=True)
for co in (65, 108):
    conv = torch.nn.Conv2d(256, co, 1).to(dev).half().to(memory_format=torch.channels_last)
    x = torch.randn(24, 256, 16, 44, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = conv(x)
    for name, g in (('channels_last grad', torch.randn_like(y)), ('NCHW grad', torch.randn_like(y).contiguous())):
        print(f'conv 256->{co} backward with {name}: {t(lambda: torch.autograd.grad(y, (x, conv.weight, conv.bias), g, retain_graph=True)):7.1f} us', flush=True)
