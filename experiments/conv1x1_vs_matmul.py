"""1x1 convolution on NCHW (MIOpen) vs the same contraction as a batched matmul (hipBLASLt), fwd+bwd."""
import torch, torch.nn.functional as F
dev = torch.device('cuda', 0)
B, C, H, W = 4, 256, 200, 200
x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
w = (torch.randn(C, C, 1, 1, device=dev) * 0.05).requires_grad_()
b = torch.zeros(C, device=dev, requires_grad=True)
g = torch.randn(B, C, H, W, device=dev)
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def conv():
    x.grad = w.grad = None
    F.conv2d(x, w, b).backward(g)
def mm():
    x.grad = w.grad = None
    (torch.matmul(w.flatten(1), x.flatten(2)) + b.view(1, -1, 1)).view(B, C, H, W).backward(g)
def mm_addmm():
    x.grad = w.grad = None
    torch.baddbmm(b.view(1, -1, 1), w.flatten(1).unsqueeze(0).expand(B, C, C), x.flatten(2)).view(B, C, H, W).backward(g)
y1 = F.conv2d(x, w, b); y2 = (torch.matmul(w.flatten(1), x.flatten(2)) + b.view(1, -1, 1)).view(B, C, H, W)
print('max diff', (y1 - y2).abs().max().item())
print(f'conv2d 1x1 fwd+bwd {t(conv):.0f} us; matmul {t(mm):.0f} us; baddbmm {t(mm_addmm):.0f} us')
bn = torch.nn.BatchNorm2d(C).to(dev).train()
def bnf():
    x.grad = None
    bn(x).backward(g)
print(f'batchnorm train fwd+bwd {t(bnf):.0f} us')
