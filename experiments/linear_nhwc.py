"""The SFA spatial branch as linear/batch_norm on an (N = B*H*W, C) layout vs conv2d/BatchNorm2d on NCHW."""
import torch, torch.nn as nn, torch.nn.functional as F
dev = torch.device('cuda', 0)
B, C, H, W = 4, 256, 200, 200
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
seq = nn.Sequential(nn.Conv2d(C, C, 1), nn.BatchNorm2d(C), nn.ReLU(inplace=True), nn.Conv2d(C, C, 1), nn.BatchNorm2d(C)).to(dev).train()
x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
g = torch.randn(B, C, H, W, device=dev)
def nchw():
    x.grad = None; seq.zero_grad(set_to_none=True)
    seq(x).backward(g)
xn = x.detach().permute(0, 2, 3, 1).reshape(-1, C).contiguous().requires_grad_()
gn = g.permute(0, 2, 3, 1).reshape(-1, C).contiguous()
def nc():
    xn.grad = None; seq.zero_grad(set_to_none=True)
    y = F.linear(xn, seq[0].weight.flatten(1), seq[0].bias)
    y = F.batch_norm(y, seq[1].running_mean, seq[1].running_var, seq[1].weight, seq[1].bias, True, 0.1, 1e-5)
    y = F.relu(y, inplace=True)
    y = F.linear(y, seq[3].weight.flatten(1), seq[3].bias)
    y = F.batch_norm(y, seq[4].running_mean, seq[4].running_var, seq[4].weight, seq[4].bias, True, 0.1, 1e-5)
    y.backward(gn)
    return y
def cl():
    x.grad = None; seq.zero_grad(set_to_none=True)
    seq(xcl).backward(gcl)
xcl = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_()
gcl = g.contiguous(memory_format=torch.channels_last)
y1 = seq(x); y2 = nc().view(B, H, W, C).permute(0, 3, 1, 2)
print('max diff', (y1 - y2).abs().max().item())
print(f'NCHW conv/BN2d fwd+bwd {t(nchw):.0f} us;  (N,C) linear/batch_norm {t(nc):.0f} us;  channels_last conv/BN2d {t(cl):.0f} us')
