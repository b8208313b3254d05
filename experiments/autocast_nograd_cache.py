"""Does a weight cast cached by autocast under no_grad (adjacent frames first) starve the key frame's
weights of gradients?  DHD_stereo processes the non-key frames first under torch.no_grad()."""
import torch
dev = torch.device('cuda:0')
lin = torch.nn.Linear(8, 8).to(dev)
conv = torch.nn.Conv2d(4, 4, 3, padding=1).to(dev)
x = torch.randn(4, 8, device=dev, requires_grad=True)
y = torch.randn(2, 4, 6, 6, device=dev, requires_grad=True)
for first_nograd in (False, True):
    lin.zero_grad(); conv.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        if first_nograd:
            with torch.no_grad():
                lin(x); conv(y)
        out = lin(x).float().sum() + conv(y).float().sum()
    out.backward()
    print('no_grad pass first:', first_nograd, '| linear weight grad:', None if lin.weight.grad is None else float(lin.weight.grad.abs().sum()),
          '| conv weight grad:', None if conv.weight.grad is None else float(conv.weight.grad.abs().sum()))
