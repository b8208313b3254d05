"""Does holding a grad-tracking output of an EARLIER eager step crash HIP graph capture with plain PyTorch ops?"""
import subprocess, sys
CODE = r'''
import sys, torch
mode = sys.argv[1]
dev = torch.device('cuda:0')
class F2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return x * 2
    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * 2
lin = torch.nn.Conv2d(8, 8, 1).to(dev)
x = torch.randn(2, 8, 20, 24, device=dev, requires_grad=True)
gy = torch.randn(2, 8, 20, 24, device=dev)
def step():
    x.grad = None
    for p in lin.parameters(): p.grad = None
    y = F2.apply(x) if mode == 'custom' else lin(x)
    y.backward(gy)
    return y
first = step()
torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): cap = step()
g.replay(); torch.cuda.synchronize(); print('OK', flush=True)
'''
for w in ('custom', 'conv'):
    res = []
    for _ in range(3):
        r = subprocess.run([sys.executable, '-c', CODE, w], capture_output=True, text=True)
        res.append((r.returncode, (r.stdout.strip().splitlines() or [''])[-1]))
    print(w, res)
