"""one shape, ten forward + backward passes of channels_last BatchNorm + ReLU (see bn_nhwc_kernels.sh): bn_nhwc_one.py hip|torch relu|add n c h w"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dhd_amd.batchnorm import BatchNorm2d
impl, mode = sys.argv[1:3]
shape = tuple(int(v) for v in sys.argv[3:7])
dev = torch.device('cuda:0')
x = torch.randn(shape, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last).requires_grad_()
res = torch.randn_like(x).requires_grad_() if mode == 'add' else None
g = torch.randn_like(x)
bn = BatchNorm2d(shape[1]).to(dev).train()
bn.use_nhwc = impl == 'hip'
for _ in range(10):
    y = bn(x, relu=mode == 'relu', residual=res)
    y.backward(g)
    x.grad = None
    if res is not None:
        res.grad = None
torch.cuda.synchronize()
