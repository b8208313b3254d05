"""SFA attention stage alone on a half x (B,512,200,200), forward + backward, for rocprofv3 runs.
usage: sfa_half.py [batch] [iters] [fp16|bf16] [half_storage 0|1]"""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dhd_amd.mix import channel_spatial_stage
b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dt = {'fp16': torch.float16, 'bf16': torch.bfloat16}[sys.argv[3] if len(sys.argv) > 3 else 'fp16']
dev = torch.device('cuda:0')
torch.manual_seed(0)
st = channel_spatial_stage(512).to(dev)
st.half_storage = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
x = torch.randn(b, 512, 200, 200, device=dev).to(dt).requires_grad_()
g = torch.randn(b, 256, 200, 200, device=dev).to(dt)
for it in range(n + 2):
    if it == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    out = st(x)
    out.backward(g)
    x.grad = None
torch.cuda.synchronize()
print('stage fwd+bwd ms', (time.perf_counter() - t0) / n * 1e3, 'half_storage', st.half_storage, dt)
