"""SFA attention stage alone (B,512,200,200) forward + backward, for rocprofv3 runs."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dhd_amd.mix import channel_spatial_stage
b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda:0')
torch.manual_seed(0)
st = channel_spatial_stage(512).to(dev)
if len(sys.argv) > 3:
    st.gemm = sys.argv[3]           # bf16x3 | bf16x6 | f32 (dhd_sfa_weights.gemm, per call)
x = torch.randn(b, 512, 200, 200, device=dev, requires_grad=True)
g = torch.randn(b, 256, 200, 200, device=dev)
for it in range(n + 2):
    if it == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    out = st(x)
    out.backward(g)
    x.grad = None
torch.cuda.synchronize()
print('stage fwd+bwd ms', (time.perf_counter() - t0) / n * 1e3)
