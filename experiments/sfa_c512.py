"""SFA stage at C = 512 (DHD-M / DHD-L: x is (B,1024,200,200)): time per GEMM mode."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dhd_amd import _lib
from dhd_amd.mix import channel_spatial_stage
dev = torch.device('cuda:0')
b = int(sys.argv[1]) if len(sys.argv) > 1 else 3
torch.manual_seed(0)
st = channel_spatial_stage(1024).to(dev)
x = torch.randn(b, 1024, 200, 200, device=dev, requires_grad=True)
g = torch.randn(b, 512, 200, 200, device=dev)
for mode in ('bf16x6', 'bf16x3'):
    st.gemm = mode
    for it in range(8):
        if it == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        for p in st.parameters():
            p.grad = None
        x.grad = None
        st(x).backward(g)
    torch.cuda.synchronize()
    print('C=512 B=%d %s stage fwd+bwd ms %.3f' % (b, mode, (time.perf_counter() - t0) / 5 * 1e3))
