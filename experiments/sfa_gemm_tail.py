"""SFA stage forward+backward with (mode 1) and without (mode 2) the tail launch of the GEMMs."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dhd_amd.mix import channel_spatial_stage
from dhd_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
for c2, bs in ((512, (1, 2, 3, 4, 6, 8)), (1024, (1, 2, 3))):
    torch.manual_seed(0)
    st = channel_spatial_stage(c2).to(dev)
    for b in bs:
        x = torch.randn(b, c2, 200, 200, device=dev, requires_grad=True)
        g = torch.randn(b, c2 // 2, 200, 200, device=dev)
        line = f'C={c2 // 2} B={b}:'
        for mode in (2, 1):
            _lib.check(lib.dhd_sfa_set_gemm_mode(mode), 'mode')
            n = 20
            for it in range(n + 3):
                if it == 3:
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                for p in st.parameters(): p.grad = None
                out = st(x)
                out.backward(g)
                x.grad = None
            torch.cuda.synchronize()
            line += f'  mode {mode}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms'
        print(line, flush=True)
        del x, g
