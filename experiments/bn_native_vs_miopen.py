"""BatchNorm2d training forward+backward: MIOpen (default dispatch) vs PyTorch's native kernels vs dhd_amd.batchnorm
(torch.backends.cudnn.flags(enabled=False) around the call), on shapes of the DHD-S dense modules."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dhd_amd.batchnorm import BatchNorm2d as HipBN
dev = torch.device('cuda:0')
shapes = [(24, 64, 128, 352), (24, 256, 64, 176), (24, 512, 32, 88), (24, 1024, 16, 44), (4, 128, 200, 200), (4, 256, 100, 100), (4, 64, 200, 200)]
for dt in (torch.float16, torch.float32):
    for shp in shapes:
        x = torch.randn(*shp, device=dev, dtype=dt, requires_grad=True)
        bn = torch.nn.BatchNorm2d(shp[1]).to(dev)
        g = torch.randn_like(x)
        res = []
        for native in (False, True):
            def step():
                with torch.backends.cudnn.flags(enabled=not native):
                    y = bn(x)
                y.backward(g)
                x.grad = None
            for _ in range(5): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): step()
            torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 30 * 1e6)
        hbn = HipBN(shp[1]).to(dev)
        def hstep():
            y = hbn(x)
            y.backward(g)
            x.grad = None
        for _ in range(5): hstep()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): hstep()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 30 * 1e6)
        gb = x.numel() * x.element_size() * 5 / 1e9   # read x twice + write y; read x, g + write gx (+1 more read)
        print(f'{str(dt)[6:]:8s} {str(shp):22s} miopen {res[0]:8.1f} us   native {res[1]:8.1f} us   hip {res[2]:8.1f} us   (~{gb * 8 / 5 / (res[2] * 1e-6) / 1e3:.1f} TB/s hip, 8 passes)')
