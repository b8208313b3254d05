import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dhd_amd import _lib, mghs_op
lib = _lib.load()
lib.dhd_debug_set_prepare_ablation.argtypes = [ctypes.c_int]
dev = torch.device('cuda', 0)
hp = bench.HotPath(dev, 4, 1000, False)
cfg = hp.cfg
band = mghs_op.height_band(hp.height, cfg['height_range'], cfg['mask_range'])
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, m in (('full', 0), ('no counting atomics', 1)):
    lib.dhd_debug_set_prepare_ablation(m)
    print(f'{name:22s} prepare {timeit(lambda: mghs_op.prepare(hp.plan, hp.calib, band, hp.ws)):7.1f} us')
lib.dhd_debug_set_prepare_ablation(0)
