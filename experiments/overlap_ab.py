"""A/B of the overlapped forward (ablation bit 256 = sequential)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dhd_amd import _lib, mghs_op
lib = _lib.load()
lib.dhd_debug_set_ablation.argtypes = [ctypes.c_int]
dev = torch.device('cuda', 0)
for B in (4, 1):
    hp = bench.HotPath(dev, B, 1000, False)
    cfg = hp.cfg
    band = mghs_op.height_band(hp.height, cfg['height_range'], cfg['mask_range'])
    feat = mghs_op._nchw_to_nhwc(hp.feat)
    mghs_op.prepare(hp.plan, hp.calib, band, hp.ws)
    def timeit(fn, n=30):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    for rep in range(2):
        for name, mask in (('overlapped', 0), ('sequential', 256)):
            lib.dhd_debug_set_ablation(mask)
            print(f'B={B} {name:11s} fwd {timeit(lambda: mghs_op.pool_forward(hp.plan, hp.depth, feat, hp.ws)):7.1f} us', flush=True)
    lib.dhd_debug_set_ablation(0)
    del hp
