"""fwd/bwd pooling time of whichever library DHD_AMD_LIB points to."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dhd_amd import _lib, mghs_op
dev = torch.device('cuda', 0)
for B in [int(b) for b in os.environ.get('BS', '4').split(',')]:
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
    f = timeit(lambda: mghs_op.pool_forward(hp.plan, hp.depth, feat, hp.ws))
    b = timeit(lambda: mghs_op.pool_backward(hp.plan, hp.depth, feat, hp.out_grads, hp.ws))
    p = timeit(lambda: mghs_op.prepare(hp.plan, hp.calib, band, hp.ws))
    print(f'{os.path.basename(_lib.LIB_PATH):34s} B={B} fwd {f:7.1f} us ({hp.pool_fwd_bytes/f/1e6:6.2f} TB/s)  bwd {b:7.1f} us  prepare {p:6.1f} us', flush=True)
    del hp
