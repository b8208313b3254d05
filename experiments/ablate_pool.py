"""Price the phases of the pooling kernels (needs `make -C dhd_amd/csrc ablate`).
usage: DHD_AMD_LIB=dhd_amd/csrc/libdhd_amd_ablate.so python experiments/ablate_pool.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dhd_amd import _lib, mghs_op
lib = _lib.load()
lib.dhd_debug_set_ablation.argtypes = [ctypes.c_int]
dev = torch.device('cuda', 0)
hp = bench.HotPath(dev, int(os.environ.get('B', 4)), 1000, False)
cfg = hp.cfg
band = mghs_op.height_band(hp.height, cfg['height_range'], cfg['mask_range'])
feat = mghs_op._nchw_to_nhwc(hp.feat)
mghs_op.prepare(hp.plan, hp.calib, band, hp.ws)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, mask in (('full', 0), ('fwd stream: no table load', 8), ('fwd stream: no stores', 16), ('fwd stream: zero stream only (no patches)', 32), ('bwd entry: no feat atomics', 1)):
    lib.dhd_debug_set_ablation(mask)
    f = timeit(lambda: mghs_op.pool_forward(hp.plan, hp.depth, feat, hp.ws))
    b = timeit(lambda: mghs_op.pool_backward(hp.plan, hp.depth, feat, hp.out_grads, hp.ws))
    print(f'{name:36s} fwd {f:8.1f} us   bwd {b:8.1f} us', flush=True)
lib.dhd_debug_set_ablation(0)
for name, fn in (('prepare', lambda: mghs_op.prepare(hp.plan, hp.calib, band, hp.ws)),
                 ('memset 704MB', lambda: [o.zero_() for o in hp.out_grads])):
    print(f'{name:36s} {timeit(fn):8.1f} us', flush=True)
