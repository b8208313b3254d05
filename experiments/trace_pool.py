"""Per-workgroup timeline of mghs_pool_fwd (ablation build)."""
import ctypes, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dhd_amd import _lib, mghs_op
lib = _lib.load()
lib.dhd_debug_set_trace.argtypes = [ctypes.c_void_p]
lib.dhd_debug_set_ablation.argtypes = [ctypes.c_int]
dev = torch.device('cuda', 0)
B = int(os.environ.get('B', 4))
hp = bench.HotPath(dev, B, 1000, False)
cfg = hp.cfg
band = mghs_op.height_band(hp.height, cfg['height_range'], cfg['mask_range'])
feat = mghs_op._nchw_to_nhwc(hp.feat)
mghs_op.prepare(hp.plan, hp.calib, band, hp.ws)
nblk = 200000
for mask in (0, 16, 8):
    lib.dhd_debug_set_ablation(mask)
    for _ in range(3): mghs_op.pool_forward(hp.plan, hp.depth, feat, hp.ws)
    buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
    lib.dhd_debug_set_trace(ctypes.c_void_p(buf.data_ptr()))
    torch.cuda.synchronize()
    mghs_op.pool_forward(hp.plan, hp.depth, feat, hp.ws)
    torch.cuda.synchronize()
    lib.dhd_debug_set_trace(ctypes.c_void_p(0))
    t = buf.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 1] > 0]
    t0 = t[:, 1].min()
    us = lambda c: (c - t0) / 100.0
    start, zero, gath, end = us(t[:, 1]), us(t[:, 2]), us(t[:, 3]), us(np.where(t[:, 4] > 0, t[:, 4], t[:, 3]))
    npts = t[:, 6]
    heavy = npts > 300
    print(f'--- ablate={mask}: {len(t)} blocks, kernel span {end.max():.1f} us; heavy blocks {heavy.sum()}')
    for name, m in (('heavy', heavy), ('light', ~heavy)):
        if m.sum() == 0: continue
        print(f'  {name}: zero {np.mean(zero[m]-start[m]):.2f}  gather {np.mean(gath[m]-zero[m]):.2f} (max {np.max(gath[m]-zero[m]):.2f})  '
              f'writeout {np.mean(end[m]-gath[m]):.2f} (max {np.max(end[m]-gath[m]):.2f})  start: mean {np.mean(start[m]):.1f} max {np.max(start[m]):.1f}  end max {np.max(end[m]):.1f}')
    if heavy.sum():
        g = (gath - zero)[heavy]; n = npts[heavy]
        print('  heavy gather us per 100 points:', np.round(np.percentile(100 * g / n, [10, 50, 90]), 2), ' points/blk p50,max', np.median(n), n.max())
    xcc = t[:, 5] & 0xf
    print('  blocks per xcc', np.bincount(xcc, minlength=8), ' heavy per xcc', np.bincount(xcc[heavy], minlength=8))
    hist, _ = np.histogram(end, bins=10, range=(0, end.max()))
    print('  blocks finishing per tenth of the span', hist)
