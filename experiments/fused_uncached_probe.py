import sys, time, torch, importlib
sys.path.insert(0, '.')
import bench
from dhd_amd import mghs_op
bm = importlib.import_module('dhd_amd.bev_pool_v2')
dev = torch.device('cuda:0')
hp = bench.HotPath(dev, 4, 0, False)
B = hp.B; N, D, fh, fw, Cc = hp.dims
rank, _ = mghs_op.voxel_index(hp.plan, hp.calib, 0)
pid = torch.nonzero(rank >= 0).flatten()
rb = rank[pid].long(); order = torch.argsort(rb, stable=True)
rb, rd = rb[order].int().contiguous(), pid[order].int().contiguous()
rf = ((rd.long() // (D * fh * fw)) * (fh * fw) + rd.long() % (fh * fw)).int().contiguous()
_, ln = torch.unique_consecutive(rb, return_counts=True)
st = (torch.cumsum(ln, 0) - ln).int().contiguous(); ln = ln.int().contiguous()
depth = hp.depth.view(B, N, D, fh, fw).clone().requires_grad_()
feat = mghs_op._nchw_to_nhwc(hp.feat).view(B, N, fh, fw, Cc).clone().requires_grad_()
shape = (B, 1, 200, 200, Cc); og = torch.randn(B, Cc, 1, 200, 200, device=dev)
for name, clr in (('none', ()), ('state', ('_state_cache',)), ('regroup', ('_regroup_cache',)), ('both', ('_state_cache', '_regroup_cache'))):
    for it in range(25):
        if it == 5:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        depth.grad = feat.grad = None
        for c in clr: getattr(bm, c).clear()
        bm.bev_pool_v2(depth, feat, rd, rf, rb, shape, st, ln, fused=True).backward(og)
    torch.cuda.synchronize()
    print(name, '%.1f us' % ((time.perf_counter() - t0) / 20 * 1e6))
