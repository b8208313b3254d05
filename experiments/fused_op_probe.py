"""Host / device time of bev_pool_v2(..., fused=True) on the benchmark's full-height grid (run under rocprofv3 for the kernels)."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from dhd_amd import mghs_op
from dhd_amd.bev_pool_v2 import bev_pool_v2

dev = torch.device('cuda:0')
hp = bench.HotPath(dev, 4, 0, False)
B = hp.B
N, D, fh, fw, Cc = hp.dims
rank, _ = mghs_op.voxel_index(hp.plan, hp.calib, 0)
pid = torch.nonzero(rank >= 0).flatten()
rb = rank[pid].long()
order = torch.argsort(rb, stable=True)
rb, rd = rb[order].int().contiguous(), pid[order].int().contiguous()
rf = ((rd.long() // (D * fh * fw)) * (fh * fw) + rd.long() % (fh * fw)).int().contiguous()
_, ln = torch.unique_consecutive(rb, return_counts=True)
st = (torch.cumsum(ln, 0) - ln).int().contiguous()
ln = ln.int().contiguous()
depth = hp.depth.view(B, N, D, fh, fw).clone().requires_grad_()
feat = mghs_op._nchw_to_nhwc(hp.feat).view(B, N, fh, fw, Cc).clone().requires_grad_()
shape = (B, 1, 200, 200, Cc)
og = torch.randn(B, Cc, 1, 200, 200, device=dev)
for fused in (False, True):
    for steps, sync in ((50, False), (50, True)):
        for it in range(5 + steps):
            if it == 5:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                tf = 0.0
            depth.grad = feat.grad = None
            t1 = time.perf_counter()
            out = bev_pool_v2(depth, feat, rd, rf, rb, shape, st, ln, fused=fused)
            if sync:
                torch.cuda.synchronize()
            t2 = time.perf_counter()
            out.backward(og)
            if sync:
                torch.cuda.synchronize()
            if it >= 5:
                tf += t2 - t1
        torch.cuda.synchronize()
        tot = (time.perf_counter() - t0) / steps * 1e6
        print('fused=%s sync=%s: total %.1f us per call, forward part %.1f us' % (fused, sync, tot, tf / steps * 1e6))

import cProfile, pstats
def loop(n):
    for _ in range(n):
        depth.grad = feat.grad = None
        bev_pool_v2(depth, feat, rd, rf, rb, shape, st, ln, fused=True).backward(og)
    torch.cuda.synchronize()
loop(20)
pr = cProfile.Profile()
pr.enable()
loop(500)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)

# host cost of the C call alone (5 launches + a memset) and of the Python around it
from dhd_amd import _lib
from dhd_amd import bev_pool_v2 as _  # noqa
import importlib
bm = importlib.import_module('dhd_amd.bev_pool_v2')
lib = _lib.load()
sizes = bm._fused_sizes(lib, Cc, B, 1, 200, 200, ln.numel())
out = torch.empty(B, Cc, 1, 200, 200, device=dev)
state = torch.empty(sizes[0], dtype=torch.uint8, device=dev)
scratch = torch.empty(sizes[1], dtype=torch.uint8, device=dev)
d0, f0 = depth.detach(), feat.detach()
s = torch.cuda.current_stream(dev).cuda_stream
args = (d0.data_ptr(), f0.data_ptr(), out.data_ptr(), rd.data_ptr(), rf.data_ptr(), rb.data_ptr(), ln.data_ptr(), st.data_ptr(), Cc, ln.numel(),
        B, 1, 200, 200, state.data_ptr(), sizes[0], scratch.data_ptr(), sizes[1], s)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        lib.dhd_bev_pool_v2_fused_forward(*args)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('C forward call: host %.1f us per call (200 calls queued), drained after %.1f us per call' % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
t0 = time.perf_counter()
for _ in range(2000):
    torch.empty(B, Cc, 1, 200, 200, device=dev)
print('torch.empty: %.2f us' % ((time.perf_counter() - t0) / 2000 * 1e6))
t0 = time.perf_counter()
for _ in range(2000):
    torch.cuda.current_stream(dev).cuda_stream
print('current_stream: %.2f us' % ((time.perf_counter() - t0) / 2000 * 1e6))
t0 = time.perf_counter()
for _ in range(2000):
    torch.zeros(1000000, device=dev)
torch.cuda.synchronize()
print('torch.zeros 4 MB: %.2f us' % ((time.perf_counter() - t0) / 2000 * 1e6))
