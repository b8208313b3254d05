"""Per-iteration HIP-event times of the operator drop-in's direct C-ABI calls (debugging aid for bench.operator_roofline)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from dhd_amd import _lib, mghs_op

dev = torch.device('cuda', 0)
hp = bench.HotPath(dev, 4, 1000, False)
lib = _lib.load()
B = hp.B; N, D, fh, fw, Cc = hp.dims
rank, _ = mghs_op.voxel_index(hp.plan, hp.calib, 0)
pid = torch.nonzero(rank >= 0).flatten()
rb = rank[pid].long(); order = torch.argsort(rb, stable=True)
rb, rd = rb[order].int().contiguous(), pid[order].int().contiguous()
pix = (rd.long() // (D * fh * fw)) * (fh * fw) + rd.long() % (fh * fw)
rf = pix.int().contiguous()
o2 = torch.argsort(rf, stable=True)
rb2, rd2, rf2 = rb[o2].contiguous(), rd[o2].contiguous(), rf[o2].contiguous()
_, ln2 = torch.unique_consecutive(rf2, return_counts=True)
st2 = (torch.cumsum(ln2, 0) - ln2).int().contiguous(); ln2 = ln2.int().contiguous()
depth = hp.depth.view(B, N, D, fh, fw)
feat = mghs_op._nchw_to_nhwc(hp.feat).view(B, N, fh, fw, Cc)
og = torch.randn(B, 1, 200, 200, Cc, device=dev)
dgrad, fgrad = torch.empty_like(depth), torch.empty_like(feat)
s = _lib.stream_ptr(dev)
ts = []
for it in range(40):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); dgrad.zero_(); fgrad.zero_(); e[1].record()
    _lib.check(lib.dhd_bev_pool_v2_backward(_lib.ptr(og), _lib.ptr(dgrad), _lib.ptr(fgrad), _lib.ptr(depth), _lib.ptr(feat), _lib.ptr(rd2),
                                            _lib.ptr(rf2), _lib.ptr(rb2), _lib.ptr(ln2), _lib.ptr(st2), Cc, int(ln2.numel()), s), 'bwd')
    e[2].record(); ts.append(e)
torch.cuda.synchronize()
print('fills us', [round(a[0].elapsed_time(a[1]) * 1e3, 1) for a in ts])
print('kernel us', [round(a[1].elapsed_time(a[2]) * 1e3, 1) for a in ts])
