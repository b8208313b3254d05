"""How much of the MGHS output is written as all-zero segments (4 rows x 200 voxels of one (grid, sample, z), all 64 channels)?
The share of the writer's bytes that does not depend on the gather (experiments for a gather / zero-fill overlap)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch
dev = torch.device('cuda', 0)
for geom, b in (('dhd-s', 4), ('dhd-l', 2)):
    hp = bench.HotPath(dev, b, 1000, False, geom)
    outs, _, _ = hp.step(False)
    torch.cuda.synchronize()
    tot = emp = 0
    for o in outs:
        B, CZ, ny, nx = o.shape
        nz = CZ // 64
        seg = o.view(B, nz, 64, ny // 4, 4, nx).abs().amax(dim=(2, 4, 5)) == 0
        tot += seg.numel(); emp += int(seg.sum())
    print(geom, 'B', b, 'segments', tot, 'empty', emp, 'fraction %.3f' % (emp / tot))
