"""Per-wave phase clocks of the resident one-input GEMM (y2 = conv2(relu(bn1(y1)))); needs the ablation build
   make -C dhd_amd/csrc ablate ABLATE_DEFS=-DRESABL=8 ABLATE_TAG=_tl     and    DHD_AMD_LIB=.../libdhd_amd_ablate_tl.so
Columns (shader clocks, summed over the wave's steps): wait for the step's loads | prologue + split | load issue + LDS
fragment reads + MFMA issue | epilogue (LDS patch + stores) | whole kernel."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dhd_amd import _lib
from dhd_amd.mix import channel_spatial_stage
b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda:0')
lib = _lib.load()
torch.manual_seed(0)
st = channel_spatial_stage(512).to(dev).train()
x = torch.randn(b, 512, 200, 200, device=dev, requires_grad=True)
gy = torch.randn(b, 256, 200, 200, device=dev)
for _ in range(3):
    st(x).backward(gy)
    x.grad = None
torch.cuda.synchronize()
n = 4 * 256 * 8 * 8
buf = (C.c_ulonglong * n)()
lib.dhd_debug_res_timeline.argtypes = [C.c_void_p, C.c_int]
assert lib.dhd_debug_res_timeline(buf, n) == 0
tt = np.frombuffer(buf, dtype=np.uint64).reshape(4, 256, 8, 8).astype(np.float64)
names = ['wait loads', 'prologue+split', 'issue+lds+mfma', 'epilogue', 'total']
for vi, vn in enumerate(['y2 = conv2(relu(bn1 y1))  [one input, bias + stats]', 'y1 = conv1(blend x)  [two inputs, bias + stats]',
                         'dgrad conv2  [two inputs, ReLU mask]', 'dgrad conv1  [two inputs]']):
    t = tt[vi]
    steps, tiles, tot = t[..., 5], t[..., 6], t[..., 4]
    print('== ' + vn)
    print('tiles per wave %d..%d; kernel span %.0f clocks; slowest wave %.0f, mean %.0f' % (
        tiles.min(), tiles.max(), (t[..., 7] + tot).max() - t[..., 7].min(), tot.max(), tot.mean()))
    for i, nm in enumerate(names[:4]):
        v = t[..., i]
        per = v / (steps if i < 3 else tiles)
        print('  %-16s %5.1f %% of wave total   per %s: mean %7.0f  p10 %7.0f  p90 %7.0f' % (
            nm, 100 * v.mean() / tot.mean(), 'step' if i < 3 else 'tile', per.mean(), np.percentile(per, 10), np.percentile(per, 90)))
    print('  unaccounted      %5.1f %%' % (100 * (1 - t[..., :4].sum(-1).mean() / tot.mean())))
    for g in (0, 1):
        sel = t[(np.arange(256) // 8) % 2 == g]
        print('  team member %d: mean prologue %.0f, epilogue %.0f, total %.0f' % (g, sel[..., 1].mean(), sel[..., 3].mean(), sel[..., 4].mean()))
