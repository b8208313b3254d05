"""Debug aid: poison recycled device memory (0xff = NaN) to expose reads of never-written scratch state."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dhd_amd import _lib, mix
from dhd_amd.mix import channel_spatial_stage
dev = torch.device('cuda:0')
torch.manual_seed(0)
lib = _lib.load()
c, b, h, w = 256, 1, 40, 40
hw, r = h * w, 2 * c // 16
nsc = lib.dhd_sfa_stage_scratch_bytes(b, c, hw, r)
al = lambda n: (n + 63) // 64 * 64
cc, plane = c * c, b * c * hw
names = ['wp1', 'wp2', 'wp1t', 'wp2t', 'part', 'da1', 'da2', 'tab_g2', 'tab_g1', 'dpre2', 'dh', 'ds', 'mean_part', 'g2', 'g1', 'du', 'wpart']
sizes = [2 * cc] * 4 + [b * 4 * 2 * c, b * 4 * c, b * 4 * c, b * 3 * c, b * 3 * c, b * c, b * r, b * 2 * c, b * 2 * c * 4, plane, plane, plane, 256 * cc]
offs, o = [], 0
for s in sizes:
    offs.append(o); o += al(s)
assert o * 4 == nsc, (o * 4, nsc)
st = channel_spatial_stage(2 * c).to(dev)
x = torch.randn(b, 2 * c, h, w, device=dev, requires_grad=True)
nwt = (hw + 31) // 32
snames = ['s', 'h', 'a1', 'tab_a', 'mean1', 'rstd1', 'scsh1', 'tab1', 'mean2', 'rstd2', 'scsh2', 'mask', 'y1', 'y2']
ssizes = [b * 2 * c, b * r, b * c, b * 3 * c, c, c, 2 * c, b * 3 * c, c, c, 2 * c, b * c * nwt, plane, plane]
soffs, o = [], 0
for z in ssizes:
    soffs.append(o); o += al(z)
def run(mode):
    _lib.check(lib.dhd_sfa_set_gemm_mode(mode), 'mode')
    buf = mix._stage_scratch(dev, nsc); buf.zero_()
    out = st(x); g = torch.randn_like(out)
    sv = out.grad_fn.saved_tensors[1].view(torch.float32)
    out.backward(g)
    f = buf.view(torch.float32)
    print('mode', mode, 'out', torch.isfinite(out).all().item(), 'gx', torch.isfinite(x.grad).all().item())
    i = names.index('tab_g1'); t = f[offs[i]:offs[i] + sizes[i]]
    badidx = (~torch.isfinite(t)).nonzero().flatten().tolist()
    print('  tab_g1 bad idx', badidx)
    i = names.index('part'); t = f[offs[i]:offs[i] + sizes[i]].view(-1, 2, c)
    for ch in sorted(set(k % c for k in badidx)):
        print('  ch', ch, 'part s1', t[:, 0, ch].tolist(), 's2', t[:, 1, ch].tolist())
        for n in ('mean1', 'rstd1'):
            j = snames.index(n); print('   ', n, sv[soffs[j] + ch].item())
        j = snames.index('y1'); y1 = sv[soffs[j]:soffs[j] + plane].view(b, c, hw)
        i2 = names.index('g1'); g1 = f[offs[i2]:offs[i2] + plane].view(b, c, hw)
        print('    y1 ch finite', torch.isfinite(y1[:, ch]).all().item(), y1[:, ch].abs().max().item(), 'g1 ch absmax', g1[:, ch].abs().max().item())
    x.grad = None
run(1); run(1)
