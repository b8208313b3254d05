"""Accuracy (vs float64 PyTorch) and speed of the SFA stage under the GEMM modes (dhd_sfa_set_gemm_mode):
0 f32 MFMA, 2 bf16x6 streamed weights, 1 bf16x6 resident weights, 3 bf16x3 resident weights."""
import copy, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dhd_amd import _lib
from dhd_amd.mix import channel_spatial_stage
dev = torch.device('cuda:0')

def plain(st, x):
    c = st.channels
    xb, xv = torch.split(x, c, dim=1)
    a1 = st.fc(x.mean(-1).mean(-1))[:, :, None, None]
    xb1, xv1 = a1 * xb, (1 - a1) * xv
    a2 = torch.sigmoid(st.spacial_leanring(xb1 + xv1))
    return a2 * xb1 + (1 - a2) * xv1

torch.manual_seed(0)
b = 2
st = channel_spatial_stage(512).to(dev)
x = torch.randn(b, 512, 200, 200, device=dev)
g = torch.randn(b, 256, 200, 200, device=dev)
ref = copy.deepcopy(st).double()
xd = x.double().requires_grad_()
od = plain(ref, xd); od.backward(g.double())
res = {}
for mode, name in ((0, 'f32 MFMA'), (2, 'x6 stream'), (1, 'x6 resident'), (3, 'x3 resident'), (None, 'torch fp32')):
    m = copy.deepcopy(st)
    xx = x.clone().requires_grad_()
    if mode is None:
        o = plain(m, xx)
    else:
        _lib.check(_lib.load().dhd_sfa_set_gemm_mode(mode), 'mode')
        o = m(xx)
    o.backward(g)
    eo = (o.double() - od).abs().max().item()
    if mode == 2:
        keep = [o.detach().clone(), xx.grad.clone()]
    if mode == 1:
        print('   resident x6 bit-identical to streamed x6:', torch.equal(o, keep[0]), torch.equal(xx.grad, keep[1]))
    d = (xx.grad.double() - xd.grad).abs()
    med = d.flatten()[::97].median().item()
    frac = (d > 1e-4).float().mean().item()
    w1 = m.spacial_leanring[0].weight.grad.double() - ref.spacial_leanring[0].weight.grad
    w2 = m.spacial_leanring[3].weight.grad.double() - ref.spacial_leanring[3].weight.grad
    print(f'{name:10s} out max err {eo:.2e} | gx median err {med:.2e}, frac>1e-4 {frac:.2e} | dW1 rel {w1.norm().item()/ref.spacial_leanring[0].weight.grad.norm().item():.2e} dW2 rel {w2.norm().item()/ref.spacial_leanring[3].weight.grad.norm().item():.2e}')
for mode in (0, 2, 1, 3):
    _lib.check(_lib.load().dhd_sfa_set_gemm_mode(mode), 'mode')
    xx = torch.randn(4, 512, 200, 200, device=dev, requires_grad=True)
    gg = torch.randn(4, 256, 200, 200, device=dev)
    for it in range(12):
        if it == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        o = st(xx); o.backward(gg); xx.grad = None
    torch.cuda.synchronize()
    print('mode', mode, 'stage fwd+bwd B=4 ms', (time.perf_counter() - t0) / 10 * 1e3)
