"""Accuracy (vs float64 PyTorch) and speed of the SFA stage under the GEMM precisions (dhd_sfa_weights.gemm, per call):
f32 (f32 MFMA), bf16x6, bf16x3 (default)."""
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
for mode, name in (('f32', 'f32 MFMA'), ('bf16x6', 'bf16x6'), ('bf16x3', 'bf16x3'), (None, 'torch fp32')):
    m = copy.deepcopy(st)
    xx = x.clone().requires_grad_()
    if mode is None:
        o = plain(m, xx)
    else:
        m.gemm = mode
        o = m(xx)
    o.backward(g)
    eo = (o.double() - od).abs().max().item()
    d = (xx.grad.double() - xd.grad).abs()
    med = d.flatten()[::97].median().item()
    frac = (d > 1e-4).float().mean().item()
    w1 = m.spacial_leanring[0].weight.grad.double() - ref.spacial_leanring[0].weight.grad
    w2 = m.spacial_leanring[3].weight.grad.double() - ref.spacial_leanring[3].weight.grad
    print(f'{name:10s} out max err {eo:.2e} | gx median err {med:.2e}, frac>1e-4 {frac:.2e} | dW1 rel {w1.norm().item()/ref.spacial_leanring[0].weight.grad.norm().item():.2e} dW2 rel {w2.norm().item()/ref.spacial_leanring[3].weight.grad.norm().item():.2e}')
for mode in ('f32', 'bf16x6', 'bf16x3'):
    st.gemm = mode
    xx = torch.randn(4, 512, 200, 200, device=dev, requires_grad=True)
    gg = torch.randn(4, 256, 200, 200, device=dev)
    for it in range(12):
        if it == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        o = st(xx); o.backward(gg); xx.grad = None
    torch.cuda.synchronize()
    print('mode', mode, 'stage fwd+bwd B=4 ms', (time.perf_counter() - t0) / 10 * 1e3)
