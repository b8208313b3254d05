"""Debug aid: where does the fused SFA stage's input gradient differ from plain PyTorch, and is every
such pixel a ReLU tie (a pre-activation within rounding of zero)?"""
import copy, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dhd_amd.mix import channel_spatial_stage
gpu = torch.device('cuda:0')
for (c, b, h, w, seed) in [(128, 3, 36, 40, 164), (256, 1, 200, 200, 1)]:
    torch.manual_seed(seed)
    st = channel_spatial_stage(2 * c).to(gpu)
    if c == 128:
        with torch.no_grad():
            for bn in (st.spacial_leanring[1], st.spacial_leanring[4]):
                bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
                bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.5, 1.5)
        x = (torch.randn(b, 2 * c, h, w, device=gpu) * 0.7 + 0.1).requires_grad_()
    else:
        x = torch.randn(b, 2 * c, h, w, device=gpu, requires_grad=True)
    ref = copy.deepcopy(st)
    out = st(x); g = torch.randn_like(out); out.backward(g)
    x2 = x.detach().clone().requires_grad_()
    xb, xv = torch.split(x2, c, dim=1)
    a1 = ref.fc(x2.mean(-1).mean(-1))[:, :, None, None]
    xb1, xv1 = a1 * xb, (1 - a1) * xv
    sp = ref.spacial_leanring
    z = sp[1](sp[0](xb1 + xv1))            # pre-ReLU
    a2 = torch.sigmoid(sp[4](sp[3](torch.relu(z))))
    o2 = a2 * xb1 + (1 - a2) * xv1
    o2.backward(g)
    e = (x.grad - x2.grad).abs()
    bad = (e > 1e-4).nonzero()
    pix = torch.unique(bad[:, [0, 2, 3]], dim=0)
    print(c, b, h, w, 'bad elements', bad.shape[0], 'bad pixels', pix.shape[0])
    for (bb, yy, xx) in pix.tolist():
        zz = z[bb, :, yy, xx].detach().abs()
        print('   pixel', bb, yy, xx, 'min |pre-ReLU| over channels', zz.min().item(), 'second', zz.sort().values[1].item())
