import sys, torch, copy
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from dhd_amd.mix import channel_spatial_stage
from test_gpu_parity import _plain_stage
gpu = torch.device('cuda:0')
for dtype in (torch.float16, torch.bfloat16):
  for (c, b, h, w) in [(256, 2, 52, 60), (128, 3, 36, 40), (256, 2, 18, 28)]:
    for train in (False, True):
        torch.manual_seed(1)
        st = channel_spatial_stage(2 * c).to(gpu).train(train)
        xh = (torch.randn(b, 2 * c, h, w, device=gpu) * 0.7 + 0.1).to(dtype)
        gh = torch.randn(b, c, h, w, device=gpu).to(dtype)
        ref = copy.deepcopy(st).double()
        xd = xh.double().requires_grad_()
        od = _plain_stage(ref, xd); od.backward(gh.double())
        x = xh.clone().requires_grad_()
        out = st(x); out.backward(gh)
        def e(a, r): return ((a.double() - r).norm() / r.norm().clamp_min(1e-30)).item()
        print(dtype, c, b, h, w, 'train' if train else 'eval', 'out %.3e gx %.3e' % (e(out, od), e(x.grad, xd.grad)),
              ' '.join('%s %.2e' % (k.split('.')[-2][-1] + k.split('.')[-1][0], e(p.grad, q.grad)) for (k, p), q in zip(st.named_parameters(), ref.parameters())))
