"""Stress of the launch tails of the SFA stage's backward (BnTail: per-channel tickets, agent-scope stores / loads): the stage
forward + backward at the full size, N times on the same inputs; every parameter gradient and the input gradient must be
bit-identical each time (a stale partial sum read by a channel's last workgroup would change dgamma / dbeta / the BatchNorm
backward coefficients and everything downstream)."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dhd_amd.mix import channel_spatial_stage
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda:0')
torch.manual_seed(0)
st = channel_spatial_stage(512).to(dev).train()
x = torch.randn(4, 512, 200, 200, device=dev, requires_grad=True)
g = torch.randn(4, 256, 200, 200, device=dev)
ref = None
bad = 0
for it in range(n):
    for p in st.parameters():
        p.grad = None
    x.grad = None
    with torch.no_grad():   # same BatchNorm state every time
        for m in st.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.zero_(); m.running_var.fill_(1.0); m.num_batches_tracked.zero_()
    out = st(x)
    out.backward(g)
    h = hashlib.sha256()
    for t in [x.grad] + [p.grad for p in st.parameters()]:
        h.update(t.detach().cpu().numpy().tobytes()) if t.numel() < 1 << 20 else h.update(t.detach().double().sum().cpu().numpy().tobytes() + t.detach().flatten()[::4099].cpu().numpy().tobytes())
    d = h.hexdigest()
    if ref is None:
        ref = d
    elif d != ref:
        bad += 1
print('iterations', n, 'mismatches', bad)
sys.exit(1 if bad else 0)
