"""img_view_transformer (MGHS.forward: depth_net, HeightNet incl. DCN, the two softmaxes, lift + pool) ALONE, forward + backward,
on the inputs it receives in a real DHD-S fp16 step (captured by a hook), in the layout / autocast dtype of the end-to-end job.
Prints eager and HIP-graph-replay time per forward+backward (HIP events).  Under `rocprofv3 --kernel-trace` the timed eager
iterations follow a 0.3 s pause, so that a script can cut the steady-state window out of the trace
(experiments/prof_view_transformer.sh -> profiles/r6/view_transformer_breakdown.txt).
usage: python experiments/view_transformer_alone.py [--iters 20] [--amp fp16|off] [--batch 4]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (sets the MIOpen find-db path before torch loads MIOpen)
import torch  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument('--iters', type=int, default=20)
p.add_argument('--amp', default='fp16')
p.add_argument('--batch', type=int, default=4)
p.add_argument('--no-graph', action='store_true')
a = p.parse_args()
dev = torch.device('cuda', 0)
job = bench.EndToEnd(dev, a.batch, 1000, 1, a.amp, 'dhd-s', True, graph=False)
vt = job.model.img_view_transformer
cap = {}
h = vt.register_forward_pre_hook(lambda m, args: cap.setdefault('in', args))
with torch.autocast('cuda', dtype=job.amp, enabled=job.amp is not None):
    job.model(return_loss=True, **job.kw)
h.remove()
inp = [t.detach() if torch.is_tensor(t) else t for t in cap['in'][0]]
x0 = inp[0]
print('input feature map', tuple(x0.shape), x0.dtype, 'channels_last' if x0.dim() == 5 and not x0[0].is_contiguous() else 'contiguous', flush=True)
x = x0.clone().requires_grad_()
grads = None


def step():
    global grads
    for q in vt.parameters():
        q.grad = None
    x.grad = None
    with torch.autocast('cuda', dtype=job.amp, enabled=job.amp is not None):
        outs = vt([x] + inp[1:])
    outs = [o for o in outs if torch.is_tensor(o) and o.requires_grad]
    if grads is None:
        grads = [torch.ones_like(o) for o in outs]
    torch.autograd.backward(outs, grads)


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for _ in range(5):
    step()
torch.cuda.synchronize()
time.sleep(0.3)
eager = timed(step, a.iters)
rec = dict(ms_fwd_bwd_eager=round(eager, 3), iters=a.iters, amp=a.amp, batch=a.batch, layout=job.layout)
if not a.no_graph:
    time.sleep(0.3)
    try:
        from dhd_amd.graph import GraphedStep
        g = GraphedStep(step, warmup=2)
        for _ in range(3):
            g()
        rec['ms_fwd_bwd_graph'] = round(timed(g, a.iters), 3)
    except Exception as exc:  # noqa: BLE001
        rec['graph_error'] = f'{type(exc).__name__}: {exc}'[:200]
print(json.dumps(rec), flush=True)
