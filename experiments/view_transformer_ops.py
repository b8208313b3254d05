"""torch.profiler view of img_view_transformer alone (same set-up as view_transformer_alone.py): aten ops by GPU time WITH input
shapes, to find where the casts / copies / reductions of the module come from.  usage: python experiments/view_transformer_ops.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

dev = torch.device('cuda', 0)
job = bench.EndToEnd(dev, 4, 1000, 1, 'fp16', 'dhd-s', True, graph=False)
vt = job.model.img_view_transformer
cap = {}
h = vt.register_forward_pre_hook(lambda m, args: cap.setdefault('in', args))
with torch.autocast('cuda', dtype=job.amp):
    job.model(return_loss=True, **job.kw)
h.remove()
inp = [t.detach() if torch.is_tensor(t) else t for t in cap['in'][0]]
x = inp[0].clone().requires_grad_()
grads = None


def step():
    global grads
    for q in vt.parameters():
        q.grad = None
    x.grad = None
    with torch.autocast('cuda', dtype=job.amp):
        outs = vt([x] + inp[1:])
    outs = [o for o in outs if torch.is_tensor(o) and o.requires_grad]
    if grads is None:
        grads = [torch.ones_like(o) for o in outs]
    torch.autograd.backward(outs, grads)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by='self_cuda_time_total', row_limit=70, max_name_column_width=40, max_shapes_column_width=90))
