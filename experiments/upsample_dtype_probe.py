"""Which dtype do the nn.Upsample modules of DHD-S see under fp16 autocast?  """
import sys, torch
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device('cuda:0')
job = bench.EndToEnd(dev, 1, 1000, 1, 'fp16', 'dhd-s', False, graph=False)
for n, m in job.model.named_modules():
    if isinstance(m, torch.nn.Upsample):
        m.register_forward_hook(lambda mod, a, o, n=n: print('upsample', n, a[0].dtype, tuple(a[0].shape), '->', o.dtype, tuple(o.shape)))
with torch.autocast('cuda', dtype=torch.float16):
    job.model(return_loss=True, **job.kw)
