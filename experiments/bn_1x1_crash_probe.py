"""A segfault met in round 6: torch.nn.BatchNorm2d (MIOpen) in training mode on a half (N, C, 1, 1) tensor that carries channels_last
strides (what a channels_last convolution returns for a 1 x 1 map: ASPP's global-pool branch) -- each case in its own process."""
import subprocess, sys
CODE = '''
import torch
n, strides = %d, %r
bn = torch.nn.BatchNorm2d(256).cuda().train()
x = torch.randn(n, 256, 1, 1, device='cuda', dtype=torch.%s)
if strides == 'cl':
    x = x.as_strided((n, 256, 1, 1), (256, 1, 256, 256))
with torch.autocast('cuda', dtype=torch.float16, enabled=%s):
    y = bn(x.requires_grad_())
y.float().sum().backward()
torch.cuda.synchronize()
print('ok', y.dtype, tuple(y.stride()))
'''
for n in (2, 24):
    for strides in ('plain', 'cl'):
        for dt, ac in (('float16', 'True'), ('float32', 'False')):
            r = subprocess.run([sys.executable, '-c', CODE % (n, strides, dt, ac)], capture_output=True, text=True)
            print(f'N={n:2d} strides={strides:5s} {dt}: rc {r.returncode}', (r.stdout.strip() or r.stderr.strip()[-120:]).replace('\n', ' '))
