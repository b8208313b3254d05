import sys, torch, copy, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from dhd_amd import mix, _lib
from dhd_amd.mix import channel_spatial_stage
gpu = torch.device('cuda:0')
c, b, h, w = 128, 1, 8, 8
dtype = torch.float16
torch.manual_seed(1)
st = channel_spatial_stage(2 * c).to(gpu).eval()
xh = (torch.randn(b, 2 * c, h, w, device=gpu) * 0.7 + 0.1).to(dtype)
saved_holder = {}
orig = torch.empty
def spy(*a, **k):
    t = orig(*a, **k)
    if k.get('dtype') == torch.uint8 and 'saved' not in saved_holder: saved_holder['saved'] = t
    return t
torch.empty = spy
out = st(xh)
torch.empty = orig
torch.cuda.synchronize()
sv = saved_holder['saved']
hw = h * w
def al(n): return (n + 255) & ~255
o = 0
def take(n):
    global o
    at = o; o += al(n); return at
r = 2 * c // 16
S = {}
for name, n in [('s', b*2*c*4), ('h', b*r*4), ('a1', b*c*4), ('tab_a', b*3*c*4), ('mean1', c*4), ('rstd1', c*4), ('scsh1', 2*c*4), ('tab1', b*3*c*4),
                ('mean2', c*4), ('rstd2', c*4), ('scsh2', 2*c*4), ('loc1', (2*c+1)*8), ('loc2', (2*c+1)*8), ('tick', (2*c+64)*4), ('wp1t', c*c*2), ('wp2t', c*c*2),
                ('mask', b*((hw+63)//64)*(c//32)*64*4), ('y1', b*c*hw*2), ('y2', b*c*hw*2)]:
    S[name] = (take(n), n)
print('total', o, sv.numel())
def f32(name): a, n = S[name]; return sv[a:a+n].view(torch.float32)
def f16(name): a, n = S[name]; return sv[a:a+n].view(dtype)
print('s nan', f32('s').isnan().sum().item(), 'ref', (f32('s').view(b, 2*c) - xh.float().mean((2,3))).abs().max().item())
print('a1 nan', f32('a1').isnan().sum().item())
y1 = f16('y1').view(b, c, h, w)
print('y1 nan', y1.isnan().sum().item(), y1.float().abs().max().item())
a1 = f32('a1').view(b, c, 1, 1)
u = a1 * xh[:, :c].float() + (1 - a1) * xh[:, c:].float()
y1r = torch.nn.functional.conv2d(u, st.spacial_leanring[0].weight, st.spacial_leanring[0].bias)
print('y1 err', (y1.float() - y1r).abs().max().item())
print(y1[0, :4, 0, :8]); print(y1r[0, :4, 0, :8])
y2 = f16('y2').view(b, c, h, w)
print('y2 nan', y2.isnan().sum().item())
print('out nan', out.isnan().sum().item())
idx = y2.isnan().nonzero()
print(idx)
sc1 = f32('scsh1'); print('scsh1', sc1[:4], sc1[c:c+4])
z = torch.relu(y1.float() * sc1[:c].view(1, c, 1, 1) + sc1[c:].view(1, c, 1, 1))
y2r = torch.nn.functional.conv2d(z, st.spacial_leanring[3].weight, st.spacial_leanring[3].bias)
ok = ~y2.isnan()
print('y2 err (non-nan)', (y2.float() - y2r)[ok].abs().max().item())
for i in idx[:4]:
    print(i.tolist(), y2r[tuple(i.tolist())].item())
m = sv[S['mask'][0]:S['mask'][0] + S['mask'][1]].view(torch.int32)
print('mask words', m.numel(), m[:8])
