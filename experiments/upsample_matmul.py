"""Bilinear x2 (align_corners=True) upsampling of FPN_LSS's big tensor: ATen's kernel vs two interpolation-matrix GEMMs."""
import sys, time, torch, torch.nn.functional as F
dev = torch.device('cuda:0')

def interp_matrix(n_in, n_out, dtype):
    pos = torch.arange(n_out, dtype=torch.float64) * (n_in - 1) / (n_out - 1)
    lo = pos.floor().clamp(max=n_in - 2).long()
    w = (pos - lo).to(torch.float64)
    a = torch.zeros(n_out, n_in, dtype=torch.float64)
    a[torch.arange(n_out), lo] = 1 - w
    a[torch.arange(n_out), lo + 1] += w
    return a.to(dtype).to(dev)

def up_mm(x, ah, awt):
    b, c, h, w = x.shape
    y = (x.reshape(-1, w) @ awt).view(b * c, h, -1)          # along W: one flat GEMM
    return torch.matmul(ah, y).view(b, c, ah.shape[0], -1)    # along H: batched GEMM

for dtype in (torch.float16, torch.float32):
    x = torch.randn(4, 512, 100, 100, device=dev, dtype=dtype, requires_grad=True)
    ah, awt = interp_matrix(100, 200, dtype), interp_matrix(100, 200, dtype).t().contiguous()
    ref = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
    out = up_mm(x, ah, awt)
    print(dtype, 'max diff', (out.float() - ref.float()).abs().max().item())
    g = torch.randn_like(ref)
    for name, fn in (('aten', lambda: F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)), ('matmul', lambda: up_mm(x, ah, awt))):
        for it in range(13):
            if it == 3:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            x.grad = None
            fn().backward(g)
        torch.cuda.synchronize()
        print('  ', name, 'fwd+bwd ms', (time.perf_counter() - t0) / 10 * 1e3)
