"""Is the float32-storage SFA stage of this tree bit-identical to round 4's (folding launches must not change a
bit in bf16x6 mode)?  Runs the stage in a given package tree and saves every result; run once per tree, then compare.
usage: bit_identity_vs_r4.py <package root> <out.pt>     |     bit_identity_vs_r4.py --compare a.pt b.pt"""
import sys, torch
if sys.argv[1] == '--compare':
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    print('compared', len(a), 'tensors;', 'ALL BIT-IDENTICAL' if not bad else 'DIFFERENT: %s' % bad[:8])
    sys.exit(1 if bad else 0)
sys.path.insert(0, sys.argv[1])
from dhd_amd.mix import channel_spatial_stage
dev = torch.device('cuda:0')
res = {}
for gemm in ('bf16x6', 'bf16x3'):
    for (c, b, h, w, train) in [(256, 4, 200, 200, True), (256, 2, 52, 60, True), (128, 3, 36, 40, True), (512, 1, 24, 40, True), (256, 2, 52, 60, False)]:
        torch.manual_seed(c + h)
        st = channel_spatial_stage(2 * c).to(dev).train(train)
        st.gemm = gemm
        x = (torch.randn(b, 2 * c, h, w, device=dev) * 0.7 + 0.1).requires_grad_()
        g = torch.randn(b, c, h, w, device=dev)
        out = st(x)
        out.backward(g)
        tag = f'{gemm}.{c}.{b}.{h}.{w}.{int(train)}.'
        res[tag + 'out'] = out.detach().cpu()
        res[tag + 'gx'] = x.grad.cpu()
        for k, p in st.named_parameters():
            res[tag + 'grad.' + k] = p.grad.cpu()
        for k, v in st.named_buffers():
            res[tag + 'buf.' + k] = v.cpu()
torch.save(res, sys.argv[2])
print('saved', len(res), 'tensors from', sys.argv[1])
