"""Occupancy losses at DHD-S size (B=4: 2.56 M voxels x 18): HIP operator vs the vectorised PyTorch formulation."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dhd_amd.detector import CrossEntropyLoss, NUSC_CLASS_FREQUENCIES, geo_scal_loss_with_mask, sem_scal_loss_with_mask
from dhd_amd.occ_loss import occ_losses
dev = torch.device('cuda:0')
m = 4 * 200 * 200 * 16
z = torch.randn(m, 18, device=dev)
t = torch.randint(0, 18, (m,), device=dev)
cam = torch.rand(m, device=dev) < 0.3
cw = torch.from_numpy((1 / np.log(NUSC_CLASS_FREQUENCIES + 0.001)).astype(np.float32)).to(dev)
t8, c8 = t.to(torch.uint8), cam.to(torch.uint8)
def hip():
    a = z.requires_grad_()
    sum(occ_losses(a, t8, c8, cw)).backward(); a.grad = None
def ref():
    a = z.requires_grad_()
    counts = torch.bincount(t[cam], minlength=256)[:18]
    avg = (counts.double() * cw.double()).sum().float()
    l = CrossEntropyLoss(class_weight=cw)(a, t, weight=cam.int(), avg_factor=avg) + sem_scal_loss_with_mask(a, t, cam.int()) + geo_scal_loss_with_mask(a, t, cam.int(), non_empty_idx=17)
    l.backward(); a.grad = None
for name, f in (('hip', hip), ('torch', ref)):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); print(name, 'fwd+bwd ms', (time.perf_counter() - t0) / 10 * 1e3)
