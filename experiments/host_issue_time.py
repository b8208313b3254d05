"""How long does the host need to ISSUE one MGHS-only hot-path step, against the GPU time of the step?"""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from dhd_amd import mghs_op
dev = torch.device('cuda', 0)
hp = bench.HotPath(dev, 4, 1000, False, 'dhd-s')
for _ in range(10):
    hp.step(False)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(50):
        hp.step(False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'issue {1e3 * (t1 - t0) / 50:.3f} ms/step, total {1e3 * (t2 - t0) / 50:.3f} ms/step')
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    hp.step(False)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
