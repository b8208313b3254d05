"""Where does whole-step HIP-graph capture of the DHD-S step stop?  Prints the Python traceback of the first op that is
not capturable (experiments only)."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dhd_amd.graph import GraphedStep
dev = torch.device('cuda:0')
job = bench.EndToEnd(dev, int(os.environ.get('B', 2)), 1000, 1, os.environ.get('AMP', 'fp16'), os.environ.get('MODEL', 'dhd-s'), True, graph=True)
for _ in range(3):
    job._eager_step()
torch.cuda.synchronize()
try:
    g = GraphedStep(job._eager_step, warmup=1)
    print('captured ok')
    for _ in range(3):
        print(float(g()))
except Exception:
    traceback.print_exc()
