"""One-off job (gpurun): MIOpen's exhaustive FIND over the convolution problems of the DHD-S training step, so that the user find-db
it leaves can be committed (dhd_amd/miopen_db/) and later runs pick the measured-fastest solver in immediate mode instead of the
heuristic's choice (this image ships no gfx950 find-db at all: /opt/rocm/share/miopen/db has none).  VERDICT r4 item 2b.
usage: miopen_find_job.py <db dir> [fp16|bf16|fp32|both] [batch] [layout] [model]      (layout as bench.py --layout; the db dir is seeded with the committed db)"""
import os, sys, time
db = os.path.abspath(sys.argv[1])
os.makedirs(db, exist_ok=True)
os.environ['MIOPEN_USER_DB_PATH'] = db
os.environ.setdefault('MIOPEN_FIND_MODE', 'NORMAL')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.backends.cudnn.benchmark = True       # PyTorch then calls miopenFindConvolution*Algorithm for every new problem
import bench
which = sys.argv[2] if len(sys.argv) > 2 else 'both'
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4
layout = sys.argv[4] if len(sys.argv) > 4 else 'nchw'
model = sys.argv[5] if len(sys.argv) > 5 else 'dhd-s'
dev = torch.device('cuda:0')
for amp in (('fp16', 'off') if which == 'both' else (('off',) if which == 'fp32' else (which,))):
    t0 = time.time()
    job = bench.EndToEnd(dev, batch, 1000, 1, amp, model, True, graph=False, layout=layout)
    for i in range(2):
        job.step(False)
        torch.cuda.synchronize()
        print(amp, 'step', i, 'done after', round(time.time() - t0, 1), 's', flush=True)
    del job
    torch.cuda.empty_cache()
print('db files:', [(f, os.path.getsize(os.path.join(db, f))) for f in os.listdir(db)])
