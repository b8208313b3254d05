"""Is the image encoder (ResNet-50 + CustomFPN, DHD-S, 24 images of 256 x 704, float16 autocast, forward + backward) faster in
channels_last once MIOpen has a find-db for the NHWC problems?  Runs MIOpen's FIND for the problems the committed db does not
hold (the channels_last ones), then times both layouts in one process.
usage: backbone_layout_probe.py <db dir>      (the db dir is seeded with dhd_amd/miopen_db and keeps what the find adds)"""
import os, shutil, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
db = os.path.abspath(sys.argv[1])
os.makedirs(db, exist_ok=True)
for f in os.listdir(os.path.join(root, 'dhd_amd', 'miopen_db')):
    if not os.path.exists(os.path.join(db, f)):
        shutil.copy(os.path.join(root, 'dhd_amd', 'miopen_db', f), db)
os.environ['MIOPEN_USER_DB_PATH'] = db
os.environ['DHD_NO_MIOPEN_DB'] = '1'
sys.path.insert(0, root)
import torch
import torch.nn as nn
torch.backends.cudnn.benchmark = True
import dhd_amd
from dhd_amd.detector import dhd_s_model_cfg, build_backbone, build_neck
from dhd_amd.batchnorm import BatchNorm2d

dev = torch.device('cuda:0')
cfg = dhd_s_model_cfg()


def build(cl):
    torch.manual_seed(0)
    net = nn.ModuleList([build_backbone(cfg['img_backbone']), build_neck(cfg['img_neck'])]).to(dev).train()
    if cl:
        net = net.to(memory_format=torch.channels_last)
    return net


def step(net, x):
    with torch.autocast('cuda', dtype=torch.float16):
        y = net[1](net[0](x))
        y = y[0] if isinstance(y, (list, tuple)) else y
    y.float().square().mean().backward()


res = {}
for name, cl, hip_bn in (('nchw', False, True), ('channels_last', True, False), ('channels_last+MIOPEN_NHWC_BN', True, False)):
    if 'NHWC_BN' in name:
        os.environ['PYTORCH_MIOPEN_SUGGEST_NHWC_BATCHNORM'] = '1'
    BatchNorm2d.use_hip = hip_bn
    net = build(cl)
    x = torch.randn(24, 3, 256, 704, device=dev)
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    t0 = time.time()
    for i in range(3):
        step(net, x)
        torch.cuda.synchronize()
        print(name, 'warm step', i, round(time.time() - t0, 1), 's', flush=True)
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(5):
            step(net, x)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    res[name] = sorted(ts)[len(ts) // 2]
    print(name, 'ms per fwd+bwd', [round(t, 2) for t in ts], flush=True)
    del net
    torch.cuda.empty_cache()
print(res)
print('db files:', [(f, os.path.getsize(os.path.join(db, f))) for f in os.listdir(db)])
