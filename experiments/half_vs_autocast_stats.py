"""Ratio of the half-storage stage's error to torch.autocast's (both against float64) over several seeds: is a ratio > 1 at a
small shape noise (which pre-ReLU activations flip) or a bias?  usage: half_vs_autocast_stats.py [fp16|bf16]"""
import sys, torch, copy, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from dhd_amd.mix import channel_spatial_stage
from test_gpu_parity import _stage_errors_against_float64, _run_ours, _run_autocast
gpu = torch.device('cuda:0')
dtype = {'fp16': torch.float16, 'bf16': torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else 'fp16']
for (c, b, h, w) in [(128, 2, 10, 16), (256, 2, 16, 24), (256, 2, 52, 60)]:
    for train in (True, False):
        logs = {}
        for seed in range(12):
            torch.manual_seed(100 + seed)
            st = channel_spatial_stage(2 * c).to(gpu).train(train)
            xh = (torch.randn(b, 2 * c, h, w, device=gpu) * 0.7 + 0.1).to(dtype)
            gh = torch.randn(b, c, h, w, device=gpu).to(dtype)
            mine = _stage_errors_against_float64(st, xh, gh, _run_ours)
            auto = _stage_errors_against_float64(st, xh, gh, _run_autocast)
            for k in mine:
                if mine[k][1] > 1e-6 and auto[k][0] > 0:
                    logs.setdefault(k, []).append(math.log(mine[k][0] / auto[k][0]))
        print(dtype, c, b, h, w, 'train' if train else 'eval', ' '.join('%s %.2f(%.2f..%.2f)' % (k.replace('spacial_leanring.', 's').replace('weight', 'w').replace('bias', 'b').replace('running_', 'r'),
              math.exp(sum(v) / len(v)), math.exp(min(v)), math.exp(max(v))) for k, v in logs.items()))
