"""world_size-2 checks on CPU (gloo): sample sharding, the timing collectives bench.py uses, and
gradient averaging of the dense producer (HeightNet) under DDP -- the N>1 path of the hot path is
'shard the samples, all-reduce only parameter gradients'."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from dhd_amd import dist as D
    from dhd_amd import HeightNet
    r, _, w = D.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    lo, hi = D.shard_range(7, r, w)
    D.barrier()
    slow = D.max_over_ranks(1.0 + r)
    total = D.sum_over_ranks(hi - lo)
    # a failure on one rank is known to every rank (bench.py abandons an end-to-end leg on all ranks together and rank 0
    # reports errors it did not see itself)
    assert D.gather_errors('boom' if r == 1 else None) == ['rank 1: boom'] and D.gather_errors(None) == []
    assert D.gather_values(10 * r) == [0, 10]
    rep = D.rank_report(ms_per_step_own=1.0 + r)
    assert rep['backend'] == 'gloo' and rep['world_size'] == rep['world_size_env'] == 2 and [x['rank'] for x in rep['ranks']] == [0, 1]
    assert [x['ms_per_step_own'] for x in rep['ranks']] == [1.0, 2.0] and rep['ranks'][r]['pid'] == os.getpid()
    # DDP over the dense producer: each rank sees different samples, gradients come out averaged
    torch.manual_seed(0)
    net = HeightNet(16, 16, 9, use_dcn=False, use_aspp=False)
    ddp = torch.nn.parallel.DistributedDataParallel(net)
    g = torch.Generator().manual_seed(100 + r)
    x, mlp = torch.randn(2, 16, 4, 6, generator=g), torch.randn(1, 2, 27, generator=g)
    ddp(x, mlp).square().mean().backward()
    grad = net.depth_conv[-1].weight.grad.clone()
    # the runner's model is a wrapper whose .module is the detector; with the detector itself wrapped once more
    # (DDP) the EMA hook must unwrap both levels (core/hook/ema.py:39-40,52-53): one SGD step, one EMA update
    import logging
    from dhd_amd import build_hook

    class Wrapper(torch.nn.Module):
        def __init__(self, module):
            super().__init__()
            self.module = module

    class Runner:
        pass
    runner = Runner()
    runner.model, runner.epoch, runner.rank, runner.logger = Wrapper(ddp), 0, rank, logging.getLogger('t')
    hook = build_hook(dict(type='MEGVIIEMAHook', init_updates=10560))
    hook.before_run(runner)
    assert runner.ema_model.ema is not net and type(runner.ema_model.ema).__name__ == 'HeightNet'
    with torch.no_grad():
        for p in net.parameters():
            if p.grad is not None:
                p -= 0.1 * p.grad
    hook.after_train_iter(runner)
    ema_w = runner.ema_model.ema.depth_conv[-1].weight
    d = 0.9990 * (1 - __import__('math').exp(-10561 / 2000))
    moved = float((ema_w - net.depth_conv[-1].weight).abs().max())
    want = float((d * (net.depth_conv[-1].weight + 0.1 * grad) + (1 - d) * net.depth_conv[-1].weight - ema_w).abs().max())
    # SyncbnControlHook converts every BatchNorm, the SFA stage's two included: with more than one rank the stage's
    # statistics are cross-rank sums (the stage operator then runs in its phased form, tests/test_gpu_parity.py)
    from dhd_amd.mix import channel_spatial_stage, needs_cross_rank_statistics
    st = channel_spatial_stage(256)
    assert not needs_cross_rank_statistics(st)
    st = torch.nn.SyncBatchNorm.convert_sync_batchnorm(st)
    assert isinstance(st.spacial_leanring[1], torch.nn.SyncBatchNorm) and needs_cross_rank_statistics(st)
    assert not needs_cross_rank_statistics(st.eval())
    q.put((rank, (lo, hi), slow, total, grad.flatten().tolist(), ema_w.detach().flatten().tolist(), moved, want))  # plain data
    D.shutdown()


def test_gloo_world_size_2_sharding_timing_and_ddp():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == (0, 4) and res[1][1] == (4, 7)          # disjoint cover of the 7 samples
    assert res[0][2] == res[1][2] == 2.0                        # MAX over ranks
    assert res[0][3] == res[1][3] == 7.0
    g0, g1 = torch.tensor(res[0][4]), torch.tensor(res[1][4])
    assert torch.allclose(g0, g1) and g0.abs().sum() > 0  # averaged gradients agree
    e0, e1 = torch.tensor(res[0][5]), torch.tensor(res[1][5])
    assert torch.equal(e0, e1)                                  # the EMA copies of the replicas stay identical
    assert res[0][6] > 0 and res[0][7] < 1e-6                    # ema = d * old + (1 - d) * new


def test_shard_range_properties():
    sys.path.insert(0, ROOT)
    from dhd_amd.dist import shard_range
    for n in (0, 1, 5, 8, 33):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bench_two_ranks_gloo_stub_e2e_scaling_record():
    """`bench.py --gpus 2 --dist-backend gloo --workload e2e --stub-model` end to end WITHOUT a GPU: the self-launch through
    torch.distributed.run, the gloo process group, DDP, the one-JSON-line contract and the `e2e_scaling` record (single-GPU step
    of the same binary next to the DDP step, exposed all-reduce time, per-rank times, the reason no graph was attempted).  The
    stand-in detector (bench._StubDetector) makes the numbers meaningless -- the line says "stub" -- the plumbing is the same
    code the first RCCL run of the real detector goes through."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dist-backend', 'gloo', '--workload', 'e2e',
                          '--stub-model', '--steps', '3', '--warmup', '1', '--batch', '2'], capture_output=True, text=True, timeout=600,
                         cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['stub'] is True and d['n_gpus'] == 2 and d['steps'] == 3 and d['unit'] == 'samples/s' and d['scaling'] == 'weak'
    assert abs(d['value'] - 2 * 2 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    assert 'DDP x2' in d['config']['parallelism'] and d['config']['global_batch'] == 4
    sc = d['e2e_scaling']
    assert sc['n_gpus'] == 2 and sc['samples_per_gpu'] == 2 and sc['bucket_mb'] == 64 and 'gloo' in sc['backend']
    leg = sc['legs']['fp32']
    for k in ('ddp_ms_per_step_eager', 'single_gpu_ms_per_step_eager', 'ms_per_step_no_allreduce', 'exposed_allreduce_ms',
              'ddp_samples_per_s_eager', 'single_gpu_samples_per_s_eager'):
        assert leg[k] is not None and leg[k] >= 0, k
    assert leg['ddp_ms_per_step_graph'] is None and 'not attempted' in leg['ddp_graph_error']
    assert len(leg['ms_per_step_by_rank']) == 2
    assert abs(leg['ddp_samples_per_s_eager'] - 4 / (leg['ddp_ms_per_step_eager'] * 1e-3)) < 1e-6 * leg['ddp_samples_per_s_eager']
    dist = d['distributed']
    assert dist['backend'] == 'gloo' and dist['world_size'] == 2 and [r['rank'] for r in dist['ranks']] == [0, 1]


def test_bench_watchdog_prints_the_partial_line_and_leaves():
    """bench.Watchdog, the guard around the first DDP graph replay over RCCL: when the guarded block outlives its deadline the
    callback runs once (rank 0 prints the line from the eager measurements) and the process exits 0 without unwinding."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "with bench.Watchdog(0.5, lambda why: print('PARTIAL', why, flush=True)):\n"
            "    time.sleep(30)\n"
            "print('NOT REACHED')\n") % root
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and 'PARTIAL watchdog' in out.stdout and 'NOT REACHED' not in out.stdout
    code = code.replace('time.sleep(30)', 'pass').replace("'NOT REACHED'", "'DONE'")
    out = subprocess.run([sys.executable, '-c', code + "time.sleep(1.0)\n"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and 'PARTIAL' not in out.stdout and 'DONE' in out.stdout
