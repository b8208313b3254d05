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
