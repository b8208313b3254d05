"""Process-group plumbing for the sample-sharded (data-parallel) hot path.

The MGHS/SFA path has no cross-sample reduction (every index carries the batch id,
models/necks/lss_heightmap.py:335-337,351-352; SFA's mean is per sample, mix.py:41), so ranks
take disjoint samples and exchange nothing on the data path.  The only collectives are the
ones a training step needs around it: a barrier + MAX of the elapsed time for measurement, and
the gradient all-reduce of the dense modules' parameters (RCCL when the backend is "nccl").
One process per GPU, launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def init_from_env(backend=None, device=None):
    """Initialise the default process group from the launcher's environment (no-op for 1 rank).
    backend: 'nccl' (= RCCL on ROCm) for GPU ranks, 'gloo' for CPU tests."""
    rank, local, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        kw = {}
        if backend == 'nccl' and device is not None:
            kw['device_id'] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) of `n_items` independent samples for `rank`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device='cpu'):
    """MAX of a python float over all ranks (the slowest rank defines the step time)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    if dist.get_backend() == 'gloo':
        device = 'cpu'
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device='cpu'):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    if dist.get_backend() == 'gloo':
        device = 'cpu'
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_values(value):
    """The python value of every rank, as a list ordered by rank (this rank's alone without a process group)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [value]
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, value)
    return got


def gather_errors(err):
    """`err`: this rank's error text or None.  Returns the list of 'rank r: text' over ALL ranks (empty = every rank is fine),
    identical on every rank: a leg that failed on any rank can be abandoned by all of them together, and rank 0 can report a
    failure it did not see itself.  To be called at points every rank reaches (not from inside a failed collective)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [] if err is None else [f'rank 0: {err}']
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, err)
    return [f'rank {r}: {e}' for r, e in enumerate(got) if e is not None]


def rank_report(**mine):
    """What a first multi-GPU run must say about itself so that a wrong topology is visible in the output: the backend as
    torch.distributed reports it (backend "nccl" is RCCL on ROCm), the world size the process group has, and one record per rank
    -- rank, local rank, device index / name / PCI bus id, RCCL's version, plus whatever the caller measured on that rank
    (`mine`: e.g. its own step time next to the MAX over ranks that defines the headline).  Identical on every rank."""
    rank, local, world_env = env_world()
    import socket
    rec = dict(rank=rank, local_rank=local, pid=os.getpid(), host=socket.gethostname())
    if torch.cuda.is_available():
        i = torch.cuda.current_device()
        prop = torch.cuda.get_device_properties(i)
        rec.update(device=i, device_name=prop.name, visible_devices=torch.cuda.device_count(),
                   pci_bus_id=getattr(prop, 'pci_bus_id', None), pci_domain_id=getattr(prop, 'pci_domain_id', None),
                   pci_device_id=getattr(prop, 'pci_device_id', None), hbm_bytes=prop.total_memory)
    rec.update(mine)
    live = dist.is_initialized()
    out = dict(backend=dist.get_backend() if live else None, world_size=dist.get_world_size() if live else 1, world_size_env=world_env,
               launcher=dict(MASTER_ADDR=os.environ.get('MASTER_ADDR'), MASTER_PORT=os.environ.get('MASTER_PORT')))
    try:
        out['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version()) if torch.cuda.is_available() else None
    except Exception as exc:  # noqa: BLE001
        out['rccl_version'] = f'unavailable ({type(exc).__name__})'
    if live and dist.get_world_size() > 1:
        got = [None] * dist.get_world_size()
        dist.all_gather_object(got, rec)
        out['ranks'] = got
        # one key type per device: (host, PCI domain, bus, device) where the runtime reports a bus id (0 is a valid bus), else
        # (host, local device index) -- bus numbers repeat across hosts and PCI domains
        def dev_key(r):
            if r.get('pci_bus_id') is not None:
                return (r.get('host'), r.get('pci_domain_id'), r.get('pci_bus_id'), r.get('pci_device_id'))
            return (r.get('host'), 'index', r.get('device'))
        out['distinct_devices'] = len({dev_key(r) for r in got})
    else:
        out['ranks'] = [rec]
        out['distinct_devices'] = 1
    return out


def shutdown():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
