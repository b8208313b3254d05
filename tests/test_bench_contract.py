"""The driver's contract with bench.py (one JSON line, named fields), checked on the GPU with a short run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_fields(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1',
                          '--no-e2e', '--cpu-samples', '1', '--repeats', '3', '--fresh-procs', '1'], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['unit'] == 'samples/s' and d['scaling'] == 'weak' and d['vs_baseline'] is None and d['dtype'] == 'f32'
    assert 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - 4 * 3 / (d['ms_per_step'] * 3e-3)) < 1e-6 * d['value']
    # the protocol of the timed region: R repeats of the K-step loop, the headline is the median loop; per-part statistics; the
    # same again from a fresh process
    assert d['repeats'] == 3 and len(d['ms_per_step_all']) == 3 and d['ms_per_step_min'] <= d['ms_per_step'] <= d['ms_per_step_max']
    for part in ('writer_ms', 'mghs_bwd_ms', 'sfa_fwd_ms', 'sfa_bwd_ms'):
        st = d['parts'][part]
        assert 0 < st['min'] <= st['median'] <= st['max']
    assert abs(d['roofline']['launch_ms'] - d['parts']['writer_ms']['median']) < 1e-12
    fp = d['fresh_processes']
    assert len(fp) == 1 and 'error' not in fp[0] and fp[0]['repeats'] == 3 and fp[0]['ms_per_step']['median'] > 0
    assert set(d['event_samples_dropped']) == {'dropped', 'total'} and d['event_samples_dropped']['dropped'] <= 0.1 * d['event_samples_dropped']['total'] + 1
    for name in ('roofline', 'roofline_bwd', 'roofline_operator', 'roofline_sfa_stage'):
        r = d[name]
        assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
        frac = r['frac_effective'] if name == 'roofline_bwd' else r['frac']   # (algorithmic bytes over time; frac_hbm = counter bytes)
        assert abs(frac - r['achieved'] / r['peak']) < 1e-9 and 0.02 < frac < 1.0, (name, frac)
        if name == 'roofline_bwd':
            assert 'frac' not in r and (r['frac_hbm'] is None or 0.02 < r['frac_hbm'] < 1.0)
        assert r['traffic'] is None or r['traffic'] > 0
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['unit'] == 'samples/s' and c['cores'] >= 1 and c['value'] > 0 and 'sample' in c
    assert d['value'] > 50 * c['value']            # sanity: the GPU path is not the CPU twin


@pytest.mark.gpu
def test_bench_refuses_more_ranks_than_gpus(gpu):
    import torch
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '1', '--warmup', '0'],
                         capture_output=True, text=True, timeout=300, cwd=ROOT,
                         env={k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')})
    assert out.returncode != 0 and 'GPU(s) are visible' in (out.stderr + out.stdout)


def test_committed_pmc_summary_is_reported_only_for_the_kernel_sources_it_was_collected_from():
    """bench.py reports `traffic` only while profiles/<round>/pmc_summary.json carries the hash of the kernel sources the
    library is built from; after a kernel edit without a fresh counter pass (profiles/collect.sh on the GPU box) the field
    must read null -- never a stale number."""
    import glob
    import warnings
    sys.path.insert(0, ROOT)
    import bench
    sha = bench.kernel_source_sha256()
    current = [f for f in glob.glob(os.path.join(ROOT, 'profiles', '*', 'pmc_summary.json'))
               if json.load(open(f)).get('source_sha256') == sha]
    if not current:
        warnings.warn('profiles/*/pmc_summary.json is stale for the current kernel sources: `traffic` reads null until '
                      'profiles/collect.sh has run on the GPU box')
        assert bench.pmc_traffic('mghs_stream_fwd', 4) is None and bench.sfa_forward_traffic(4) is None
        return
    assert bench.pmc_traffic('mghs_stream_fwd', 4) is not None
    both = [bench.pmc_traffic(k, 4) for k in ('mghs_stream_bwd', 'mghs_pixel_bwd')]
    assert None not in both
    fwd = bench.sfa_forward_traffic(4)
    assert fwd is not None and 4 * 328e6 < fwd < 2 * 4 * 328e6     # algorithmic 328 MB per sample


def test_event_mean_leaves_out_host_stalls():
    """bench.event_mean: an interval between two HIP events also contains the time the stream sat idle while the host was
    paused (a 36-39 ms garbage collection between an event and the next launch made a 26 us kernel read 1.9 ms); samples above
    three times the median are dropped, everything else is a plain mean."""
    sys.path.insert(0, ROOT)
    import bench

    class Ev:
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t

    pairs = [(Ev(0.0), Ev(0.026)) for _ in range(19)] + [(Ev(0.0), Ev(38.0))]
    assert abs(bench.event_mean(pairs) - 0.026) < 1e-12
    pairs = [(Ev(0.0), Ev(t)) for t in (0.10, 0.12, 0.14, 0.29)]      # ordinary spread: nothing dropped
    assert abs(bench.event_mean(pairs) - 0.1625) < 1e-12
    import gc
    was = gc.isenabled()
    with bench.no_gc():
        assert not gc.isenabled()
    assert gc.isenabled() == was
