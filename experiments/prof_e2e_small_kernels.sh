#!/bin/bash
# which kernels of the DHD-S fp16 step are many and small?  (steady-state window of an eager run)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_e2e
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_e2e -o e -- python $R/bench.py --workload e2e --amp fp16 --steps 6 --warmup 4 --no-graph 2>&1 | grep '^{' | cut -c1-120
python - <<'PY'
import collections, csv, glob, os
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_e2e/**/e_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
t_end = max(int(r['End_Timestamp']) for r in rows)
win = 0.45e9
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s >= t_end - win:
        k = r['Kernel_Name'][:150]
        acc[k][0] += 1; acc[k][1] += e - s
steps = 5
print('kernels with >= 20 launches per step, by total time (per step):')
for k, (n, d) in sorted(acc.items(), key=lambda x: -x[1][1]):
    if n / steps >= 20:
        print(f'{k[:120]:120s} {n/steps:7.1f}/step avg {d/n/1e3:6.1f} us total {d/steps/1e6:6.2f} ms/step')
PY
rm -rf $R/gpurun_out/prof_e2e
