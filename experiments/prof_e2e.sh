#!/bin/bash
# rocprofv3 kernel stats of the end-to-end DHD-S step; prints the top kernels per step
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_e2e
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_e2e -o e -- python $GRAFT_REPO_ROOT/bench.py --workload e2e --steps 6 --warmup 4 --cpu-samples 0 "$@" 2>&1 | grep '^{' | cut -c1-200
python - <<'PY'
# steady state only: kernels that start in the last 0.4 s of the run (MIOpen's algorithm search runs during warm-up)
import collections, csv, glob, os
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_e2e/**/e_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
t_end = max(int(r['End_Timestamp']) for r in rows)
win = float(os.environ.get('WIN', '0.4')) * 1e9
acc = collections.defaultdict(lambda: [0, 0.0])
tot = 0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s >= t_end - win:
        k = r['Kernel_Name'][:110]
        acc[k][0] += 1; acc[k][1] += e - s; tot += e - s
print('GPU busy fraction in the window', round(tot / win, 3))
for k, (n, d) in sorted(acc.items(), key=lambda x: -x[1][1])[:int(os.environ.get('TOP', '45'))]:
    print(f'{k:110s} {n:5d} {d/1e6:8.2f} ms {100*d/win:5.1f}%')
PY
