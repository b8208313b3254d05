#!/bin/bash
# rocprofv3 kernel stats of the default bench (run on the GPU box through gpurun); prints per-step kernel times
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_hp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_hp -o hp -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --cpu-samples 0 "$@" 2>&1 | grep '^{' | cut -c1-220
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_hp/**/hp_kernel_stats.csv', recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    c = int(r['Calls'])
    if c < 20: continue
    per = float(r['TotalDurationNs']) / 23 / 1000; tot += per
    if per > 8: print(f"{r['Name'][:90]:90s} {c:4d} {float(r['AverageNs'])/1000:8.1f} {per:8.1f}")
print('total us/step', tot)
PY
