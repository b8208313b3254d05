#!/bin/bash
# kernel statistics of the DHD-S fp16 step (eager, default layout): the 60 largest kernels, ms per step
# usage (gpurun): bash experiments/prof_e2e_top.sh [extra bench args]   -> gpurun_out/e2e_top.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_top
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_top -o e -- python $R/bench.py --workload e2e --amp fp16 --steps 6 --warmup 2 --no-graph "$@" > /dev/null 2>&1
cp $(find $R/gpurun_out/prof_top -name 'e_kernel_stats.csv') $R/gpurun_out/e2e_top.csv
rm -rf $R/gpurun_out/prof_top
python - > $R/gpurun_out/e2e_top.txt <<'PY'
import csv, os
R = os.environ['GRAFT_REPO_ROOT']
rows = list(csv.DictReader(open(f'{R}/gpurun_out/e2e_top.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 8e6
print('kernel time per step %.2f ms (8 steps)' % tot)
for r in rows[:60]:
    print(f"{float(r['TotalDurationNs'])/8e6:7.3f} ms ({int(r['Calls'])//8:4d} per step, avg {float(r['AverageNs'])/1e3:7.1f} us)  {r['Name'][:120]}")
PY
cat $R/gpurun_out/e2e_top.txt
