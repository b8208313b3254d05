#!/bin/bash
# which kernels pay for a dense stack in channels_last?  kernel statistics of the DHD-S fp16 step (eager) under two layouts, largest differences
# usage (gpurun): bash experiments/prof_e2e_layout_diff.sh <layout A> <layout B>   -> gpurun_out/e2e_layout_diff.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in A B; do
  lay=$1; [ $v = B ] && lay=$2
  rm -rf $R/gpurun_out/prof_lay_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lay_$v -o e -- python $R/bench.py --workload e2e --amp fp16 --steps 6 --warmup 2 --no-graph --layout $lay > /dev/null 2>&1
  cp $(find $R/gpurun_out/prof_lay_$v -name 'e_kernel_stats.csv') $R/gpurun_out/lay_$v.csv
  rm -rf $R/gpurun_out/prof_lay_$v
done
python - "$1" "$2" > $R/gpurun_out/e2e_layout_diff.txt <<'PY'
import csv, os, sys, collections
R = os.environ['GRAFT_REPO_ROOT']
def load(v):
    d = collections.Counter(); n = collections.Counter()
    for r in csv.DictReader(open(f'{R}/gpurun_out/lay_{v}.csv')):
        k = r['Name'][:110]; d[k] += float(r['TotalDurationNs']) / 8e6; n[k] += int(r['Calls'])
    return d, n
(a, na), (b, nb) = load('A'), load('B')
print('A =', sys.argv[1], ' B =', sys.argv[2], ' (ms per step, 8 steps each)')
print('total A %.2f  B %.2f' % (sum(a.values()), sum(b.values())))
keys = sorted(set(a) | set(b), key=lambda k: -abs(b[k] - a[k]))
for k in keys[:40]:
    print(f'{b[k]-a[k]:+8.3f}  A {a[k]:7.3f} ({na[k]:5d})  B {b[k]:7.3f} ({nb[k]:5d})  {k}')
print('--- largest kernels of B')
for k in sorted(b, key=lambda k: -b[k])[:45]:
    print(f'{b[k]:7.3f} ms ({nb[k] // 8:4d} launches per step)  {k}')
PY
cat $R/gpurun_out/e2e_layout_diff.txt
