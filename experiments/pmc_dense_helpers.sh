#!/bin/bash
# The dense callers' helper kernels of round 5 (channels_last BatchNorm, layout transpose, bilinear upsample, Swin window rows, stereo
# cost volume) one family at a time: kernel durations (rocprofv3 --stats) and HBM traffic from two counter-only passes
# (FETCH_SIZE, WRITE_SIZE; bytes = (2 FETCH + WRITE) * 1024 as in profiles/collect.sh).   -> gpurun_out/dense_helpers_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/dense_helpers_pmc.txt
: > $OUT
for op in bn transpose upsample window cost_volume; do
  rm -rf /tmp/dh_s /tmp/dh_f /tmp/dh_w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dh_s -o s -- python $R/experiments/dense_helpers_one.py $op > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/dh_f -o f -- python $R/experiments/dense_helpers_one.py $op > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/dh_w -o w -- python $R/experiments/dense_helpers_one.py $op > /dev/null 2>&1
  python - "$op" >> $OUT <<'PY'
import collections, csv, glob, sys
pat = ('bn_cl', 'transpose_batched', 'up_fwd', 'up_bwd', 'window_rows', 'stereo_cost')
def counters(d, name):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(f'/tmp/{d}/**/*_counter_collection.csv', recursive=True)[0])):
        if r['Counter_Name'] == name and any(p in r['Kernel_Name'] for p in pat):
            acc[r['Kernel_Name'][:100]].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}
f, w = counters('dh_f', 'FETCH_SIZE'), counters('dh_w', 'WRITE_SIZE')
print(f'== {sys.argv[1]}')
for r in csv.DictReader(open(glob.glob('/tmp/dh_s/**/s_kernel_stats.csv', recursive=True)[0])):
    n = r['Name'][:100]
    if not any(p in n for p in pat):
        continue
    us = float(r['AverageNs']) / 1e3
    b = (2 * f.get(n, 0.0) + w.get(n, 0.0)) * 1024
    print(f'{us:8.1f} us  {b / 1e6:8.1f} MB  {b / us / 1e6:6.2f} TB/s  x{int(r["Calls"]) // 10} per run  {n}')
PY
done
cat $OUT
