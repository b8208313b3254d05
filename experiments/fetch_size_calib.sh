#!/bin/bash
# usage (gpurun): bash experiments/fetch_size_calib.sh  -> gpurun_out/fetch_size_calib.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/fetch_size_calib.txt
$R/experiments/fetch_size_calib.bin > $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fsc
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/fsc -o c -- $R/experiments/fetch_size_calib.bin > /dev/null 2>&1
  python - "$C" >> $OUT <<'PY'
import collections, csv, glob, sys
f = glob.glob('/tmp/fsc/**/c_counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == sys.argv[1]:
        acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(f'{sys.argv[1]:11s} {k:18s} launches {len(v)}  mean {sum(v)/len(v):12.1f} KB  = {sum(v)/len(v)/786432.0:6.3f} of the 786432 KB moved')
PY
done
cat $OUT
