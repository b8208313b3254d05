#!/bin/bash
# A/B of the merged column / band sums launch (DHD_MGHS_MERGED_SUMS=1) at the DHD-L geometry, MGHS-only step, alternating, 3 rounds
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/merged_sums_ab.txt
python -m pytest tests/test_gpu_reference_fixtures.py tests/test_gpu_parity.py -q -k "dhdl or dhd_l or g15 or column" 2>&1 | tail -2 > $O
for i in 1 2 3; do
  for m in base merged; do
    if [ $m = base ]; then export DHD_MGHS_SEPARATE_SUMS=1; else unset DHD_MGHS_SEPARATE_SUMS; fi
    python bench.py --geometry dhd-l --batch 2 --no-sfa --child --fresh-procs 0 --cpu-samples 0 --no-e2e --no-operator --steps 20 --warmup 5 --repeats 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$m round $i: step', round(d['ms_per_step']['median'],4), 'min', round(d['ms_per_step']['min'],4), {k: round(v['median'],4) for k,v in d['parts'].items()})" >> $O
  done
done
cat $O
