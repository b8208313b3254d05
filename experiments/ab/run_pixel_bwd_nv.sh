#!/bin/bash
# mghs_pixel_bwd: one pixel per wave (NV = 1) vs the column form (NV = 2, 4), MGHS-only step, alternating
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
for nv in 1 2 4; do
  export DHD_PIXEL_BWD_NV=$nv
  for g in "dhd-l 2" "dhd-s 4"; do set -- $g
  python $R/bench.py --no-e2e --no-operator --no-sfa --cpu-samples 0 --no-dhdl --fresh-procs 0 --steps 40 --warmup 5 --repeats 5 --geometry $1 --batch $2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('nv$nv', '$1', round(d['ms_per_step'],4), 'bwd', round(d['parts']['mghs_bwd_ms']['median'],4))"
  done
done; done
