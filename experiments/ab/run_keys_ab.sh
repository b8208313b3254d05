#!/bin/bash
# column-major vs row-major key / rank / p_slot arrays: MGHS-only step at the DHD-L (B = 2) and DHD-S (B = 4) geometries, alternating
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in rowmajor colmajor; do
  if [ $v = rowmajor ]; then export DHD_AMD_LIB=$R/experiments/ab/libdhd_amd_rowmajor.so; else unset DHD_AMD_LIB; fi
  for g in "dhd-l 2" "dhd-s 4"; do set -- $g
  python $R/bench.py --no-e2e --no-operator --no-sfa --cpu-samples 0 --no-dhdl --fresh-procs 0 --steps 40 --warmup 5 --repeats 5 --geometry $1 --batch $2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$1', round(d['ms_per_step'],4), 'lift_us', round(d['prepare']['lift_us'],1), 'static', round(d['prepare']['lift_static_us'],1), 'bwd', round(d['parts']['mghs_bwd_ms']['median'],4))"
  done
done; done
