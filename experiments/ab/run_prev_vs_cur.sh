#!/bin/bash
# alternate two builds of libdhd_amd.so on one box: hot-path step (float32 headline) and the autocast hot path
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in prev cur; do
  if [ $v = prev ]; then export DHD_AMD_LIB=$R/experiments/ab/libdhd_amd_prev.so; else unset DHD_AMD_LIB; fi
  python $R/bench.py --no-e2e --no-operator --cpu-samples 0 --no-dhdl --fresh-procs 0 --steps 40 --warmup 5 --repeats 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), 'sfa fwd/bwd', round(d['parts']['sfa_fwd_ms']['median'],4), round(d['parts']['sfa_bwd_ms']['median'],4), 'amp half_io stage', round(d['hotpath_amp']['half_io']['sfa_stage_ms'],4))"
done; done
