#!/bin/bash
# DHD-M (temporal stereo, B = 3) end to end in fp16: MIOpen's FIND for its channels_last problems, then NCHW (no find-db entries) vs channels_last
R=$GRAFT_REPO_ROOT
DB=$R/gpurun_out/miopen_dhdm_db
mkdir -p $DB && cp $R/dhd_amd/miopen_db/*.txt $DB/
timeout 1700 python $R/experiments/miopen_find_job.py $DB fp16 3 channels_last dhd-m > $R/gpurun_out/find_dhdm.log 2>&1
tail -3 $R/gpurun_out/find_dhdm.log | cut -c1-300
export MIOPEN_USER_DB_PATH=$DB
for rep in 1 2; do
for lay in nchw channels_last; do
  python $R/bench.py --workload e2e --model dhd-m --amp fp16 --batch 3 --layout $lay --steps 5 --warmup 3 2>$R/gpurun_out/e2e_dhdm_layout.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dhd-m fp16 $lay', round(d['ms_per_step'],2), 'ms', round(d['value'],2), 'samples/s', d['config'].get('hip_graph'))"
done; done
tail -2 $R/gpurun_out/e2e_dhdm_layout.err | cut -c1-300
