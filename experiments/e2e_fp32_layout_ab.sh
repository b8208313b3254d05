#!/bin/bash
# DHD-S end to end in float32: MIOpen's FIND for the channels_last problems (db -> gpurun_out/miopen_fp32_cl_db), then NCHW vs channels_last
R=$GRAFT_REPO_ROOT
DB=$R/gpurun_out/miopen_fp32_cl_db
mkdir -p $DB && cp $R/dhd_amd/miopen_db/*.txt $DB/
timeout 1500 python $R/experiments/miopen_find_job.py $DB fp32 4 channels_last > $R/gpurun_out/find_fp32_cl.log 2>&1
tail -3 $R/gpurun_out/find_fp32_cl.log
export MIOPEN_USER_DB_PATH=$DB
for rep in 1 2; do
for lay in nchw channels_last; do
  python $R/bench.py --workload e2e --amp off --layout $lay --steps 6 --warmup 3 2>$R/gpurun_out/e2e_fp32_layout.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fp32 $lay', round(d['ms_per_step'],2), 'ms', round(d['value'],2), 'samples/s')"
done; done
