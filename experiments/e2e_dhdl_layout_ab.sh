#!/bin/bash
# DHD-L (Swin-B, 512 x 1408, temporal stereo, B = 2) end to end in bf16: MIOpen's FIND for its problems in both layouts, then NCHW vs channels_last
R=$GRAFT_REPO_ROOT
DB=$R/gpurun_out/miopen_dhdl_db
mkdir -p $DB && cp $R/dhd_amd/miopen_db/*.txt $DB/
for lay in nchw channels_last; do
  timeout 1500 python $R/experiments/miopen_find_job.py $DB bf16 2 $lay dhd-l > $R/gpurun_out/find_dhdl_$lay.log 2>&1
  tail -2 $R/gpurun_out/find_dhdl_$lay.log | cut -c1-200
done
for rep in 1 2; do
for v in "nchw nodb" "nchw db" "channels_last db"; do set -- $v
  if [ $2 = db ]; then export MIOPEN_USER_DB_PATH=$DB; else unset MIOPEN_USER_DB_PATH; fi
  python $R/bench.py --workload e2e --model dhd-l --amp bf16 --batch 2 --layout $1 --steps 4 --warmup 3 2>$R/gpurun_out/e2e_dhdl_layout.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dhd-l bf16 $1 $2', round(d['ms_per_step'],2), 'ms', round(d['value'],2), 'samples/s', d['config'].get('hip_graph'))"
done; done
tail -2 $R/gpurun_out/e2e_dhdl_layout.err | cut -c1-300
