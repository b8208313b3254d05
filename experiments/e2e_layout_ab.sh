#!/bin/bash
# DHD-S end to end, fp16 autocast, whole-step HIP graph: dense modules in NCHW vs channels_last (detector.use_channels_last), alternating.
# Stage 1 runs MIOpen's FIND for the NHWC problems the committed db does not hold yet and leaves the db in gpurun_out/miopen_cl_db2.
R=$GRAFT_REPO_ROOT
DB=$R/gpurun_out/miopen_cl_db2
LAYOUT=${1:-channels_last}
mkdir -p $DB && cp $R/dhd_amd/miopen_db/*.txt $DB/
timeout 2200 python $R/experiments/miopen_find_job.py $DB fp16 4 $LAYOUT > $R/gpurun_out/find_cl.log 2>&1
tail -5 $R/gpurun_out/find_cl.log
export MIOPEN_USER_DB_PATH=$DB
for rep in 1 2; do
for lay in nchw $LAYOUT; do
  python $R/bench.py --workload e2e --amp fp16 --layout $lay --steps 10 --warmup 4 2>$R/gpurun_out/e2e_layout_$lay.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lay', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'samples/s', d.get('graph'), d.get('loss'))"
done; done
tail -3 $R/gpurun_out/e2e_layout_$LAYOUT.err
