#!/bin/bash
# A/B of the round-5 levers on the end-to-end DHD-S fp16 step, one box, alternating: gpurun_out/e2e_levers_ab.txt
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/e2e_levers_ab.txt
: > $out
run() { echo "== $1" >> $out; env $2 python $R/bench.py --workload e2e --amp fp16 --steps 10 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), 'samples/s', round(d['ms_per_step'],2), 'ms/step')" >> $out; }
for rep in 1 2; do
run "all levers (find-db, half upsample, half-storage SFA)" "X=1"
run "without the MIOpen find-db" "DHD_NO_MIOPEN_DB=1"
run "with autocast's float32 upsample" "DHD_PLAIN_UPSAMPLE=1"
run "with round 4's SFA stage (half edges, float32 storage)" "DHD_SFA_HALF_STORAGE=0"
run "none of the three (round 4's configuration)" "DHD_NO_MIOPEN_DB=1 DHD_PLAIN_UPSAMPLE=1 DHD_SFA_HALF_STORAGE=0"
done
cat $out
