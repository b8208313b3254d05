#!/bin/bash
# A/B of dhd_amd.batchnorm.BatchNorm2d's routing thresholds on the end-to-end DHD-S fp16 step UNDER THE HIP GRAPH (round 2 tuned them
# with eager launches, where the operator's host time counts): gpurun_out/e2e_bn_routing_ab.txt
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/e2e_bn_routing_ab.txt
: > $out
run() { echo "== DHD_BN_ROUTING=$1" >> $out; DHD_BN_ROUTING=$1 python $R/bench.py --workload e2e --amp fp16 --steps 10 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), 'samples/s', round(d['ms_per_step'],2), 'ms/step', 'graph', d['config']['hip_graph'])" >> $out; }
for rep in 1 2; do
run "16777216,128,67108864"
run "4194304,128,33554432"
run "1048576,256,16777216"
run "262144,4096,4194304"
run "1,65536,1"
done
cat $out
