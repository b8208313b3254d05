#!/bin/bash
# DHD-S end to end, fp16 autocast, whole-step HIP graph, default layout: two environments alternating.
# usage: e2e_now_vs_env.sh "NAME_A" "ENV_A=1 ..." "NAME_B" "ENV_B=..."
R=$GRAFT_REPO_ROOT
run() { env $2 python $R/bench.py --workload e2e --amp fp16 --steps 10 --warmup 4 2>$R/gpurun_out/e2e_ab.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'samples/s', 'loss', round(d['config']['final_loss'],4), d['config']['layout'])"; }
for rep in 1 2 3; do run "$1" "$2"; run "$3" "$4"; done
tail -3 $R/gpurun_out/e2e_ab.err
