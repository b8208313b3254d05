#!/bin/bash
# DHD-L bf16 B = 2 end to end, two environments alternating: e2e_dhdl_env_ab.sh NAME_A "ENV_A=1" NAME_B "ENV_B=1"
R=$GRAFT_REPO_ROOT
run() { env $2 python $R/bench.py --workload e2e --model dhd-l --amp bf16 --batch 2 --steps 4 --warmup 3 2>$R/gpurun_out/e2e_dhdl_env.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), 'ms', round(d['value'],3), 'samples/s', 'loss', round(d['config']['final_loss'],4))"; }
for rep in 1 2; do run "$1" "$2"; run "$3" "$4"; done
tail -2 $R/gpurun_out/e2e_dhdl_env.err | cut -c1-300
