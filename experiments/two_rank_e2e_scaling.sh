#!/bin/bash
# Round 6: the e2e_scaling record of `bench.py --gpus 2` with the REAL detector, two ranks sharing the one GPU of a development box
# over gloo (RCCL refuses two ranks on one device).  Plumbing evidence only: single-GPU step of the same binary, DDP eager, no_sync,
# and the reason the DDP graph is not attempted over gloo.  -> gpurun_out/two_rank_e2e_scaling.json
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --gpus 2 --dist-backend gloo --steps 3 --warmup 1 --batch 1 --cpu-samples 0 --no-operator --no-dhdl --fresh-procs 0 --repeats 1 2>gpurun_out/two_rank_e2e_scaling.err | grep '^{' > gpurun_out/two_rank_e2e_scaling.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/two_rank_e2e_scaling.json'))
print('n_gpus', d['n_gpus'], d['config']['parallelism'][:160])
print(json.dumps(d.get('e2e_scaling'), indent=1)[:3000])
PY
tail -3 gpurun_out/two_rank_e2e_scaling.err | cut -c1-300
