#!/bin/bash
# the remaining round-5 records: MGHS-only kernel stats at the DHD-L / DHD-M geometries, end-to-end lines of DHD-M and DHD-L, GPU test log
R=$GRAFT_REPO_ROOT
cd $R
bash experiments/prof_geometry.sh dhd-l 2 > gpurun_out/geom_dhdl.txt 2>&1
bash experiments/prof_geometry.sh dhd-m 3 > gpurun_out/geom_dhdm.txt 2>&1
python bench.py --workload e2e --model dhd-m --amp fp16 --batch 3 --steps 5 --warmup 3 2>/dev/null | tail -1 > gpurun_out/e2e_dhdm_fp16.json
python bench.py --workload e2e --model dhd-l --amp bf16 --batch 2 --steps 5 --warmup 3 2>/dev/null | tail -1 > gpurun_out/e2e_dhdl_bf16.json
python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/gpu_tests.log
cut -c1-260 gpurun_out/e2e_dhdm_fp16.json gpurun_out/e2e_dhdl_bf16.json; cat gpurun_out/gpu_tests.log; tail -2 gpurun_out/geom_dhdl.txt | cut -c1-200
