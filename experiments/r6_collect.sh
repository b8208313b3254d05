#!/bin/bash
# The ONE collection of round 6 (one gpurun call = one box), after the last kernel edit:
#   profiles/collect.sh            kernel stats, PMC passes (FETCH_SIZE / WRITE_SIZE, per-kernel fetch factor), EMA, the plain bench line
#   experiments/prof_geometry.sh   MGHS-only kernel stats at the DHD-L / DHD-M geometries
#   experiments/pmc_dhdl_mghs.sh   occupancy / stall / L2 counters of the point-proportional kernels at the DHD-L geometry
#   experiments/prof_view_transformer.sh   img_view_transformer alone: eager / graph time + kernel-category breakdown
#   experiments/prof_e2e_categories.sh, prof_e2e_top.sh   the whole DHD-S fp16 step
#   DHD-M / DHD-L end-to-end lines, the GPU test log
# results: gpurun_out/profiles_new/ and gpurun_out/*.txt -> copy into profiles/r6/
R=$GRAFT_REPO_ROOT
cd $R
ROUND=r6 PREV_PROFILES=$R/profiles/r5 bash profiles/collect.sh hotpath ema > gpurun_out/collect.log 2>&1
bash experiments/prof_geometry.sh dhd-l 2 > gpurun_out/geom_dhdl.txt 2>&1
bash experiments/prof_geometry.sh dhd-m 3 > gpurun_out/geom_dhdm.txt 2>&1
bash experiments/pmc_dhdl_mghs.sh > /dev/null 2>&1
bash experiments/prof_view_transformer.sh final > /dev/null 2>&1
bash experiments/prof_e2e_categories.sh > /dev/null 2>&1
bash experiments/prof_e2e_top.sh > /dev/null 2>&1
python bench.py --workload e2e --model dhd-m --amp fp16 --batch 3 --steps 5 --warmup 3 2>/dev/null | tail -1 > gpurun_out/e2e_dhdm_fp16.json
python bench.py --workload e2e --model dhd-l --amp bf16 --batch 2 --steps 5 --warmup 3 2>/dev/null | tail -1 > gpurun_out/e2e_dhdl_bf16.json
python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/gpu_tests.log
cat gpurun_out/gpu_tests.log; tail -3 gpurun_out/collect.log | cut -c1-300; head -4 gpurun_out/view_transformer_breakdown_final.txt | cut -c1-200
