#!/bin/bash
# SQ counters of the SFA stage kernels (counters only, one pass), printed per kernel
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_sfa
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sfa -o p -- python $GRAFT_REPO_ROOT/experiments/sfa_only.py 4 3 ${GEMM_MODE:-3} 2>&1 | tail -2
python - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc_sfa/**/p_counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if 'pw_' not in k: continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k)
    print('   ', {c: f'{v:.3g}' for c, v in m.items()})
PY
