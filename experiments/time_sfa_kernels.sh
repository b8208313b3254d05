#!/bin/bash
# per-kernel times of the SFA stage alone (rocprofv3 kernel stats); env: GEMM_MODE (0/1/2), B (batch), ALLK, DHD_AMD_LIB
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_sfa1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_sfa1 -o s -- python $GRAFT_REPO_ROOT/experiments/sfa_only.py ${B:-4} 10 ${GEMM_MODE:-1} 2>&1 | grep 'stage fwd'
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_sfa1/**/s_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'pw_' in r['Name'] or os.environ.get('ALLK'): print(f"{r['Name'][:80]:80s} {int(r['Calls']):4d} {float(r['AverageNs'])/1000:8.1f}")
PY
