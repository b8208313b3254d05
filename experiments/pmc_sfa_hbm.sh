#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and L2 hit counters of the SFA GEMM kernels; env GEMM_MODE, B
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  tag=$(echo $C | cut -d' ' -f1)
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_hbm_$tag
  rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "pw_" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_hbm_$tag -o p -- python $GRAFT_REPO_ROOT/experiments/sfa_only.py ${B:-4} 3 ${GEMM_MODE:-3} > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc_hbm_*/**/p_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:86]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    hbm = (2 * m.get('FETCH_SIZE', 0) + m.get('WRITE_SIZE', 0)) * 1024 / 1e6
    print(f'{k:86s} fetch(x2) {2 * m.get("FETCH_SIZE", 0) * 1024 / 1e6:7.1f} MB write {m.get("WRITE_SIZE", 0) * 1024 / 1e6:7.1f} MB  L2 hit {m.get("TCC_HIT_sum", 0):.3g} miss {m.get("TCC_MISS_sum", 0):.3g} req {m.get("TCC_REQ_sum", 0):.3g}')
PY
