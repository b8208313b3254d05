#!/bin/bash
# experiments/ab_kernels.sh "<regex of kernel names>" GEOMETRY BATCH lib1.so lib2.so ... : rocprofv3 kernel averages of the MGHS-only
# step for several builds of the library on one box (DHD_AMD_LIB)
set -u
PAT=$1; G=$2; B=$3; shift 3
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
for LIB in "$@"; do
  OUT=$R/gpurun_out/abk; rm -rf $OUT; mkdir -p $OUT
  DHD_AMD_LIB=$R/$LIB rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python $R/bench.py --steps 10 --warmup 2 --repeats 3 --fresh-procs 0 --no-dhdl --cpu-samples 0 --no-e2e ${BENCH_FLAGS:---no-sfa} --no-operator --geometry $G --batch $B 2>/dev/null | grep '^{' > $OUT/bench.json
  echo "== $LIB: $(python -c "import json;d=json.load(open('$OUT/bench.json'));print('ms_per_step', round(d['ms_per_step'],4))")"
  python - "$OUT" "$PAT" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/k_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[2], r['Name']) and float(r['AverageNs']) > 1e3 * float(__import__('os').environ.get('MIN_US', '0')):
        print(f"   {r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
done
