set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_quick
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hp -o hp -- python $R/bench.py --steps 20 --warmup 3 --cpu-samples 0 --no-e2e --no-operator 2>/dev/null | grep '^{' > $OUT/bench_hp.json
cp $(find $OUT/hp -name 'hp_kernel_stats.csv') $OUT/hotpath_kernel_stats.csv
rm -rf $OUT/hp
python - <<'PY'
import csv, os
f = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_quick/hotpath_kernel_stats.csv'
rows = list(csv.DictReader(open(f)))
for r in rows[:40]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
