set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_sfa_half
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for dt in fp16 bf16; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$dt -o p -- python $R/experiments/sfa_half.py 4 20 $dt 1 > $OUT/run_$dt.log 2>&1
cp $(find $OUT/$dt -name 'p_kernel_stats.csv') $OUT/kernel_stats_$dt.csv
rm -rf $OUT/$dt
tail -1 $OUT/run_$dt.log
python - $OUT/kernel_stats_$dt.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
done
python $R/experiments/sfa_half.py 4 20 fp16 0
