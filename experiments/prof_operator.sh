# rocprofv3 kernel statistics of the operator drop-in's kernels (bench.py --no-sfa: MGHS loop + the operator timing)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_operator
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/op -o op -- python $R/bench.py --steps 10 --warmup 3 --cpu-samples 0 --no-e2e --no-sfa 2>/dev/null | grep '^{' > $OUT/bench_op.json
cp $(find $OUT/op -name 'op_kernel_stats.csv') $OUT/operator_kernel_stats.csv
rm -rf $OUT/op
python - <<'PY'
import csv, os, json
d = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_operator/'
for r in list(csv.DictReader(open(d + 'operator_kernel_stats.csv')))[:30]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
j = json.load(open(d + 'bench_op.json'))
print({k: j['roofline_operator'][k] for k in ('forward_ms', 'backward_ms', 'python_op_fwd_bwd_ms', 'frac')})
PY
