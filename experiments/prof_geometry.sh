# rocprofv3 kernel statistics of the MGHS-only step at another view-transform geometry: prof_geometry.sh dhd-l [batch]
set -u
G=${1:-dhd-l}; B=${2:-4}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$G
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/g -o g -- python $R/bench.py --steps 20 --warmup 3 --cpu-samples 0 --no-e2e --no-sfa --no-operator --fresh-procs 0 --no-dhdl --geometry $G --batch $B 2>/dev/null | grep '^{' > $OUT/bench.json
cp $(find $OUT/g -name 'g_kernel_stats.csv') $OUT/kernel_stats.csv
rm -rf $OUT/g
python - "$OUT" <<'PY'
import csv, sys, json
d = sys.argv[1]
for r in list(csv.DictReader(open(d + '/kernel_stats.csv')))[:16]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
j = json.load(open(d + '/bench.json'))
print(j['ms_per_step'], j['value'], {k: j['roofline'][k] for k in ('launch_ms', 'frac', 'frac_of_fill', 'algorithmic_bytes')}, j.get('prepare'))
PY
