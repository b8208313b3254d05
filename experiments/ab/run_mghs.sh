# A/B of two builds on one box, MGHS only, several batch sizes: bash experiments/ab/run_mghs.sh
R=$GRAFT_REPO_ROOT
for b in ${BATCHES:-4 3 5}; do for rep in 1 2; do for v in ${VARIANTS:-A B}; do
  DHD_AMD_LIB=$R/experiments/variants/libdhd_amd_$v.so python $R/bench.py --steps 40 --warmup 10 --cpu-samples 0 --no-e2e --no-operator --no-sfa --fresh-procs 0 --no-dhdl --repeats 5 ${GEOM:-} --batch $b 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', 'B=$b', 'step', round(d['ms_per_step'],4), 'stream_fwd us', round(r['launch_ms']*1e3,1), 'frac', round(r['frac'],3), 'of_fill', round(r['frac_of_fill'],3), 'bwd us', round(d['roofline_bwd']['launch_ms']*1e3,1))"
done; done; done
