# writer / reader channel-split experiment: DHD_STREAM_SPLIT_ALL / DHD_STREAM_SPLIT_BWD select the split of every segment
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do for sp in ${SPLITS:-0,0 4,0 4,2 4,4}; do
  export DHD_STREAM_SPLIT_ALL=${sp%,*} DHD_STREAM_SPLIT_BWD=${sp#*,}
  DHD_AMD_LIB=$R/experiments/ab/libdhd_amd_X.so python $R/bench.py --steps 40 --warmup 10 --cpu-samples 0 --no-e2e --no-operator --no-sfa ${BARGS:-} 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('split $sp', 'step', round(d['ms_per_step'],4), 'stream_fwd us', round(r['launch_ms']*1e3,1), 'frac', round(r['frac'],3), 'of_fill', round(r['frac_of_fill'],3), 'bwd us', round(d['roofline_bwd']['launch_ms']*1e3,1))"
done; done
