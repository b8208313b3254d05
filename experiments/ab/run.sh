# A/B of two builds of the library on one box: bash experiments/ab/run.sh [bench args]
# (experiments/ab/libdhd_amd_A.so, libdhd_amd_B.so: built by hand from two source states; not committed)
R=$GRAFT_REPO_ROOT
for rep in $(seq 1 ${REPS:-3}); do
for v in ${VARIANTS:-A B}; do
  DHD_AMD_LIB=$R/experiments/ab/libdhd_amd_$v.so python $R/bench.py --steps 40 --warmup 10 --cpu-samples 0 --no-e2e --no-operator "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d.get('roofline_sfa_stage',{}); print('$v', round(d['ms_per_step'],4), round(d.get('ms_per_step_bf16x6',0),4), 'sfa fwd', round(s.get('launch_ms',0),4), 'bwd', round(s.get('backward_ms',0),4), 'stream_fwd', round(d['roofline']['launch_ms'],4))"
done; done
