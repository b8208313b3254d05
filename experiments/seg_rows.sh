#!/bin/bash
# streaming-segment height (rows per workgroup = contiguous bytes per channel run) vs time of the two streaming kernels
for v in default 8 10 20; do
  if [ $v = default ]; then unset DHD_AMD_LIB; else export DHD_AMD_LIB=$GRAFT_REPO_ROOT/experiments/build/libdhd_seg$v.so; fi
  echo "== seg rows $v"
  python $GRAFT_REPO_ROOT/bench.py --no-sfa --cpu-samples 0 --steps 30 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],4), 'stream_fwd ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],3))"
  timeout 200 python -m pytest $GRAFT_REPO_ROOT/tests/test_gpu_parity.py -x -q -m gpu -k "golden" 2>&1 | tail -1
done
