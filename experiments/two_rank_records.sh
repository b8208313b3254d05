#!/bin/bash
# Two ranks sharing the one GPU of a development box over gloo, WITH the end-to-end leg (the only one that has a collective):
# eager and --ddp-graph, and a bucket-size sweep.  Plumbing evidence only -- gloo stages through the host, the numbers say nothing
# about RCCL over xGMI.   -> gpurun_out/two_ranks/*.json
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/two_ranks
rm -rf $OUT && mkdir -p $OUT
cd $R
common="--gpus 2 --dist-backend gloo --steps 3 --warmup 1 --batch 1 --cpu-samples 0 --no-operator --no-dhdl --fresh-procs 0 --repeats 1"
python bench.py $common --bucket-mb 64 2>$OUT/eager_b64.err | grep '^{' > $OUT/eager_bucket64.json
python bench.py $common --bucket-mb 64 --ddp-graph 2>$OUT/graph_b64.err | grep '^{' > $OUT/ddp_graph_bucket64.json
python bench.py $common --bucket-mb 25 2>$OUT/eager_b25.err | grep '^{' > $OUT/eager_bucket25.json
python bench.py $common --bucket-mb 128 2>$OUT/eager_b128.err | grep '^{' > $OUT/eager_bucket128.json
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/two_ranks/*.json')):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), 'unreadable', e); continue
    e = d.get('e2e', {})
    print(os.path.basename(f), 'n_gpus', d['n_gpus'], 'backend', d['distributed']['backend'], 'world', d['distributed']['world_size'])
    for tag in ('fp32', 'fp16'):
        r = e.get(tag, {})
        print('  ', tag, {k: r.get(k) for k in ('ms_per_step', 'ms_per_step_by_rank', 'ms_per_step_no_allreduce', 'exposed_allreduce_ms', 'bucket_mb', 'hip_graph', 'hip_graph_error', 'ddp_graph_requested', 'error')})
PY
tail -3 $OUT/*.err | cut -c1-300
