#!/bin/bash
# DHD-S end to end, fp16 autocast, whole-step HIP graph: which dense stacks gain from channels_last (detector.use_channels_last)?
# Uses the committed find-db (dhd_amd/miopen_db holds both the NCHW and the NHWC problems).
R=$GRAFT_REPO_ROOT
E=img_backbone
U=img_voxel_encoder0,img_voxel_encoder1,img_voxel_encoder2
for rep in 1 2; do
for lay in ${LAYOUTS:-nchw channels_last:$E channels_last:$E,img_view_transformer channels_last:$E,img_bev_encoder_backbone channels_last:$E,$U channels_last:$E,occ_head channels_last}; do
  python $R/bench.py --workload e2e --amp fp16 --layout $lay --steps 10 --warmup 4 2>$R/gpurun_out/e2e_layout_parts.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lay', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'samples/s')"
done; done
tail -3 $R/gpurun_out/e2e_layout_parts.err
