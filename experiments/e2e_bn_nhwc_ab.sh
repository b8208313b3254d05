#!/bin/bash
# DHD-S end to end, fp16 autocast, whole-step HIP graph: channels_last BatchNorm (+ReLU, + residual) on the library's kernels
# (default) vs MIOpen's NHWC BatchNorm + torch's element-wise ReLU / add (DHD_BN_NO_NHWC=1), alternating; then the BEV encoder's layout again
R=$GRAFT_REPO_ROOT
D=channels_last:img_backbone,img_voxel_encoder0,img_voxel_encoder1,img_voxel_encoder2,occ_head
run() { python $R/bench.py --workload e2e --amp fp16 --layout $2 --steps 10 --warmup 4 2>$R/gpurun_out/e2e_bn_nhwc.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'samples/s', 'loss', round(d['config']['final_loss'],4))"; }
for rep in 1 2; do
  DHD_BN_NO_NHWC=1 run miopen_bn $D
  run hip_nhwc_bn $D
  run hip_nhwc_bn+bev $D,img_bev_encoder_backbone
  run hip_nhwc_bn+bev+vt $D,img_bev_encoder_backbone,img_view_transformer
done
tail -3 $R/gpurun_out/e2e_bn_nhwc.err
