#!/bin/bash
# DHD-S end to end, fp16 autocast, whole-step HIP graph: nn.Upsample on the library's bilinear kernels (default) vs torch's
# (DHD_PLAIN_UPSAMPLE=1), alternating; then the layouts of the remaining NCHW stacks again (their upsample / BatchNorm costs changed)
R=$GRAFT_REPO_ROOT
D=channels_last:img_backbone,img_voxel_encoder0,img_voxel_encoder1,img_voxel_encoder2,occ_head
run() { python $R/bench.py --workload e2e --amp fp16 --layout $2 --steps 10 --warmup 4 2>$R/gpurun_out/e2e_upsample.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'samples/s', 'loss', round(d['config']['final_loss'],4))"; }
for rep in 1 2; do
  DHD_PLAIN_UPSAMPLE=1 run torch_upsample $D
  run hip_upsample $D
  run hip_upsample+bev $D,img_bev_encoder_backbone
  run hip_upsample+bev+vt channels_last
done
tail -3 $R/gpurun_out/e2e_upsample.err
