#!/bin/bash
# rocprofv3 kernel trace of the end-to-end DHD-S fp16 step; steady-state window only; kernels grouped into categories
# usage (gpurun): bash experiments/prof_e2e_categories.sh [extra bench args]   -> gpurun_out/e2e_categories.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_e2e
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_e2e -o e -- python $R/bench.py --workload e2e --amp fp16 --steps 6 --warmup 4 --no-graph "$@" 2>&1 | grep '^{' | cut -c1-200 > $R/gpurun_out/e2e_categories.txt
python - >> $R/gpurun_out/e2e_categories.txt <<'PY'
import collections, csv, glob, os, re
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_e2e/**/e_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
t_end = max(int(r['End_Timestamp']) for r in rows)
win = float(os.environ.get('WIN', '0.35')) * 1e9
cats = [('convolution / GEMM (MFMA)', r'igemm|Cijk_|Winograd|SP3AsmConv|miopenSp3AsmConv|gemm|xdlops|Conv.*Xdl|naive_conv|wrw|DeviceGroupedConv|kernel_grouped_conv'),
        ('layout transposes around NHWC solvers (MIOpen) + NCHW <-> channels_last at the operators (transpose_batched)', r'batched_transpose|transpose_'),
        ('batch norm (MIOpen + dhd bn kernels)', r'BatchNorm|bn_affine|bn_train|bn_bwd|bn_stat|bn_plane|bn_cl_|bn_forward|bn_backward'),
        ('casts half <-> float', r'float16_copy|float16tofloat32|bfloat16_copy|copy_kernel'),
        ('dhd_amd HIP kernels (MGHS, SFA stage, losses, EMA)', r'mghs_|pw_gemm|pw_wgrad|blend|plane_mean|pair_sums|stage_gx|fc_forward|fc_backward|wgrad_reduce|occ_loss|label_|bin_bce|ema_update|deform_|bev_pool|lift|scan|sparse_bin|up_fwd_|up_bwd_'),
        ('optimizer / grad clip (multi-tensor)', r'multi_tensor|FusedOptimizer|Lamb|adam'),
        ('pooling / upsample / pad (torch)', r'pool|upsample|pad'),
        ('float32 element-wise / reductions', r'<float|float,|c10::Half, float'),
        ('half element-wise / reductions', r'c10::Half|__half|BFloat16')]
acc = collections.OrderedDict((c, [0, 0.0]) for c, _ in cats)
acc['other'] = [0, 0.0]
tot = 0.0
other = collections.Counter()
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s < t_end - win:
        continue
    k = r['Kernel_Name']
    for c, pat in cats:
        if re.search(pat, k):
            acc[c][0] += 1; acc[c][1] += e - s
            break
    else:
        acc['other'][0] += 1; acc['other'][1] += e - s; other[k[:80]] += e - s
    tot += e - s
print('window %.2f s of steady state (eager, no graph); GPU busy fraction %.3f' % (win / 1e9, tot / win))
for c, (n, d) in acc.items():
    print(f'{c:60s} {n:6d} launches {d/1e6:9.2f} ms {100*d/tot:5.1f} % of kernel time')
print('largest "other":', [(k, round(v / 1e6, 2)) for k, v in other.most_common(6)])
PY
rm -rf $R/gpurun_out/prof_e2e
