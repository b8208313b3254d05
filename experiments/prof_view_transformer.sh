#!/bin/bash
# Kernel-time breakdown of img_view_transformer (MGHS.forward fwd+bwd, DHD-S fp16, B = 4) run alone on the inputs of a real step:
# rocprofv3 kernel trace of experiments/view_transformer_alone.py, the eager steady-state window (after its 0.3 s pause) grouped
# by what the kernels are.  usage (gpurun): bash experiments/prof_view_transformer.sh [tag] -> gpurun_out/view_transformer_breakdown[_tag].txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:+_$1}
OUT=$R/gpurun_out/view_transformer_breakdown$TAG.txt
ITERS=20
rm -rf $R/gpurun_out/prof_vt
python $R/experiments/view_transformer_alone.py --iters $ITERS 2>/dev/null | tail -2 > $OUT      # un-profiled timing first
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_vt -o v -- python $R/experiments/view_transformer_alone.py --iters $ITERS --no-graph > /dev/null 2>&1
ITERS=$ITERS python - >> $OUT <<'PY'
import collections, csv, glob, os, re
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_vt/**/v_kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))))
iters = int(os.environ['ITERS'])
# the timed window starts after the LAST pause of >= 0.25 s between two kernels
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 0.25e9:
        cut = i
rows = rows[cut:]
cats = [('MGHS lift + pool (csrc/mghs_*.hip)', r'mghs_|lift'),
        ('DCN sampling (csrc/deform.hip)', r'deform_'),
        ('depth / height softmax + band (csrc/lift.hip dh_softmax_*)', r'dh_softmax'),
        ('softmax (torch)', r'[Ss]oft[Mm]ax'),
        ('convolution / GEMM (MIOpen, hipBLASLt, CK)', r'igemm|Cijk_|Winograd|SP3AsmConv|miopenSp3AsmConv|gemm|xdlops|Conv.*Xdl|naive_conv|wrw|DeviceGroupedConv|kernel_grouped_conv|conv_'),
        ('layout transposes (MIOpen batched_transpose, transpose_batched)', r'batched_transpose|transpose_'),
        ('batch norm (dhd bn_* kernels, MIOpen)', r'BatchNorm|bn_'),
        ('casts half <-> float', r'float16_copy|float16tofloat32|bfloat16_copy|copy_kernel|direct_copy'),
        ('reductions (torch)', r'reduce_kernel'),
        ('element-wise (torch)', r'elementwise|vectorized'),
        ('fill / memset / copyBuffer', r'fillBuffer|FillFunctor|copyBuffer')]
acc = collections.OrderedDict((c, [0, 0.0]) for c, _ in cats)
acc['other'] = [0, 0.0]
per = collections.Counter(); cnt = collections.Counter()
tot = 0.0
for s, e, k in rows:
    for c, pat in cats:
        if re.search(pat, k):
            acc[c][0] += 1; acc[c][1] += e - s
            break
    else:
        acc['other'][0] += 1; acc['other'][1] += e - s
    per[k[:110]] += e - s; cnt[k[:110]] += 1
    tot += e - s
span = rows[-1][1] - rows[0][0]
print(f'eager window: {len(rows)} kernels in {span/1e6:.1f} ms = {iters} x forward+backward; kernel time {tot/1e6/iters:.3f} ms per fwd+bwd, '
      f'{len(rows)/iters:.0f} launches per fwd+bwd, GPU busy {tot/span:.2f} of the eager window')
for c, (n, d) in acc.items():
    print(f'  {c:66s} {n/iters:7.1f} launches {d/1e6/iters:8.3f} ms {100*d/tot:5.1f} %')
print('top kernels (ms per fwd+bwd, launches per fwd+bwd):')
for k, d in per.most_common(28):
    print(f'  {d/1e6/iters:7.3f} ms {cnt[k]/iters:6.1f} x  {k}')
PY
rm -rf $R/gpurun_out/prof_vt
cat $OUT
