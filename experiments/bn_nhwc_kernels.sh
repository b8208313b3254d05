#!/bin/bash
# GPU time (kernel durations from rocprofv3, not host-bound event timing) of one channels_last BatchNorm + ReLU forward + backward per
# shape: the library's kernels vs torch (MIOpen NHWC BatchNorm + element-wise ReLU kernels).   -> gpurun_out/bn_nhwc_kernels.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/bn_nhwc_kernels.txt
: > $OUT
IFS=';' read -ra LIST <<< "${SHAPES:-24 64 128 352;24 64 64 176;24 256 64 176;24 128 32 88;24 512 32 88;24 256 16 44;24 1024 16 44;24 512 8 22;24 2048 8 22;4 64 200 200;4 512 25 25}"
for shape in "${LIST[@]}"; do
for mode in relu add; do
for impl in ${IMPLS:-hip torch}; do
  rm -rf /tmp/bnk
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bnk -o k -- python $R/experiments/bn_nhwc_one.py $impl $mode $shape > /dev/null 2>&1
  python - "$impl" "$mode" "$shape" >> $OUT <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/bnk/**/k_kernel_stats.csv', recursive=True)[0]
tot = 0.0; parts = []
for r in csv.DictReader(open(f)):
    n = r['Name']
    if int(r['Calls']) % 10 or 'randn' in n or 'distribution' in n:   # 10 timed iterations; setup kernels run once or twice
        continue
    per = float(r['TotalDurationNs']) / 10e3
    tot += per; parts.append((per, int(r['Calls']) // 10, n[:48]))
parts.sort(reverse=True)
print(f"{sys.argv[3]:18s} {sys.argv[2]:5s} {sys.argv[1]:5s} {tot:8.1f} us  " + '  '.join(f'{p:.1f}x{c} {n}' for p, c, n in parts[:7]))
PY
done; done; done
cat $OUT
