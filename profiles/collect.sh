#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/profiles_new/ (run on the GPU box through gpurun,
# from the repo root); copy the results into profiles/<round>/ afterwards.
#   1. kernel stats of the default bench (MGHS + SFA stage) and of --no-sfa
#   2. PMC passes (counters only, FETCH_SIZE and WRITE_SIZE separately) of the default bench
#   3. a plain bench line outside the profiler (run last, after the PMC summary has been written)
#   3b. `bench.py --gpus 2 --dist-backend gloo`: the N > 1 path with two ranks sharing the GPU
#   4. kernel stats + the two PMC passes of `bench.py --workload ema` (its kernel is merged into pmc_summary.json)
# Usage: collect.sh [hotpath] [ema]   (default: both)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_new
WHAT="${*:-hotpath ema}"
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [[ " $WHAT " == *" hotpath "* ]]; then
B="python $R/bench.py --steps 20 --warmup 3 --cpu-samples 0 --no-e2e"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hp -o hp -- $B 2>/dev/null | grep '^{' > $OUT/bench_hotpath_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mo -o mo -- $B --no-sfa 2>/dev/null | grep '^{' > $OUT/bench_mghs_only_under_rocprof.json
P="python $R/bench.py --steps 5 --warmup 2 --pmc-pass"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pf -o pf -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pw -o pw -- $P > /dev/null 2>&1
cp $(find $OUT/hp -name 'hp_kernel_stats.csv') $OUT/hotpath_kernel_stats.csv
cp $(find $OUT/mo -name 'mo_kernel_stats.csv') $OUT/mghs_only_kernel_stats.csv
cp $(find $OUT/pf -name 'pf_counter_collection.csv') $OUT/pmc_fetch_size.csv
cp $(find $OUT/pw -name 'pw_counter_collection.csv') $OUT/pmc_write_size.csv
rm -rf $OUT/hp $OUT/mo $OUT/pf $OUT/pw
fi
cd /tmp
if [[ " $WHAT " == *" hotpath "* ]]; then
# round 5: the half-storage SFA stage alone (its kernels have their own names: no clash with the float32 hot path)
H="python $R/experiments/sfa_half.py 4 6 fp16 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hs -o hs -- python $R/experiments/sfa_half.py 4 20 fp16 1 > $OUT/sfa_half_fp16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hb -o hb -- python $R/experiments/sfa_half.py 4 20 bf16 1 > $OUT/sfa_half_bf16.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/hf -o hf -- $H > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/hw -o hw -- $H > /dev/null 2>&1
cp $(find $OUT/hs -name 'hs_kernel_stats.csv') $OUT/sfa_half_fp16_kernel_stats.csv
cp $(find $OUT/hb -name 'hb_kernel_stats.csv') $OUT/sfa_half_bf16_kernel_stats.csv
head -1 $(find $OUT/hf -name 'hf_counter_collection.csv') > $OUT/pmc_sfa_half.csv
grep -h "_h_kernel\|cuh_kernel\|wgrad_h\|wgrad_reduce\|fc_forward\|fc_backward\|bn_stats_finalize" $(find $OUT/hf -name 'hf_counter_collection.csv') $(find $OUT/hw -name 'hw_counter_collection.csv') >> $OUT/pmc_sfa_half.csv
rm -rf $OUT/hs $OUT/hb $OUT/hf $OUT/hw
fi
cd /tmp
if [[ " $WHAT " == *" ema "* ]]; then
# counters only for the EMA kernel: the detector's construction launches ~30 k kernels, each of which
# would otherwise be serialised for counter collection (35 min)
E="python $R/bench.py --workload ema --steps 5 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/es -o es -- $E 2>/dev/null | grep '^{' > $OUT/bench_ema_under_rocprof.json
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex ema_update_kernel --output-format csv -d $OUT/ef -o ef -- $E > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex ema_update_kernel --output-format csv -d $OUT/ew -o ew -- $E > /dev/null 2>&1
cp $(find $OUT/es -name 'es_kernel_stats.csv') $OUT/ema_kernel_stats.csv
head -1 $(find $OUT/ef -name 'ef_counter_collection.csv') > $OUT/pmc_ema.csv
grep -h ema_update_kernel $(find $OUT/ef -name 'ef_counter_collection.csv') $(find $OUT/ew -name 'ew_counter_collection.csv') >> $OUT/pmc_ema.csv
rm -rf $OUT/es $OUT/ef $OUT/ew
fi
PREV=${PREV_PROFILES:-$R/profiles/r5}
ROUND=${ROUND:-r6}
[ -f $OUT/pmc_fetch_size.csv ] || cp $PREV/pmc_fetch_size.csv $PREV/pmc_write_size.csv $OUT/
[ -f $OUT/pmc_ema.csv ] || cp $PREV/pmc_ema.csv $OUT/ 2>/dev/null || head -1 $OUT/pmc_fetch_size.csv > $OUT/pmc_ema.csv
cd $R
python - <<'PY'
import collections, csv, json, os, re, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from bench import kernel_source_sha256
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/profiles_new'
def means(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            name = re.sub(r'^.*?([A-Za-z_0-9]+)(<.*)?\(.*$', r'\1', r['Kernel_Name'].split('(')[0] + '(') if False else r['Kernel_Name']
            m = re.search(r'([A-Za-z_0-9]+)(?:<[^(]*>)?\(', name)
            key = (m.group(1) if m else name)
            t = re.search(r'<([^(]*)>\(', name)
            if t and key.startswith('pw_'): key += '<' + t.group(1).replace(' ', '') + '>'
            acc[key].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}
f, w = means(out + '/pmc_fetch_size.csv', 'FETCH_SIZE'), means(out + '/pmc_write_size.csv', 'WRITE_SIZE')
f.update(means(out + '/pmc_ema.csv', 'FETCH_SIZE')); w.update(means(out + '/pmc_ema.csv', 'WRITE_SIZE'))  # bench.py --workload ema
ks = {}
FETCH_FACTOR = {'mghs_stream_fwd': 1.0}
for k in sorted(set(f) | set(w)):
    if k.startswith('Cijk') or 'at::native' in k or k.startswith('__amd'): continue
    # gfx950: FETCH_SIZE tallies a 128-byte request at 64 bytes -- HALF the bytes of any fully coalesced read (16 B or 4 B per lane), but
    # EXACTLY the bytes of a read made of 64-byte pieces (experiments/fetch_size_calib.hip, profiles/r6/fetch_size_calib.txt: 0.500 /
    # 0.500 / 1.000 of 768 MiB; WRITE_SIZE 1.003).  The writer's reads are such pieces: every (segment, channel part) workgroup takes
    # 16 channels x 4 B of each 256-byte vsum row; its coalesced index reads (nzvox, <= 3 MB) are then undercounted by <= 3 MB.
    ff = FETCH_FACTOR.get(k, 2.0)
    ks[k] = dict(FETCH_SIZE_KB=f.get(k, 0.0), WRITE_SIZE_KB=w.get(k, 0.0), fetch_factor=ff,
                 hbm_bytes_per_launch=int((ff * f.get(k, 0.0) + w.get(k, 0.0)) * 1024))
json.dump(dict(samples_per_gpu=4, source_sha256=kernel_source_sha256(),
               command='rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 5 --warmup 2 --pmc-pass (two separate passes)',
               correction='bytes = (fetch_factor*FETCH_SIZE + WRITE_SIZE) * 1024; fetch_factor 2 (gfx950: FETCH_SIZE reports half of a coalesced '
                          'read) except where a kernel reads 64-byte pieces, which the counter reports exactly (mghs_stream_fwd: 1; calibration '
                          'profiles/r6/fetch_size_calib.txt)',
               kernels=ks), open(out + '/pmc_summary.json', 'w'), indent=1)
print(json.dumps({k: v['hbm_bytes_per_launch'] for k, v in ks.items()}, indent=0)[:3000])
if os.path.exists(out + '/pmc_sfa_half.csv'):
    import subprocess
    def demangle(n):
        return subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip() if n.startswith('_Z') else n
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(out + '/pmc_sfa_half.csv')):
        acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    hs = {}
    for k, d in acc.items():
        fz = sum(d.get('FETCH_SIZE', [0])) / max(1, len(d.get('FETCH_SIZE', [0])))
        wz = sum(d.get('WRITE_SIZE', [0])) / max(1, len(d.get('WRITE_SIZE', [0])))
        hs[demangle(k)[:160]] = dict(FETCH_SIZE_KB=fz, WRITE_SIZE_KB=wz, hbm_bytes_per_launch=int((2 * fz + wz) * 1024))
    json.dump(dict(samples_per_gpu=4, dtype='float16', source_sha256=kernel_source_sha256(),
                   command='rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python experiments/sfa_half.py 4 6 fp16 1 (two separate passes)',
                   correction='bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024', kernels=hs), open(out + '/pmc_sfa_half_summary.json', 'w'), indent=1)
    print(json.dumps({k[:70]: v['hbm_bytes_per_launch'] for k, v in hs.items()}, indent=0)[:3000])
PY
# the plain bench line last, with the fresh PMC summary in place (bench.py reports `traffic` only for a matching source hash)
if [[ " $WHAT " == *" hotpath "* ]]; then
mkdir -p $R/profiles/$ROUND && cp $OUT/pmc_summary.json $R/profiles/$ROUND/pmc_summary.json
cd $R && python bench.py > $OUT/bench_default.json 2>$OUT/bench_default.err
# the N > 1 code path on the one-GPU box: two ranks sharing the GPU over gloo (bench.py spawns its own ranks)
python bench.py --gpus 2 --dist-backend gloo --steps 3 --warmup 1 --batch 1 --cpu-samples 0 --no-operator --no-e2e --no-dhdl 2>$OUT/bench_two_ranks.err | grep '^{' > $OUT/bench_two_ranks_one_gpu_gloo.json
fi
ls -la $OUT
