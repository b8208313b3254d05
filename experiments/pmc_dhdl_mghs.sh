#!/bin/bash
# occupancy and stall counters of the point-proportional MGHS kernels at the DHD-L geometry (configs[3]/[4]:
# 6 x 512x1408 -> 32x88, D = 88, B = 2), MGHS-only step.  Three counter passes (SQ, TCP/TCC, GRBM), --kernel-trace only.
# usage (gpurun): bash experiments/pmc_dhdl_mghs.sh -> gpurun_out/pmc_dhdl_mghs.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_dhdl_mghs.txt
B="python $R/bench.py --geometry dhd-l --batch 2 --no-sfa --steps 5 --warmup 2 --pmc-pass --no-operator"
RE='mghs_pixel_bwd|mghs_col_sums|mghs_geom_count|mghs_gather_sums|mghs_scatter_planes|mghs_stream|mghs_sums'
rm -rf /tmp/pd
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-include-regex "$RE" --output-format csv -d /tmp/pd/a -o a -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-include-regex "$RE" --output-format csv -d /tmp/pd/b -o b -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS --kernel-include-regex "$RE" --output-format csv -d /tmp/pd/c -o c -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd/s -o s -- $B > /dev/null 2>&1
python - > $OUT <<'PY'
import collections, csv, glob, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob('/tmp/pd/*/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(mghs_[a-z_0-9]+(?:<[^>]*>)?)', r['Kernel_Name'])
        k = m.group(1) if m else r['Kernel_Name'][:40]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        meta[k] = (r.get('VGPR_Count'), r.get('Accum_VGPR_Count'), r.get('SGPR_Count'), r.get('LDS_Block_Size'), r.get('Workgroup_Size'), r.get('Grid_Size'))
dur = {}
for f in glob.glob('/tmp/pd/s/**/s_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(mghs_[a-z_0-9]+(?:<[^>]*>)?)', r['Name'])
        if m: dur[m.group(1)] = float(r['AverageNs']) / 1e3
print('DHD-L geometry (6 x 32x88, D = 88, B = 2), MGHS-only step; per launch means; SQ_* cycle counters are quad-cycles summed over waves')
for k in sorted(acc, key=lambda k: -dur.get(k, 0)):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    vg, ag, sg, lds, wg, grid = meta[k]
    waves = c.get('SQ_WAVES', 0)
    wc = c.get('SQ_WAVE_CYCLES', 1)
    regs = int(vg or 0) + int(ag or 0)           # arch + accumulation registers: one unified file on gfx950
    alloc = (regs + 7) // 8 * 8
    print(f'\n{k}: {dur.get(k, float("nan")):.1f} us | VGPR {vg} + AGPR {ag} (alloc {alloc}: {min(8, 512 // max(alloc, 1))} waves/SIMD by registers) SGPR {sg} LDS {lds} B  workgroup {wg} grid {grid}  waves {waves:.0f}')
    if 'SQ_WAVE_CYCLES' in c:
        print(f'   wave-cycles: active-inst {c.get("SQ_ACTIVE_INST_ANY", 0) / wc:.2f}  wait-any (s_waitcnt / barrier) {c.get("SQ_WAIT_ANY", 0) / wc:.2f}  '
              f'wait-inst-any (issue stall) {c.get("SQ_WAIT_INST_ANY", 0) / wc:.2f}  |  VALU insts/wave {c.get("SQ_INSTS_VALU", 0) / max(waves, 1):.0f}  '
              f'VMEM-read insts/wave {c.get("SQ_INSTS_VMEM_RD", 0) / max(waves, 1):.1f}')
        if k in dur:   # SQ_WAVE_CYCLES counts quad-cycles summed over waves: x 4 / (kernel time x clock x 1024 SIMDs) = mean waves per SIMD
            clk = c.get('GRBM_GUI_ACTIVE', 0) / 8 / (dur[k] * 1e3) if 'GRBM_GUI_ACTIVE' in c else 2.1
            print(f'   mean resident waves per SIMD over the kernel ~ {4 * wc / (dur[k] * 1e3 * clk * 1024):.2f}  (of 8)')
    if 'TCC_HIT_sum' in c:
        h, m_ = c['TCC_HIT_sum'], c.get('TCC_MISS_sum', 0)
        print(f'   L2: requests from L1 {c.get("TCP_TCC_READ_REQ_sum", 0):.3g}  hit rate {h / max(h + m_, 1):.3f}  fabric read requests {c.get("TCC_EA0_RDREQ_sum", 0):.3g}')
    if 'GRBM_GUI_ACTIVE' in c and k in dur:
        print(f'   GRBM_GUI_ACTIVE {c["GRBM_GUI_ACTIVE"]:.3g} cycles over 8 XCDs -> effective clock {c["GRBM_GUI_ACTIVE"] / 8 / dur[k] / 1e3:.2f} GHz (under the profiler)  LDS insts/wave {c.get("SQ_INSTS_LDS", 0) / max(waves, 1):.0f}')
PY
rm -rf /tmp/pd
cat $OUT
