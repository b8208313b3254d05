#!/bin/bash
# Stall / pipe counters of the SFA stage's resident GEMM kernels, several counter-only passes (no trace domains besides the
# kernel trace), means per launch and kernel variant.  SQ counters only: a pass with TA_* / TCP_* counters did not finish in
# ten minutes on the GPU box.  Usage: bash experiments/pmc_gemm_stalls.sh [regex]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_gemm; rm -rf $OUT; mkdir -p $OUT
RX=${1:-pw_gemm_res_kernel}
i=0
for set in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
 "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM" \
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_ANY SQ_INSTS_SALU SQ_INST_CYCLES_SALU" \
 "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_CYCLES" ; do
  i=$((i+1))
  timeout 170 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "$RX" --output-format csv -d $OUT/p$i -o p -- python $R/experiments/sfa_only.py 4 3 bf16x3 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections, re
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc_gemm'
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/**/p_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'pw_gemm_res_kernel<([^>]*)>', r['Kernel_Name'])
        k = m.group(1).replace(' ', '') if m else r['Kernel_Name'][:60]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    d = {c: sum(v) / len(v) for c, v in acc[k].items()}
    print(k)
    for c in sorted(d): print(f'    {c:36s} {d[c]:.4g}')
PY
