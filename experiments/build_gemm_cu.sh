#!/bin/bash
# builds experiments/build/gemm_cu_bench and prints the register / scratch use of every pw_gemm_cu_kernel instantiation
cd "$(dirname "$0")/.." && mkdir -p experiments/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wall -Wno-unused -save-temps=obj experiments/gemm_cu_bench.hip -o experiments/build/gemm_cu_bench 2>&1 | grep -E "error" ; test -x experiments/build/gemm_cu_bench
python3 - <<'PY'
import re
s=open('experiments/build/gemm_cu_bench-hip-amdgcn-amd-amdhsa-gfx950.s').read()
for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', s):
    n=m.group(1)
    if 'pw_gemm' in n:
        t=re.search(r'ILi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELi(\d)ELb(\d)ELi(\d)ELi(\d)ELi(\d)E',n).groups()
        print('KCN,W,TWO,RELU,EPI,REC,AUX,ORD,NACC=',','.join(t),'scratch',m.group(2),'sgpr',m.group(3),'vgpr',m.group(4),'spill',m.group(5))
PY
