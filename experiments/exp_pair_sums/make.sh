#!/bin/bash
# VERDICT r5 item 3, priced by measurement instead of arithmetic: what does the data-gradient GEMM of the half-storage stage
# (pw_gemm_cuh_kernel, EPI = 1) cost when its epilogue also does pair_sums_h's work -- load the y1 tile of the rows it is about to store
# and accumulate S1 = sum g1, S2 = sum g1 (y1 - mean) per channel (one row of partial sums per workgroup)?
# The product sources are NOT edited: this script copies dhd_amd/csrc into experiments/build/exp_pair_sums/, patches the copy of sfa_half.h
# (patch.py) and builds experiments/gemm_cuh_bench.hip against the copy, once plain (A) and once with -DDHD_EXP_PAIR (B).
set -eu
cd "$(dirname "$0")/../.."
D=experiments/build/exp_pair_sums
rm -rf $D && mkdir -p $D/dhd_amd/csrc $D/include $D/experiments
cp dhd_amd/csrc/*.h dhd_amd/csrc/*.hip $D/dhd_amd/csrc/
cp include/dhd_amd.h $D/include/
cp experiments/gemm_cuh_bench.hip $D/experiments/
python experiments/exp_pair_sums/patch.py $D
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -Wno-unused-function"
/opt/rocm/bin/hipcc $F $D/experiments/gemm_cuh_bench.hip -o experiments/exp_pair_sums/bench_A.bin
/opt/rocm/bin/hipcc $F -DDHD_EXP_PAIR $D/experiments/gemm_cuh_bench.hip -o experiments/exp_pair_sums/bench_B.bin
echo built
