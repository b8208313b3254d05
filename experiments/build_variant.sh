#!/bin/bash
# experiments/build_variant.sh NAME FILE.hip "-DFLAG=..." : libdhd_amd with ONE translation unit rebuilt with extra flags ->
# experiments/variants/libdhd_amd_NAME.so (select with DHD_AMD_LIB=...).  The other objects are those of the last `make`.
set -eu
cd "$(dirname "$0")/../dhd_amd/csrc"
NAME=$1; SRC=$2; shift 2
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -Wall -Wno-unused-function"
mkdir -p ../../experiments/variants
/opt/rocm/bin/hipcc $FLAGS "$@" -c $SRC -o ../../experiments/variants/${SRC%.hip}_$NAME.o
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../experiments/variants/libdhd_amd_$NAME.so $OBJS ../../experiments/variants/${SRC%.hip}_$NAME.o
echo built experiments/variants/libdhd_amd_$NAME.so
