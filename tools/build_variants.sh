#!/bin/bash
# A/B builds of libawsm_hip.so with extra -D flags for kernels_shade.hip, into build/variants/ (git-ignored; travels with gpurun).
# usage: tools/build_variants.sh NAME "-DFLAG=1 ..." [NAME2 "..."] ...   then on the GPU box: tools/ab_bench.sh build/variants/lib_NAME.so ...
# (a NAME that starts with g_ applies its flags to kernels_geometry.hip instead, one that starts with h_ to awsm_hip.cpp — e.g.
#  h_debug "-DAWSM_DEBUG_SWITCHES": the AWSM_DEBUG_KNOCKOUT / AWSM_SHADE_CU_MASK experiments, which the product library does not carry)
set -e
cd "$(dirname "$0")/../awsm-renderer_amd/csrc"
OUT=../../build/variants
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function"
make -s ../libawsm_hip.so >/dev/null
while [ $# -ge 2 ]; do
  NAME=$1; DEFS=$2; shift 2
  if [[ $NAME == g_* ]]; then SRC=kernels_geometry.hip; OTHER="awsm_hip.o kernels_shade.o"; elif [[ $NAME == h_* ]]; then SRC=awsm_hip.cpp; OTHER="kernels_geometry.o kernels_shade.o"; else SRC=kernels_shade.hip; OTHER="awsm_hip.o kernels_geometry.o"; fi
  ( /opt/rocm/bin/hipcc $FLAGS $DEFS -c -o $OUT/var_$NAME.o $SRC 2>&1 | grep -v hip-link || true
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$NAME.so $OTHER $OUT/var_$NAME.o && rm -f $OUT/var_$NAME.o && echo built $OUT/lib_$NAME.so ) &
  while [ $(jobs -r | wc -l) -ge 3 ]; do sleep 1; done
done
wait
