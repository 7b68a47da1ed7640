#!/bin/bash
# Diagnostic build of libawsm_hip.so with in-kernel s_memrealtime stamps in the geometry kernels (-DAWSM_STAMP) -> build/variants/lib_STAMP.so
set -e
cd "$(dirname "$0")/../awsm-renderer_amd/csrc"
OUT=../../build/variants; mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -DAWSM_STAMP"
for f in awsm_hip.cpp kernels_geometry.hip kernels_shade.hip; do /opt/rocm/bin/hipcc $FLAGS -c -o $OUT/stamp_${f%.*}.o $f 2>&1 | grep -v hip-link || true; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_STAMP.so $OUT/stamp_awsm_hip.o $OUT/stamp_kernels_geometry.o $OUT/stamp_kernels_shade.o && rm -f $OUT/stamp_*.o && echo built $OUT/lib_STAMP.so
