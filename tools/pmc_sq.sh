#!/bin/bash
# SQ-counter pass for one library variant. Usage: tools/pmc_sq.sh <outdir> <lib.so>
set -o pipefail
OUT=$1
export AWSM_HIP_LIB=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -- python3 tools/quick_bench.py 3840 2160 3 > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES --output-format csv -d $OUT/sq2 -- python3 tools/quick_bench.py 3840 2160 3 > $OUT/sq2.log 2>&1
echo done
