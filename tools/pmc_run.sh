#!/bin/bash
# PMC passes (separate from --kernel-trace/--stats runs, as the guide prescribes). Usage: tools/pmc_run.sh <outdir> <steps>
set -o pipefail
OUT=${1:-gpurun_out/pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -- python3 tools/quick_bench.py 3840 2160 3 > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/quick_bench.py 3840 2160 3 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/write -- python3 tools/quick_bench.py 3840 2160 3 > $OUT/write.log 2>&1
find $OUT -name "*.csv" | head -20
