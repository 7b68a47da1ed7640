#!/bin/bash
# L1/L2 counters for one library variant. Usage: tools/pmc_mem.sh <outdir> <lib.so>
set -o pipefail
OUT=$1
export AWSM_HIP_LIB=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/tcp -- python3 tools/quick_bench.py 3840 2160 3 > $OUT/tcp.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- python3 tools/quick_bench.py 3840 2160 3 > $OUT/tcc.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d $OUT/ea -- python3 tools/quick_bench.py 3840 2160 3 > $OUT/ea.log 2>&1
tail -2 $OUT/tcp.log; echo done
