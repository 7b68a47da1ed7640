#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel stats of the same command, then PMC passes
# (counters in their own runs, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Usage: tools/profile_round.sh <tag>
set -o pipefail
TAG=${1:-r01_x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
tail -c 600 $OUT/bench.json; echo
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-overlap --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 2
echo "stats done"
B="python3 bench.py --no-overlap --no-cpu-baseline --static-camera --steps 3 --warmup 1 --profile-frames 3"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1 || exit 4
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 || exit 5
echo "pmc done"
