#!/bin/bash
# A/B of library variants in one bench mode: per library the free-running rate, rocprofv3 kernel times (no overlap) and the SQ instruction counters.
# Usage: tools/ab_mode.sh <tag> "<bench args>" lib1.so [lib2.so ...]      (libraries from tools/build_variants.sh; "-" = the in-tree library)
set -o pipefail
TAG=$1; ARGS=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  name=$(basename $lib .so); [ "$lib" = "-" ] && name=tree
  OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/$name; mkdir -p $OUT
  X=""; if [ "$lib" != "-" ]; then export AWSM_HIP_LIB=$GRAFT_REPO_ROOT/$lib; X="--allow-variant-lib"; else unset AWSM_HIP_LIB; fi
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 100 --warmup 10 $X $ARGS > $OUT/bench.json 2> $OUT/bench.err || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-overlap --no-cpu-baseline --steps 40 --warmup 10 $X $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 2
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py --no-overlap --no-cpu-baseline --static-camera --steps 3 --warmup 1 --profile-frames 3 $X $ARGS > $OUT/pmc_sq.log 2>&1 || exit 3
  echo "=== $name"; python3 tools/mode_summary.py $OUT ${AB_LINES:-6}
done
