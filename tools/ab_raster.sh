#!/bin/bash
# k_raster_tile under variants: SQ counters + durations.  Usage: tools/ab_raster.sh <outdir> name=lib.so[,ENV=val...] ...
# (program directly after `--`: the profiler's preload initialises the GPU, an env/bash hop would be an exec from a GPU process)
set -o pipefail
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
for spec in "$@"; do
  name=${spec%%=*}; rest=${spec#*=}
  lib=${rest%%,*}
  envs=""
  if [[ $rest == *,* ]]; then envs=${rest#*,}; fi
  (
    export AWSM_HIP_LIB=$lib
    IFS=','; for kv in $envs; do export "$kv"; done; unset IFS
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/$name.sq -- python3 tools/quick_bench.py 3840 2160 4 $AB_QB_ARGS > $OUT/$name.sq.log 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name.kt -- python3 tools/quick_bench.py 3840 2160 20 $AB_QB_ARGS > $OUT/$name.kt.log 2>&1
  )
  echo "== $name"
  python3 tools/pmc_summary.py $OUT/$name.sq | grep -A 9 "k_raster_tile" | head -12
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/$name.kt/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])) if f else []:
    if "k_raster_tile" in r["Name"] or "k_bin" in r["Name"] or "k_deform" in r["Name"]:
        print("   %-44s calls %4s avg %8.1f us" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  tail -3 $OUT/$name.kt.log | head -2
done
