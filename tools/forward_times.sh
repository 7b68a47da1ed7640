#!/bin/bash
# per-kernel average durations of the transparent-pass scratch bench.  Usage: tools/forward_times.sh <outdir> [detail] [msaa] [mip]
OUT=$GRAFT_REPO_ROOT/${1:-gpurun_out/ft}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/transparent_bench.py 3840 2160 ${2:-4} ${3:-0} ${4:-0} > $OUT/bench.log 2> $OUT/err.log || exit 1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "awsm" in r["Name"]:
        print("%-44s calls %4s avg %8.1f us" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -1 $OUT/bench.log
