#!/bin/bash
# per-kernel average durations of a short bench run (rocprofv3 --kernel-trace --stats). Usage: tools/kernel_times.sh <outdir>
OUT=${1:-gpurun_out/kt}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-overlap --no-cpu-baseline --steps 60 --warmup 5 --profile-frames 5 $AWSM_KT_ARGS > $OUT/bench.json 2> $OUT/err.log || exit 1
python3 - <<PY
import csv, glob, json
f = glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Name"].startswith(("awsm", "void awsm")):
        print("%-40s calls %4s avg %8.1f us" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3))
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("bench fps", d["value"], "ms/step", d["ms_per_step"])
PY
