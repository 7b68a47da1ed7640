#!/bin/bash
# k_raster_tile durations under environment variants of the current library.  Usage: tools/ab_raster_times.sh <outdir> name=ENV=val[,ENV=val...] ...
set -o pipefail
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  (
    IFS=','; for kv in $envs; do export "$kv"; done; unset IFS
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name.kt -- python3 tools/quick_bench.py 3840 2160 20 $AB_QB_ARGS > $OUT/$name.kt.log 2>&1
  )
  python3 - <<PY
import csv, glob, re
f = glob.glob("$OUT/$name.kt/**/*kernel_stats.csv", recursive=True)
t = {}
for r in csv.DictReader(open(f[0])) if f else []:
    for k in ("k_raster_tile", "k_bin<true>", "k_bin<false>", "k_deform"):
        if k in r["Name"]: t[k] = float(r["AverageNs"]) / 1e3
log = open("$OUT/$name.kt.log").read()
m = re.findall(r"'raster_entries_culled': ([0-9.]+)", log)
print("%-28s raster %6.1f us  fill %5.1f  count %5.1f  transform %5.1f   culled %s" % ("$name", t.get("k_raster_tile", 0), t.get("k_bin<true>", 0), t.get("k_bin<false>", 0), t.get("k_deform", 0), m[-1] if m else "?"))
PY
done
