#!/bin/bash
# bench.py frames/s under environment variants of the current library.  Usage: tools/ab_env_fps.sh <outdir> name=ENV=val[,ENV=val...] ...   (AB_BENCH_ARGS: extra bench.py flags)
OUT=$1; shift
mkdir -p $OUT
for rep in 1 2; do
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  (
    IFS=','; for kv in $envs; do export "$kv"; done; unset IFS
    python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --profile-frames 3 $AB_BENCH_ARGS > $OUT/$name.$rep.json 2> $OUT/$name.$rep.err
  )
  python3 - <<PY
import json
try:
    d = json.load(open("$OUT/$name.$rep.json"))
    print("%-24s rep $rep  %8.1f frames/s   culled %s  kernels %s" % ("$name", d["value"], d["frame_stats"].get("raster_entries_culled"), {k: round(v * 1e3, 1) for k, v in d["roofline"]["all_kernels_ms"].items()}))
except Exception as e:
    print("$name failed", e)
PY
done
done
