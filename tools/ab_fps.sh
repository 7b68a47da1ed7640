#!/bin/bash
# frames/s of library variants in one bench mode, 200 steps each.  Usage: tools/ab_fps.sh "<bench args>" lib1.so ... ("-" = the in-tree library)
ARGS=$1; shift
for lib in "$@"; do
  if [ "$lib" != "-" ]; then export AWSM_HIP_LIB=$GRAFT_REPO_ROOT/$lib; X="--allow-variant-lib"; else unset AWSM_HIP_LIB; X=""; fi
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 200 --warmup 20 $X $ARGS | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$ARGS] $lib:', round(d['value'], 1), 'frames/s', {k: round(v * 1000) for k, v in d['roofline']['all_kernels_ms'].items()})"
done
