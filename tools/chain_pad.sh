#!/bin/bash
# Is the overlapped frame bound by the length of the geometry stream's chain?  Pads the chain with an idle wavefront of N microseconds
# (AWSM_DEBUG_CHAIN_PAD_US, kernels_geometry.hip; read by a debug build only: tools/build_variants.sh g_debug "-DAWSM_DEBUG_SWITCHES" first)
# and prints the frame period.  Usage: tools/chain_pad.sh "<bench args>" 0 20 40
ARGS=$1; shift
for p in "$@"; do
  AWSM_DEBUG_CHAIN_PAD_US=$p AWSM_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/lib_g_debug.so timeout -k 10 200 python3 bench.py --no-cpu-baseline --allow-variant-lib --steps 200 --warmup 20 $ARGS | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pad $p us [$ARGS]:', round(d['value'], 1), 'frames/s,', round(d['ms_per_step'] * 1000, 1), 'us per frame')"
done
