#!/bin/bash
# What does each stage of the geometry pass take from the opaque pass it runs beside?  AWSM_DEBUG_KNOCKOUT (awsm_hip.cpp, enqueue_geometry) leaves out
# the raster (4), binning + raster (6) or the whole geometry pass (7) from the ninth frame on; with a static camera the slot's buffers keep the
# same keys, so the opaque pass does identical work.  The switch is compiled only into a debug build of the library:
#   tools/build_variants.sh h_debug "-DAWSM_DEBUG_SWITCHES"      (here, before gpurun)
# Usage: tools/knockout.sh <outdir>
OUT=${1:-gpurun_out/knockout}
mkdir -p $OUT
fps() { python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'], 1), 'frames/s', round(d['ms_per_step'] * 1e3, 1), 'us')"; }
{
for mode in "" "--msaa 4 --mipmap"; do
  for ko in 0 4 6 7 0; do
    echo -n "knockout $ko $mode: "
    AWSM_DEBUG_KNOCKOUT=$ko AWSM_HIP_LIB=build/variants/lib_h_debug.so timeout -k 10 200 python3 bench.py --gpus 1 --no-cpu-baseline --allow-variant-lib --static-camera --steps 200 --warmup 20 $mode 2>/dev/null | fps || exit 1
  done
done
} > $OUT/fps.txt 2>&1
cat $OUT/fps.txt
