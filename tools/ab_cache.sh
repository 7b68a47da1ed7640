#!/bin/bash
# Geometry cache on / off, same box, interleaved: frames/s of the default bench and of the reference's default mode.
# Usage: tools/ab_cache.sh <outdir>
OUT=${1:-gpurun_out/ab_cache}
mkdir -p $OUT
fps() { python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['roofline']['all_kernels_ms']; print(round(d['value'], 1), 'frames/s', round(d['ms_per_step'] * 1e3, 1), 'us;  transform', round(k['k_deform_transform'] * 1e3, 1), 'bin', round(k['k_bin'] * 1e3, 1), 'raster', round(k['k_raster_tile'] * 1e3, 1), 'shade', round(k['k_shade'] * 1e3, 1), 'us; cache blocks', d['frame_stats']['geometry_cache_blocks'], '/', d['frame_stats']['geometry_blocks'])"; }
{
for mode in "" "--msaa 4 --mipmap" "--static-camera"; do
  for rep in 1 2; do
    for cache in 1 0; do
      echo -n "cache $cache $mode: "
      AWSM_GEOMETRY_CACHE=$cache timeout -k 10 200 python3 bench.py --gpus 1 --no-cpu-baseline --steps 200 --warmup 20 $mode 2>/dev/null | fps || exit 1
    done
  done
done
} > $OUT/fps.txt 2>&1
cat $OUT/fps.txt
