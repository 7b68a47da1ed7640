#!/bin/bash
# The frame rates DESIGN.md's mode table quotes, in one call.  Usage: tools/fps_modes.sh <outdir>
OUT=${1:-gpurun_out/fps_modes}; mkdir -p $OUT
fps() { python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'], 1), 'frames/s')"; }
{
run() { echo -n "$1: "; timeout -k 10 300 python3 bench.py --gpus 1 --no-cpu-baseline $1 2>/dev/null | fps || exit 1; }
run "--steps 20 --warmup 5"
run "--steps 200 --warmup 20"
run "--steps 200 --warmup 20 --msaa 4"
run "--steps 200 --warmup 20 --mipmap"
run "--steps 200 --warmup 20 --msaa 4 --mipmap"
run "--steps 200 --warmup 20 --msaa 4 --mipmap --anisotropic"
run "--steps 200 --warmup 20 --config 2"
run "--steps 200 --warmup 20 --config 3"
run "--steps 20 --warmup 5"
} > $OUT/fps.txt 2>&1
cat $OUT/fps.txt
