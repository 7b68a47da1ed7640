#!/bin/bash
# (AWSM_SHADE_CU_MASK is read only by a debug build: tools/build_variants.sh h_debug "-DAWSM_DEBUG_SWITCHES" first)
# Do the two streams of the overlapped frame run better on partly disjoint CUs?  (Round 4: the frame follows resource use, and the geometry
# kernels — latency-bound, low issue rate — take wave slots from the lean kernel on every CU.)  hipExtStreamCreateWithCUMask on the caller's stream
# (AWSM_BENCH_STREAM_CU_MASK: the geometry passes) and / or on the library's shade streams (AWSM_SHADE_CU_MASK).
# Usage: tools/ab_cu_mask.sh <outdir>
OUT=${1:-gpurun_out/cu_mask}
mkdir -p $OUT
F=ffffffff
ALL=$F,$F,$F,$F,$F,$F,$F,$F
{
for m in "" "$F" "$F,$F,$F,$F" "55555555,55555555,55555555,55555555,55555555,55555555,55555555,55555555" "ff,ff,ff,ff,ff,ff,ff,ff" "0,0,$F,$F,$F,$F,$F,$F"; do
  echo "== probe mask '$m'"; timeout -k 5 60 tools/cu_mask_probe.bin $m || exit 1
done
} > $OUT/probe.txt 2>&1
fps() { python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'], 1), 'frames/s')"; }
{
run() {  # label, geometry mask, shade mask, extra args
  local g=$2 s=$3
  for rep in 1; do
    echo -n "$1 $4: "
    env ${g:+AWSM_BENCH_STREAM_CU_MASK=$g} ${s:+AWSM_SHADE_CU_MASK=$s} AWSM_HIP_LIB=build/variants/lib_h_debug.so timeout -k 10 200 python3 bench.py --gpus 1 --no-cpu-baseline --allow-variant-lib --steps 200 --warmup 20 $4 2>/dev/null | fps || exit 1
  done
}
H=$F,$F,$F,$F                      # bits 0..127
Q=$F,$F                            # bits 0..63
T=$F,$F,$F                         # bits 0..95
S6=$F,$F,$F,$F,$F,$F               # bits 0..191: CUs 0-5 of every shader engine of every XCD (bit i: XCD i % 8, engine (i / 8) % 4, CU i / 32)
S7=$F,$F,$F,$F,$F,$F,$F            # bits 0..223
UP6=0,0,$F,$F,$F,$F,$F,$F          # bits 64..255
UP4=0,0,0,0,$F,$F,$F,$F            # bits 128..255
run "no masks          " "" ""
run "geometry low128   " "$H" ""
run "geometry low192   " "$S6" ""
run "geometry low224   " "$S7" ""
run "shade low224      " "" "$S7"
run "shade low192      " "" "$S6"
run "geometry low64, shade up192 (disjoint)" "$Q" "$UP6"
run "geometry low128, shade up128 (disjoint)" "$H" "$UP4"
run "geometry low128, shade up192 (overlap 64)" "$H" "$UP6"
run "no masks, msaa+mips" "" "" "--msaa 4 --mipmap"
run "geometry low192, msaa+mips" "$S6" "" "--msaa 4 --mipmap"
run "geometry low128, msaa+mips" "$H" "" "--msaa 4 --mipmap"
run "shade low192, msaa+mips" "" "$S6" "--msaa 4 --mipmap"
run "no masks (again)   " "" ""
} > $OUT/fps.txt 2>&1
cat $OUT/probe.txt | grep -v "^xcd" ; cat $OUT/fps.txt
