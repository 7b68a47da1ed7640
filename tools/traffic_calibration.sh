#!/bin/bash
# What do FETCH_SIZE / TCC_EA0_RDREQ say for known byte counts in the lean kernel's own access mix, and what do they say for the headline frame?
# (VERDICT r3 "next" #3.)  Usage: tools/traffic_calibration.sh <outdir>.   Separate --pmc passes, program directly after `--`.
set -o pipefail
OUT=${1:-gpurun_out/traffic}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_[A-Z_]*MALL[A-Z_]*\|TCC_BUBBLE[A-Z_]*\|TCC_TAG_STALL[A-Z_]*" | sort -u > $OUT/counters_available.txt
PASS_A="FETCH_SIZE TCC_EA0_RDREQ_32B_sum"
PASS_B="TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum"
PASS_C="WRITE_SIZE TCC_EA0_RDREQ_DRAM_sum"
run() {  # name, program...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc $PASS_A --output-format csv -d $OUT/$name.a -- "$@" > $OUT/$name.a.log 2>&1
  rocprofv3 --kernel-trace --pmc $PASS_B --output-format csv -d $OUT/$name.b -- "$@" > $OUT/$name.b.log 2>&1
  rocprofv3 --kernel-trace --pmc $PASS_C --output-format csv -d $OUT/$name.c -- "$@" > $OUT/$name.c.log 2>&1
  "$@" > $OUT/$name.plain.log 2>&1
}
run scatter8 tools/gather_probe.bin 2048 16
run stream8  tools/gather_probe.bin 2048 0 1
run records  tools/gather_probe.bin 2048 64 2
run frame    python3 tools/quick_bench.py 3840 2160 3
python3 - <<PY
import csv, glob, collections
def counters(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out
for name in ("scatter8", "stream8", "records", "frame"):
    print("==", name)
    merged = collections.defaultdict(dict)
    for p in "abc":
        for k, c in counters("$OUT/%s.%s" % (name, p)).items():
            for cn, v in c.items():
                merged[k][cn] = sum(v) / len(v)
    for k, c in sorted(merged.items()):
        if name == "frame" and "k_shade_lean" not in k and "k_raster" not in k and "k_deform" not in k: continue
        print("  ", k[:60], {cn: ("%.4g" % v) for cn, v in sorted(c.items())})
    if name != "frame":
        print("   ", open("$OUT/%s.plain.log" % name).read().strip().splitlines()[-1])
PY
