#!/bin/bash
# LDS side of k_raster_tile: instructions, active cycles, bank-conflict cycles.  Usage: tools/pmc_lds.sh <outdir> [quick_bench args: W H steps msaa]
# (program directly after `--`; counters in their own pass)
OUT=${1:-gpurun_out/pmc_lds}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o "SQ_LDS_[A-Z_]*\|SQ_ACTIVE_INST_LDS\|SQ_INST_CYCLES_[A-Z_]*\|SQ_INSTS_LDS" | sort -u > $OUT/lds_counters_available.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -- python3 tools/quick_bench.py "${@:-3840 2160 4 0}" > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS --output-format csv -d $OUT/b -- python3 tools/quick_bench.py "${@:-3840 2160 4 0}" > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("a", "b"):
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ctr[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(ctr.items()):
        if "raster" in k or "lean" in k:
            print(d, k[:44], {n: round(sum(v) / len(v) / 1e6, 2) for n, v in sorted(c.items())})
PY
cat $OUT/lds_counters_available.txt | tr '\n' ' '
