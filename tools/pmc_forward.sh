#!/bin/bash
# SQ counters of the transparent pass's tile kernel.  Usage: tools/pmc_forward.sh <outdir> [detail]
set -o pipefail
OUT=$GRAFT_REPO_ROOT/$1
D=${2:-1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/sq -- python3 tools/transparent_bench.py 3840 2160 $D 0 0 > $OUT/sq.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq2 -- python3 tools/transparent_bench.py 3840 2160 $D 0 0 > $OUT/sq2.log 2>&1 || exit 2
python3 - <<PY
import csv, glob, collections
for sub in ("sq", "sq2"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k in acc:
        if "forward" in k or "k_shade" in k:
            print(k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
