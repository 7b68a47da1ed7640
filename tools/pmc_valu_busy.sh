#!/bin/bash
# VALU busy time of the frame's kernels: SQ_ACTIVE_INST_VALU (cycles the VALU executes, summed over SIMDs / 4) against GRBM_GUI_ACTIVE
# and SQ_BUSY_CYCLES, in a PMC pass of its own.  Usage (GPU box): tools/pmc_valu_busy.sh <tag>
set -o pipefail
TAG=${1:-r02_x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-overlap --no-cpu-baseline --static-camera --steps 3 --warmup 1 --profile-frames 3"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_busy -- $B > $OUT/pmc_busy.log 2>&1 || { tail -5 $OUT/pmc_busy.log; exit 3; }
python3 - <<PY
import csv, glob, collections
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_busy/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if name.startswith("awsm::"): ctr[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in sorted(ctr.items()):
    print(k, {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
