#!/bin/bash
# The memory-side read requests of the lean kernel's access patterns and of the headline frame by SIZE (32 / 64 / 128 bytes).  Usage: tools/traffic_sizes.sh <outdir>
set -o pipefail
OUT=${1:-gpurun_out/traffic_sizes}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
P="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/scatter8 -- tools/gather_probe.bin 2048 16 > $OUT/scatter8.log 2>&1
rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/stream8 -- tools/gather_probe.bin 2048 0 1 > $OUT/stream8.log 2>&1
rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/records -- tools/gather_probe.bin 2048 64 2 > $OUT/records.log 2>&1
rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/frame -- python3 tools/quick_bench.py 3840 2160 3 > $OUT/frame.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum --output-format csv -d $OUT/frame_wr -- python3 tools/quick_bench.py 3840 2160 3 > $OUT/frame_wr.log 2>&1
python3 tools/pmc_summary.py $OUT/scatter8 $OUT/stream8 $OUT/records 2>/dev/null | grep -v "^awsm" ; python3 - <<PY
import csv, glob, collections
for d in ("scatter8", "stream8", "records", "frame", "frame_wr"):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", d)
    for k, c in sorted(out.items()):
        if "fillBuffer" in k: continue
        v = {cn: sum(x) / len(x) for cn, x in c.items()}
        line = "   %-44s " % k[:44] + " ".join("%s=%.4g" % (cn.replace("TCC_EA0_", "").replace("_sum", ""), x) for cn, x in sorted(v.items()))
        if "TCC_EA0_RDREQ_64B_sum" in v:
            n32, n64, n128, n = v.get("TCC_EA0_RDREQ_32B_sum", 0), v["TCC_EA0_RDREQ_64B_sum"], v["TCC_EA0_RDREQ_128B_sum"], v["TCC_EA0_RDREQ_sum"]
            line += "  | bytes by size = %.4g MB (other sizes: %.4g requests)" % ((32 * n32 + 64 * n64 + 128 * n128) / 1e6, n - n32 - n64 - n128)
        print(line)
PY
