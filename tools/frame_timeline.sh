#!/bin/bash
# Timeline of one steady-state frame of the bench (overlapped, stage timers off): kernel, stream, duration, gap to the previous kernel
# on the same queue.  Usage: tools/frame_timeline.sh <outdir> [bench args]
OUT=$GRAFT_REPO_ROOT/${1:-gpurun_out/tl}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --steps 60 --warmup 10 --profile-frames 1 "$@" > $OUT/bench.json 2> $OUT/err.log || exit 1
python3 - <<PY
import csv, glob, statistics
f = sorted(glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f))]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-34:], r["Queue_Id"]) for r in rows)
starts = [i for i, k in enumerate(ks) if "k_deform_transform" in k[2]]
mid = starts[len(starts) // 2]
nxt = starts[len(starts) // 2 + 1]
per = [(ks[b][0] - ks[a][0]) / 1e3 for a, b in zip(starts[15:-3], starts[16:-2])]
print("frame period us: median %.1f" % statistics.median(per))
last_end = {}
t0 = ks[mid][0]
for k in ks[mid:nxt + 8]:
    gap = (k[0] - last_end[k[3]]) / 1e3 if k[3] in last_end else 0.0
    print("%8.1f  q%-3s %-34s dur %7.1f  gap %6.1f" % ((k[0] - t0) / 1e3, k[3], k[2], (k[1] - k[0]) / 1e3, gap))
    last_end[k[3]] = k[1]
PY
