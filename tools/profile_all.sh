#!/bin/bash
# One round's evidence in one gpurun call: the headline set (bench line, kernel stats, PMC passes), the default mode's set, the two frame timelines,
# the mode table and the driver's own command.  Usage: tools/profile_all.sh <tag>     (then tools/profile_collect.py on this box's copy of gpurun_out/)
TAG=${1:-r04_b}
tools/profile_round.sh $TAG > gpurun_out/${TAG}_round.log 2>&1 || { echo "profile_round failed"; tail -5 gpurun_out/${TAG}_round.log; exit 1; }
echo "round done"
tools/profile_mode.sh ${TAG}_msaa_mips --msaa 4 --mipmap > gpurun_out/${TAG}_mode.log 2>&1 || { echo "profile_mode failed"; tail -5 gpurun_out/${TAG}_mode.log; exit 2; }
echo "mode done"
tools/frame_timeline.sh gpurun_out/${TAG}_tl > gpurun_out/${TAG}_timeline.txt 2>&1 || { echo "timeline failed"; exit 3; }
tools/frame_timeline.sh gpurun_out/${TAG}_tl_msaa_mips --msaa 4 --mipmap > gpurun_out/${TAG}_timeline_msaa_mips.txt 2>&1 || { echo "timeline 2 failed"; exit 4; }
echo "timelines done"
tools/fps_modes.sh gpurun_out/${TAG}_fps_modes > /dev/null 2>&1 || { echo "fps_modes failed"; exit 5; }
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_driver_cmd.json 2> gpurun_out/${TAG}_driver_cmd.err || { echo "driver cmd failed"; exit 6; }
cat gpurun_out/${TAG}_fps_modes/fps.txt; tail -c 400 gpurun_out/${TAG}_driver_cmd.json
