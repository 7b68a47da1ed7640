"""Per-frame device-clock periods of the driver's command (bench.py --gpus 1 --steps 20 --warmup 5 --trace): where do the 25 frames spend their time?
Usage (GPU box): python tools/driver_trace.py [extra bench args]"""
import json, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--trace"] + sys.argv[1:],
                   capture_output=True, text=True, cwd=root)
d = json.loads(p.stdout.strip().splitlines()[-1])
rows = d["frame_trace"]["device_ms_geometry_begin_done_shade_done"]
host = d["frame_trace"]["host_ms"]
print("value", round(d["value"], 1), "frames/s")
prev = None
for i, r in enumerate(rows):
    g0, g1, s1 = r
    print("frame %2d: geometry begins %8.3f  takes %6.3f  shade done %8.3f  period %s" % (i, g0, g1 - g0, s1, "%.3f" % (s1 - prev) if prev is not None else "-"))
    prev = s1
print("host:", [(k, round(t, 2)) for k, t in host if k in ("loop_begin", "t0", "enqueued", "t1")])
