"""Host time per frame of the bench loop (camera move + awsm_host_render enqueue), measured on a frame whose GPU work is negligible
(the atrium's draw list at 160x90): what the frame rate is capped at when the GPU is not the limit.  GPU box: python tools/host_overhead.py"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from awsm_renderer_amd import scenes
from awsm_renderer_amd.host import Renderer

import sys as _s
FULL = len(_s.argv) > 1 and _s.argv[1] == "full"      # full: the 4K frame itself (is the host ahead of the GPU, or held back by it?)
scene = scenes.atrium_scene(3840, 2160) if FULL else scenes.atrium_scene(160, 90, tex_scale=1 / 32)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    r = Renderer(scene, device=0, stream=stream.cuda_stream, lut_size=64, overlap_frames=True)
    r.render(sync=True)
    r.host.set_render_timings(False)
    inv_view = np.linalg.inv(np.asarray(scene.view, dtype=np.float64).T)
    eye0 = np.asarray(scene.camera_position, dtype=np.float64)
    fwd, right, up = -inv_view[:3, 2], inv_view[:3, 0], inv_view[:3, 1]
    t_cam_math = t_cam = t_render = 0.0
    N = 400 if FULL else 2000
    per = []
    for i in range(N + 100):
        if i == 100:
            torch.cuda.synchronize(); t_cam_math = t_cam = t_render = 0.0; t_all = time.perf_counter()
        t0 = time.perf_counter()
        a = 2.0 * math.pi * (i % 240) / 240.0
        eye = eye0 + 0.6 * (math.cos(a) * right + math.sin(a) * up)
        view = scenes.look_at_rh(tuple(eye), tuple(eye + 30.0 * fwd))
        t1 = time.perf_counter()
        r.host.camera_update(view, scene.proj, tuple(eye))
        t2 = time.perf_counter()
        r.host.render(sync=False)
        t3 = time.perf_counter()
        t_cam_math += t1 - t0; t_cam += t2 - t1; t_render += t3 - t2
        if i >= 100: per.append((t3 - t2) * 1e6)
    t_enq = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_tot = time.perf_counter() - t_all
    print(f"per frame (us): camera math {t_cam_math / N * 1e6:.1f}  camera_update {t_cam / N * 1e6:.1f}  render enqueue {t_render / N * 1e6:.1f}  "
          f"loop {t_enq / N * 1e6:.1f}  incl. final sync {t_tot / N * 1e6:.1f}")
    per = np.array(per)
    print("render enqueue per frame (us): min %.0f  median %.0f  p90 %.0f  max %.0f; first 12 after warm-up: %s" % (per.min(), np.median(per), np.percentile(per, 90), per.max(), np.round(per[:12]).astype(int).tolist()))
