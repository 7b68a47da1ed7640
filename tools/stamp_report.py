"""Runs the 4K atrium on build/variants/lib_STAMP.so and reports, per geometry kernel, the phase durations of its workgroups (GPU box)."""
import ctypes as C, os, sys
os.environ["AWSM_HIP_LIB"] = "build/variants/lib_STAMP.so"
sys.path.insert(0, ".")
import numpy as np
from awsm_renderer_amd import scenes
from awsm_renderer_amd.hip_backend import HipDevice
from tests import helpers
sc = scenes.atrium_scene(3840, 2160, tex_scale=1 / 16)
model = helpers.build_model(sc)
dev = HipDevice()
msaa = int(sys.argv[1]) if len(sys.argv) > 1 else 0      # 4: k_raster_tile<4>
dev.resize(3840, 2160, msaa)
dev.upload_mirrors(model.mirrors())
for i, t in enumerate(model.texture_arrays()):
    dev.texture_array_upload(i, t["texels"])
for i, s in enumerate(sc.samplers):
    dev.sampler_set(i, s)
dev.env_upload(sc.skybox_rgba, sc.prefiltered_rgb, sc.irradiance_rgb)
dev.brdf_lut_generate(64, 64)
dev.set_stage_timers(False) if hasattr(dev, "set_stage_timers") else None
draws = model.collect_draws()
for _ in range(5):
    dev.geometry_pass(draws); dev.opaque_pass(); dev.frame_end()
st = np.zeros((4, 16384, 8), dtype=np.uint64)
dev.lib.awsm_hip_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert dev.lib.awsm_hip_debug_read_stamps(dev.ctx, st.ctypes.data_as(C.c_void_p)) == 0
names = ["k_bin<count>", "k_bin_scan", "k_bin<fill>", "k_raster_tile"]
t0 = st[st > 0].min()
for k in range(4):
    s = st[k].astype(np.float64)
    used = s[:, 0] > 0
    if not used.any():
        continue
    s = (s[used] - float(t0)) * 0.01          # us (100 MHz)
    s[s < 0] = np.nan
    last = np.nanmax(s, axis=1)
    print(f"{names[k]}: {used.sum()} workgroups; first start {np.nanmin(s[:, 0]):.2f} us, last start {np.nanmax(s[:, 0]):.2f}, last end {np.nanmax(last):.2f}; workgroup life: median {np.nanmedian(last - s[:, 0]):.2f} us, max {np.nanmax(last - s[:, 0]):.2f}")
    if k in (0, 2) and np.isfinite(s[:, 7]).any():
        d = s[:, 7] - s[:, 0]; e = s[:, 1] - s[:, 7]
        print(f"    start -> triangle loaded / set up: median {np.nanmedian(d):.2f} us, p95 {np.nanpercentile(d, 95):.2f}, max {np.nanmax(d):.2f};  -> end of phase 0: median {np.nanmedian(e):.2f}, p95 {np.nanpercentile(e, 95):.2f}, max {np.nanmax(e):.2f}")
        slow = np.argsort(-(s[:, 1] - s[:, 0]))[:12]
        print("    slowest workgroups in phase 0:", [(int(i), round(float(s[i, 7] - s[i, 0]), 1), round(float(s[i, 1] - s[i, 7]), 1)) for i in slow])
    if k == 3:      # how full the machine is over the kernel's life: workgroups alive per 5-us bin (256 CUs x up to 7), and where the time of a workgroup goes
        t_begin, t_end = np.nanmin(s[:, 0]), np.nanmax(last)
        edges = np.arange(t_begin, t_end + 5.0, 5.0)
        alive = [int(((s[:, 0] < e + 5.0) & (last > e)).sum()) for e in edges[:-1]]
        print(f"    kernel {t_end - t_begin:.1f} us; sum of workgroup lives {np.nansum(last - s[:, 0]):.0f} us = {np.nansum(last - s[:, 0]) / (t_end - t_begin):.0f} workgroups alive on average; alive per 5-us bin: {alive}")
        order = np.argsort(-(last - s[:, 0]))[:8]
        print("    longest workgroups (id, start, life):", [(int(i), round(float(s[i, 0] - t_begin), 1), round(float(last[i] - s[i, 0]), 1)) for i in order])
        ends = np.sort(last)[-8:] - t_begin
        print("    last workgroups end at:", [round(float(e), 1) for e in ends])
    for j in range(1, 7):
        d = s[:, j] - s[:, j - 1]
        if np.isfinite(d).any():
            print(f"    phase {j - 1}->{j}: median {np.nanmedian(d):.2f} us, p95 {np.nanpercentile(d, 95):.2f}, max {np.nanmax(d):.2f}  (n={np.isfinite(d).sum()})")
dev.close()
