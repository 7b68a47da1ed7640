"""Scratch measurement harness (uses the test-infrastructure model to feed the device; bench.py is the real one)."""
import sys, time, json; sys.path.insert(0, '.')
import numpy as np
from awsm_renderer_amd import scenes
from awsm_renderer_amd.hip_backend import HipDevice
from tests import helpers

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
msaa = int(sys.argv[4]) if len(sys.argv) > 4 else 0
tex_scale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
t0 = time.time()
sc = scenes.atrium_scene(W, H, tex_scale=tex_scale)
model = helpers.build_model(sc)
print("scene+model %.1fs" % (time.time() - t0), flush=True)
dev = HipDevice()
print(dev.device_info())
dev.resize(W, H, msaa)
dev.upload_mirrors(model.mirrors())
for i, t in enumerate(model.texture_arrays()):
    dev.texture_array_upload(i, t["texels"])
for i, s in enumerate(sc.samplers):
    dev.sampler_set(i, s)
dev.env_upload(sc.skybox_rgba, sc.prefiltered_rgb, sc.irradiance_rgb)
dev.brdf_lut_generate(1024, 1024)
draws = HipDevice.make_draws(model.collect_draws())
n = len(model.collect_draws())
for _ in range(3):
    dev.geometry_pass(draws, n); dev.opaque_pass(); st = dev.frame_end()
print("warm", st, flush=True)
acc = {}
for _ in range(steps):
    dev.geometry_pass(draws, n); dev.opaque_pass(); st = dev.frame_end()
    for k, v in st.items():
        acc[k] = acc.get(k, 0) + v
print({k: v / steps for k, v in acc.items()})
t = time.time()
for _ in range(steps):
    dev.geometry_pass(draws, n); dev.opaque_pass()
t_enq = (time.time() - t) / steps
st = dev.frame_end()
dt = (time.time() - t) / steps
print("host enqueue: %.3f ms/frame" % (t_enq * 1e3))
print("pipelined: %.3f ms/frame  %.1f fps  %.1f Mpix/s" % (dt * 1e3, 1 / dt, W * H / dt / 1e6))
