"""Scratch measurement of the transparent pass (uses the test-infrastructure model to feed the device)."""
import sys, time; sys.path.insert(0, '.')
from awsm_renderer_amd import scenes
from awsm_renderer_amd.hip_backend import HipDevice
from tests import helpers

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
detail = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
msaa = int(sys.argv[4]) if len(sys.argv) > 4 else 0
mip = int(sys.argv[5]) if len(sys.argv) > 5 else 0
steps = 20
sc = scenes.transparent_scene(W, H, tex_size=512, detail=detail)
model = helpers.build_model(sc)
dev = HipDevice()
dev.resize(W, H, msaa)
dev.upload_mirrors(model.mirrors())
for i, t in enumerate(model.texture_arrays()):
    if mip:
        dev.texture_array_upload(i, t["texels"], mips=max(t["width"], t["height"]).bit_length())
        dev.texture_array_generate_mips(i, t["kinds"])
    else:
        dev.texture_array_upload(i, t["texels"])
for i, s in enumerate(sc.samplers):
    dev.sampler_set(i, s)
dev.env_upload(sc.skybox_rgba, sc.prefiltered_rgb, sc.irradiance_rgb)
dev.brdf_lut_generate(256, 256)
od = model.collect_draws(); td = model.collect_transparent_draws()
draws = HipDevice.make_draws(od); tdraws = HipDevice.make_draws(td)
acc = {}
for it in range(steps + 3):
    dev.geometry_pass(draws, len(od)); dev.opaque_pass(mipmap=mip); dev.transparent_pass(tdraws, len(td)); st = dev.frame_end()
    if it >= 3:
        for k, v in st.items():
            acc[k] = acc.get(k, 0) + v
print(f"{W}x{H} detail {detail} msaa {msaa} mip {mip}:", {k: round(v / steps, 4) for k, v in acc.items()})
