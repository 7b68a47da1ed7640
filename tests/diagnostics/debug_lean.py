"""Worst pixels of the lean opaque route against the oracle and against the general route (GPU box)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from awsm_renderer_amd import scenes
from awsm_renderer_amd.hip_backend import HipDevice
from oracle import oracle_lib
from tests import helpers

name = sys.argv[1] if len(sys.argv) > 1 else "helmet"
sc = {"helmet": lambda: scenes.helmet_scene(480, 270, segments=64, rings=48, tex_size=128),
      "atrium": lambda: scenes.atrium_scene(640, 360, detail=0.25, tex_scale=1 / 32),
      "zoo": lambda: scenes.material_zoo_scene(400, 300)}[name]()
lut = oracle_lib.brdf_lut(64, 64)
model = helpers.build_model(sc)
orc = helpers.oracle_frame(model, lut)
dev, _ = helpers.hip_frame(model, lut)
lean = dev.read_opaque_f32().astype(np.float64)
dev.close()
dev2, _ = helpers.hip_frame(model, lut, dev=HipDevice(parity_tap=True, general_shade_only=True))
gen = dev2.read_opaque_f32().astype(np.float64)
dev2.close()
ref = orc.rgba32f.astype(np.float64)
bound = 1e-4 * np.maximum(1.0, np.abs(ref))
for label, img in (("lean", lean), ("general", gen)):
    d = np.abs(img - ref)[..., :3] / bound[..., :3]
    worst = np.argsort(d.max(axis=2).ravel())[::-1][:6]
    print(label, "max err/bound", d.max(), "over", int((d.max(axis=2) > 1).sum()))
    for w in worst:
        y, x = divmod(int(w), sc.width)
        print("  px", x, y, "ref", ref[y, x, :3], "got", img[y, x, :3], "err/bound", d[y, x])
dl = np.abs(lean - gen)[..., :3]
print("lean vs general max abs", dl.max(), "mean", dl.mean())
