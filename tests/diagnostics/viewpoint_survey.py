"""Random viewpoints inside the atrium: lean route and general route against the oracle and against each other (GPU box).  Prints, per
viewpoint, the number of pixels beyond the shading tolerance and the worst ratio — the data behind the criteria of
tests/test_gpu_parity.py::test_random_viewpoints_gbuffer_exact_and_routes_agree."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from awsm_renderer_amd import scenes
from awsm_renderer_amd.hip_backend import HipDevice
from awsm_renderer_amd.scenes import look_at_rh
from oracle import oracle_lib
from tests import helpers
n_views = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(20260104)
sc = scenes.atrium_scene(1280, 720, detail=0.5, tex_scale=1 / 16)
lut = oracle_lib.brdf_lut(64, 64)
ld, gd = HipDevice(parity_tap=True), HipDevice(parity_tap=True, general_shade_only=True)
for k in range(n_views):
    eye = (float(rng.uniform(-5.5, 5.5)), float(rng.uniform(0.3, 9.5)), float(rng.uniform(-17.0, 17.0)))
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    if abs(d[1]) > 0.95:
        d = np.array([0.6, 0.5, -0.62])
    target = tuple(float(v) for v in np.asarray(eye) + 10.0 * d)
    sc.view, sc.camera_position = look_at_rh(eye, target), eye
    model = helpers.build_model(sc)
    orc = helpers.oracle_frame(model, lut, threads=64)
    helpers.hip_frame(model, lut, dev=ld); helpers.hip_frame(model, lut, dev=gd)
    a, b, o = ld.read_opaque_f32().astype(np.float64), gd.read_opaque_f32().astype(np.float64), orc.rgba32f.astype(np.float64)
    keys_ok = bool((ld.read_visibility() == orc.keys).all()) if hasattr(orc, "keys") else None
    bound = 1e-4 * np.maximum(1.0, np.abs(o))
    ra, rb = np.abs(a - o) / bound, np.abs(b - o) / bound
    rl = np.abs(a - b) / (1e-5 * np.maximum(1.0, np.abs(b)))
    print("view %2d: lean vs oracle: %3d px over 1e-4, worst %.2f | general vs oracle: %3d px, worst %.2f | lean vs general: %3d px over 1e-5, worst %.1f | keys %s"
          % (k, int((ra > 1).any(axis=2).sum()), ra.max(), int((rb > 1).any(axis=2).sum()), rb.max(), int((rl > 1).any(axis=2).sum()), rl.max(), keys_ok), flush=True)
