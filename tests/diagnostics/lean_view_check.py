"""Diagnostic (GPU box): the random views of test_random_viewpoints_gbuffer_exact_and_routes_agree, lean route against the oracle under the conditioned
bound, for the library named by AWSM_HIP_LIB.  usage: python tests/diagnostics/lean_view_check.py [view ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from awsm_renderer_amd import scenes
from awsm_renderer_amd.hip_backend import HipDevice
from awsm_renderer_amd.scenes import look_at_rh
from oracle import oracle_lib
from tests import helpers

views = [int(a) for a in sys.argv[1:]] or list(range(12))
rng = np.random.default_rng(20260104)
sc = scenes.atrium_scene(1280, 720, detail=0.5, tex_scale=1 / 16)
lut = oracle_lib.brdf_lut(64, 64)
dev = HipDevice(parity_tap=True)
T = os.cpu_count() or 16
for k in range(12):
    eye = (float(rng.uniform(-5.5, 5.5)), float(rng.uniform(0.3, 9.5)), float(rng.uniform(-17.0, 17.0)))
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    if abs(d[1]) > 0.95:
        d = np.array([0.6, 0.5, -0.62])
    target = tuple(float(v) for v in np.asarray(eye) + 10.0 * d)
    if k not in views:
        continue
    sc.view, sc.camera_position = look_at_rh(eye, target), eye
    model = helpers.build_model(sc)
    helpers.hip_frame(model, lut, dev=dev)
    a = dev.read_opaque_f32().astype(np.float64)
    fr = oracle_lib.frame_from_model(model, lut).run(T)
    o = fr.rgba32f.astype(np.float64)
    cond = fr.conditioning(T)
    base = 1e-4 * np.maximum(1.0, np.abs(o))
    err = np.abs(a - o)
    over = (err > base + cond).any(axis=2)
    ob = (err > base).any(axis=2)
    print(f"view {k}: over plain {int(ob.sum())}, over conditioned {int(over.sum())}, worst ratio {float((err / (base + cond)).max()):.2f}", flush=True)
    ys, xs = np.nonzero(over)
    for y, x in list(zip(ys, xs))[:6]:
        print(f"   ({x},{y}) hip {a[y, x, :3]} oracle {o[y, x, :3]} err {err[y, x, :3].max():.3e} cond {cond[y, x, :3].max():.3e}")
dev.close()
