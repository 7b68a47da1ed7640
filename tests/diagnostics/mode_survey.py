"""Random viewpoints through the other modes of the path (GPU box): the material zoo (every optional PBR block, unlit, debug views, all sampler
modes, point + spot lights), the helmet, the skinned + morphed strip, the atrium with MSAA x4, with gradient mipmaps and with both, and the
transparent scene with its forward pass — each against the oracle.  Prints the compare_frames / compare_composite summary per view.
usage: python tests/diagnostics/mode_survey.py [views_per_mode] [mode ...]"""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from awsm_renderer_amd import scenes
from awsm_renderer_amd.hip_backend import HipDevice
from awsm_renderer_amd.scenes import look_at_rh
from oracle import oracle_lib
from tests import helpers

rng = np.random.default_rng(20260105)


def orbit_eye(center, rmin, rmax):
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    r = rng.uniform(rmin, rmax)
    eye = np.asarray(center) + r * d
    tgt = np.asarray(center) + rng.uniform(-0.6, 0.6, size=3)
    if abs(d[1]) > 0.97:
        eye = np.asarray(center) + r * np.array([0.5, 0.6, 0.62])
    return tuple(float(v) for v in eye), tuple(float(v) for v in tgt)


def inside_atrium():
    eye = (float(rng.uniform(-5.5, 5.5)), float(rng.uniform(0.3, 9.5)), float(rng.uniform(-17.0, 17.0)))
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    if abs(d[1]) > 0.95:
        d = np.array([0.6, 0.5, -0.62])
    return eye, tuple(float(v) for v in np.asarray(eye) + 10.0 * d)


MODES = {
    "zoo": (lambda: scenes.material_zoo_scene(640, 360), lambda: orbit_eye((0, 0, 0), 1.2, 6.0), {}),
    "helmet": (lambda: scenes.helmet_scene(640, 360, segments=48, rings=40, tex_size=128), lambda: orbit_eye((0, 0, 0), 1.3, 5.0), {}),
    "skinned": (lambda: scenes.skinned_morph_scene(640, 360, around=32, along=120, tex_size=64), lambda: orbit_eye((0, 0, 0), 1.0, 6.0), {}),
    "atrium_msaa": (lambda: scenes.atrium_scene(640, 360, detail=0.35, tex_scale=1 / 16), inside_atrium, {"msaa": 4}),
    "atrium_mips": (lambda: scenes.atrium_scene(640, 360, detail=0.35, tex_scale=1 / 16), inside_atrium, {"mipmap": True}),
    "atrium_msaa_mips": (lambda: scenes.atrium_scene(640, 360, detail=0.35, tex_scale=1 / 16), inside_atrium, {"msaa": 4, "mipmap": True}),
    "zoo_mips": (lambda: scenes.material_zoo_scene(640, 360), lambda: orbit_eye((0, 0, 0), 1.2, 6.0), {"mipmap": True}),
    "atrium_aniso": (lambda: scenes.atrium_scene(640, 360, detail=0.35, tex_scale=1 / 8), inside_atrium, {"mipmap": True, "anisotropic": True}),
    "atrium_msaa_aniso": (lambda: scenes.atrium_scene(640, 360, detail=0.35, tex_scale=1 / 8), inside_atrium, {"msaa": 4, "mipmap": True, "anisotropic": True}),
    "zoo_aniso": (lambda: scenes.material_zoo_scene(640, 360), lambda: orbit_eye((0, 0, 0), 1.2, 6.0), {"mipmap": True, "anisotropic": True}),
    "transparent": (lambda: scenes.transparent_scene(640, 360), lambda: orbit_eye((0, -0.3, 0), 1.5, 7.0), {"transparent": True}),
    "transparent_msaa": (lambda: scenes.transparent_scene(640, 360), lambda: orbit_eye((0, -0.3, 0), 1.5, 7.0), {"transparent": True, "msaa": 4}),
}

def survey(n_views, only=(), lut=None, seed=20260105):
    """Yields (mode, view, eye, compare_frames dict, compare_composite dict or None) for n_views random viewpoints per mode."""
    global rng
    rng = np.random.default_rng(seed)
    lut = oracle_lib.brdf_lut(64, 64) if lut is None else lut
    for name, (make, viewpoint, kw) in MODES.items():
        if only and name not in only:
            continue
        kw = dict(kw)
        sc = make()
        transparent = kw.pop("transparent", False)
        dev = HipDevice(parity_tap=True, anisotropic=bool(kw.get("anisotropic")))      # AWSM_CFG_ANISOTROPIC is a property of the context
        for k in range(n_views):
            eye, tgt = viewpoint()
            sc.view, sc.camera_position = look_at_rh(eye, tgt), eye
            model = helpers.build_model(sc)
            orc = oracle_lib.frame_from_model(model, lut, **kw).run(64)
            if transparent:
                orc.forward(model.collect_transparent_draws(), 64)
            helpers.hip_frame(model, lut, dev=dev, transparent=transparent, **kw)
            c = helpers.compare_frames(orc, dev, cond=orc.conditioning(64))      # the opaque image; conditioned bound (helpers.compare_frames)
            if not kw.get("msaa"):      # the STRICT G-buffer texel, value for value (awsm_hip_read_gbuffer: single-sampled frames)
                go, gh = orc.gbuffer(64), dev.read_gbuffer()
                hit = orc.keys != np.uint64(0xFFFFFFFFFFFFFFFF)
                c["gbuffer_mismatch"] = int(((go.view(np.uint32) != gh.view(np.uint32)) & hit[..., None]).any(axis=-1).sum())
            yield name, k, eye, c, (helpers.compare_composite(orc, dev) if transparent else None)
        dev.close()


if __name__ == "__main__":
    for name, k, eye, c, cc in survey(int(sys.argv[1]) if len(sys.argv) > 1 else 4, set(sys.argv[2:])):
        line = "%-18s view %d eye (%.2f %.2f %.2f): covered %7d keys %d verts %d/%d gbuffer %s | rgb over %4d worst %8.2f alpha %d f16ulp %d" % (
            name, k, *eye, c["covered"], c["key_mismatch"], c["clip_mismatch"], c["nt_mismatch"], c.get("gbuffer_mismatch", "-"), c["rgb_over_tol"], c["rgb_max_rel_to_bound"], c["alpha_mismatch"], c["f16_max_ulp"])
        if cc:
            line += " | composite: touched %d over2ulp %d max_ulp %d over_bound %d alpha %d untouched_changed %d" % (
                cc["touched_pixels"], cc["pixels_over_2ulp"], cc["max_ulp"], cc["pixels_over_bound"], cc["alpha_mismatch"], cc["untouched_changed"])
        print(line, flush=True)
