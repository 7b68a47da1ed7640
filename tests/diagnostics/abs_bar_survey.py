"""Which operation is behind the pixels over the ABSOLUTE 1e-4 colour bar?  GPU box.

Renders the full-size frames of BASELINE configs[1] / [3] (the latter also in the reference's default mode), compares every pixel with the oracle and,
for each pixel whose difference exceeds 1e-4 absolute, prints how far the oracle's OWN colour moves under each of its five input perturbations
(oracle_shade.c: oracle_set_perturbation — 1 / 2: decoded normal tilted 16 ulp along T / B, 3 / 4: reconstructed position shifted 16 ulp across the view ray,
5: n.h of the GGX lobe 16 ulp down), i.e. which input the pixel is ill-conditioned in.  Usage: python tests/diagnostics/abs_bar_survey.py [helmet|atrium|default]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from awsm_renderer_amd import scenes
from oracle import oracle_lib
from tests import helpers

which = sys.argv[1:] or ["atrium", "default", "helmet"]
lut = oracle_lib.brdf_lut(64, 64)
threads = max(8, min(len(os.sched_getaffinity(0)), 128))
for name in which:
    sc = scenes.helmet_scene() if name == "helmet" else scenes.atrium_scene(3840, 2160)
    kw = dict(msaa=4, mipmap=True) if name == "default" else {}
    model = helpers.build_model(sc)
    dev, _ = helpers.hip_frame(model, lut, **kw)
    orc = helpers.oracle_frame(model, lut, threads=threads, **kw)
    got = dev.read_opaque_f32().astype(np.float64)
    ref = orc.rgba32f.astype(np.float64)
    dev.close()
    diff = np.abs(got - ref)[..., :3]
    over = (diff > 1e-4).any(axis=-1)
    ys, xs = np.nonzero(over)
    print(f"== {name}: {len(ys)} pixels over 1e-4 absolute; {int(((np.abs(ref[..., :3]) <= 1.0).all(axis=-1) & over).sum())} of them with every channel <= 1.0", flush=True)
    resp = []
    base32 = orc.rgba32f.copy()
    for k in (1, 2, 3, 4, 5):
        oracle_lib.lib().oracle_set_perturbation(C.c_int(k))
        orc.shade(threads)
        resp.append(np.abs(orc.rgba32f.astype(np.float64) - base32.astype(np.float64))[..., :3].max(axis=-1))
    oracle_lib.lib().oracle_set_perturbation(C.c_int(0))
    resp = np.stack(resp)        # [5][H][W]
    tally = np.zeros(6, dtype=int)
    for y, x in zip(ys, xs):
        r = resp[:, y, x]
        d = diff[y, x].max()
        dom = int(r.argmax()) + 1 if r.max() > 0.25 * d else 0
        tally[dom] += 1
        print("  (%4d,%4d) ref %-28s diff %.2e | response to normal-T %.1e normal-B %.1e pos-1 %.1e pos-2 %.1e n.h %.1e  -> %s" % (
            x, y, np.array2string(ref[y, x, :3], precision=4), d, r[0], r[1], r[2], r[3], r[4],
            ("unexplained", "normal", "normal", "position", "position", "n.h")[dom]), flush=True)
    print("  tally: unexplained %d, normal %d, position %d, n.h %d" % (tally[0], tally[1] + tally[2], tally[3] + tally[4], tally[5]), flush=True)
