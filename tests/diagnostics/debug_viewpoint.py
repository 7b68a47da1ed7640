"""One random viewpoint of tests/diagnostics/viewpoint_survey.py in detail (GPU box): the pixels where the HIP path and the oracle disagree."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from awsm_renderer_amd import scenes
from awsm_renderer_amd.hip_backend import HipDevice
from awsm_renderer_amd.scenes import look_at_rh
from oracle import oracle_lib
from tests import helpers
want = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rng = np.random.default_rng(20260104)
sc = scenes.atrium_scene(1280, 720, detail=0.5, tex_scale=1 / 16)
lut = oracle_lib.brdf_lut(64, 64)
for k in range(want + 1):
    eye = (float(rng.uniform(-5.5, 5.5)), float(rng.uniform(0.3, 9.5)), float(rng.uniform(-17.0, 17.0)))
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    if abs(d[1]) > 0.95:
        d = np.array([0.6, 0.5, -0.62])
    target = tuple(float(v) for v in np.asarray(eye) + 10.0 * d)
print("view", want, "eye", eye, "target", target)
sc.view, sc.camera_position = look_at_rh(eye, target), eye
dbg = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # material debug view: 1 base colour, 2 metallic-roughness, 4 normal, 8 occlusion, 16 emissive
for m in sc.materials:
    m.debug_bitmask = dbg & 63
    if dbg & 64:
        m.normal_tex = None      # debug 4 then shows the G-buffer normal itself
if dbg & 128:      # every normal map = the constant (1, 0, 0): debug 4 then shows the G-buffer tangent; 256: (0, 1, 0) -> the bitangent
    for m in sc.materials:
        if m.normal_tex is not None:
            sc.textures[m.normal_tex.texture][...] = (255, 128, 128, 255)
if dbg & 256:
    for m in sc.materials:
        if m.normal_tex is not None:
            sc.textures[m.normal_tex.texture][...] = (128, 255, 128, 255)
model = helpers.build_model(sc)
orc = helpers.oracle_frame(model, lut, threads=64)
gd = HipDevice(parity_tap=True, general_shade_only=True)
helpers.hip_frame(model, lut, dev=gd)
b, o = gd.read_opaque_f32().astype(np.float64), orc.rgba32f.astype(np.float64)
keys = gd.read_visibility()
r = (np.abs(b - o) / (1e-4 * np.maximum(1.0, np.abs(o)))).max(axis=2)
ys, xs = np.nonzero(r > 1)
draws = model.collect_draws()
first = np.cumsum([0] + [int(dr["tri_count"]) * max(1, int(dr.get("inst_count", 1) or 1)) for dr in draws]) if isinstance(draws[0], dict) else None
print(len(ys), "pixels over the bound")
for y, x in list(zip(ys, xs))[:24]:
    rank = int(0xFFFFFFFF - (int(keys[y, x]) & 0xFFFFFFFF)); depth = np.uint32(int(keys[y, x]) >> 32).view(np.float32)
    print("(%4d,%3d) ratio %7.1f hip %s oracle %s rank %d depth %.6f" % (x, y, r[y, x], np.round(b[y, x, :3], 4), np.round(o[y, x, :3], 4), rank, depth))
# G-buffer: the STRICT reconstruction value for value
go, gh = orc.gbuffer(64) if hasattr(orc, "gbuffer") else None, gd.read_gbuffer()
if go is not None:
    hit = keys != np.uint64(0xFFFFFFFFFFFFFFFF)
    neq = (go.view(np.uint32) != gh.view(np.uint32)) & hit[..., None]
    print("G-buffer values that differ (packed_nt.xyzw, bx, by):", neq.sum(axis=(0, 1)).tolist(), "of", int(hit.sum()), "covered pixels")
    ys2, xs2 = np.nonzero(neq.any(axis=2))
    for y, x in list(zip(ys2, xs2))[:12]:
        print("(%4d,%3d) hip %s oracle %s" % (x, y, np.round(gh[y, x], 6).tolist(), np.round(go[y, x], 6).tolist()))
import math
def decode(pk):
    ex, ey = pk[0] * 2 - 1, pk[1] * 2 - 1
    n = np.array([ex, ey, 1 - abs(ex) - abs(ey)]); t = min(max(-n[2], 0), 1)
    n[0] += -t if n[0] >= 0 else t; n[1] += -t if n[1] >= 0 else t; n /= np.linalg.norm(n)
    theta = pk[2] * 2 * math.pi - math.pi
    a = 1 / (1 + n[2]); bb = -n[0] * n[1] * a
    tt = np.array([1 - n[0] * n[0] * a, bb, -n[0]]); tb = np.array([bb, 1 - n[1] * n[1] * a, -n[1]])
    T = tt * math.cos(theta) + tb * math.sin(theta); T /= np.linalg.norm(T)
    return n, T, theta
for y, x in list(zip(ys, xs))[:6]:
    n, T, th = decode(gh[y, x].astype(np.float64))
    print("(%d,%d) packed %s -> N %s T %s theta %.6f | view colour hip %s oracle %s" % (x, y, gh[y, x].tolist(), np.round(n, 5), np.round(T, 5), th, np.round(b[y, x, :3] * 2 - 1, 5), np.round(o[y, x, :3] * 2 - 1, 5)))
