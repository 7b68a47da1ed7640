"""Analytic known-answer checks for the parts of the oracle the reference holds no vectors for ("parity unpinned").  They are
not a pin — nothing but the reference's own output could be — but a second, independent derivation in numpy (closed forms from the glTF /
WebGPU formulas the WGSL implements) catches a transcription slip that oracle and kernels would share.  Reference lines are cited per check."""
import math

import numpy as np
import pytest

from awsm_renderer_amd import scenes
from awsm_renderer_amd.scene_desc import MaterialDesc, NodeDesc, PrimitiveDesc, SceneDesc
from oracle import oracle_lib, scene_model as sm
from tests import helpers

F = np.float32


def _quad_scene(material, lights, n=65, z=0.0, eye=(0.0, 0.0, 3.0), tilt=None):
    pos = np.array([[-1, -1, z], [1, -1, z], [1, 1, z], [-1, 1, z]], dtype=F)
    nrm = np.tile(np.array([[0, 0, 1]], dtype=F), (4, 1))
    prim = PrimitiveDesc(positions=pos, normals=nrm, indices=np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32), material=0,
                         uvs=[np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=F)])
    node = NodeDesc(parent=0, primitives=[prim]) if tilt is None else NodeDesc(parent=0, rotation=tilt, primitives=[prim])
    return SceneDesc(nodes=[NodeDesc(), node], materials=[material], samplers=[dict(scenes.REPEAT_LINEAR)], lights=lights, width=n, height=n,
                     view=scenes.look_at_rh(eye, (0, 0, 0)), proj=scenes.perspective_rh(math.radians(40), 1.0, 0.1, 50.0), camera_position=eye,
                     prefiltered_rgb=(0, 0, 0), irradiance_rgb=(0, 0, 0))


def _cook_torrance(base, metallic, roughness, n, v, l, radiance, occlusion=1.0):
    """brdf.wgsl:104-140,308-381 (glTF 2.0 appendix B with this renderer's constants): Schlick Fresnel, GGX D = a^2 / (pi d^2 + 1e-4),
    Schlick-GGX k = (alpha + 1)^2 / 8, spec = D G / max(4 n.l n.v, 1e-4), k_d = 1 - max(F); alpha = roughness^2; f0 = 0.04 for ior 1.5."""
    n, v, l = (np.asarray(x, dtype=np.float64) / np.linalg.norm(x) for x in (n, v, l))
    h = (v + l) / np.linalg.norm(v + l)
    ndl, ndv, ndh, vdh = max(n @ l, 0.0), max(n @ v, 1e-4), max(n @ h, 0.0), max(v @ h, 0.0)
    rough = max(min(max(roughness, 0.0), 1.0), 0.04)
    alpha = max(rough * rough, 0.001)
    f0 = 0.04 * (1.0 - metallic) + np.asarray(base, dtype=np.float64) * metallic
    Fr = f0 + (1.0 - f0) * (1.0 - vdh) ** 5
    d = ndh * ndh * (alpha * alpha - 1.0) + 1.0
    D = alpha * alpha / (math.pi * d * d + 1e-4)
    k = (alpha + 1.0) ** 2 / 8.0
    G = (ndv / (ndv * (1 - k) + k)) * (ndl / (ndl * (1 - k) + k))
    spec = D * G / max(4.0 * ndl * ndv, 1e-4)
    kd = 1.0 - Fr.max()
    return (np.asarray(base, dtype=np.float64) * (1.0 - metallic) / math.pi * kd + Fr * spec) * np.asarray(radiance, dtype=np.float64) * ndl * occlusion


def test_one_directional_light_on_a_facing_quad_matches_the_closed_form():
    """lights.rs:354-473 + lights.wgsl:70-152 + brdf.wgsl:308-381 + standard.wgsl:11-62 at the pixel on the optical axis (odd frame size:
    its centre is the axis, so N = V = (0, 0, 1) exactly; black IBL cubes, so the pixel is the one light's Cook-Torrance term)."""
    lut = oracle_lib.brdf_lut(32, 32)
    for base, metallic, rough, ldir, colour, intensity in (((0.8, 0.5, 0.2, 1.0), 0.0, 0.5, (0, 0, -1), (1, 1, 1), 2.0),
                                                          ((0.9, 0.9, 0.9, 1.0), 1.0, 0.3, (0.3, -0.2, -1.0), (1.0, 0.8, 0.6), 3.0),
                                                          ((0.2, 0.7, 0.4, 1.0), 0.4, 0.9, (-0.5, 0.5, -0.6), (0.5, 0.9, 1.0), 1.5)):
        mat = MaterialDesc(base_color_factor=base, metallic_factor=metallic, roughness_factor=rough)
        sc = _quad_scene(mat, [{"kind": "directional", "color": colour, "intensity": intensity, "direction": ldir}])
        fr = helpers.oracle_frame(helpers.build_model(sc), lut)
        c = sc.width // 2
        got = fr.rgba32f[c, c, :3].astype(np.float64)
        want = _cook_torrance(base[:3], metallic, rough, (0, 0, 1), (0, 0, 1), -np.asarray(ldir, dtype=np.float64), np.asarray(colour) * intensity)
        assert np.allclose(got, want, rtol=2e-4, atol=2e-6), (got, want)
        assert fr.rgba32f[c, c, 3] == 1.0


def test_point_and_spot_attenuation_closed_forms():
    """math.wgsl:12-19 inverse_square (range 0: 1 / max(d^2, 0.01); else clamp((1 - d^2 / r^2)^2, 0, 1) / (d^2 + 1)) and lights.wgsl:95-118 spot
    smooth falloff ((cos - outer) / (inner - outer))^2, the two cone values being cosines: lights.rs:447-468 writes inner_angle / outer_angle
    unchanged and lights.wgsl:96-97 compares them with cos_l."""
    lut = oracle_lib.brdf_lut(32, 32)
    mat = MaterialDesc(base_color_factor=(0.6, 0.6, 0.6, 1.0), metallic_factor=0.0, roughness_factor=0.7)
    for light, att in (({"kind": "point", "color": (1, 1, 1), "intensity": 5.0, "position": (0.0, 0.0, 2.0), "range": 0.0}, 1.0 / 4.0),
                       ({"kind": "point", "color": (1, 1, 1), "intensity": 5.0, "position": (0.0, 0.0, 2.0), "range": 4.0}, (1.0 - 4.0 / 16.0) ** 2 / 5.0),
                       ({"kind": "spot", "color": (1, 1, 1), "intensity": 5.0, "position": (0.0, 0.0, 2.0), "range": 0.0, "direction": (0.0, 0.0, -1.0),
                         "inner_angle": math.cos(math.radians(10)), "outer_angle": math.cos(math.radians(30))}, 1.0 / 4.0),
                       ({"kind": "spot", "color": (1, 1, 1), "intensity": 5.0, "position": (0.0, 0.0, 2.0), "range": 0.0,
                         "direction": (math.sin(math.radians(20)), 0.0, -math.cos(math.radians(20))),
                         "inner_angle": math.cos(math.radians(10)), "outer_angle": math.cos(math.radians(30))},
                        0.25 * ((math.cos(math.radians(20)) - math.cos(math.radians(30))) / (math.cos(math.radians(10)) - math.cos(math.radians(30)))) ** 2)):
        sc = _quad_scene(mat, [light])
        fr = helpers.oracle_frame(helpers.build_model(sc), lut)
        c = sc.width // 2
        want = _cook_torrance((0.6, 0.6, 0.6), 0.0, 0.7, (0, 0, 1), (0, 0, 1), (0, 0, 1), np.ones(3) * 5.0 * att)
        assert np.allclose(fr.rgba32f[c, c, :3], want, rtol=2e-4, atol=2e-6), (light["kind"], fr.rgba32f[c, c, :3], want)


def test_ibl_only_pixel_is_the_split_sum():
    """brdf.wgsl:389-514 with no punctual light and uniform cubes E (irradiance) and P (prefiltered): colour = base / pi * E * (1 - F_max) (1 - m)
    + P (F0 A + f90 B) + emissive, A, B = the LUT at (n.v, roughness) — here n.v = 1 on the optical axis."""
    lut = oracle_lib.brdf_lut(64, 64)
    base, metallic, rough = (0.7, 0.3, 0.2), 0.25, 0.6
    mat = MaterialDesc(base_color_factor=base + (1.0,), metallic_factor=metallic, roughness_factor=rough, emissive_factor=(0.05, 0.1, 0.15))
    sc = _quad_scene(mat, [])
    sc.irradiance_rgb, sc.prefiltered_rgb = (0.5, 0.6, 0.7), (1.5, 1.0, 0.5)
    fr = helpers.oracle_frame(helpers.build_model(sc), lut)
    c = sc.width // 2
    lut_f = lut.view(np.float16).astype(np.float64)                     # [row = v][col = u][A, B]; brdf.wgsl:293-302 samples at (n.v, roughness), linear, clamp
    W = lut.shape[1]
    y = rough * lut.shape[0] - 0.5
    j0 = int(math.floor(y)); fy = y - j0
    AB = lut_f[j0, W - 1] * (1 - fy) + lut_f[j0 + 1, W - 1] * fy        # u = 1: the last column (clamped)
    f0 = 0.04 * (1 - metallic) + np.asarray(base) * metallic
    Fv = f0                                                              # (1 - n.v)^5 = 0
    want = np.asarray(base) / math.pi * np.asarray(sc.irradiance_rgb) * (1 - Fv.max()) * (1 - metallic) + np.asarray(sc.prefiltered_rgb) * (f0 * AB[0] + AB[1]) + np.asarray(mat.emissive_factor)
    assert np.allclose(fr.rgba32f[c, c, :3], want, rtol=3e-4, atol=2e-6), (fr.rgba32f[c, c, :3], want)


def test_camera_ubo_and_transform_mirrors_closed_forms():
    """camera.rs:111-227 + glam 0.31 perspective_rh (depth 0..1): m00 = f / aspect, m11 = f, m22 = far / (near - far), m23 = -1, m32 = near far / (near - far);
    transforms.rs:396-410: world = parent * local, TRS -> Mat4::from_scale_rotation_translation."""
    sc = scenes.box_scene(64, 48)
    model = helpers.build_model(sc)
    cam = np.frombuffer(bytes(model.mirrors()[sm.BUF_CAMERA]), dtype=np.float32)
    proj = cam[16:32].reshape(4, 4)                                      # [col][row]
    f = 1.0 / math.tan(math.radians(45) / 2)
    near, far = 0.1, 100.0
    want = np.zeros((4, 4))
    want[0, 0], want[1, 1], want[2, 2], want[2, 3], want[3, 2] = f / (64 / 48), f, far / (near - far), -1.0, near * far / (near - far)
    assert np.allclose(proj, want, rtol=1e-6, atol=1e-7)
    view, view_proj, inv_view = cam[0:16].reshape(4, 4), cam[32:48].reshape(4, 4), cam[80:96].reshape(4, 4)
    assert np.allclose(view_proj.T, proj.T.astype(np.float64) @ view.T.astype(np.float64), rtol=1e-5, atol=1e-6)      # view_proj = proj * view
    assert np.allclose(inv_view.T @ view.T, np.eye(4), atol=1e-5)
    assert np.allclose(cam[96:99], sc.camera_position, atol=1e-6)
    # the box's node chain: a -90 degree rotation about X maps +y to -z (glTF Box: "Y up" asset stored Z up)
    world = np.frombuffer(bytes(model.mirrors()[sm.BUF_TRANSFORMS]), dtype=np.float32).reshape(-1, 4, 4)
    rot = np.array([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)          # [col][row]
    assert any(np.allclose(w, rot, atol=1e-6) for w in world)


def test_raster_top_left_rule_and_depth_on_a_hand_computed_case():
    """pipeline.rs:337-344 (TriangleList, CCW front, LessEqual) under the raster contract: a pixel-aligned right triangle covers exactly the
    pixels whose centres lie inside or on its top / left edges; depth = the plane's value at the pixel centre."""
    s = oracle_lib.OracleScene
    # an 8x8 frame, identity view-projection: clip = position; triangle (0,0) (8,0) (0,8) in pixels, CCW in NDC (y up) = front facing
    def ndc(px, py):
        return (px / 4.0 - 1.0, 1.0 - py / 4.0)
    pos = np.array([[*ndc(0, 8), 0.25], [*ndc(8, 8), 0.75], [*ndc(0, 0), 0.25]], dtype=F)
    prim = PrimitiveDesc(positions=pos, normals=np.tile(np.array([[0, 0, 1]], dtype=F), (3, 1)), indices=np.array([[0, 1, 2]], dtype=np.uint32), material=0)
    eye4 = np.eye(4, dtype=F)
    sc = SceneDesc(nodes=[NodeDesc(), NodeDesc(parent=0, primitives=[prim])], materials=[MaterialDesc()], samplers=[dict(scenes.REPEAT_LINEAR)], lights=[],
                   width=8, height=8, view=eye4, proj=eye4, camera_position=(0, 0, 1))
    fr = helpers.oracle_frame(helpers.build_model(sc), oracle_lib.brdf_lut(8, 8))
    hit = fr.keys != helpers.NO_HIT
    # the hypotenuse runs from (8, 8) to (0, 0): pixel (x, y) is inside iff its centre has x + 0.5 < y + 0.5, i.e. x < y; on the diagonal (x == y)
    # the centre lies ON the edge, which is neither a top nor a left edge of this triangle -> not covered
    want = np.array([[x < y for x in range(8)] for y in range(8)])
    assert (hit == want).all(), hit.astype(int)
    depth = (fr.keys >> np.uint64(32)).astype(np.uint32).view(np.float32)
    for (y, x) in ((7, 0), (7, 6), (4, 1)):
        assert abs(float(depth[y, x]) - (0.25 + 0.5 * (x + 0.5) / 8.0)) < 1e-6          # z rises linearly along x


def test_depth_key_test_as_one_add_and_one_unsigned_compare():
    """raster_setup.hpp depth_key_bits: the key's depth test "zn >= 0 && zn <= 1, -0 stored as +0" (oracle_geometry.c tri_sample) is computed on the device as
    bits(zn + 0.0f) <= 0x3F800000 — restated here in numpy and compared with the oracle's spelling over every special value, every binade boundary and
    four million random bit patterns (both the decision and the stored bits)."""
    rng = np.random.default_rng(7)
    special = np.array([0x00000000, 0x80000000, 0x00000001, 0x80000001, 0x007FFFFF, 0x807FFFFF, 0x00800000, 0x80800000, 0x3F7FFFFF, 0x3F800000, 0x3F800001,
                        0xBF800000, 0x7F7FFFFF, 0x7F800000, 0xFF800000, 0x7F800001, 0x7FC00000, 0xFFC00000, 0xFFFFFFFF], dtype=np.uint32)
    edges = np.array([(e << 23) + d for e in range(256) for d in (0, 1, 0x7FFFFF)] + [0x80000000 + (e << 23) + d for e in range(256) for d in (0, 1, 0x7FFFFF)], dtype=np.uint64).astype(np.uint32)
    bits = np.concatenate([special, edges, rng.integers(0, 1 << 32, size=4_000_000, dtype=np.uint64).astype(np.uint32),
                           rng.integers(0x3F000000, 0x3F800010, size=200_000, dtype=np.uint64).astype(np.uint32)])
    zn = bits.view(np.float32)
    with np.errstate(invalid="ignore"):
        want_in = (zn >= np.float32(0.0)) & (zn <= np.float32(1.0))
        want_bits = np.where(zn == np.float32(0.0), np.float32(0.0), zn).view(np.uint32)        # -0 -> +0
        got_bits = (zn + np.float32(0.0)).view(np.uint32)
    got_in = got_bits <= np.uint32(0x3F800000)
    assert (got_in == want_in).all()
    assert (got_bits[want_in] == want_bits[want_in]).all()
    # depth_key_bits_sum: with zc canonicalised once, zc0 + dz needs no second canonicalisation — a sum is -0 only when both terms are
    a = np.concatenate([special, rng.integers(0, 1 << 32, size=500_000, dtype=np.uint64).astype(np.uint32)]).view(np.float32)
    b = np.concatenate([special[::-1], rng.integers(0, 1 << 32, size=500_000, dtype=np.uint64).astype(np.uint32)]).view(np.float32)
    b = np.where(rng.random(b.size) < 0.2, -a, b).astype(np.float32)                                       # cancellations too
    with np.errstate(invalid="ignore", over="ignore"):
        ref = a + b
        ref = np.where(ref == np.float32(0.0), np.float32(0.0), ref)                                       # the oracle: zn = zc + dz, then -0 -> +0
        got = (a + np.float32(0.0)) + b
    same = (ref.view(np.uint32) == got.view(np.uint32)) | (np.isnan(ref) & np.isnan(got))
    assert same.all()


def test_msaa_sample_depth_is_the_plane_at_the_sample():
    """The multisampled raster rule (oracle_geometry.c tri_sample_msaa): coverage from the exact edge values at the sample, depth = the
    plane at the pixel's corner + the sample's step along the gradient.  Against the planes themselves, in f64, at WebGPU's four standard positions: 120
    random triangles with random depths in one 96x64 frame, identity view-projection.  Every sample's depth must be within a few f32 steps of its winning
    triangle's plane (the three-term form this rule replaced did no better); a sample strictly inside some triangle is covered, one strictly outside all
    of them is not."""
    rng = np.random.default_rng(404)
    W, H, N = 96, 64, 120
    offs = np.array([[0.375, 0.125], [0.875, 0.375], [0.125, 0.625], [0.625, 0.875]])
    pos, idx = [], []
    while len(idx) < N:
        p = rng.uniform([-4, -4], [W + 4, H + 4], size=(3, 2))
        d1, d2 = p[1] - p[0], p[2] - p[0]
        if abs(d1[0] * d2[1] - d1[1] * d2[0]) < 8.0:
            continue
        z = rng.uniform(0.05, 0.95, size=3)
        o = len(pos)
        pos += [[px / (W / 2) - 1.0, 1.0 - py / (H / 2), zz] for (px, py), zz in zip(p, z)]
        idx.append([o, o + 1, o + 2])
    pos = np.array(pos, dtype=F)
    prim = PrimitiveDesc(positions=pos, normals=np.tile(np.array([[0, 0, 1]], dtype=F), (len(pos), 1)), indices=np.array(idx, dtype=np.uint32), material=0)
    eye4 = np.eye(4, dtype=F)
    sc = SceneDesc(nodes=[NodeDesc(), NodeDesc(parent=0, primitives=[prim])], materials=[MaterialDesc(double_sided=True)], samplers=[dict(scenes.REPEAT_LINEAR)],
                   lights=[], width=W, height=H, view=eye4, proj=eye4, camera_position=(0, 0, 1))
    fr = helpers.oracle_frame(helpers.build_model(sc), oracle_lib.brdf_lut(8, 8), msaa=4, threads=4)
    keys = fr.keys                                                   # [H, W, 4]
    depth = (keys >> np.uint64(32)).astype(np.uint32).view(np.float32)
    rank = (np.uint64(0xFFFFFFFF) - (keys & np.uint64(0xFFFFFFFF))).astype(np.int64)
    hit = keys != helpers.NO_HIT
    ys, xs = np.mgrid[0:H, 0:W]
    any_inside = np.zeros((H, W, 4), dtype=bool)
    all_outside = np.ones((H, W, 4), dtype=bool)
    worst = 0.0
    for t in range(N):
        v = pos[3 * t:3 * t + 3]
        # the snapped triangle, as the contract snaps it (1/256 px, round to nearest even), and its plane in f64
        q = np.rint((np.stack([(v[:, 0] + F(1)) * F(W / 2), (F(1) - v[:, 1]) * F(H / 2)], axis=1) * F(256)).astype(np.float64)) / 256.0
        plane = np.linalg.solve(np.array([[q[0, 0], q[0, 1], 1.0], [q[1, 0], q[1, 1], 1.0], [q[2, 0], q[2, 1], 1.0]]), v[:, 2].astype(np.float64))
        sgn = np.sign((q[1, 0] - q[0, 0]) * (q[2, 1] - q[0, 1]) - (q[1, 1] - q[0, 1]) * (q[2, 0] - q[0, 0]))
        for k in range(4):
            sx, sy = xs + offs[k, 0], ys + offs[k, 1]
            e = [sgn * ((q[(i + 2) % 3, 0] - q[(i + 1) % 3, 0]) * (sy - q[(i + 1) % 3, 1]) - (q[(i + 2) % 3, 1] - q[(i + 1) % 3, 1]) * (sx - q[(i + 1) % 3, 0])) for i in range(3)]
            any_inside[..., k] |= np.all([ei > 1e-9 for ei in e], axis=0)
            all_outside[..., k] &= np.any([ei < -1e-9 for ei in e], axis=0)
            mine = hit[..., k] & (rank[..., k] == t)
            if mine.any():
                err = np.abs(depth[..., k].astype(np.float64) - (plane[0] * sx + plane[1] * sy + plane[2]))[mine]
                worst = max(worst, float(err.max() / np.spacing(np.float32(1.0))))
    assert hit[any_inside].all() and not hit[all_outside].any()
    assert hit.mean() > 0.5 and worst < 4.0, (float(hit.mean()), worst)       # in steps of f32 at 1.0 (the depths lie in [0.05, 0.95])


@pytest.mark.gpu
def test_the_hip_path_meets_the_same_closed_forms_without_the_oracle():
    """The HIP path against the closed forms directly (no oracle between them): the directional, ranged point and tilted spot cases above,
    read from the f32 parity tap at the pixel on the optical axis; tolerance = north_star's 1e-4 relative."""
    lut = oracle_lib.brdf_lut(32, 32)
    c20, c10, c30 = (math.cos(math.radians(a)) for a in (20, 10, 30))
    cases = (
        (MaterialDesc(base_color_factor=(0.9, 0.9, 0.9, 1.0), metallic_factor=1.0, roughness_factor=0.3),
         {"kind": "directional", "color": (1.0, 0.8, 0.6), "intensity": 3.0, "direction": (0.3, -0.2, -1.0)},
         ((0.9, 0.9, 0.9), 1.0, 0.3, (-0.3, 0.2, 1.0), np.asarray((1.0, 0.8, 0.6)) * 3.0)),
        (MaterialDesc(base_color_factor=(0.6, 0.6, 0.6, 1.0), metallic_factor=0.0, roughness_factor=0.7),
         {"kind": "point", "color": (1, 1, 1), "intensity": 5.0, "position": (0.0, 0.0, 2.0), "range": 4.0},
         ((0.6, 0.6, 0.6), 0.0, 0.7, (0, 0, 1), np.ones(3) * 5.0 * (1.0 - 4.0 / 16.0) ** 2 / 5.0)),
        (MaterialDesc(base_color_factor=(0.6, 0.6, 0.6, 1.0), metallic_factor=0.0, roughness_factor=0.7),
         {"kind": "spot", "color": (1, 1, 1), "intensity": 5.0, "position": (0.0, 0.0, 2.0), "range": 0.0,
          "direction": (math.sin(math.radians(20)), 0.0, -c20), "inner_angle": c10, "outer_angle": c30},
         ((0.6, 0.6, 0.6), 0.0, 0.7, (0, 0, 1), np.ones(3) * 5.0 * 0.25 * ((c20 - c30) / (c10 - c30)) ** 2)),
    )
    for mat, light, (base, metallic, rough, l, radiance) in cases:
        sc = _quad_scene(mat, [light])
        dev, _ = helpers.hip_frame(helpers.build_model(sc), lut)
        c = sc.width // 2
        got = dev.read_opaque_f32()[c, c, :3].astype(np.float64)
        want = _cook_torrance(base, metallic, rough, (0, 0, 1), (0, 0, 1), np.asarray(l, dtype=np.float64), radiance)
        assert np.allclose(got, want, rtol=2e-4, atol=2e-6), (light["kind"], got, want)


def test_brdf_lut_against_a_quadrature_of_the_split_sum_integrals():
    """crates/renderer-core/src/brdf_lut/shader.wgsl:46-78 estimates A = E[(1 - Fc) G_vis], B = E[Fc G_vis] over GGX-importance-sampled half
    vectors with 1,024 Hammersley points.  Here the same expectation as a midpoint rule over the unit square of the sampling variables
    (no Hammersley sequence, f64, no f16 rounding): the two must agree to the estimator's error.  Texel row j holds v = 1 - (j + 0.5) / H
    (the full-screen triangle's uv, shader.wgsl:4-12), column i holds n.v = (i + 0.5) / W."""
    W = H = 32
    lut = oracle_lib.brdf_lut(W, H).view(np.float16).astype(np.float64)
    n = 384
    x1 = (np.arange(n) + 0.5) / n
    X, Y = np.meshgrid(x1, x1, indexing="ij")                      # xi.x -> phi, xi.y -> theta
    worst = 0.0
    for (i, j) in ((0, 0), (W - 1, 0), (0, H - 1), (W - 1, H - 1), (W // 2, H // 2), (5, 20), (27, 9), (16, 2), (3, 29)):
        nov = min(max((i + 0.5) / W, 1e-3), 1 - 1e-3)
        rough = min(max(1.0 - (j + 0.5) / H, 1e-3), 1 - 1e-3)
        alpha = rough * rough
        v = np.array([math.sqrt(max(0.0, 1 - nov * nov)), 0.0, nov])
        cos_t = np.sqrt((1 - Y) / (1 + (alpha * alpha - 1) * Y))
        sin_t = np.sqrt(np.maximum(0.0, 1 - cos_t * cos_t))
        phi = 2 * math.pi * X
        h = np.stack([np.cos(phi) * sin_t, np.sin(phi) * sin_t, cos_t], axis=-1)
        voh_signed = h @ v
        l = 2 * voh_signed[..., None] * h - v
        l = l / np.linalg.norm(l, axis=-1, keepdims=True)
        nol, noh, voh = np.maximum(l[..., 2], 0), np.maximum(h[..., 2], 0), np.maximum(voh_signed, 0)
        a_ = max(alpha, 0.001)
        k = (a_ + 1) ** 2 / 8
        g = (nov / (nov * (1 - k) + k)) * (nol / (nol * (1 - k) + k))
        g_vis = np.where(nol > 0, g * voh / np.maximum(noh * nov, 1e-4), 0.0)
        fc = (1 - voh) ** 5
        A, B = float(((1 - fc) * g_vis).mean()), float((fc * g_vis).mean())
        worst = max(worst, abs(lut[j, i, 0] - A), abs(lut[j, i, 1] - B))
        assert abs(lut[j, i, 0] - A) < 0.02 and abs(lut[j, i, 1] - B) < 0.02, ((i, j), lut[j, i], (A, B))
    assert worst < 0.02


def test_bilinear_repeat_sampling_of_a_known_texture():
    """texture_uvs.wgsl:144-187 at level 0 with a linear / repeat sampler: a 4x4 texture whose red channel is the column index and green the
    row index, on a quad that maps uv = position.  At the pixel on the optical axis uv = (0.5, 0.5): x = u W - 0.5 = 1.5 -> texels 1 and 2
    averaged -> base colour r = 1.5 / 255 * 17 (texel value = 17 * index), and the same for g.  Unlit material: the pixel is the texel."""
    tex = np.zeros((1, 4, 4, 4), np.uint8)
    for yy in range(4):
        for xx in range(4):
            tex[0, yy, xx] = (17 * xx, 17 * yy, 0, 255)
    mat = MaterialDesc(kind="unlit", base_color_factor=(1.0, 1.0, 1.0, 1.0))
    sc = _quad_scene(mat, [])
    sc.textures, sc.samplers = [tex[0]], [dict(scenes.REPEAT_LINEAR)]
    from awsm_renderer_amd.scene_desc import TextureRef
    mat.base_color_tex = TextureRef(texture=0, sampler=0, uv_index=0)
    prim = [p for nd in sc.nodes for p in nd.primitives][0]
    prim.uvs = [((np.asarray(prim.positions)[:, :2] + 1.0) * 0.5).astype(F)]       # the quad spans [-1, 1]^2
    fr = helpers.oracle_frame(helpers.build_model(sc), oracle_lib.brdf_lut(16, 16))
    c = sc.width // 2
    got = fr.rgba32f[c, c, :3].astype(np.float64)
    assert np.allclose(got[:2], [1.5 * 17 / 255, 1.5 * 17 / 255], atol=2e-4), got
