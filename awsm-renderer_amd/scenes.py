"""
scenes.py — seeded synthetic scenes for BASELINE.json's configs.

No glTF assets exist in this environment or in the reference (its demo streams Khronos samples over HTTP), so
every config is generated procedurally with the same statistics the named assets have:

  C1  box_scene()           Khronos "Box" topology: 24 verts, 12 tris, root rotation -90deg about X -> child mesh node
  C2  helmet_scene()        "DamagedHelmet-class": ~15k-tri displaced sphere, 1 PBR material, 5 textures, TANGENT supplied
  C3  skinned_morph_scene() "BrainStem-class" 61k-tri skinned tube (18 joints) + 12-tri cube with 2 morph targets
  C4  atrium_scene()        "Sponza-class": 262,144 tris in 103 primitives, 25 materials, 69 textures in 3 size classes

Texture colour data is what the reference's pool would hold after its load-time sRGB->linear + requantise-to-RGBA8
step (crates/renderer-core/src/texture/convert_srgb.rs:52-76): plain linear RGBA8.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import numpy as np

from .scene_desc import MaterialDesc, NodeDesc, PrimitiveDesc, SceneDesc, SkinDesc, TextureRef

F = np.float32

# crates/frontend/src/pages/app/scene.rs:656-677 — the demo's default punctual lights
DEFAULT_LIGHTS = [
    {"kind": "directional", "color": (1.0, 0.97, 0.92), "intensity": 1.4, "direction": (0.1, -0.35, -1.0)},
    {"kind": "directional", "color": (0.9, 0.95, 1.0), "intensity": 0.6, "direction": (0.0, -0.2, -1.0)},
    {"kind": "directional", "color": (0.8, 0.9, 1.0), "intensity": 0.7, "direction": (-0.05, -0.25, 1.0)},
    {"kind": "directional", "color": (1.0, 0.96, 0.9), "intensity": 0.5, "direction": (-1.0, -0.2, 0.2)},
]
REPEAT_LINEAR = {"address_mode_u": 1, "address_mode_v": 1, "mag_filter": 1, "min_filter": 1, "mipmap_filter": 1, "max_anisotropy": 16}
CLAMP_LINEAR = {"address_mode_u": 0, "address_mode_v": 0, "mag_filter": 1, "min_filter": 1, "mipmap_filter": 1, "max_anisotropy": 16}
MIRROR_NEAREST = {"address_mode_u": 2, "address_mode_v": 2, "mag_filter": 0, "min_filter": 0, "mipmap_filter": 0, "max_anisotropy": 1}


# ------------------------------------------------------------------------------------------------ camera helpers (f32)

def perspective_rh(fov_y, aspect, z_near, z_far):
    """glam Mat4::perspective_rh (0..1 depth), [col][row]."""
    half = F(0.5) * F(fov_y)
    h = F(math.cos(float(half))) / F(math.sin(float(half)))
    w = h / F(aspect)
    r = F(z_far) / (F(z_near) - F(z_far))
    m = np.zeros((4, 4), dtype=F)
    m[0][0], m[1][1], m[2][2], m[2][3], m[3][2] = w, h, r, -1.0, r * F(z_near)
    return m


def orthographic_rh(left, right, bottom, top, near, far):
    rw, rh, r = F(1.0) / (F(right) - F(left)), F(1.0) / (F(top) - F(bottom)), F(1.0) / (F(near) - F(far))
    m = np.zeros((4, 4), dtype=F)
    m[0][0], m[1][1], m[2][2] = rw + rw, rh + rh, r
    m[3] = np.array([-(F(left) + F(right)) * rw, -(F(top) + F(bottom)) * rh, r * F(near), 1.0], dtype=F)
    return m


def look_at_rh(eye, center, up=(0, 1, 0)):
    eye, center, up = (np.asarray(v, dtype=F) for v in (eye, center, up))
    f = center - eye
    f = (f / np.sqrt((f * f).sum(dtype=F))).astype(F)
    s = np.cross(f, up).astype(F)
    s = (s / np.sqrt((s * s).sum(dtype=F))).astype(F)
    u = np.cross(s, f).astype(F)
    m = np.zeros((4, 4), dtype=F)
    m[0] = [s[0], u[0], -f[0], 0]
    m[1] = [s[1], u[1], -f[1], 0]
    m[2] = [s[2], u[2], -f[2], 0]
    m[3] = [-np.dot(eye, s), -np.dot(eye, u), np.dot(eye, f), 1]
    return m.astype(F)


def quat_axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    s = math.sin(angle / 2)
    return (float(axis[0] * s), float(axis[1] * s), float(axis[2] * s), float(math.cos(angle / 2)))


# ------------------------------------------------------------------------------------------------ textures

def value_noise_rgba8(rng, size, cells=16, base=(0.5, 0.5, 0.5), amp=(0.4, 0.4, 0.4), alpha=255, kind="color"):
    """Cheap band-limited noise: a coarse random lattice bilinearly upsampled (periodic), plus a finer octave."""
    def octave(c):
        g = rng.random((c, c, 3), dtype=np.float32)
        gp = np.concatenate([g, g[:1]], axis=0)
        gp = np.concatenate([gp, gp[:, :1]], axis=1)
        t = (np.arange(size, dtype=np.float32) + 0.5) * (c / size)
        i0 = np.floor(t).astype(np.int32)
        f = (t - i0)[:, None]
        rows = gp[i0] * (1 - f[:, :, None]) + gp[i0 + 1] * f[:, :, None]          # (size, c+1, 3)
        cols = rows[:, i0] * (1 - f[None, :, :]) + rows[:, i0 + 1] * f[None, :, :]  # (size, size, 3)
        return cols
    n = 0.7 * octave(cells) + 0.3 * octave(cells * 4)
    if kind == "normal":
        dx, dy = (n[..., 0] - 0.5) * 0.6, (n[..., 1] - 0.5) * 0.6
        nz = np.sqrt(np.clip(1.0 - dx * dx - dy * dy, 0.0, 1.0))
        rgb = np.stack([dx * 0.5 + 0.5, dy * 0.5 + 0.5, nz * 0.5 + 0.5], axis=-1)
    else:
        rgb = np.asarray(base, dtype=np.float32) + (n - 0.5) * 2.0 * np.asarray(amp, dtype=np.float32)
    out = np.empty((size, size, 4), dtype=np.uint8)
    out[..., :3] = np.clip(np.rint(rgb * 255.0), 0, 255).astype(np.uint8)
    out[..., 3] = alpha
    return out


# ------------------------------------------------------------------------------------------------ geometry helpers

def grid_patch(fn, nu, nv, uv_scale=(1.0, 1.0), wrap_u=False):
    """Tessellate p = fn(u, v), u,v in [0,1], into nu x nv quads (2 tris each, CCW for n = dP/du x dP/dv).
    Returns positions, normals, tangents(w=+1), uvs, indices.  Derivatives are central differences in f64."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    U, V = np.meshgrid(u, v, indexing="xy")            # (nv+1, nu+1)
    P = fn(U, V)
    h = 1e-4
    dPu = (fn(U + h, V) - fn(U - h, V)) / (2 * h)
    dPv = (fn(U, V + h) - fn(U, V - h)) / (2 * h)
    N = np.cross(dPu, dPv)
    nl = np.linalg.norm(N, axis=-1, keepdims=True)
    N = np.where(nl > 1e-12, N / np.maximum(nl, 1e-12), np.array([0.0, 1.0, 0.0]))
    T = dPu - N * (dPu * N).sum(-1, keepdims=True)
    tl = np.linalg.norm(T, axis=-1, keepdims=True)
    T = np.where(tl > 1e-12, T / np.maximum(tl, 1e-12), np.array([1.0, 0.0, 0.0]))
    pos = P.reshape(-1, 3).astype(F)
    nrm = N.reshape(-1, 3).astype(F)
    tan = np.concatenate([T.reshape(-1, 3), np.ones((pos.shape[0], 1))], axis=1).astype(F)
    uvs = np.stack([U * uv_scale[0], V * uv_scale[1]], axis=-1).reshape(-1, 2).astype(F)
    i = np.arange(nu)[None, :] + (nu + 1) * np.arange(nv)[:, None]
    a, b, c, d = i, i + 1, i + nu + 1, i + nu + 2
    idx = np.stack([a, b, d, a, d, c], axis=-1).reshape(-1, 3).astype(np.uint32)
    return pos, nrm, tan, uvs, idx


def _prim(patch, material, **kw):
    pos, nrm, tan, uvs, idx = patch
    return PrimitiveDesc(positions=pos, normals=nrm, tangents=tan, uvs=[uvs], indices=idx, material=material, **kw)


# ------------------------------------------------------------------------------------------------ C1: Box

def box_scene(width=256, height=256) -> SceneDesc:
    """Khronos Box: 24 vertices (pos + normal), 36 indices, node 0 = rotation -90deg about X, child node 1 = mesh;
    material baseColorFactor (0.8, 0, 0, 1), metallicFactor 0."""
    faces = [((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (-1, 0, 0), (0, 1, 0)), ((1, 0, 0), (0, 0, -1), (0, 1, 0)),
             ((-1, 0, 0), (0, 0, 1), (0, 1, 0)), ((0, 1, 0), (1, 0, 0), (0, 0, -1)), ((0, -1, 0), (1, 0, 0), (0, 0, 1))]
    pos, nrm, idx = [], [], []
    for f, (n, a, b) in enumerate(faces):
        n, a, b = (np.array(v, dtype=np.float64) for v in (n, a, b))
        for sa, sb in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
            pos.append(0.5 * (n + sa * a + sb * b))
            nrm.append(n)
        o = 4 * f
        idx += [[o, o + 1, o + 2], [o, o + 2, o + 3]]
    prim = PrimitiveDesc(positions=np.array(pos, dtype=F), normals=np.array(nrm, dtype=F), indices=np.array(idx, dtype=np.uint32), material=0)
    nodes = [NodeDesc(rotation=quat_axis_angle((1, 0, 0), -math.pi / 2)), NodeDesc(parent=0, primitives=[prim])]
    mats = [MaterialDesc(base_color_factor=(0.8, 0.0, 0.0, 1.0), metallic_factor=0.0)]
    eye = (1.6, 1.2, 2.2)
    return SceneDesc(nodes=nodes, materials=mats, samplers=[dict(REPEAT_LINEAR)], lights=list(DEFAULT_LIGHTS), width=width, height=height,
                     view=look_at_rh(eye, (0, 0, 0)), proj=perspective_rh(math.radians(45), width / height, 0.1, 100.0), camera_position=eye)


# ------------------------------------------------------------------------------------------------ C2: helmet-class

def helmet_scene(width=1920, height=1080, segments=96, rings=80, tex_size=2048, seed=0xA35A0002, material_overrides: Optional[dict] = None) -> SceneDesc:
    rng = np.random.default_rng(seed)
    k = rng.uniform(1.5, 4.0, size=(6, 3))
    ph = rng.uniform(0, 2 * math.pi, size=6)

    def fn(U, V):
        theta, phi = U * 2 * math.pi, (0.02 + 0.96 * V) * math.pi   # open poles: no degenerate triangles
        d = np.stack([np.sin(phi) * np.cos(theta), np.cos(phi), np.sin(phi) * np.sin(theta)], axis=-1)
        r = 1.0
        for i in range(6):
            r = r + 0.05 * np.sin(d @ k[i] * 2.0 + ph[i])
        return d * r[..., None]

    patch = grid_patch(lambda U, V: fn(1.0 - U, V), segments, rings, uv_scale=(2.0, 1.0))
    textures = [
        value_noise_rgba8(rng, tex_size, 24, base=(0.55, 0.5, 0.45), amp=(0.35, 0.3, 0.3)),                 # base colour
        value_noise_rgba8(rng, tex_size, 32, base=(0.5, 0.55, 0.5), amp=(0.0, 0.35, 0.5)),                  # G rough, B metal
        value_noise_rgba8(rng, tex_size, 48, kind="normal"),
        value_noise_rgba8(rng, tex_size, 12, base=(0.8, 0.8, 0.8), amp=(0.2, 0.2, 0.2)),                    # occlusion (R)
        value_noise_rgba8(rng, tex_size, 8, base=(0.08, 0.05, 0.02), amp=(0.08, 0.05, 0.02)),               # emissive
    ]
    mat = MaterialDesc(base_color_tex=TextureRef(0), metallic_roughness_tex=TextureRef(1), normal_tex=TextureRef(2),
                       occlusion_tex=TextureRef(3), emissive_tex=TextureRef(4), emissive_factor=(1.0, 1.0, 1.0))
    if material_overrides:
        for kk, vv in material_overrides.items():
            setattr(mat, kk, vv)
    nodes = [NodeDesc(rotation=quat_axis_angle((0, 1, 0), 0.6), primitives=[_prim(patch, 0)])]
    eye = (0.0, 0.6, 2.5 * 1.2)
    return SceneDesc(nodes=nodes, materials=[mat], textures=textures, samplers=[dict(REPEAT_LINEAR)], lights=list(DEFAULT_LIGHTS), width=width,
                     height=height, view=look_at_rh(eye, (0, 0, 0)), proj=perspective_rh(math.radians(45), width / height, 0.1, 100.0),
                     camera_position=eye)


# ------------------------------------------------------------------------------------------------ C3: skinned + morph

def skinned_morph_scene(width=1920, height=1080, around=64, along=480, joints=18, seed=0xA35A0003, tex_size=512) -> SceneDesc:
    rng = np.random.default_rng(seed)
    length = 4.0

    def tube(U, V):
        theta = U * 2 * math.pi
        r = 0.35 + 0.05 * np.sin(V * 14.0)
        return np.stack([r * np.cos(theta), V * length - length / 2, -r * np.sin(theta)], axis=-1)

    pos, nrm, tan, uvs, idx = grid_patch(tube, around, along, uv_scale=(2.0, 8.0))
    V = pos.shape[0]
    # joints along the tube axis; each vertex bound to its 4 nearest joints with Dirichlet(4) weights
    jy = np.linspace(-length / 2, length / 2, joints)
    d = np.abs(pos[:, 1:2] - jy[None, :])
    near = np.argsort(d, axis=1)[:, :4].astype(np.uint32)
    w = rng.dirichlet(np.ones(4), size=V).astype(F)
    w = (-np.sort(-w, axis=1)).astype(F)
    tube_prim = PrimitiveDesc(positions=pos, normals=nrm, tangents=tan, uvs=[uvs], indices=idx, material=0, joints=[near], weights=[w])

    nodes: List[NodeDesc] = [NodeDesc()]                    # 0: rig root
    joint_nodes = []
    t_anim = 0.37                                           # fixed animation time
    for j in range(joints):                                 # chain: each joint is the child of the previous
        parent = 0 if j == 0 else joint_nodes[-1]
        trans = (0.0, float(jy[0]), 0.0) if j == 0 else (0.0, float(jy[1] - jy[0]), 0.0)
        ang = 0.12 * math.sin(t_anim * 3.0 + 0.5 * j)
        nodes.append(NodeDesc(translation=trans, rotation=quat_axis_angle((0, 0, 1), ang), parent=parent))
        joint_nodes.append(len(nodes) - 1)
    inv_bind = np.zeros((joints, 4, 4), dtype=F)
    for j in range(joints):
        m = np.eye(4, dtype=F)
        m[3][1] = -jy[j]                                    # inverse of the bind-pose translation (column 3)
        inv_bind[j] = m
    nodes.append(NodeDesc(parent=0, primitives=[tube_prim], skin=0))

    # morph cube: 24 verts, 12 tris, 2 targets (positions + normals)
    box = box_scene().nodes[1].primitives[0]
    cp, cn = box.positions.copy(), box.normals.copy()
    t0p = (cp * np.array([0.6, 0.0, 0.0], dtype=F)).astype(F)
    t1p = (cn * F(0.35)).astype(F)
    t0n = np.zeros_like(cn)
    t1n = (rng.normal(size=cn.shape) * 0.1).astype(F)
    cube_uv = np.tile(np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=F), (6, 1))
    cube = PrimitiveDesc(positions=cp, normals=cn, uvs=[cube_uv], indices=box.indices.copy(), material=1,
                         morph_targets=[{"positions": t0p, "normals": t0n}, {"positions": t1p, "normals": t1n}],
                         morph_weights=np.zeros(2, dtype=F), animated_morph_weights=np.array([0.3, 0.7], dtype=F))
    nodes.append(NodeDesc(translation=(1.6, 0.0, 0.0), rotation=quat_axis_angle((0.3, 1, 0.1), 0.7), scale=(0.8, 0.8, 0.8), primitives=[cube]))

    textures = [value_noise_rgba8(rng, tex_size, 16, base=(0.6, 0.45, 0.4), amp=(0.3, 0.25, 0.2)),
                value_noise_rgba8(rng, tex_size, 24, kind="normal"),
                value_noise_rgba8(rng, tex_size, 8, base=(0.3, 0.5, 0.7), amp=(0.25, 0.25, 0.25))]
    mats = [MaterialDesc(base_color_tex=TextureRef(0), normal_tex=TextureRef(1), metallic_factor=0.1, roughness_factor=0.6),
            MaterialDesc(base_color_tex=TextureRef(2), metallic_factor=0.0, roughness_factor=0.4, double_sided=True)]
    eye = (0.8, 0.4, 5.2)
    return SceneDesc(nodes=nodes, materials=mats, textures=textures, samplers=[dict(REPEAT_LINEAR)], skins=[SkinDesc(joints=joint_nodes, inverse_bind=inv_bind)],
                     lights=list(DEFAULT_LIGHTS), width=width, height=height, view=look_at_rh(eye, (0.4, 0, 0)),
                     proj=perspective_rh(math.radians(45), width / height, 0.1, 100.0), camera_position=eye)


# ------------------------------------------------------------------------------------------------ C4: Sponza-class atrium

def atrium_scene(width=3840, height=2160, detail=1.0, tex_scale=1.0, seed=0xA35A0004) -> SceneDesc:
    """262,144 triangles in 103 primitives at detail=1 (detail scales the tessellation, for small test cases);
    25 materials; 69 textures in three size classes (1024^2 x23, 2048^2 x12, 512^2 x34 at tex_scale=1)."""
    rng = np.random.default_rng(seed)
    q = lambda n: max(2, int(round(n * detail)))   # noqa: E731
    L, Wd, Hh = 36.0, 12.0, 10.0                   # long axis = z
    prims: List[Tuple[tuple, int, bool]] = []      # (patch, material, double_sided marker via material)

    def plane(origin, du, dv, nu, nv, uv=(1, 1)):
        o, a, b = (np.array(v, dtype=np.float64) for v in (origin, du, dv))
        return grid_patch(lambda U, V: o + U[..., None] * a + V[..., None] * b, nu, nv, uv_scale=uv)

    n_mat = 25
    m_floor, m_ceil, m_wall = 0, 1, 2
    # floor (normal +y): dP/du x dP/dv = x cross -z = +y
    prims.append((plane((-Wd / 2, 0, L / 2), (Wd, 0, 0), (0, 0, -L), q(64), q(64), uv=(6, 18)), m_floor))
    # ceiling, two halves (normal -y)
    prims.append((plane((-Wd / 2, Hh, -L / 2), (Wd, 0, 0), (0, 0, L / 2), q(32), q(64), uv=(6, 9)), m_ceil))
    prims.append((plane((-Wd / 2, Hh, 0), (Wd, 0, 0), (0, 0, L / 2), q(32), q(64), uv=(6, 9)), m_ceil))
    # four walls facing inward (normal = dP/du x dP/dv)
    prims.append((plane((-Wd / 2, 0, L / 2), (0, 0, -L), (0, Hh, 0), q(64), q(32), uv=(18, 5)), m_wall))      # x = -W/2, normal +x
    prims.append((plane((Wd / 2, 0, -L / 2), (0, 0, L), (0, Hh, 0), q(64), q(32), uv=(18, 5)), m_wall + 1))   # x = +W/2, normal -x
    prims.append((plane((-Wd / 2, 0, -L / 2), (Wd, 0, 0), (0, Hh, 0), q(64), q(32), uv=(6, 5)), m_wall + 2))  # z = -L/2, normal +z
    prims.append((plane((Wd / 2, 0, L / 2), (-Wd, 0, 0), (0, Hh, 0), q(64), q(32), uv=(6, 5)), m_wall + 3))   # z = +L/2, normal -z
    # 32 columns: two rows of 16
    for i in range(32):
        side = -1 if i % 2 == 0 else 1
        cz = -L / 2 + 1.5 + (i // 2) * ((L - 3.0) / 15)
        cx, rad = side * 3.2, 0.42

        def col(U, V, cx=cx, cz=cz, rad=rad):
            th = U * 2 * math.pi
            r = rad * (1.0 + 0.08 * np.cos(12 * th)) * (1.0 + 0.25 * (np.exp(-30 * V) + np.exp(-30 * (1 - V))))
            return np.stack([cx + r * np.cos(th), V * 7.0, cz - r * np.sin(th)], axis=-1)
        prims.append((grid_patch(col, q(32), q(32), uv_scale=(2, 6)), 6 + (i % 4)))
    # 32 arches between columns and wall (half tori)
    for i in range(32):
        side = -1 if i % 2 == 0 else 1
        cz = -L / 2 + 1.5 + (i // 2) * ((L - 3.0) / 15)
        x0, x1 = side * 3.2, side * (Wd / 2)

        def arch(U, V, x0=x0, x1=x1, cz=cz, side=side):
            ang = V * math.pi
            xc, R = (x0 + x1) / 2, abs(x1 - x0) / 2
            th = U * 2 * math.pi
            rr = 0.28
            cx_ = xc - side * R * np.cos(ang)
            cy_ = 7.0 + R * np.sin(ang)
            # tube around the arch centre line; local frame (radial in the arch plane, z)
            rx, ry = -side * np.cos(ang), np.sin(ang)
            return np.stack([cx_ + rr * np.cos(th) * rx, cy_ + rr * np.cos(th) * ry, cz + side * rr * np.sin(th) * np.ones_like(ang)], axis=-1)
        prims.append((grid_patch(arch, q(32), q(32), uv_scale=(1, 4)), 10 + (i % 4)))
    # 16 drapes (double sided, wavy)
    for i in range(16):
        side = -1 if i % 2 == 0 else 1
        cz = -L / 2 + 2.6 + (i // 2) * ((L - 5.2) / 7)
        ph = rng.uniform(0, 6.28)

        def drape(U, V, side=side, cz=cz, ph=ph):
            x = side * (1.1 + 0.9 * U)
            z = cz + 0.25 * np.sin(10 * U + ph) * (0.2 + V) + 0.1 * np.sin(23 * U)
            return np.stack([x * np.ones_like(V), 8.6 - 4.2 * V, z], axis=-1)
        prims.append((grid_patch(drape, q(32), q(64), uv_scale=(1, 2)), 14 + (i % 5)))
    # 16 props (bumpy spheres / vases) along the centre line
    for i in range(16):
        cz = -L / 2 + 2.0 + i * ((L - 4.0) / 15)
        cx = 1.4 * math.sin(i * 1.7)
        kk = rng.uniform(2.0, 5.0, size=3)

        def prop(U, V, cx=cx, cz=cz, kk=kk):
            th, phi = U * 2 * math.pi, (0.03 + 0.94 * V) * math.pi
            d = np.stack([np.sin(phi) * np.cos(th), np.cos(phi), -np.sin(phi) * np.sin(th)], axis=-1)
            r = 0.55 * (1.0 + 0.12 * np.sin(d @ kk * 3.0))
            return d * r[..., None] + np.array([cx, 0.8, cz])
        prims.append((grid_patch(prop, q(32), q(32), uv_scale=(2, 1)), 19 + (i % 6)))
    assert len(prims) == 103

    # ---- 69 textures: 25 base colour, 25 normal, 12 metallic-roughness, 5 occlusion, 2 emissive ----
    s_big, s_mid, s_small = (max(8, int(v * tex_scale)) for v in (2048, 1024, 512))
    sizes = [s_big] * 12 + [s_mid] * 23 + [s_small] * 34
    order = rng.permutation(69)
    sizes = [sizes[j] for j in order]
    textures, tex_id = [], 0
    base_tex, nrm_tex, mr_tex, occ_tex, em_tex = {}, {}, {}, {}, {}
    for m in range(n_mat):
        hue = rng.uniform(0.25, 0.75, size=3)
        textures.append(value_noise_rgba8(rng, sizes[tex_id], 8 + 4 * (m % 5), base=tuple(hue), amp=(0.25, 0.25, 0.25))); base_tex[m] = tex_id; tex_id += 1
    for m in range(n_mat):
        textures.append(value_noise_rgba8(rng, sizes[tex_id], 16 + 8 * (m % 4), kind="normal")); nrm_tex[m] = tex_id; tex_id += 1
    for m in range(12):
        textures.append(value_noise_rgba8(rng, sizes[tex_id], 16, base=(0.5, 0.6, 0.3), amp=(0.0, 0.3, 0.3))); mr_tex[2 * m] = tex_id; tex_id += 1
    for m in range(5):
        textures.append(value_noise_rgba8(rng, sizes[tex_id], 6, base=(0.8, 0.8, 0.8), amp=(0.2, 0.2, 0.2))); occ_tex[5 * m] = tex_id; tex_id += 1
    for m in range(2):
        textures.append(value_noise_rgba8(rng, sizes[tex_id], 5, base=(0.1, 0.07, 0.03), amp=(0.1, 0.07, 0.03))); em_tex[19 + 3 * m] = tex_id; tex_id += 1
    assert tex_id == 69
    mats = []
    for m in range(n_mat):
        md = MaterialDesc(base_color_tex=TextureRef(base_tex[m]), normal_tex=TextureRef(nrm_tex[m]),
                          metallic_factor=1.0 if m in mr_tex else float(rng.uniform(0.0, 0.3)), roughness_factor=1.0 if m in mr_tex else float(rng.uniform(0.35, 0.9)),
                          double_sided=(14 <= m <= 18))
        if m in mr_tex:
            md.metallic_roughness_tex = TextureRef(mr_tex[m])
        if m in occ_tex:
            md.occlusion_tex = TextureRef(occ_tex[m])
        if m in em_tex:
            md.emissive_tex = TextureRef(em_tex[m]); md.emissive_factor = (1.0, 1.0, 1.0)
        mats.append(md)

    nodes = [NodeDesc()]                                              # scene root
    for patch, material in prims:
        nodes.append(NodeDesc(parent=0, primitives=[_prim(patch, material)]))
    eye = (0.4, 3.1, L / 2 - 1.0)
    target = (-0.2, 3.4, -L / 2)
    lights = list(DEFAULT_LIGHTS) + [{"kind": "point", "color": (1.0, 0.8, 0.6), "intensity": 30.0, "position": (0.0, 6.0, 0.0), "range": 30.0}]
    return SceneDesc(nodes=nodes, materials=mats, textures=textures, samplers=[dict(REPEAT_LINEAR)], lights=lights, width=width, height=height,
                     view=look_at_rh(eye, target), proj=perspective_rh(math.radians(60), width / height, 0.1, 100.0), camera_position=eye)


def cube_face_directions(n: int) -> np.ndarray:
    """(6, n, n, 3) unit directions through the texel centres of a cube's faces, layer order +X -X +Y -Y +Z -Z, with the (sc, tc)
    parametrisation WebGPU / Vulkan / D3D share (s, t = (i + 0.5) / n; row 0 is the top of the image)."""
    c = (np.arange(n, dtype=np.float64) + 0.5) / n * 2.0 - 1.0
    sc, tc = np.meshgrid(c, c)                      # sc varies along x (columns), tc along y (rows)
    one = np.ones_like(sc)
    faces = [(one, -tc, -sc), (-one, -tc, sc), (sc, one, tc), (sc, -one, -tc), (sc, -tc, one), (-sc, -tc, -one)]
    d = np.stack([np.stack(f, axis=-1) for f in faces])
    return d / np.linalg.norm(d, axis=-1, keepdims=True)


def procedural_environment(size: int = 64, irradiance_size: int = 16, seed: int = 0xA35A0008) -> dict:
    """A non-uniform HDR environment for the texel-cubemap path: sky gradient, a sun lobe far above 1.0, coloured bands and a darker
    ground.  Not a physically prefiltered set — the renderer only samples these cubes; what is in them does not matter for parity:
    skybox one level, "prefiltered" a full chain whose lobes widen per level, irradiance a small smooth cube."""
    rng = np.random.default_rng(seed)
    sun = np.array([0.35, 0.75, -0.55]); sun /= np.linalg.norm(sun)
    tint = rng.uniform(0.6, 1.0, size=(3, 3))

    def radiance(d, sharp):
        y = d[..., 1:2]
        up = np.clip(y * 0.5 + 0.5, 0.0, 1.0)
        sky = (1.0 - up) * np.array([0.9, 0.75, 0.6]) + up * np.array([0.15, 0.35, 0.9])
        ground = np.array([0.18, 0.14, 0.10]) * (1.0 + 0.5 * np.sin(6.0 * d[..., 0:1]) * np.cos(5.0 * d[..., 2:3]))
        base = np.where(y > 0.0, sky, ground)
        bands = 0.25 * (np.sin(9.0 * d[..., 0:1] + 1.0) * tint[0] + np.sin(7.0 * d[..., 2:3] + 2.0) * tint[1] + np.sin(5.0 * d[..., 1:2]) * tint[2]) / max(1.0, 16.0 / sharp)
        lobe = np.exp(-sharp * (1.0 - np.clip((d * sun).sum(axis=-1, keepdims=True), -1.0, 1.0)))
        rgb = np.clip(base + bands, 0.0, None) + lobe * np.array([30.0, 26.0, 20.0]) * min(1.0, sharp / 64.0)
        return np.concatenate([rgb, np.ones_like(y)], axis=-1).astype(np.float16)

    levels = int(math.log2(size)) + 1
    return {"skybox": [radiance(cube_face_directions(size), 256.0)],
            "prefiltered": [radiance(cube_face_directions(max(size >> l, 1)), 256.0 / (4.0 ** l)) for l in range(levels)],
            "irradiance": [radiance(cube_face_directions(irradiance_size), 1.5)]}


def total_triangles(scene: SceneDesc) -> int:
    return int(sum(np.asarray(p.indices).reshape(-1, 3).shape[0] for n in scene.nodes for p in n.primitives))


# ------------------------------------------------------------------------------------------------ material zoo (parity coverage)

def material_zoo_scene(width=640, height=360, tex_size=64, seed=0xA35A0005) -> SceneDesc:
    """A grid of bumpy spheres, one per shading feature of the opaque pass: every optional PBR block
    (vertex colour, emissive strength, ior, specular, volume+transmission texture factor 0, clearcoat, sheen), unlit,
    debug views, a second UV set, texture transforms, clamp / mirror / nearest samplers, non-power-of-two textures,
    point + spot lights, double-sided and single-sided materials.  Not a BASELINE config; it exists so that the parity
    tests reach every branch of compute.wgsl / material_color_calc.wgsl / brdf.wgsl."""
    rng = np.random.default_rng(seed)
    npot = max(12, (tex_size * 3) // 4 + 1)            # a non-power-of-two size -> second pool array, generic wrap path
    textures = [
        value_noise_rgba8(rng, tex_size, 8, base=(0.6, 0.5, 0.4), amp=(0.3, 0.3, 0.3)),      # 0 colour
        value_noise_rgba8(rng, tex_size, 12, kind="normal"),                                 # 1 normal
        value_noise_rgba8(rng, tex_size, 10, base=(0.5, 0.5, 0.5), amp=(0.4, 0.4, 0.4)),     # 2 generic data (all channels vary)
        value_noise_rgba8(rng, npot, 6, base=(0.5, 0.6, 0.5), amp=(0.3, 0.3, 0.3)),          # 3 non-pow2 colour
        value_noise_rgba8(rng, npot, 9, kind="normal"),                                      # 4 non-pow2 normal
    ]
    textures[2][..., 3] = rng.integers(60, 255, size=textures[2].shape[:2], dtype=np.uint8)  # alpha channel carries data too
    samplers = [dict(REPEAT_LINEAR), dict(CLAMP_LINEAR), dict(MIRROR_NEAREST),
                {"address_mode_u": 2, "address_mode_v": 0, "mag_filter": 1, "min_filter": 1, "mipmap_filter": 1, "max_anisotropy": 1}]
    T = TextureRef
    xf = {"offset": (0.13, -0.21), "origin": (0.5, 0.5), "rotation": 0.4, "scale": (1.7, 0.6)}
    mats = [
        MaterialDesc(base_color_tex=T(0), normal_tex=T(1), metallic_factor=0.0, roughness_factor=0.5),                                      # 0 plain
        MaterialDesc(base_color_tex=T(0), metallic_roughness_tex=T(2), normal_tex=T(1), occlusion_tex=T(2), emissive_tex=T(0), emissive_factor=(0.5, 0.4, 0.3),
                     occlusion_strength=0.7, normal_scale=1.4),                                                                             # 1 all five core textures
        MaterialDesc(base_color_factor=(0.9, 0.8, 0.7, 1.0), vertex_color_set=0, metallic_factor=0.2, roughness_factor=0.6),                # 2 vertex colour
        MaterialDesc(base_color_tex=T(3, sampler=1), normal_tex=T(4, sampler=3), metallic_factor=0.9, roughness_factor=0.3),                # 3 non-pow2, clamp, mirror/clamp mix
        MaterialDesc(base_color_tex=T(0, sampler=2, transform=xf), emissive_tex=T(3, sampler=2), emissive_factor=(0.3, 0.3, 0.3), emissive_strength=2.5,
                     roughness_factor=0.8, metallic_factor=0.0),                                                                             # 4 mirror+nearest, texture transform, emissive strength
        MaterialDesc(base_color_tex=T(0, uv_index=1), normal_tex=T(1, uv_index=1, transform=xf), ior=1.9, roughness_factor=0.35, metallic_factor=0.0),  # 5 second UV set, ior
        MaterialDesc(base_color_tex=T(0), specular={"tex": T(2), "factor": 0.8, "color_tex": T(0), "color_factor": (1.0, 0.7, 0.4)},
                     metallic_factor=0.0, roughness_factor=0.4),                                                                             # 6 KHR_materials_specular
        MaterialDesc(base_color_tex=T(0), clearcoat={"tex": T(2), "factor": 0.9, "roughness_tex": T(2), "roughness_factor": 0.5, "normal_tex": T(1), "normal_scale": 0.8},
                     metallic_factor=0.3, roughness_factor=0.7),                                                                             # 7 clearcoat (all three textures)
        MaterialDesc(base_color_factor=(0.5, 0.2, 0.6, 1.0), sheen={"roughness_tex": T(2), "roughness_factor": 0.6, "color_tex": T(0), "color_factor": (0.9, 0.8, 1.0)},
                     metallic_factor=0.0, roughness_factor=0.9, double_sided=True),                                                          # 8 sheen, double sided
        MaterialDesc(kind="unlit", base_color_tex=T(0), base_color_factor=(0.8, 0.9, 1.0, 1.0), emissive_tex=T(3, sampler=1), emissive_factor=(0.2, 0.1, 0.0)),  # 9 unlit
        MaterialDesc(base_color_tex=T(0), normal_tex=T(1), debug_bitmask=4),                                                                 # 10 debug: normals
        MaterialDesc(base_color_tex=T(0), metallic_roughness_tex=T(2), debug_bitmask=2),                                                     # 11 debug: metallic/roughness
        MaterialDesc(base_color_tex=T(0), volume={"thickness_tex": T(2), "thickness_factor": 0.5, "attenuation_distance": 2.0, "attenuation_color": (0.8, 0.9, 0.7)},
                     transmission={"tex": None, "factor": 0.0}, clearcoat={"factor": 0.6, "roughness_factor": 0.1}, sheen={"roughness_factor": 0.3, "color_factor": (0.2, 0.3, 0.1)},
                     specular={"factor": 0.5}, ior=1.33, emissive_strength=1.5, emissive_factor=(0.05, 0.05, 0.1), metallic_factor=0.1, roughness_factor=0.5),  # 12 every block at once
        MaterialDesc(base_color_tex=T(63), normal_tex=T(1, sampler=9), roughness_factor=0.5, metallic_factor=0.0),                          # 13 dangling texture / sampler ids -> SkipTexture
    ]
    nodes = [NodeDesc()]
    cols = 7
    for m in range(len(mats)):
        gx, gy = m % cols, m // cols
        kk = rng.uniform(2.0, 5.0, size=3)

        def ball(U, V, kk=kk):
            th, phi = U * 2 * math.pi, (0.03 + 0.94 * V) * math.pi
            d = np.stack([np.sin(phi) * np.cos(th), np.cos(phi), -np.sin(phi) * np.sin(th)], axis=-1)
            return d * (0.42 * (1.0 + 0.1 * np.sin(d @ kk * 3.0)))[..., None]
        pos, nrm, tan, uvs, idx = grid_patch(ball, 20, 14, uv_scale=(2.0, 1.0))
        uv1 = (uvs[:, ::-1] * np.array([1.5, 0.75], dtype=F) + np.array([0.25, -0.1], dtype=F)).astype(F)
        col = np.concatenate([0.5 + 0.5 * nrm, np.ones((pos.shape[0], 1), dtype=F)], axis=1).astype(F)
        prim = PrimitiveDesc(positions=pos, normals=nrm, tangents=tan, uvs=[uvs, uv1], colors=[col], indices=idx, material=m)
        nodes.append(NodeDesc(translation=((gx - (cols - 1) / 2) * 1.05, (0.5 - gy) * 1.05, 0.0), rotation=quat_axis_angle((0.2, 1, 0.1), 0.3 * m),
                              scale=(1.0, 1.0 + 0.1 * (m % 3), 1.0), parent=0, primitives=[prim]))
    lights = list(DEFAULT_LIGHTS[:2]) + [
        {"kind": "point", "color": (1.0, 0.7, 0.5), "intensity": 8.0, "position": (-2.0, 1.5, 2.0), "range": 12.0},
        {"kind": "point", "color": (0.5, 0.7, 1.0), "intensity": 5.0, "position": (2.5, -1.0, 1.5), "range": 0.0},
        {"kind": "spot", "color": (0.9, 1.0, 0.8), "intensity": 20.0, "position": (0.0, 0.0, 3.5), "direction": (0.0, 0.0, -1.0), "range": 20.0,
         "inner_angle": 0.98, "outer_angle": 0.90},
    ]
    eye = (0.3, 0.2, 5.2)
    return SceneDesc(nodes=nodes, materials=mats, textures=textures, samplers=samplers, lights=lights, width=width, height=height,
                     view=look_at_rh(eye, (0, 0, 0)), proj=perspective_rh(math.radians(45), width / height, 0.1, 100.0), camera_position=eye,
                     skybox_rgba=(0.02, 0.03, 0.05, 1.0), prefiltered_rgb=(0.9, 0.95, 1.0), irradiance_rgb=(0.8, 0.85, 0.9))


def ortho_scene(width=320, height=240) -> SceneDesc:
    """The helmet-class mesh under an orthographic camera (standard.wgsl:41-49, skybox.wgsl:13-29 ortho branches)."""
    sc = helmet_scene(width, height, segments=32, rings=24, tex_size=32)
    a = width / height
    sc.proj = orthographic_rh(-1.6 * a, 1.6 * a, -1.6, 1.6, 0.1, 50.0)
    return sc


# ------------------------------------------------------------------------------------------------ instancing

def instanced_scene(width=640, height=360, seed=0xA35A0006) -> SceneDesc:
    """A textured floor plus two instanced meshes (a bumpy sphere x 24 on a ring, double-sided; a box x 9 in a grid, back-face
    culled, non-uniformly scaled and rotated) and one ordinary mesh between them in submission order.  Not a BASELINE
    config; it exercises GPU instancing: per-instance mat4s (instances.rs), model * instance in apply_vertex
    (apply_vertex.wgsl:47-59), one draw per mesh with an instance count (meshes/mesh.rs:91-121)."""
    rng = np.random.default_rng(seed)
    textures = [value_noise_rgba8(rng, 128, 8, base=(0.6, 0.5, 0.4), amp=(0.3, 0.3, 0.3)), value_noise_rgba8(rng, 128, 16, kind="normal")]
    mats = [MaterialDesc(base_color_tex=TextureRef(0), normal_tex=TextureRef(1), metallic_factor=0.1, roughness_factor=0.6),
            MaterialDesc(base_color_factor=(0.9, 0.3, 0.2, 1.0), metallic_factor=0.8, roughness_factor=0.35, double_sided=True),
            MaterialDesc(base_color_factor=(0.2, 0.5, 0.9, 1.0), base_color_tex=TextureRef(0), metallic_factor=0.0, roughness_factor=0.8)]

    def plane(U, V):
        return np.stack([(U - 0.5) * 16.0, np.zeros_like(U), (0.5 - V) * 16.0], axis=-1)

    def ball(U, V):
        th, phi = U * 2 * math.pi, (0.02 + 0.96 * V) * math.pi
        d = np.stack([np.sin(phi) * np.cos(th), np.cos(phi), -np.sin(phi) * np.sin(th)], axis=-1)
        return d * (0.35 * (1.0 + 0.15 * np.sin(6 * th) * np.sin(4 * phi)))[..., None]

    def quat_y(a):
        return (0.0, math.sin(a / 2), 0.0, math.cos(a / 2))

    ring = [((4.5 * math.cos(2 * math.pi * i / 24), 0.8 + 0.3 * math.sin(i), 4.5 * math.sin(2 * math.pi * i / 24)), quat_y(0.3 * i), (1.0 + 0.3 * (i % 3),) * 3)
            for i in range(24)]
    grid = [((-2.0 + 2.0 * (i % 3), 0.5, -2.0 + 2.0 * (i // 3)), quat_y(0.4 * i), (0.6, 0.4 + 0.2 * (i % 4), 0.9)) for i in range(9)]
    box = box_scene().nodes[1].primitives[0]
    box_prim = PrimitiveDesc(positions=box.positions, normals=box.normals, indices=box.indices, material=2, uvs=[np.zeros((len(box.positions), 2), dtype=np.float32) + 0.37],
                             instances=grid)
    nodes = [NodeDesc(),
             NodeDesc(parent=0, primitives=[_prim(grid_patch(plane, 16, 16, uv_scale=(4, 4)), 0)]),
             NodeDesc(parent=0, translation=(0.0, 0.2, 0.0), primitives=[_prim(grid_patch(ball, 24, 16, uv_scale=(2, 1)), 1, instances=ring)]),
             NodeDesc(parent=0, translation=(0.0, 1.6, 0.0), primitives=[_prim(grid_patch(ball, 24, 16, uv_scale=(2, 1)), 0)]),
             NodeDesc(parent=0, rotation=quat_y(0.5), primitives=[box_prim])]
    eye = (7.5, 5.0, 8.5)
    return SceneDesc(nodes=nodes, materials=mats, textures=textures, samplers=[dict(REPEAT_LINEAR)], lights=list(DEFAULT_LIGHTS), width=width, height=height,
                     view=look_at_rh(eye, (0.0, 0.6, 0.0)), proj=perspective_rh(math.radians(50), width / height, 0.1, 100.0), camera_position=eye)


# ------------------------------------------------------------------------------------------------ transparent pass

def transparent_scene(width=640, height=360, tex_size=64, seed=0xA35A0007, detail=1.0) -> SceneDesc:
    """An opaque backdrop (textured wall + floor + three spheres) seen through a row of transparent objects, one per branch of
    the forward pass (material_transparent_wgsl/fragment.wgsl + helpers): alpha blend with a textured alpha, a double-sided
    blended shell whose back faces show through its front faces, ALPHA_MODE_MASK with a cutoff, vertex-colour alpha, unlit
    blend, KHR_materials_transmission (smooth: one refracted tap of the opaque image; rough: the 25-tap blur; thin: no
    refraction), volume attenuation, a blended mesh that also carries morph targets, a blended instanced mesh, and two
    overlapping blended quads to pin the back-to-front order.  Not a BASELINE config."""
    rng = np.random.default_rng(seed)
    textures = [
        value_noise_rgba8(rng, tex_size, 8, base=(0.6, 0.5, 0.4), amp=(0.3, 0.3, 0.3)),      # 0 colour, opaque alpha
        value_noise_rgba8(rng, tex_size, 12, kind="normal"),                                 # 1 normal
        value_noise_rgba8(rng, tex_size, 6, base=(0.5, 0.6, 0.7), amp=(0.3, 0.3, 0.3)),      # 2 colour with varying alpha
        value_noise_rgba8(rng, tex_size, 10, base=(0.5, 0.5, 0.5), amp=(0.4, 0.4, 0.4)),     # 3 data (transmission / thickness)
    ]
    a = value_noise_rgba8(rng, tex_size, 5, base=(0.5, 0.5, 0.5), amp=(0.5, 0.5, 0.5))[..., 0]
    textures[2][..., 3] = a
    T = TextureRef
    mats = [
        MaterialDesc(base_color_tex=T(0), normal_tex=T(1), metallic_factor=0.0, roughness_factor=0.6),                                        # 0 opaque backdrop
        MaterialDesc(base_color_factor=(0.9, 0.3, 0.2, 1.0), metallic_factor=0.6, roughness_factor=0.35),                                    # 1 opaque sphere
        MaterialDesc(base_color_tex=T(2), base_color_factor=(1.0, 1.0, 1.0, 0.8), normal_tex=T(1), alpha_mode="blend", metallic_factor=0.0, roughness_factor=0.4),   # 2 blend, textured alpha
        MaterialDesc(base_color_factor=(0.2, 0.6, 0.9, 0.45), alpha_mode="blend", double_sided=True, metallic_factor=0.1, roughness_factor=0.3),                 # 3 blend, double sided
        MaterialDesc(base_color_tex=T(2), alpha_mode="mask", alpha_cutoff=0.5, double_sided=True, metallic_factor=0.0, roughness_factor=0.7),                    # 4 mask
        MaterialDesc(base_color_factor=(1.0, 1.0, 1.0, 0.9), vertex_color_set=0, alpha_mode="blend", metallic_factor=0.0, roughness_factor=0.8),                 # 5 vertex colour (alpha in the colour)
        MaterialDesc(kind="unlit", base_color_tex=T(2), base_color_factor=(0.9, 1.0, 0.8, 0.7), emissive_factor=(0.1, 0.0, 0.1), alpha_mode="blend"),            # 6 unlit blend
        MaterialDesc(base_color_factor=(0.95, 0.98, 1.0, 1.0), transmission={"factor": 0.95}, volume={"thickness_factor": 0.35, "attenuation_distance": 1.5, "attenuation_color": (0.7, 0.9, 0.8)},
                     ior=1.5, metallic_factor=0.0, roughness_factor=0.02),                                                                    # 7 smooth glass: refraction, one tap
        MaterialDesc(base_color_factor=(1.0, 0.9, 0.8, 1.0), transmission={"factor": 0.8, "tex": T(3)}, volume={"thickness_factor": 0.25, "thickness_tex": T(3)},
                     ior=1.7, metallic_factor=0.0, roughness_factor=0.45, normal_tex=T(1)),                                                   # 8 rough glass: 25-tap blur, textured factors
        MaterialDesc(base_color_factor=(0.8, 1.0, 0.9, 1.0), transmission={"factor": 0.6}, metallic_factor=0.0, roughness_factor=0.2,
                     clearcoat={"factor": 0.5, "roughness_factor": 0.1}),                                                                     # 9 thin-walled transmission (no volume): unrefracted tap
        MaterialDesc(base_color_factor=(0.9, 0.8, 0.2, 0.6), alpha_mode="blend", metallic_factor=0.0, roughness_factor=0.5),                 # 10 blend + morph targets
        MaterialDesc(base_color_factor=(0.7, 0.2, 0.8, 0.5), alpha_mode="blend", metallic_factor=0.0, roughness_factor=0.5, double_sided=True),   # 11 blend, instanced
        MaterialDesc(base_color_factor=(1.0, 0.1, 0.1, 0.5), alpha_mode="blend", double_sided=True, roughness_factor=0.9, metallic_factor=0.0),  # 12 overlapping quad A
        MaterialDesc(base_color_factor=(0.1, 1.0, 0.1, 0.5), alpha_mode="blend", double_sided=True, roughness_factor=0.9, metallic_factor=0.0),  # 13 overlapping quad B
    ]

    def wall(U, V):
        return np.stack([(U - 0.5) * 14.0, (V - 0.5) * 8.0, np.full_like(U, -3.0) + 0.3 * np.sin(U * 9.0) * np.sin(V * 7.0)], axis=-1)

    def floor(U, V):
        return np.stack([(U - 0.5) * 14.0, np.full_like(U, -1.6), (0.5 - V) * 10.0 - 1.0], axis=-1)

    def ball_fn(radius, bump, k):
        def ball(U, V):
            th, phi = U * 2 * math.pi, (0.03 + 0.94 * V) * math.pi
            d = np.stack([np.sin(phi) * np.cos(th), np.cos(phi), -np.sin(phi) * np.sin(th)], axis=-1)
            return d * (radius * (1.0 + bump * np.sin(d @ k * 3.0)))[..., None]
        return ball

    def quad_fn(w, h):
        def quad(U, V):
            return np.stack([(U - 0.5) * w, (V - 0.5) * h, np.zeros_like(U)], axis=-1)
        return quad

    def prim(fn, nu, nv, material, with_color=False, **kw):
        nu, nv = max(2, int(round(nu * detail))), max(2, int(round(nv * detail)))     # detail: tessellation factor (benchmarks)
        pos, nrm, tan, uvs, idx = grid_patch(fn, nu, nv, uv_scale=(2.0, 1.0))
        colors = []
        if with_color:
            colors = [np.concatenate([0.5 + 0.5 * nrm, (0.35 + 0.6 * uvs[:, 1:2] / 1.0).clip(0, 1)], axis=1).astype(F)]
        return PrimitiveDesc(positions=pos, normals=nrm, tangents=tan, uvs=[uvs], colors=colors, indices=idx, material=material, **kw)

    nodes = [NodeDesc(),
             NodeDesc(parent=0, primitives=[prim(wall, 24, 16, 0)]),
             NodeDesc(parent=0, primitives=[prim(floor, 16, 16, 0)])]
    for i, x in enumerate((-3.0, 0.2, 3.2)):
        nodes.append(NodeDesc(parent=0, translation=(x, -0.4 + 0.5 * i, -1.6), primitives=[prim(ball_fn(0.7, 0.1, np.array([2.0, 3.0, 4.0])), 20, 14, 1 if i != 1 else 0)]))
    # front row of transparent objects
    xs = np.linspace(-4.2, 4.2, 9)
    row = [2, 3, 4, 5, 6, 7, 8, 9]
    for j, m in enumerate(row):
        kk = rng.uniform(2.0, 5.0, size=3)
        p = prim(ball_fn(0.42, 0.08, kk), 20, 14, m, with_color=(m == 5))
        nodes.append(NodeDesc(parent=0, translation=(float(xs[j]), 0.55 if j % 2 else -0.45, 1.0 + 0.15 * j), rotation=quat_axis_angle((0.2, 1, 0.1), 0.4 * j), primitives=[p]))
    # morph-target blended sphere
    base = prim(ball_fn(0.4, 0.05, np.array([3.0, 2.0, 5.0])), 18, 12, 10)
    tgt = {"positions": (base.normals * 0.15 * np.sin(base.positions[:, 1:2] * 9.0)).astype(F), "normals": (0.2 * np.cos(base.positions * 5.0)).astype(F)}
    base.morph_targets = [tgt]
    base.morph_weights = np.array([0.8], dtype=F)
    nodes.append(NodeDesc(parent=0, translation=(float(xs[8]), -0.4, 1.4), primitives=[base]))
    # instanced blended shards
    inst = [((-3.5 + 1.4 * i, 1.7 + 0.1 * (i % 2), 0.4 + 0.2 * i), quat_axis_angle((0.3, 1.0, 0.2), 0.7 * i), (0.5, 0.35 + 0.1 * (i % 3), 0.5)) for i in range(6)]
    nodes.append(NodeDesc(parent=0, primitives=[prim(ball_fn(0.5, 0.2, np.array([4.0, 1.0, 2.0])), 10, 8, 11, instances=inst)]))
    # two interpenetrating blended quads: submission order = back to front by the closest AABB corner, whatever the per-pixel order is
    nodes.append(NodeDesc(parent=0, translation=(-0.6, -0.9, 2.2), rotation=quat_axis_angle((0, 1, 0), 0.5), primitives=[prim(quad_fn(1.6, 0.9), 4, 3, 12)]))
    nodes.append(NodeDesc(parent=0, translation=(-0.3, -0.8, 2.3), rotation=quat_axis_angle((0, 1, 0), -0.6), primitives=[prim(quad_fn(1.6, 0.9), 4, 3, 13)]))
    lights = list(DEFAULT_LIGHTS[:2]) + [{"kind": "point", "color": (1.0, 0.9, 0.7), "intensity": 12.0, "position": (0.0, 2.5, 3.0), "range": 20.0}]
    eye = (0.4, 0.3, 6.0)
    return SceneDesc(nodes=nodes, materials=mats, textures=textures, samplers=[dict(REPEAT_LINEAR), dict(CLAMP_LINEAR)], lights=lights, width=width, height=height,
                     view=look_at_rh(eye, (0, 0, 0)), proj=perspective_rh(math.radians(50), width / height, 0.1, 100.0), camera_position=eye,
                     skybox_rgba=(0.02, 0.03, 0.05, 1.0), prefiltered_rgb=(0.9, 0.95, 1.0), irradiance_rgb=(0.8, 0.85, 0.9))


def hud_scene(width=640, height=360, tex_size=64) -> SceneDesc:
    """transparent_scene's world plus four hud meshes (Mesh.hud: render.rs:169-178,301-312) in front of the camera: an opaque-material panel,
    a half-transparent quad overlapping it (their mutual order is decided by hud_depth), a textured panel, and a quad that lies BEHIND the
    world's back wall — a hud mesh hides the world whatever the world's depth, so it must still show.  Not a BASELINE config."""
    sc = transparent_scene(width, height, tex_size=tex_size)

    def quad_fn(w, h):
        def quad(U, V):
            return np.stack([(U - 0.5) * w, (V - 0.5) * h, np.zeros_like(U)], axis=-1)
        return quad

    def hud_prim(w, h, material):
        pos, nrm, tan, uvs, idx = grid_patch(quad_fn(w, h), 5, 4, uv_scale=(1.0, 1.0))
        return PrimitiveDesc(positions=pos, normals=nrm, tangents=tan, uvs=[uvs], indices=idx, material=material, hud=True)

    sc.nodes.append(NodeDesc(parent=0, translation=(-1.3, 0.9, 3.6), rotation=quat_axis_angle((0, 1, 0), 0.25), primitives=[hud_prim(1.3, 0.7, 1)]))      # opaque material
    sc.nodes.append(NodeDesc(parent=0, translation=(-0.9, 0.7, 3.9), rotation=quat_axis_angle((0, 1, 0), -0.2), primitives=[hud_prim(1.2, 0.8, 12)]))     # blend 0.5, overlaps the first
    sc.nodes.append(NodeDesc(parent=0, translation=(1.5, -0.6, 3.4), rotation=quat_axis_angle((1, 0, 0), 0.3), primitives=[hud_prim(1.1, 0.8, 2)]))       # textured alpha
    sc.nodes.append(NodeDesc(parent=0, translation=(2.6, 1.6, -6.0), primitives=[hud_prim(4.0, 2.5, 13)]))                                                # behind the wall (z = -3)
    return sc
