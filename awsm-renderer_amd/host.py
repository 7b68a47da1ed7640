"""
host.py — ctypes binding of libawsm_host.so (include/awsm_host.h) and `populate()`, which feeds a SceneDesc through
the key-based API in the order the reference's populate_gltf does (crates/renderer/src/gltf/populate.rs:185-205:
node transforms -> skins -> meshes, and per primitive morph -> skin -> material -> mesh, populate/mesh.rs:89-311).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import numpy as np

from . import PACKAGE_DIR, hip_backend
from .hip_backend import AwsmDraw, AwsmEnv, AwsmFrameStats, AwsmSampler
from .scene_desc import MaterialDesc, SceneDesc, TextureRef, texture_mip_kinds

LIB_PATH = os.environ.get("AWSM_HOST_LIB") or os.path.join(PACKAGE_DIR, "libawsm_host.so")   # AWSM_HOST_LIB: instrumented builds of the same ABI (tests)
F32P = C.POINTER(C.c_float)
U32P = C.POINTER(C.c_uint32)


class TexRef(C.Structure):
    _fields_ = [("texture", C.c_int32), ("sampler", C.c_uint32), ("uv_index", C.c_uint32), ("pad", C.c_uint32), ("transform", C.c_uint64)]


class HostMaterial(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("shader", C.c_uint32), ("double_sided", C.c_uint32), ("base_color_factor", C.c_float * 4), ("metallic_factor", C.c_float),
                ("roughness_factor", C.c_float), ("normal_scale", C.c_float), ("occlusion_strength", C.c_float), ("emissive_factor", C.c_float * 3),
                ("debug_bitmask", C.c_uint32), ("base_color_tex", TexRef), ("metallic_roughness_tex", TexRef), ("normal_tex", TexRef),
                ("occlusion_tex", TexRef), ("emissive_tex", TexRef),
                ("has_vertex_color", C.c_uint32), ("vertex_color_set", C.c_uint32), ("has_emissive_strength", C.c_uint32), ("emissive_strength", C.c_float),
                ("has_ior", C.c_uint32), ("ior", C.c_float),
                ("has_specular", C.c_uint32), ("specular_factor", C.c_float), ("specular_color_factor", C.c_float * 3), ("specular_tex", TexRef), ("specular_color_tex", TexRef),
                ("has_transmission", C.c_uint32), ("transmission_factor", C.c_float), ("transmission_tex", TexRef),
                ("has_volume", C.c_uint32), ("volume_thickness_factor", C.c_float), ("volume_attenuation_distance", C.c_float),
                ("volume_attenuation_color", C.c_float * 3), ("volume_thickness_tex", TexRef),
                ("has_clearcoat", C.c_uint32), ("clearcoat_factor", C.c_float), ("clearcoat_roughness_factor", C.c_float), ("clearcoat_normal_scale", C.c_float),
                ("clearcoat_tex", TexRef), ("clearcoat_roughness_tex", TexRef), ("clearcoat_normal_tex", TexRef),
                ("has_sheen", C.c_uint32), ("sheen_roughness_factor", C.c_float), ("sheen_color_factor", C.c_float * 3), ("sheen_roughness_tex", TexRef), ("sheen_color_tex", TexRef),
                ("alpha_mode", C.c_uint32), ("alpha_cutoff", C.c_float),
                ("has_diffuse_transmission", C.c_uint32), ("diffuse_transmission_factor", C.c_float), ("diffuse_transmission_color_factor", C.c_float * 3),
                ("diffuse_transmission_tex", TexRef), ("diffuse_transmission_color_tex", TexRef),
                ("has_dispersion", C.c_uint32), ("dispersion", C.c_float),
                ("has_anisotropy", C.c_uint32), ("anisotropy_strength", C.c_float), ("anisotropy_rotation", C.c_float), ("anisotropy_tex", TexRef),
                ("has_iridescence", C.c_uint32), ("iridescence_factor", C.c_float), ("iridescence_ior", C.c_float), ("iridescence_thickness_min", C.c_float),
                ("iridescence_thickness_max", C.c_float), ("iridescence_tex", TexRef), ("iridescence_thickness_tex", TexRef)]


class MorphTarget(C.Structure):
    _fields_ = [("positions", F32P), ("normals", F32P), ("tangents", F32P)]


class HostPrimitive(C.Structure):
    _fields_ = [("vertex_count", C.c_uint32), ("triangle_count", C.c_uint32), ("positions", F32P), ("normals", F32P), ("tangents", F32P),
                ("indices", U32P), ("n_uv_sets", C.c_uint32), ("uv_sets", F32P * 8), ("n_color_sets", C.c_uint32), ("color_sets", F32P * 4),
                ("n_morph_targets", C.c_uint32), ("morph_targets", C.POINTER(MorphTarget)), ("morph_weights", F32P), ("animated_morph_weights", F32P),
                ("front_face_cw", C.c_uint32)]


class HostLight(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("color", C.c_float * 3), ("intensity", C.c_float), ("position", C.c_float * 3), ("range", C.c_float),
                ("direction", C.c_float * 3), ("inner_angle", C.c_float), ("outer_angle", C.c_float)]


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} not found: run __graft_entry__.build()")
    lib = C.CDLL(LIB_PATH)
    u64, vp, i64, sz = C.c_uint64, C.c_void_p, C.c_int64, C.c_size_t
    sig = {
        "awsm_host_create": (C.c_int, [C.c_char_p, C.c_int, vp, C.c_uint32, C.POINTER(vp)]),
        "awsm_host_destroy": (C.c_int, [vp]), "awsm_host_last_error": (C.c_char_p, [vp]), "awsm_host_device_ctx": (vp, [vp]),
        "awsm_host_transform_root": (u64, [vp]), "awsm_host_transform_insert": (u64, [vp, F32P, F32P, F32P, u64]),
        "awsm_host_transform_set_local": (C.c_int, [vp, u64, F32P, F32P, F32P]), "awsm_host_transform_set_parent": (C.c_int, [vp, u64, u64]),
        "awsm_host_transform_remove": (C.c_int, [vp, u64]), "awsm_host_transform_parent": (u64, [vp, u64]),
        "awsm_host_transform_world": (C.c_int, [vp, u64, F32P]),
        "awsm_host_texture_insert": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32]), "awsm_host_sampler_insert": (C.c_int, [vp, vp]),
        "awsm_host_texture_transform_insert": (u64, [vp, F32P, F32P, C.c_float, F32P]),
        "awsm_host_material_insert": (u64, [vp, vp]), "awsm_host_material_update": (C.c_int, [vp, u64, vp]), "awsm_host_material_offset": (i64, [vp, u64]),
        "awsm_host_skin_insert": (u64, [vp, C.POINTER(u64), C.c_uint32, F32P, C.c_uint32, C.POINTER(U32P), C.POINTER(F32P), C.c_uint32]),
        "awsm_host_mesh_insert": (u64, [vp, vp, u64, u64, u64, C.c_uint32]), "awsm_host_mesh_insert_hud": (u64, [vp, vp, u64, u64, u64, C.c_uint32]),
        "awsm_host_hud_draw_lists": (C.c_int, [vp, vp, vp, C.c_uint32, U32P]), "awsm_host_mesh_remove": (C.c_int, [vp, u64]),
        "awsm_host_light_insert": (u64, [vp, vp]), "awsm_host_light_remove": (C.c_int, [vp, u64]),
        "awsm_host_set_ibl_mip_counts": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
        "awsm_host_camera_update": (C.c_int, [vp, F32P, F32P, F32P]), "awsm_host_env": (C.c_int, [vp, vp]),
        "awsm_host_env_cube": (C.c_int, [vp, C.c_int, C.c_uint32, C.c_uint32, vp]),
        "awsm_host_set_render_hooks": (C.c_int, [vp, vp, vp, vp, vp]),
        "awsm_host_brdf_lut_generate": (C.c_int, [vp, C.c_uint32, C.c_uint32]), "awsm_host_resize": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
        "awsm_host_set_shard_rows": (C.c_int, [vp, C.c_uint32, C.c_uint32]), "awsm_host_set_shard_bands": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32]), "awsm_host_set_render_timings": (C.c_int, [vp, C.c_int]),
        "awsm_host_pick": (C.c_int, [vp, C.c_int32, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
        "awsm_host_set_anti_aliasing": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
        "awsm_host_mesh_set_instances": (C.c_int, [vp, u64, F32P, C.c_uint32]), "awsm_host_mesh_append_instances": (C.c_int, [vp, u64, F32P, C.c_uint32]),
        "awsm_host_texture_insert_kind": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32]), "awsm_host_update_transforms": (C.c_int, [vp]),
        "awsm_host_render": (C.c_int, [vp, C.c_int, vp]), "awsm_host_mirror": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(sz)]),
        "awsm_host_load_gltf": (C.c_int, [vp, C.c_char_p, C.c_int, vp, C.c_char_p, C.c_size_t]),
        "awsm_host_decode_image": (C.c_int, [C.c_char_p, C.c_size_t, vp, C.c_size_t, U32P, U32P, C.c_char_p, C.c_size_t]),
        "awsm_host_draw_list": (C.c_int, [vp, vp, C.c_uint32, U32P]), "awsm_host_transparent_draw_list": (C.c_int, [vp, vp, C.c_uint32, U32P]), "awsm_host_texture_array_count": (C.c_uint32, [vp]),
        "awsm_host_texture_array_info": (C.c_int, [vp, C.c_uint32, U32P, U32P, U32P, C.POINTER(vp)]),
        "awsm_host_upload_bytes_last_frame": (u64, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _f(arr):
    a = np.ascontiguousarray(arr, dtype=np.float32)
    return a, a.ctypes.data_as(F32P)


class HostError(RuntimeError):
    pass


class Host:
    """One AwsmHost: scene state + a device context reached through the awsm_hip_* C-ABI of `backend_path`."""

    def __init__(self, backend_path: Optional[str] = None, device: int = 0, stream: Optional[int] = None, parity_tap: bool = False, overlap_frames: bool = False,
                 anisotropic: bool = False):
        self.lib = load_library()
        backend_path = backend_path or hip_backend.LIB_PATH
        if not os.path.exists(backend_path):
            raise FileNotFoundError(f"backend library {backend_path} not found (no CPU fallback exists for the product path)")
        h = C.c_void_p()
        flags = (hip_backend.AWSM_CFG_PARITY_TAP if parity_tap else 0) | (hip_backend.AWSM_CFG_OVERLAP_FRAMES if overlap_frames else 0) | (hip_backend.AWSM_CFG_ANISOTROPIC if anisotropic else 0)
        rc = self.lib.awsm_host_create(backend_path.encode(), device, stream, flags, C.byref(h))
        if rc != 0:
            raise HostError(f"awsm_host_create({backend_path}) failed with status {rc}")
        self.h = h
        self.width = self.height = 0

    def _chk(self, rc, where):
        if rc != 0:
            raise HostError(f"{where} failed ({rc}): {(self.lib.awsm_host_last_error(self.h) or b'').decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.awsm_host_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_ctx(self) -> int:
        return self.lib.awsm_host_device_ctx(self.h)

    # ---- transforms ----
    def transform_insert(self, t, r, s, parent=0) -> int:
        (_, pt), (_, pr), (_, ps) = _f(t), _f(r), _f(s)
        k1, k2, k3 = _f(t), _f(r), _f(s)
        return self.lib.awsm_host_transform_insert(self.h, k1[1], k2[1], k3[1], parent)

    def transform_set_local(self, key, t, r, s):
        k1, k2, k3 = _f(t), _f(r), _f(s)
        self._chk(self.lib.awsm_host_transform_set_local(self.h, key, k1[1], k2[1], k3[1]), "transform_set_local")

    def transform_parent(self, key) -> int:
        return self.lib.awsm_host_transform_parent(self.h, key)

    def transform_world(self, key) -> np.ndarray:
        out = np.zeros(16, dtype=np.float32)
        self._chk(self.lib.awsm_host_transform_world(self.h, key, out.ctypes.data_as(F32P)), "transform_world")
        return out.reshape(4, 4)

    # ---- textures / samplers ----
    def texture_insert(self, image: np.ndarray, mip_kind: int = 0) -> int:
        img = np.ascontiguousarray(image, dtype=np.uint8)
        r = self.lib.awsm_host_texture_insert_kind(self.h, img.ctypes.data_as(C.c_void_p), img.shape[1], img.shape[0], mip_kind)
        if r < 0:
            self._chk(r, "texture_insert")
        return r

    def sampler_insert(self, s: dict) -> int:
        smp = AwsmSampler(s.get("address_mode_u", 1), s.get("address_mode_v", 1), s.get("mag_filter", 1), s.get("min_filter", 1),
                          s.get("mipmap_filter", 1), s.get("max_anisotropy", 1))
        r = self.lib.awsm_host_sampler_insert(self.h, C.byref(smp))
        if r < 0:
            self._chk(r, "sampler_insert")
        return r

    def texture_transform_insert(self, offset=(0, 0), origin=(0, 0), rotation=0.0, scale=(1, 1)) -> int:
        a, b, c = _f(offset), _f(origin), _f(scale)
        return self.lib.awsm_host_texture_transform_insert(self.h, a[1], b[1], float(rotation), c[1])

    # ---- materials ----
    def material_insert(self, m: HostMaterial) -> int:
        k = self.lib.awsm_host_material_insert(self.h, C.byref(m))
        if not k:
            self._chk(-1, "material_insert")
        return k

    def material_update(self, key: int, m: HostMaterial):
        self._chk(self.lib.awsm_host_material_update(self.h, key, C.byref(m)), "material_update")

    # ---- skins / meshes ----
    def skin_insert(self, joint_keys: List[int], inverse_bind: np.ndarray, joints: List[np.ndarray], weights: List[np.ndarray]) -> int:
        n = len(joint_keys)
        keys = (C.c_uint64 * n)(*joint_keys)
        ib = _f(inverse_bind)
        js = [np.ascontiguousarray(j, dtype=np.uint32) for j in joints]
        ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        jp = (U32P * len(js))(*[j.ctypes.data_as(U32P) for j in js])
        wp = (F32P * len(ws))(*[w.ctypes.data_as(F32P) for w in ws])
        k = self.lib.awsm_host_skin_insert(self.h, keys, n, ib[1], len(js), jp, wp, js[0].shape[0])
        if not k:
            self._chk(-1, "skin_insert")
        return k

    def hud_draw_lists(self):
        """(HUD geometry pass draws, HUD transparent pass draws): the hud meshes, back to front (render.rs:169-178,301-312)."""
        n = C.c_uint32()
        self._chk(self.lib.awsm_host_hud_draw_lists(self.h, None, None, 0, C.byref(n)), "hud_draw_lists")
        g, t = (AwsmDraw * max(1, n.value))(), (AwsmDraw * max(1, n.value))()
        self._chk(self.lib.awsm_host_hud_draw_lists(self.h, g, t, n.value, C.byref(n)), "hud_draw_lists")
        conv = lambda arr: [{"geom_meta_off": d.geom_meta_off, "vis_data_off": d.vis_data_off, "tri_count": d.tri_count, "flags": d.flags,
                             **({"inst_off": d.inst_off, "inst_count": d.inst_count} if d.inst_count else {})} for d in arr[: n.value]]     # noqa: E731
        return conv(g), conv(t)

    def mesh_insert(self, p, transform: int, material: int, skin: int = 0, hidden: bool = False, front_face_cw: bool = False) -> int:
        keep = []

        def fp(a):
            if a is None:
                return None
            arr = np.ascontiguousarray(a, dtype=np.float32)
            keep.append(arr)
            return arr.ctypes.data_as(F32P)

        hp = HostPrimitive()
        idx = np.ascontiguousarray(p.indices, dtype=np.uint32).reshape(-1, 3)
        keep.append(idx)
        hp.vertex_count, hp.triangle_count = p.positions.shape[0], idx.shape[0]
        hp.positions, hp.normals, hp.tangents = fp(p.positions), fp(p.normals), fp(p.tangents)
        hp.indices = idx.ctypes.data_as(U32P)
        hp.n_uv_sets = len(p.uvs)
        for i, u in enumerate(p.uvs):
            hp.uv_sets[i] = fp(u)
        hp.n_color_sets = len(p.colors)
        for i, c in enumerate(p.colors):
            hp.color_sets[i] = fp(c)
        hp.n_morph_targets = len(p.morph_targets)
        if p.morph_targets:
            mts = (MorphTarget * len(p.morph_targets))()
            for i, t in enumerate(p.morph_targets):
                mts[i] = MorphTarget(fp(t.get("positions")), fp(t.get("normals")), fp(t.get("tangents")))
            keep.append(mts)
            hp.morph_targets = C.cast(mts, C.POINTER(MorphTarget))
            hp.morph_weights = fp(p.morph_weights)
            hp.animated_morph_weights = fp(p.animated_morph_weights)
        hp.front_face_cw = 1 if front_face_cw else 0
        insert = self.lib.awsm_host_mesh_insert_hud if getattr(p, "hud", False) else self.lib.awsm_host_mesh_insert
        k = insert(self.h, C.byref(hp), transform, material, skin, 1 if hidden else 0)
        if not k:
            self._chk(-1, "mesh_insert")
        return k

    def mesh_remove(self, key: int):
        self._chk(self.lib.awsm_host_mesh_remove(self.h, key), "mesh_remove")

    # ---- lights / camera / env ----
    def light_insert(self, l: dict) -> int:
        kind = {"directional": 1, "point": 2, "spot": 3}[l["kind"]]
        hl = HostLight(kind, (C.c_float * 3)(*l["color"]), l["intensity"], (C.c_float * 3)(*l.get("position", (0, 0, 0))), l.get("range", 0.0),
                       (C.c_float * 3)(*l.get("direction", (0, 0, 0))), l.get("inner_angle", 0.0), l.get("outer_angle", 0.0))
        return self.lib.awsm_host_light_insert(self.h, C.byref(hl))

    def light_remove(self, key: int):
        self._chk(self.lib.awsm_host_light_remove(self.h, key), "light_remove")

    def set_ibl_mip_counts(self, prefiltered: int, irradiance: int):
        self._chk(self.lib.awsm_host_set_ibl_mip_counts(self.h, prefiltered, irradiance), "set_ibl_mip_counts")

    def camera_update(self, view, proj, position):
        a, b, c = _f(view), _f(proj), _f(position)
        self._chk(self.lib.awsm_host_camera_update(self.h, a[1], b[1], c[1]), "camera_update")

    def env(self, skybox=(0, 0, 0, 1), prefiltered=(1, 1, 1), irradiance=(1, 1, 1), lut_rgba16f: Optional[np.ndarray] = None):
        env = AwsmEnv()
        for i in range(4):
            env.skybox_rgba[i] = skybox[i]
        for i in range(3):
            env.prefiltered_rgb[i] = prefiltered[i]
            env.irradiance_rgb[i] = irradiance[i]
        keep = None
        if lut_rgba16f is not None:
            keep = np.ascontiguousarray(lut_rgba16f, dtype=np.uint16)
            env.brdf_lut_height, env.brdf_lut_width = keep.shape[0], keep.shape[1]
            env.brdf_lut_rgba16f = keep.ctypes.data
        self._chk(self.lib.awsm_host_env(self.h, C.byref(env)), "env")

    def env_cube(self, which: int, levels):
        """levels: [level0, ...] of (6, N_l, N_l, 4) float16 (faces +X -X +Y -Y +Z -Z), or None."""
        if not levels:
            self._chk(self.lib.awsm_host_env_cube(self.h, which, 0, 0, None), "env_cube")
            return
        flat = np.ascontiguousarray(np.concatenate([np.ascontiguousarray(a, dtype=np.float16).reshape(-1) for a in levels])).view(np.uint16)
        self._chk(self.lib.awsm_host_env_cube(self.h, which, levels[0].shape[1], len(levels), flat.ctypes.data), "env_cube")

    def brdf_lut_generate(self, w: int, h: int):
        self._chk(self.lib.awsm_host_brdf_lut_generate(self.h, w, h), "brdf_lut_generate")

    def resize(self, w: int, h: int):
        self._chk(self.lib.awsm_host_resize(self.h, w, h), "resize")
        self.width, self.height = w, h

    def set_shard_rows(self, y0: int, y1: int):
        self._chk(self.lib.awsm_host_set_shard_rows(self.h, y0, y1), "set_shard_rows")

    @staticmethod
    def _trs10(instances) -> np.ndarray:
        return np.ascontiguousarray([list(t) + list(r) + list(sc) for (t, r, sc) in instances], dtype=np.float32).reshape(-1, 10)

    def mesh_set_instances(self, mesh_key: int, instances):
        """Meshes::enable_mesh_instancing / set_mesh_instances: [(translation, rotation xyzw, scale), ...]."""
        a, ap = _f(self._trs10(instances))
        self._chk(self.lib.awsm_host_mesh_set_instances(self.h, mesh_key, ap, len(a)), "mesh_set_instances")

    def mesh_append_instances(self, mesh_key: int, instances) -> int:
        a, ap = _f(self._trs10(instances))
        r = self.lib.awsm_host_mesh_append_instances(self.h, mesh_key, ap, len(a))
        if r < 0:
            self._chk(r, "mesh_append_instances")
        return r

    def set_anti_aliasing(self, msaa_sample_count: int = 0, mipmap: bool = False):
        """AwsmRenderer::set_anti_aliasing: msaa 0 (None) or 4, gradient mipmaps on/off (the reference's default: 4, True)."""
        self._chk(self.lib.awsm_host_set_anti_aliasing(self.h, msaa_sample_count, 1 if mipmap else 0), "set_anti_aliasing")

    def pick(self, x: int, y: int):
        """AwsmRenderer::pick: the MeshKey under pixel (x, y) of the last rendered frame, or None (PickResult::Miss)."""
        hit, key = C.c_uint32(0), C.c_uint64(0)
        self._chk(self.lib.awsm_host_pick(self.h, x, y, C.byref(hit), C.byref(key)), "pick")
        return key.value if hit.value else None

    def set_shard_bands(self, n: int, r: int, compact_output: bool = False):
        self._chk(self.lib.awsm_host_set_shard_bands(self.h, n, r, 1 if compact_output else 0), "set_shard_bands")

    def set_render_timings(self, enabled: bool):
        """AwsmRendererLogging.render_timings (debug.rs:8-12): per-stage times in the frame stats (default on)."""
        self._chk(self.lib.awsm_host_set_render_timings(self.h, 1 if enabled else 0), "set_render_timings")

    # ---- frame ----
    def set_render_hooks(self, after_geometry_pass=None, after_opaque_pass=None):
        """RenderHooks: Python callables (no arguments; an exception aborts the frame) run between the passes of render()."""
        HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p)

        def wrap(fn):
            if fn is None:
                return None

            def call(_user):
                try:
                    fn()
                    return 0
                except Exception as e:      # noqa: BLE001 — reported through the status code
                    self._hook_error = e
                    return -3
            return HOOK(call)
        self._hooks = (wrap(after_geometry_pass), wrap(after_opaque_pass))     # kept alive as long as they are installed
        self._chk(self.lib.awsm_host_set_render_hooks(self.h, C.cast(self._hooks[0], C.c_void_p) if self._hooks[0] else None, None,
                                                      C.cast(self._hooks[1], C.c_void_p) if self._hooks[1] else None, None), "set_render_hooks")

    def update_transforms(self):
        self._chk(self.lib.awsm_host_update_transforms(self.h), "update_transforms")

    def render(self, sync: bool = True) -> Optional[dict]:
        st = AwsmFrameStats()
        st.struct_size = C.sizeof(AwsmFrameStats)
        self._chk(self.lib.awsm_host_render(self.h, 1 if sync else 0, C.byref(st) if sync else None), "render")
        return st.as_dict() if sync else None

    # ---- introspection ----
    def mirror(self, which: int) -> bytes:
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(self.lib.awsm_host_mirror(self.h, which, C.byref(p), C.byref(n)), "mirror")
        return C.string_at(p, n.value)

    def load_gltf(self, path: str, scene_index: int = -1) -> dict:
        """Populate this host from a .gltf / .glb file (awsm_host_load_gltf); returns the counts of what was inserted."""
        info = (C.c_uint32 * 12)()
        err = C.create_string_buffer(512)
        rc = self.lib.awsm_host_load_gltf(self.h, os.fsencode(path), scene_index, info, err, 512)
        if rc != 0:
            raise HostError(f"load_gltf({path}) failed ({rc}): {err.value.decode(errors='replace')}")
        names = ("nodes", "meshes", "materials", "images", "samplers", "skins", "lights", "triangles", "generated_tangents")
        return {k: int(info[i]) for i, k in enumerate(names)}

    def transparent_draw_list(self) -> List[dict]:
        """The world transparent pass's list: back to front; vis_data_off = offset into the transparency geometry buffer."""
        return self.draw_list(fn="awsm_host_transparent_draw_list")

    def draw_list(self, fn: str = "awsm_host_draw_list") -> List[dict]:
        n = C.c_uint32()
        f = getattr(self.lib, fn)
        self._chk(f(self.h, None, 0, C.byref(n)), "draw_list")
        arr = (AwsmDraw * max(1, n.value))()
        self._chk(f(self.h, arr, n.value, C.byref(n)), "draw_list")
        out = []
        for d in arr[:n.value]:
            e = {"geom_meta_off": d.geom_meta_off, "vis_data_off": d.vis_data_off, "tri_count": d.tri_count, "flags": d.flags}
            if d.inst_count:
                e["inst_off"], e["inst_count"] = d.inst_off, d.inst_count
            out.append(e)
        return out

    def upload_bytes_last_frame(self) -> int:
        return self.lib.awsm_host_upload_bytes_last_frame(self.h)


def _texref(ref: Optional[TextureRef], tt_keys: Dict[tuple, int], host: Host) -> TexRef:
    if ref is None:
        return TexRef(-1, 0, 0, 0, 0)
    tk = 0
    if ref.transform:
        t = ref.transform
        sig = (tuple(t.get("offset", (0, 0))), tuple(t.get("origin", (0, 0))), float(t.get("rotation", 0.0)), tuple(t.get("scale", (1, 1))))
        if sig not in tt_keys:
            tt_keys[sig] = host.texture_transform_insert(*sig)
        tk = tt_keys[sig]
    return TexRef(ref.texture, ref.sampler, ref.uv_index, 0, tk)


def material_struct(m: MaterialDesc, host: Host, tt_keys: Dict[tuple, int]) -> HostMaterial:
    tr = lambda r: _texref(r, tt_keys, host)   # noqa: E731
    hm = HostMaterial()
    hm.struct_size = C.sizeof(HostMaterial)
    hm.shader = 2 if m.kind == "unlit" else 1
    hm.double_sided = 1 if m.double_sided else 0
    hm.base_color_factor = (C.c_float * 4)(*m.base_color_factor)
    hm.metallic_factor, hm.roughness_factor, hm.normal_scale, hm.occlusion_strength = m.metallic_factor, m.roughness_factor, m.normal_scale, m.occlusion_strength
    hm.emissive_factor = (C.c_float * 3)(*m.emissive_factor)
    hm.debug_bitmask = m.debug_bitmask
    hm.alpha_mode = {"opaque": 0, "mask": 1, "blend": 2}[m.alpha_mode]
    hm.alpha_cutoff = m.alpha_cutoff
    hm.base_color_tex, hm.metallic_roughness_tex, hm.normal_tex = tr(m.base_color_tex), tr(m.metallic_roughness_tex), tr(m.normal_tex)
    hm.occlusion_tex, hm.emissive_tex = tr(m.occlusion_tex), tr(m.emissive_tex)
    none = TexRef(-1, 0, 0, 0, 0)
    for name in ("specular_tex", "specular_color_tex", "transmission_tex", "volume_thickness_tex", "clearcoat_tex", "clearcoat_roughness_tex",
                 "clearcoat_normal_tex", "sheen_roughness_tex", "sheen_color_tex", "diffuse_transmission_tex", "diffuse_transmission_color_tex", "anisotropy_tex",
                 "iridescence_tex", "iridescence_thickness_tex"):
        setattr(hm, name, none)
    if m.vertex_color_set is not None:
        hm.has_vertex_color, hm.vertex_color_set = 1, m.vertex_color_set
    if m.emissive_strength is not None:
        hm.has_emissive_strength, hm.emissive_strength = 1, m.emissive_strength
    if m.ior is not None:
        hm.has_ior, hm.ior = 1, m.ior
    if m.specular is not None:
        s = m.specular
        hm.has_specular, hm.specular_factor = 1, s.get("factor", 1.0)
        hm.specular_color_factor = (C.c_float * 3)(*s.get("color_factor", (1, 1, 1)))
        hm.specular_tex, hm.specular_color_tex = tr(s.get("tex")), tr(s.get("color_tex"))
    if m.transmission is not None:
        s = m.transmission
        hm.has_transmission, hm.transmission_factor, hm.transmission_tex = 1, s.get("factor", 0.0), tr(s.get("tex"))
    if m.volume is not None:
        s = m.volume
        hm.has_volume, hm.volume_thickness_factor, hm.volume_attenuation_distance = 1, s.get("thickness_factor", 0.0), s.get("attenuation_distance", 0.0)
        hm.volume_attenuation_color = (C.c_float * 3)(*s.get("attenuation_color", (1, 1, 1)))
        hm.volume_thickness_tex = tr(s.get("thickness_tex"))
    if m.clearcoat is not None:
        s = m.clearcoat
        hm.has_clearcoat, hm.clearcoat_factor, hm.clearcoat_roughness_factor = 1, s.get("factor", 0.0), s.get("roughness_factor", 0.0)
        hm.clearcoat_normal_scale = s.get("normal_scale", 1.0)
        hm.clearcoat_tex, hm.clearcoat_roughness_tex, hm.clearcoat_normal_tex = tr(s.get("tex")), tr(s.get("roughness_tex")), tr(s.get("normal_tex"))
    if m.sheen is not None:
        s = m.sheen
        hm.has_sheen, hm.sheen_roughness_factor = 1, s.get("roughness_factor", 0.0)
        hm.sheen_color_factor = (C.c_float * 3)(*s.get("color_factor", (0, 0, 0)))
        hm.sheen_roughness_tex, hm.sheen_color_tex = tr(s.get("roughness_tex")), tr(s.get("color_tex"))
    if m.diffuse_transmission is not None:
        s = m.diffuse_transmission
        hm.has_diffuse_transmission, hm.diffuse_transmission_factor = 1, s.get("factor", 0.0)
        hm.diffuse_transmission_color_factor = (C.c_float * 3)(*s.get("color_factor", (1, 1, 1)))
        hm.diffuse_transmission_tex, hm.diffuse_transmission_color_tex = tr(s.get("tex")), tr(s.get("color_tex"))
    if m.dispersion is not None:
        hm.has_dispersion, hm.dispersion = 1, m.dispersion
    if m.anisotropy is not None:
        s = m.anisotropy
        hm.has_anisotropy, hm.anisotropy_strength, hm.anisotropy_rotation, hm.anisotropy_tex = 1, s.get("strength", 0.0), s.get("rotation", 0.0), tr(s.get("tex"))
    if m.iridescence is not None:
        s = m.iridescence
        hm.has_iridescence, hm.iridescence_factor, hm.iridescence_ior = 1, s.get("factor", 0.0), s.get("ior", 1.3)
        hm.iridescence_thickness_min, hm.iridescence_thickness_max = s.get("thickness_min", 100.0), s.get("thickness_max", 400.0)
        hm.iridescence_tex, hm.iridescence_thickness_tex = tr(s.get("tex")), tr(s.get("thickness_tex"))
    return hm


def decode_image(data: bytes) -> np.ndarray:
    """PNG / baseline JPEG bytes -> (h, w, 4) u8 through the host library's own decoders (awsm_host_decode_image)."""
    lib = load_library()
    w, h = C.c_uint32(), C.c_uint32()
    err = C.create_string_buffer(256)
    rc = lib.awsm_host_decode_image(data, len(data), None, 0, C.byref(w), C.byref(h), err, 256)
    if rc != 0:
        raise HostError(f"decode_image failed ({rc}): {err.value.decode(errors='replace')}")
    out = np.zeros((h.value, w.value, 4), dtype=np.uint8)
    rc = lib.awsm_host_decode_image(data, len(data), out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(w), C.byref(h), err, 256)
    if rc != 0:
        raise HostError(f"decode_image failed ({rc}): {err.value.decode(errors='replace')}")
    return out


class Populated:
    """Keys produced by populate(): what the reference keeps in GltfKeyLookups."""

    def __init__(self):
        self.node_keys: List[int] = []
        self.mesh_keys: List[int] = []
        self.material_keys: Dict[int, int] = {}
        self.skin_keys: Dict[int, int] = {}
        self.light_keys: List[int] = []


def populate(host: Host, scene: SceneDesc) -> Populated:
    out = Populated()
    kinds = texture_mip_kinds(scene)
    for tex, kind in zip(scene.textures, kinds):
        host.texture_insert(tex, kind)
    for s in scene.samplers:
        host.sampler_insert(s)
    host.set_ibl_mip_counts(scene.prefiltered_mip_count, scene.irradiance_mip_count)
    children: Dict[Optional[int], List[int]] = {}
    for i, n in enumerate(scene.nodes):
        children.setdefault(n.parent, []).append(i)
    out.node_keys = [0] * len(scene.nodes)

    def add_transform(i, parent_key):
        n = scene.nodes[i]
        out.node_keys[i] = host.transform_insert(n.translation, n.rotation, n.scale, parent_key)
        for c in children.get(i, []):
            add_transform(c, out.node_keys[i])

    for r in children.get(None, []):
        add_transform(r, 0)
    joint_nodes = set(j for sk in scene.skins for j in sk.joints)
    tt_keys: Dict[tuple, int] = {}

    def add_meshes(i):
        n = scene.nodes[i]
        if n.primitives:
            tk = out.node_keys[i]
            if i in joint_nodes:   # populate/mesh.rs:36-52: a skinned mesh on a joint node gets a fresh identity transform
                tk = host.transform_insert((0, 0, 0), (0, 0, 0, 1), (1, 1, 1), host.transform_parent(tk))
            for p in n.primitives:
                skin_key = 0
                if n.skin is not None and p.joints:   # one Skins::insert per primitive, like populate_gltf_primitive
                    sk = scene.skins[n.skin]
                    skin_key = host.skin_insert([out.node_keys[j] for j in sk.joints], sk.inverse_bind, p.joints, p.weights)
                if p.material not in out.material_keys:
                    out.material_keys[p.material] = host.material_insert(material_struct(scene.materials[p.material], host, tt_keys))
                out.mesh_keys.append(host.mesh_insert(p, tk, out.material_keys[p.material], skin_key))
                if p.instances is not None:
                    host.mesh_set_instances(out.mesh_keys[-1], p.instances)
        for c in children.get(i, []):
            add_meshes(c)

    for r in children.get(None, []):
        add_meshes(r)
    for l in scene.lights:
        out.light_keys.append(host.light_insert(l))
    return out


class Renderer:
    """Convenience wrapper: Host + populated scene, frame loop = update_all -> render (crates/renderer/src/update.rs, render.rs)."""

    def __init__(self, scene: SceneDesc, backend_path: Optional[str] = None, device: int = 0, stream: Optional[int] = None, parity_tap: bool = False,
                 lut_rgba16f: Optional[np.ndarray] = None, lut_size: int = 1024, msaa: int = 0, mipmap: bool = False, overlap_frames: bool = False,
                 gltf: Optional[str] = None, anisotropic: bool = False):
        """gltf: path of a .gltf / .glb file to populate from (AwsmRenderer::populate_gltf); `scene` then only supplies the frame size,
        the camera and the environment."""
        self.scene = scene
        self.host = Host(backend_path, device, stream, parity_tap, overlap_frames, anisotropic)      # anisotropic: AWSM_CFG_ANISOTROPIC (include/awsm_hip.h)
        self.host.set_anti_aliasing(msaa, mipmap)   # AwsmRendererBuilder::with_anti_aliasing (off unless asked: BASELINE configs are single-sampled, MipmapMode::None)
        self.host.resize(scene.width, scene.height)
        if gltf is not None:
            self.host.set_ibl_mip_counts(scene.prefiltered_mip_count, scene.irradiance_mip_count)
            self.gltf_info = self.host.load_gltf(gltf)
            self.keys = None
        else:
            self.keys = populate(self.host, scene)
        if lut_rgba16f is not None:
            self.host.env(scene.skybox_rgba, scene.prefiltered_rgb, scene.irradiance_rgb, lut_rgba16f)
        else:
            self.host.env(scene.skybox_rgba, scene.prefiltered_rgb, scene.irradiance_rgb)
            self.host.brdf_lut_generate(lut_size, lut_size)   # BrdfLut::new at build() time (renderer-core brdf_lut/generate.rs)
        for k, name in enumerate(("skybox", "prefiltered", "irradiance")):
            if scene.env_cubes and scene.env_cubes.get(name):
                self.host.env_cube(k, scene.env_cubes[name])
        self.update()

    def update(self):
        self.host.update_transforms()
        self.host.camera_update(self.scene.view, self.scene.proj, self.scene.camera_position)

    def render(self, sync: bool = True):
        return self.host.render(sync)

    def close(self):
        self.host.close()
