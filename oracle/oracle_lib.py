"""
oracle_lib.py — TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/liboracle.so (oracle/c/*.c) plus a helper that
runs a full oracle frame from the mirrors of oracle.scene_model.HostModel.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Optional

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_DIR, "liboracle.so")
BUF_COUNT = 19
MAX_TEX = 64
MAX_SAMPLERS = 32


class AwsmDraw(C.Structure):
    _fields_ = [("geom_meta_off", C.c_uint32), ("vis_data_off", C.c_uint32), ("tri_count", C.c_uint32), ("flags", C.c_uint32),
                ("inst_off", C.c_uint32), ("inst_count", C.c_uint32)]


class AwsmSampler(C.Structure):
    _fields_ = [("address_mode_u", C.c_uint32), ("address_mode_v", C.c_uint32), ("mag_filter", C.c_uint32), ("min_filter", C.c_uint32),
                ("mipmap_filter", C.c_uint32), ("max_anisotropy", C.c_uint32)]


class OracleTexArray(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("layers", C.c_uint32), ("mips", C.c_uint32)]


class OracleCube(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("size", C.c_uint32), ("mips", C.c_uint32)]


CUBE_SLOTS = ("skybox", "prefiltered", "irradiance")


def pack_cube(levels) -> np.ndarray:
    """[level0, level1, ...] of (6, N_l, N_l, 4) float16 -> the flat [mip][face][y][x][4] uint16 array both the oracle and the C-ABI take."""
    size = levels[0].shape[1]
    for l, a in enumerate(levels):
        n = max(size >> l, 1)
        assert a.shape == (6, n, n, 4) and a.dtype == np.float16, (l, a.shape, a.dtype)
    return np.ascontiguousarray(np.concatenate([a.reshape(-1) for a in levels])).view(np.uint16)


class OracleScene(C.Structure):
    _fields_ = [("buf", C.c_void_p * BUF_COUNT), ("buf_size", C.c_uint64 * BUF_COUNT), ("width", C.c_uint32), ("height", C.c_uint32),
                ("y0", C.c_uint32), ("y1", C.c_uint32), ("draws", C.POINTER(AwsmDraw)), ("n_draws", C.c_uint32), ("has_opaque", C.c_uint32),
                ("n_tex_arrays", C.c_uint32), ("tex_arrays", OracleTexArray * MAX_TEX), ("n_samplers", C.c_uint32),
                ("samplers", AwsmSampler * MAX_SAMPLERS), ("skybox_rgba", C.c_float * 4), ("prefiltered_rgb", C.c_float * 4),
                ("irradiance_rgb", C.c_float * 4), ("brdf_lut_rg16f", C.c_void_p), ("lut_width", C.c_uint32), ("lut_height", C.c_uint32),
                ("msaa", C.c_uint32), ("mipmap", C.c_uint32), ("cube", OracleCube * 3)]


def build(force: bool = False) -> str:
    """Compile oracle/c/*.c -> liboracle.so (gcc, -ffp-contract=off)."""
    srcs = [os.path.join(_DIR, "c", f) for f in os.listdir(os.path.join(_DIR, "c"))] + [os.path.join(_DIR, "..", "include", "awsm_hip.h")]
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", _DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_total_vertices.restype = C.c_uint32
        _lib.oracle_mip_levels.restype = C.c_uint32
        _lib.oracle_det_atan2f.restype = C.c_float
        _lib.oracle_det_atan2f.argtypes = [C.c_float, C.c_float]
        _lib.oracle_f32_to_f16.restype = C.c_uint16
        _lib.oracle_f32_to_f16.argtypes = [C.c_float]
        _lib.oracle_f16_to_f32.restype = C.c_float
        _lib.oracle_f16_to_f32.argtypes = [C.c_uint16]
    return _lib


def brdf_lut(width: int, height: int, threads: int = 8) -> np.ndarray:
    out = np.zeros((height, width, 2), dtype=np.uint16)
    rc = lib().oracle_brdf_lut(C.c_uint32(width), C.c_uint32(height), out.ctypes.data_as(C.c_void_p), C.c_int(threads))
    assert rc == 0
    return out


def mip_levels(width: int, height: int) -> int:
    return int(lib().oracle_mip_levels(C.c_uint32(width), C.c_uint32(height)))


def mip_chain(texels: np.ndarray, kinds=None):
    """[layers, h, w, 4] level 0 -> (flat uint8 chain [level][layer][h_l][w_l][4], levels): the oracle's generate_mipmaps."""
    layers, h, w, _ = texels.shape
    L = lib()
    L.oracle_mip_chain_bytes.restype = C.c_size_t
    levels = mip_levels(w, h)
    n = int(L.oracle_mip_chain_bytes(C.c_uint32(w), C.c_uint32(h), C.c_uint32(layers), C.c_uint32(levels)))
    chain = np.zeros(n + 16, dtype=np.uint8)
    chain[: layers * h * w * 4] = np.ascontiguousarray(texels, dtype=np.uint8).reshape(-1)
    k = (C.c_uint32 * layers)(*([0] * layers if kinds is None else [int(x) for x in kinds]))
    assert L.oracle_generate_mips(C.c_uint32(w), C.c_uint32(h), C.c_uint32(layers), k, C.c_uint32(levels), chain.ctypes.data_as(C.c_void_p)) == 0
    return chain, levels


def mip_level_view(chain: np.ndarray, width: int, height: int, layers: int, level: int) -> np.ndarray:
    off = sum(layers * max(1, width >> l) * max(1, height >> l) for l in range(level)) * 4
    wl, hl = max(1, width >> level), max(1, height >> level)
    return chain[off: off + layers * hl * wl * 4].reshape(layers, hl, wl, 4)


def lut_rg_to_rgba16f(rg: np.ndarray) -> np.ndarray:
    """RG16F -> the reference's RGBA16F texel (b = 0, a = 1.0)."""
    h, w, _ = rg.shape
    out = np.zeros((h, w, 4), dtype=np.uint16)
    out[..., :2] = rg
    out[..., 3] = 0x3C00
    return out


NO_HIT_KEY = np.uint64(0xFFFFFFFFFFFFFFFF)


class OracleFrame:
    """Holds the numpy arrays an OracleScene points at and runs the three oracle stages."""

    def __init__(self, mirrors: Dict[int, bytes], draws: List[dict], width: int, height: int, tex_arrays: List[dict], samplers: List[dict],
                 lut_rg16f: np.ndarray, skybox=(0, 0, 0, 1), prefiltered=(1, 1, 1), irradiance=(1, 1, 1), rows=(0, 0), has_opaque=True, msaa=0, mipmap=False,
                 env_cubes=None):
        self._keep = []
        s = OracleScene()
        for i in range(BUF_COUNT):
            data = mirrors.get(i)
            if data is None:
                continue
            arr = np.frombuffer(bytes(data) + bytes(16), dtype=np.uint8).copy()
            self._keep.append(arr)
            s.buf[i] = arr.ctypes.data
            s.buf_size[i] = len(data)
        s.width, s.height, s.y0, s.y1 = width, height, rows[0], rows[1]
        self.draw_arr = (AwsmDraw * max(1, len(draws)))()
        for i, d in enumerate(draws):
            self.draw_arr[i] = AwsmDraw(d["geom_meta_off"], d["vis_data_off"], d["tri_count"], d["flags"], d.get("inst_off", 0), d.get("inst_count", 0))
        s.draws = C.cast(self.draw_arr, C.POINTER(AwsmDraw))
        s.n_draws = len(draws)
        s.has_opaque = 1 if has_opaque else 0
        assert msaa in (0, 4)
        s.msaa = msaa
        self.msaa = msaa
        s.n_tex_arrays = len(tex_arrays)
        s.mipmap = 1 if mipmap else 0
        self.mip_chains = []
        for i, t in enumerate(tex_arrays):
            if mipmap:     # TexturePoolArray: mipmap = true, full chain generated per layer kind (texture_pool.rs:187-320)
                chain, levels = mip_chain(t["texels"], t.get("kinds"))
                self._keep.append(chain)
                self.mip_chains.append((chain, levels))
                s.tex_arrays[i] = OracleTexArray(chain.ctypes.data, t["width"], t["height"], t["layers"], levels)
                continue
            arr = np.ascontiguousarray(t["texels"], dtype=np.uint8)
            self._keep.append(arr)
            s.tex_arrays[i] = OracleTexArray(arr.ctypes.data, t["width"], t["height"], t["layers"], 1)
        s.n_samplers = len(samplers)
        for i, sm in enumerate(samplers):
            s.samplers[i] = AwsmSampler(sm.get("address_mode_u", 1), sm.get("address_mode_v", 1), sm.get("mag_filter", 1), sm.get("min_filter", 1),
                                        sm.get("mipmap_filter", 1), sm.get("max_anisotropy", 1))
        for i in range(4):
            s.skybox_rgba[i] = skybox[i]
        for i in range(3):
            s.prefiltered_rgb[i] = prefiltered[i]
            s.irradiance_rgb[i] = irradiance[i]
        for k, name in enumerate(CUBE_SLOTS):
            levels = (env_cubes or {}).get(name)
            if levels:
                flat = pack_cube(levels)
                self._keep.append(flat)
                s.cube[k] = OracleCube(flat.ctypes.data, levels[0].shape[1], len(levels))
        self.lut = np.ascontiguousarray(lut_rg16f, dtype=np.uint16)
        s.brdf_lut_rg16f = self.lut.ctypes.data
        s.lut_height, s.lut_width = self.lut.shape[0], self.lut.shape[1]
        self.scene = s
        self.width, self.height = width, height
        self.n_verts = int(lib().oracle_total_vertices(C.byref(s)))
        self.clip = np.zeros((max(1, self.n_verts), 4), dtype=np.float32)
        self.nt = np.zeros((max(1, self.n_verts), 8), dtype=np.float32)
        self.keys = np.zeros((height, width, 4) if msaa == 4 else (height, width), dtype=np.uint64)   # msaa: [y][x][sample]
        self.rgba32f = np.zeros((height, width, 4), dtype=np.float32)
        self.rgba16f = np.zeros((height, width, 4), dtype=np.uint16)

    def transform(self):
        assert lib().oracle_transform(C.byref(self.scene), self.clip.ctypes.data_as(C.c_void_p), self.nt.ctypes.data_as(C.c_void_p)) == 0
        return self

    def raster(self, threads=8):
        assert lib().oracle_raster(C.byref(self.scene), self.clip.ctypes.data_as(C.c_void_p), self.keys.ctypes.data_as(C.c_void_p), C.c_int(threads)) == 0
        return self

    def gbuffer(self, threads=8):
        """(height, width, 6) f32: packed normal / tangent (RGBA16F values), barycentric (RG16F values) per pixel; zeros where nothing was hit."""
        out = np.zeros(self.keys.shape + (6,), dtype=np.float32)
        assert lib().oracle_gbuffer(C.byref(self.scene), self.clip.ctypes.data_as(C.c_void_p), self.nt.ctypes.data_as(C.c_void_p),
                                    self.keys.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_int(threads)) == 0
        return out

    def shade(self, threads=8):
        lib().oracle_set_anisotropic(C.c_int(1 if getattr(self, "anisotropic", False) else 0))      # AWSM_CFG_ANISOTROPIC's twin (oracle_shade.c: sample_array_grad)
        assert lib().oracle_shade(C.byref(self.scene), self.clip.ctypes.data_as(C.c_void_p), self.nt.ctypes.data_as(C.c_void_p),
                                  self.keys.ctypes.data_as(C.c_void_p), self.rgba32f.ctypes.data_as(C.c_void_p),
                                  self.rgba16f.ctypes.data_as(C.c_void_p), C.c_int(threads)) == 0
        return self

    def run(self, threads=8):
        return self.transform().raster(threads).shade(threads)

    def conditioning(self, threads=8) -> np.ndarray:
        """(height, width, 4) f64: per pixel and channel, how far the oracle's own colour moves when the decoded normal (two tilts), the
        reconstructed world position (two shifts across the view ray) or n.h in the GGX lobe is off by 16 ulps — the pixel's condition number times epsilon, measured
        (oracle_shade.c: oracle_set_perturbation).  Call after shade(); leaves rgba32f / rgba16f as shade() made them."""
        base32, base16 = self.rgba32f.copy(), self.rgba16f.copy()
        worst = np.zeros(base32.shape, dtype=np.float64)
        try:
            for k in (1, 2, 3, 4, 5):
                lib().oracle_set_perturbation(C.c_int(k))
                self.shade(threads)
                d = np.abs(self.rgba32f.astype(np.float64) - base32.astype(np.float64))
                worst = np.maximum(worst, np.where(np.isfinite(d), d, np.inf))
        finally:
            lib().oracle_set_perturbation(C.c_int(0))
            self.rgba32f[...] = base32
            self.rgba16f[...] = base16
        return worst

    def hud_geometry(self, model, threads=8) -> np.ndarray:
        """GeometryRenderPass::render(.., &renderables.hud, true) (render.rs:169-178): the hud meshes rasterised with a depth buffer of their own.
        Returns their keys (no hit = all ones) and keeps them (self.hud_keys); shade() then leaves the pixels they cover cleared
        (compute.wgsl:176-179).  A second OracleFrame over the same mirrors with the hud draw list does the work."""
        sc = model.scene
        draws = model.hud_geometry_draws
        hud = OracleFrame(model.mirrors(), draws, sc.width, sc.height, model.texture_arrays(), sc.samplers, self.lut) if draws else None
        self.hud_keys = hud.transform().raster(threads).keys.copy() if hud else np.full(self.keys.shape, NO_HIT_KEY, dtype=np.uint64)
        return self.hud_keys

    def apply_hud_clear(self):
        """after shade(): the opaque pass returns at a pixel whose visibility texel belongs to a hud mesh — it stays as clear_opaque left it"""
        covered = self.hud_keys != NO_HIT_KEY
        self.rgba32f[covered] = 0.0
        self.rgba16f[covered] = 0
        return self

    def forward(self, draws: List[dict], threads=8, hud=False):
        """World transparent pass over `draws` (HostModel.collect_transparent_draws()), after run(): fills fwd_clip / fwd_nt / fwd_wpos
        and the composite image (composite32f holds the f16 values, composite16f their bits).
        hud=True: the HUD transparent pass (render.rs:301-312, :490-521) over the composite a previous forward() left — colours LoadOp::Load
        (the composite is the image blended over), depth against hud_depth, cleared (every key reads as no hit = depth 1.0)."""
        L = lib()
        L.oracle_set_anisotropic(C.c_int(1 if getattr(self, "anisotropic", False) else 0))
        arr = (AwsmDraw * max(1, len(draws)))()
        for i, d in enumerate(draws):
            arr[i] = AwsmDraw(d["geom_meta_off"], d["vis_data_off"], d["tri_count"], d["flags"], d.get("inst_off", 0), d.get("inst_count", 0))
        n = len(draws)
        L.oracle_forward_total_vertices.restype = C.c_uint32
        nv = int(L.oracle_forward_total_vertices(arr, C.c_uint32(n)))
        prev_composite16 = self.composite16f.copy() if hud else None
        self.fwd_n_verts = nv
        self.fwd_clip = np.zeros((max(1, nv), 4), dtype=np.float32)
        self.fwd_nt = np.zeros((max(1, nv), 8), dtype=np.float32)
        self.fwd_wpos = np.zeros((max(1, nv), 4), dtype=np.float32)
        self.composite32f = np.zeros((self.height, self.width, 4), dtype=np.float32)
        self.composite16f = np.zeros((self.height, self.width, 4), dtype=np.uint16)
        self.fwd_touched = np.zeros((self.height, self.width), dtype=np.uint8)
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        keys, source = self.keys, self.rgba16f
        if hud:
            keys = np.full(self.keys.shape, NO_HIT_KEY, dtype=np.uint64)
            source = prev_composite16
        assert L.oracle_forward_transform(C.byref(self.scene), arr, C.c_uint32(n), p(self.fwd_clip), p(self.fwd_nt), p(self.fwd_wpos)) == 0
        assert L.oracle_forward(C.byref(self.scene), arr, C.c_uint32(n), p(self.fwd_clip), p(self.fwd_nt), p(self.fwd_wpos), p(keys), p(source),
                                p(self.composite32f), p(self.composite16f), p(self.fwd_touched), C.c_int(threads)) == 0
        return self

    def unpack_visibility(self):
        tri = np.zeros(self.keys.shape, dtype=np.uint32)
        meta = np.zeros(self.keys.shape, dtype=np.uint32)
        depth = np.zeros(self.keys.shape, dtype=np.float32)
        assert lib().oracle_unpack_visibility(C.byref(self.scene), self.keys.ctypes.data_as(C.c_void_p), tri.ctypes.data_as(C.c_void_p),
                                              meta.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p)) == 0
        return tri, meta, depth


def frame_with_hud_msaa(model, lut_rg16f: np.ndarray, threads=8, msaa=4, mipmap=False) -> OracleFrame:
    """An MSAA frame with hud meshes, as the reference's targets hold it when the opaque pass runs (render.rs:169-178; geometry/render_pass.rs:55-57,
    107-114): the HUD geometry pass draws over the visibility / barycentric / normal targets (LoadOp::Load) but tests and writes hud_depth, so a sample a hud
    mesh covers shows the HUD triangle and still the WORLD's depth (1.0 where the world left none).  The opaque pass (compute.wgsl:118-180,303-318,
    helpers/msaa.wgsl, helpers/material_shading.wgsl:170-210) then: leaves a pixel whose sample 0 is a hud triangle cleared; feeds hud normals and world
    depths to the edge detector; and shades hud samples like any other in msaa_resolve_samples ("this may bleed a little", compute.wgsl:180).

    Built from three OracleFrames over the same mirrors, with no new arithmetic: the world draws alone (keys = the world's depth and ids), the hud draws alone
    (their coverage, depth-tested among themselves), and both lists as ONE draw list (world first) whose keys are the merge — the hud triangle's rank in
    that list under the world's depth bits — shaded by the ordinary MSAA code.  Returns the combined frame, shaded, with .keys set back to the WORLD's keys
    (what the device keeps in its visibility buffer, what the world transparent pass tests against) and .n_verts / .clip / .nt to the world's, .hud_keys =
    the hud pass's own keys, .merged_keys = what the opaque pass read."""
    sc = model.scene
    kw = dict(skybox=sc.skybox_rgba, prefiltered=sc.prefiltered_rgb, irradiance=sc.irradiance_rgb, msaa=msaa, mipmap=mipmap, env_cubes=sc.env_cubes)
    world_draws, hud_draws = model.collect_draws(), model.hud_geometry_draws
    mk = lambda draws: OracleFrame(model.mirrors(), draws, sc.width, sc.height, model.texture_arrays(), sc.samplers, lut_rg16f, **kw)   # noqa: E731
    world = mk(world_draws).transform().raster(threads)
    hud = mk(hud_draws).transform().raster(threads)
    comb = mk(world_draws + hud_draws).transform()
    t_world = np.uint64(world.n_verts // 3)
    no = hud.keys == NO_HIT_KEY
    depth_w = np.where(world.keys == NO_HIT_KEY, np.uint64(0x3F800000), world.keys >> np.uint64(32))      # depth cleared to 1.0 (render_pass.rs:107-114)
    merged = np.where(no, world.keys, (depth_w << np.uint64(32)) | ((hud.keys & np.uint64(0xFFFFFFFF)) - t_world))      # low word = 0xFFFFFFFF - rank
    comb.keys[...] = merged
    comb.shade(threads)
    comb.merged_keys, comb.hud_keys = merged, hud.keys.copy()
    comb.keys = world.keys.copy()
    comb.n_verts, comb.clip, comb.nt = world.n_verts, world.clip, world.nt
    comb._world, comb._hud = world, hud
    return comb


def frame_from_model(model, lut_rg16f: np.ndarray, rows=(0, 0), has_opaque=True, msaa=0, mipmap=False, anisotropic=False) -> OracleFrame:
    sc = model.scene
    fr = OracleFrame(model.mirrors(), model.collect_draws(), sc.width, sc.height, model.texture_arrays(), sc.samplers, lut_rg16f,
                       skybox=sc.skybox_rgba, prefiltered=sc.prefiltered_rgb, irradiance=sc.irradiance_rgb, rows=rows, has_opaque=has_opaque, msaa=msaa, mipmap=mipmap,
                       env_cubes=sc.env_cubes)
    fr.anisotropic = anisotropic
    return fr


def sample_grad(texels: np.ndarray, sampler: dict, uv, ddx, ddy, layer=0, anisotropic=False, kinds=None) -> np.ndarray:
    """textureSampleGrad on an RGBA8 array [layers, h, w, 4] (its chain generated here) by the oracle's contract: uv / ddx / ddy (n, 2) -> (n, 4) f32."""
    chain, levels = mip_chain(texels, kinds)
    layers, h, w, _ = texels.shape
    arr = OracleTexArray(chain.ctypes.data, w, h, layers, levels)
    smp = AwsmSampler(sampler.get("address_mode_u", 1), sampler.get("address_mode_v", 1), sampler.get("mag_filter", 1), sampler.get("min_filter", 1),
                      sampler.get("mipmap_filter", 1), sampler.get("max_anisotropy", 1))
    a = [np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 2) for x in (uv, ddx, ddy)]
    out = np.zeros((a[0].shape[0], 4), dtype=np.float32)
    lib().oracle_sample_grad(C.byref(arr), C.byref(smp), C.c_uint32(layer), a[0].ctypes.data_as(C.c_void_p), a[1].ctypes.data_as(C.c_void_p), a[2].ctypes.data_as(C.c_void_p),
                             C.c_uint32(a[0].shape[0]), C.c_int(1 if anisotropic else 0), out.ctypes.data_as(C.c_void_p))
    return out


def sample_cube(levels, dirs: np.ndarray, lods: np.ndarray) -> np.ndarray:
    """textureSampleLevel on a cube by the oracle's contract: dirs (n, 3) f32, lods (n,) f32 -> (n, 4) f32."""
    flat = pack_cube(levels)
    cube = OracleCube(flat.ctypes.data, levels[0].shape[1], len(levels))
    d = np.ascontiguousarray(dirs, dtype=np.float32)
    l = np.ascontiguousarray(lods, dtype=np.float32)
    out = np.zeros((d.shape[0], 4), dtype=np.float32)
    lib().oracle_sample_cube(C.byref(cube), d.ctypes.data_as(C.c_void_p), l.ctypes.data_as(C.c_void_p), C.c_uint32(d.shape[0]), out.ctypes.data_as(C.c_void_p))
    return out
