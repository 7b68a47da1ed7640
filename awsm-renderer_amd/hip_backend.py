"""
hip_backend.py — thin ctypes binding of libawsm_hip.so (include/awsm_hip.h).  No fallback: if the HIP library
is missing or fails to load, importing callers get an exception.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import PACKAGE_DIR

LIB_PATH = os.environ.get("AWSM_HIP_LIB") or os.path.join(PACKAGE_DIR, "libawsm_hip.so")   # AWSM_HIP_LIB: A/B builds of the same ABI
BUF_COUNT = 19
AWSM_CFG_PARITY_TAP = 1
AWSM_CFG_SMALL_BIN_LIST = 2
AWSM_CFG_OVERLAP_FRAMES = 4
AWSM_CFG_GENERAL_SHADE_ONLY = 8
AWSM_CFG_ANISOTROPIC = 16

BUF_NAMES = ["TRANSFORMS", "NORMAL_MATS", "MATERIALS", "LIGHTS", "LIGHTS_INFO", "CAMERA", "SKIN_MATRICES", "SKIN_INDEX_WEIGHTS",
             "MORPH_WEIGHTS", "MORPH_VALUES", "GEOM_META", "MATERIAL_META", "VIS_GEOM_DATA", "VIS_GEOM_INDEX", "ATTR_DATA", "ATTR_INDEX",
             "TEXTURE_TRANSFORMS", "INSTANCES", "TRANSPARENCY_GEOM_DATA"]

# every symbol include/awsm_hip.h declares (tests/test_abi_symbols.py checks the header against this list too)
EXPORTS = ["awsm_hip_create", "awsm_hip_destroy", "awsm_hip_last_error", "awsm_hip_abi_version", "awsm_hip_buffer_create",
           "awsm_hip_buffer_write", "awsm_hip_resize", "awsm_hip_set_shard_rows", "awsm_hip_set_shard_bands", "awsm_hip_set_stage_timers", "awsm_hip_pick", "awsm_hip_texture_array_upload", "awsm_hip_texture_array_generate_mips", "awsm_hip_texture_array_read_level", "awsm_hip_sampler_set",
           "awsm_hip_env_upload", "awsm_hip_brdf_lut_generate", "awsm_hip_read_brdf_lut", "awsm_hip_geometry_pass", "awsm_hip_opaque_pass",
           "awsm_hip_frame_end", "awsm_hip_frame_flush", "awsm_hip_read_gbuffer", "awsm_hip_stream_handoff", "awsm_hip_bind_output", "awsm_hip_output_device_ptr", "awsm_hip_read_visibility",
           "awsm_hip_read_visibility_unpacked", "awsm_hip_read_opaque", "awsm_hip_read_opaque_f32", "awsm_hip_read_transformed",
           "awsm_hip_device_info", "awsm_hip_transparent_pass", "awsm_hip_read_composite", "awsm_hip_read_composite_f32", "awsm_hip_bind_composite",
           "awsm_hip_read_transformed_forward", "awsm_hip_visibility_digest", "awsm_hip_bind_output_rows", "awsm_hip_env_cube_upload", "awsm_hip_bind_opaque_source", "awsm_hip_msaa_halo_bands", "awsm_hip_msaa_halo_export", "awsm_hip_msaa_halo_bind",
           "awsm_hip_frame_trace", "awsm_hip_read_frame_trace", "awsm_hip_hud_geometry_pass", "awsm_hip_hud_transparent_pass"]


class AwsmConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("abi_version", C.c_uint32), ("device", C.c_int32), ("flags", C.c_uint32), ("stream", C.c_void_p)]


class AwsmDraw(C.Structure):
    _fields_ = [("geom_meta_off", C.c_uint32), ("vis_data_off", C.c_uint32), ("tri_count", C.c_uint32), ("flags", C.c_uint32),
                ("inst_off", C.c_uint32), ("inst_count", C.c_uint32)]


class AwsmOpaqueParams(C.Structure):
    _fields_ = [("mipmap", C.c_uint32), ("has_opaque", C.c_uint32)]


class AwsmSampler(C.Structure):
    _fields_ = [("address_mode_u", C.c_uint32), ("address_mode_v", C.c_uint32), ("mag_filter", C.c_uint32), ("min_filter", C.c_uint32),
                ("mipmap_filter", C.c_uint32), ("max_anisotropy", C.c_uint32)]


class AwsmEnv(C.Structure):
    _fields_ = [("skybox_rgba", C.c_float * 4), ("prefiltered_rgb", C.c_float * 4), ("irradiance_rgb", C.c_float * 4),
                ("brdf_lut_width", C.c_uint32), ("brdf_lut_height", C.c_uint32), ("brdf_lut_rgba16f", C.c_void_p)]


class AwsmFrameStats(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("ms_transform", C.c_float), ("ms_bin", C.c_float), ("ms_raster", C.c_float), ("ms_shade", C.c_float), ("ms_total", C.c_float),
                ("triangles_in", C.c_uint32), ("triangles_binned", C.c_uint32), ("bin_entries", C.c_uint32), ("covered_pixels", C.c_uint32),
                ("bin_overflow_retries", C.c_uint32), ("ms_forward", C.c_float), ("forward_triangles", C.c_uint32), ("forward_fragment_slots", C.c_uint32),
                ("ms_shade_lean", C.c_float), ("shade_general_wavefronts", C.c_uint32), ("frames_with_dropped_bin_entries", C.c_uint32),
                ("handoff_gate_timeouts", C.c_uint32), ("geometry_cache_blocks", C.c_uint32), ("geometry_blocks", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k not in ("reserved", "struct_size")}


class AwsmHipError(RuntimeError):
    def __init__(self, code: int, where: str, text: str):
        super().__init__(f"{where} failed with status {code}: {text}")
        self.code = code


_lib = None


def load_library():
    """dlopen libawsm_hip.so.  Raises if it is missing — there is no CPU fallback for the product path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(hipcc --offload-arch=gfx950); awsm-renderer_amd has no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    lib.awsm_hip_last_error.restype = C.c_char_p
    lib.awsm_hip_last_error.argtypes = [C.c_void_p]
    lib.awsm_hip_abi_version.restype = C.c_uint32
    lib.awsm_hip_output_device_ptr.restype = C.c_void_p
    lib.awsm_hip_output_device_ptr.argtypes = [C.c_void_p]
    lib.awsm_hip_buffer_create.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    lib.awsm_hip_buffer_write.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.awsm_hip_bind_output.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.awsm_hip_bind_output_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32]
    lib.awsm_hip_geometry_pass.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.awsm_hip_opaque_pass.argtypes = [C.c_void_p, C.c_void_p]
    lib.awsm_hip_frame_end.argtypes = [C.c_void_p, C.c_void_p]
    lib.awsm_hip_frame_flush.argtypes = [C.c_void_p]
    lib.awsm_hip_resize.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.awsm_hip_set_shard_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    lib.awsm_hip_set_stage_timers.argtypes = [C.c_void_p, C.c_int]
    lib.awsm_hip_set_shard_bands.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.awsm_hip_pick.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.awsm_hip_texture_array_generate_mips.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.awsm_hip_texture_array_read_level.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.awsm_hip_texture_array_upload.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    lib.awsm_hip_sampler_set.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.awsm_hip_env_upload.argtypes = [C.c_void_p, C.c_void_p]
    lib.awsm_hip_env_cube_upload.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.awsm_hip_brdf_lut_generate.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    lib.awsm_hip_read_brdf_lut.argtypes = [C.c_void_p, C.c_void_p]
    lib.awsm_hip_read_visibility.argtypes = [C.c_void_p, C.c_void_p]
    lib.awsm_hip_visibility_digest.argtypes = [C.c_void_p, C.c_void_p]
    lib.awsm_hip_read_visibility_unpacked.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.awsm_hip_read_opaque.argtypes = [C.c_void_p, C.c_void_p]
    lib.awsm_hip_read_opaque_f32.argtypes = [C.c_void_p, C.c_void_p]
    lib.awsm_hip_read_transformed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.awsm_hip_device_info.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.awsm_hip_destroy.argtypes = [C.c_void_p]
    lib.awsm_hip_transparent_pass.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.awsm_hip_read_composite.argtypes = [C.c_void_p, C.c_void_p]
    lib.awsm_hip_read_composite_f32.argtypes = [C.c_void_p, C.c_void_p]
    lib.awsm_hip_bind_composite.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.awsm_hip_bind_opaque_source.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.awsm_hip_msaa_halo_bands.argtypes = [C.c_void_p]
    lib.awsm_hip_msaa_halo_bands.restype = C.c_uint32
    lib.awsm_hip_msaa_halo_export.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.awsm_hip_msaa_halo_bind.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.awsm_hip_read_transformed_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.awsm_hip_hud_geometry_pass.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.awsm_hip_hud_transparent_pass.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.awsm_hip_frame_trace.argtypes = [C.c_void_p, C.c_uint32]
    lib.awsm_hip_read_frame_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


class HipDevice:
    """One AwsmHipCtx: one HIP device + stream."""

    def __init__(self, device: int = 0, stream: Optional[int] = None, parity_tap: bool = False, small_bin_list: bool = False, overlap_frames: bool = False,
                 general_shade_only: bool = False, anisotropic: bool = False):
        self.lib = load_library()
        flags = (AWSM_CFG_ANISOTROPIC if anisotropic else 0) | (AWSM_CFG_PARITY_TAP if parity_tap else 0) | (AWSM_CFG_SMALL_BIN_LIST if small_bin_list else 0) | (AWSM_CFG_OVERLAP_FRAMES if overlap_frames else 0) | (AWSM_CFG_GENERAL_SHADE_ONLY if general_shade_only else 0)
        cfg = AwsmConfig(C.sizeof(AwsmConfig), self.lib.awsm_hip_abi_version(), device, flags, stream)
        ctx = C.c_void_p()
        rc = self.lib.awsm_hip_create(C.byref(cfg), C.byref(ctx))
        if rc != 0:
            raise AwsmHipError(rc, "awsm_hip_create", "no usable gfx950 device" if rc == -4 else "see status code")
        self.ctx = ctx
        self.owns = True
        self.width = self.height = 0
        self.lut_size = (0, 0)

    @classmethod
    def from_ctx(cls, ctx: int, width: int, height: int) -> "HipDevice":
        """Non-owning view over an AwsmHipCtx created elsewhere (the host layer's), for the readback helpers."""
        self = cls.__new__(cls)
        self.lib, self.ctx, self.owns = load_library(), C.c_void_p(ctx), False
        self.width, self.height, self.lut_size = width, height, (0, 0)
        return self

    def _chk(self, rc: int, where: str):
        if rc != 0:
            raise AwsmHipError(rc, where, (self.lib.awsm_hip_last_error(self.ctx) or b"").decode())

    def close(self):
        if self.ctx and self.owns:
            self.lib.awsm_hip_destroy(self.ctx)
        self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- buffers ----
    def buffer_create(self, which: int, nbytes: int):
        self._chk(self.lib.awsm_hip_buffer_create(self.ctx, which, nbytes), f"buffer_create({BUF_NAMES[which]})")

    def buffer_write(self, which: int, offset: int, data):
        arr = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        self._chk(self.lib.awsm_hip_buffer_write(self.ctx, which, offset, arr.ctypes.data_as(C.c_void_p), arr.nbytes), f"buffer_write({BUF_NAMES[which]})")

    def upload_mirrors(self, mirrors: Dict[int, bytes]):
        """create + full write of every mirror (what the reference does on the first frame / after a resize)."""
        for which, data in mirrors.items():
            n = (len(data) + 3) & ~3
            self.buffer_create(which, n)
            if len(data):
                self.buffer_write(which, 0, np.frombuffer(bytes(data) + bytes(n - len(data)), dtype=np.uint8))

    # ---- targets / environment ----
    def resize(self, width: int, height: int, msaa: int = 0):
        self._chk(self.lib.awsm_hip_resize(self.ctx, width, height, msaa), "resize")
        self.width, self.height, self.msaa = width, height, (4 if msaa == 4 else 0)

    def set_shard_rows(self, y0: int, y1: int):
        self._chk(self.lib.awsm_hip_set_shard_rows(self.ctx, y0, y1), "set_shard_rows")

    def pick(self, x: int, y: int):
        """picker.rs:55-121 -> (mesh_key u64, primitive-local triangle index) or None for background / outside the frame."""
        out = (C.c_uint32 * 4)()
        self._chk(self.lib.awsm_hip_pick(self.ctx, x, y, out), "pick")
        return ((out[1] << 32) | out[2], out[3]) if out[0] else None

    def set_stage_timers(self, enabled: bool):
        """hipEventRecord between the stages of a frame (frame_end's ms_* fields); off = no bubbles between the kernels."""
        self._chk(self.lib.awsm_hip_set_stage_timers(self.ctx, 1 if enabled else 0), "set_stage_timers")

    def set_shard_bands(self, n: int, r: int, compact_output: bool = False):
        self._chk(self.lib.awsm_hip_set_shard_bands(self.ctx, n, r, 1 if compact_output else 0), "set_shard_bands")

    def texture_array_upload(self, index: int, texels: np.ndarray, mips: int = 1):
        """Level 0 of a pool array; `mips` levels are reserved (see texture_array_generate_mips)."""
        layers, h, w, _ = texels.shape
        t = np.ascontiguousarray(texels, dtype=np.uint8)
        self._chk(self.lib.awsm_hip_texture_array_upload(self.ctx, index, w, h, layers, mips, 0, t.ctypes.data_as(C.c_void_p)), "texture_array_upload")
        self._tex_shapes = getattr(self, "_tex_shapes", {})
        self._tex_shapes[index] = (layers, h, w)

    def texture_array_generate_mips(self, index: int, kinds=None):
        layers = self._tex_shapes[index][0]
        k = None if kinds is None else (C.c_uint32 * layers)(*[int(x) for x in kinds])
        self._chk(self.lib.awsm_hip_texture_array_generate_mips(self.ctx, index, k), "texture_array_generate_mips")

    def texture_array_read_level(self, index: int, level: int) -> np.ndarray:
        layers, h, w = self._tex_shapes[index]
        out = np.zeros((layers, max(1, h >> level), max(1, w >> level), 4), dtype=np.uint8)
        self._chk(self.lib.awsm_hip_texture_array_read_level(self.ctx, index, level, out.ctypes.data_as(C.c_void_p)), "texture_array_read_level")
        return out

    def sampler_set(self, index: int, s: dict):
        smp = AwsmSampler(s.get("address_mode_u", 1), s.get("address_mode_v", 1), s.get("mag_filter", 1), s.get("min_filter", 1),
                          s.get("mipmap_filter", 1), s.get("max_anisotropy", 1))
        self._chk(self.lib.awsm_hip_sampler_set(self.ctx, index, C.byref(smp)), "sampler_set")

    def env_cube_upload(self, which: int, levels):
        """levels: [level0, level1, ...] of (6, N_l, N_l, 4) float16 (faces +X -X +Y -Y +Z -Z), or None for the uniform colour."""
        if not levels:
            self._chk(self.lib.awsm_hip_env_cube_upload(self.ctx, which, 0, 0, None), "env_cube_upload")
            return
        flat = np.ascontiguousarray(np.concatenate([np.ascontiguousarray(a, dtype=np.float16).reshape(-1) for a in levels])).view(np.uint16)
        self._chk(self.lib.awsm_hip_env_cube_upload(self.ctx, which, levels[0].shape[1], len(levels), flat.ctypes.data), "env_cube_upload")

    def env_upload(self, skybox=(0, 0, 0, 1), prefiltered=(1, 1, 1), irradiance=(1, 1, 1), lut_rgba16f: Optional[np.ndarray] = None):
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
            self.lut_size = (keep.shape[1], keep.shape[0])
        self._chk(self.lib.awsm_hip_env_upload(self.ctx, C.byref(env)), "env_upload")

    def brdf_lut_generate(self, width: int, height: int):
        self._chk(self.lib.awsm_hip_brdf_lut_generate(self.ctx, width, height), "brdf_lut_generate")
        self.lut_size = (width, height)

    def read_brdf_lut(self) -> np.ndarray:
        w, h = self.lut_size
        out = np.zeros((h, w, 2), dtype=np.uint16)
        self._chk(self.lib.awsm_hip_read_brdf_lut(self.ctx, out.ctypes.data_as(C.c_void_p)), "read_brdf_lut")
        return out

    # ---- passes ----
    @staticmethod
    def make_draws(draws: Sequence[dict]):
        arr = (AwsmDraw * max(1, len(draws)))()
        for i, d in enumerate(draws):
            arr[i] = AwsmDraw(d["geom_meta_off"], d["vis_data_off"], d["tri_count"], d["flags"], d.get("inst_off", 0), d.get("inst_count", 0))
        return arr

    def geometry_pass(self, draws, n: Optional[int] = None):
        if not isinstance(draws, C.Array):
            n = len(draws)
            draws = self.make_draws(draws)
        self._chk(self.lib.awsm_hip_geometry_pass(self.ctx, draws, n if n is not None else len(draws)), "geometry_pass")

    def opaque_pass(self, has_opaque: bool = True, mipmap: int = 0):
        p = AwsmOpaqueParams(mipmap, 1 if has_opaque else 0)
        self._chk(self.lib.awsm_hip_opaque_pass(self.ctx, C.byref(p)), "opaque_pass")

    def transparent_pass(self, draws, n: Optional[int] = None):
        """World transparent pass over the back-to-front draw list (after opaque_pass); the result is the composite image."""
        if not isinstance(draws, C.Array):
            n = len(draws)
            draws = self.make_draws(draws)
        self._chk(self.lib.awsm_hip_transparent_pass(self.ctx, draws, n if n is not None else len(draws)), "transparent_pass")

    def hud_geometry_pass(self, draws):
        """The hud meshes' visibility geometry, between geometry_pass and opaque_pass (render.rs:169-178)."""
        arr = self.make_draws(draws)
        self._chk(self.lib.awsm_hip_hud_geometry_pass(self.ctx, arr, len(draws)), "hud_geometry_pass")

    def hud_transparent_pass(self, draws):
        """The hud meshes' transparency geometry over the composite, after transparent_pass (render.rs:301-312)."""
        arr = self.make_draws(draws)
        self._chk(self.lib.awsm_hip_hud_transparent_pass(self.ctx, arr, len(draws)), "hud_transparent_pass")

    def frame_end(self) -> dict:
        st = AwsmFrameStats()
        st.struct_size = C.sizeof(AwsmFrameStats)
        self._chk(self.lib.awsm_hip_frame_end(self.ctx, C.byref(st)), "frame_end")
        return st.as_dict()

    def frame_flush(self):
        """Order everything enqueued so far (incl. overlapped opaque passes) before later work on the caller's stream."""
        self._chk(self.lib.awsm_hip_frame_flush(self.ctx), "frame_flush")

    def frame_trace(self, capacity: int):
        """Device-clock stamps per frame (geometry begins / geometry done / shading done) in a ring of `capacity` frames; 0 = off."""
        self._chk(self.lib.awsm_hip_frame_trace(self.ctx, capacity), "frame_trace")

    def read_frame_trace(self, n_frames: int):
        """-> (ms[n_frames, 3] relative to the first stamp of the oldest frame, serial of the newest frame); synchronises."""
        ticks = np.zeros((n_frames, 3), dtype=np.uint64)
        serial, rate = C.c_uint32(), C.c_uint32()
        self._chk(self.lib.awsm_hip_read_frame_trace(self.ctx, ticks.ctypes.data_as(C.c_void_p), n_frames, C.byref(serial), C.byref(rate)), "read_frame_trace")
        t0 = int(ticks[ticks > 0].min()) if (ticks > 0).any() else 0
        return (ticks.astype(np.int64) - t0) / float(rate.value), int(serial.value)

    def bind_output(self, device_ptr: Optional[int], nbytes: int = 0):
        self._chk(self.lib.awsm_hip_bind_output(self.ctx, device_ptr, nbytes), "bind_output")

    def msaa_halo_bands(self) -> int:
        return int(self.lib.awsm_hip_msaa_halo_bands(self.ctx))

    def msaa_halo_export(self, device_ptr: int, nbytes: int):
        """MSAA + bands: this rank's [bands][2][width] u64 boundary keys into device memory (to be all-gathered)."""
        self._chk(self.lib.awsm_hip_msaa_halo_export(self.ctx, device_ptr, nbytes), "msaa_halo_export")

    def msaa_halo_bind(self, device_ptr: Optional[int], nbytes: int = 0):
        """The gathered [n][bands][2][width] u64 array the next opaque pass reads its neighbours' rows from."""
        self._chk(self.lib.awsm_hip_msaa_halo_bind(self.ctx, device_ptr, nbytes), "msaa_halo_bind")

    def bind_opaque_source(self, device_ptr: Optional[int], nbytes: int = 0):
        """Sharded transparent pass: the gathered full-frame opaque image (None = the context's own output)."""
        self._chk(self.lib.awsm_hip_bind_opaque_source(self.ctx, device_ptr, nbytes), "bind_opaque_source")

    def bind_output_rows(self, device_ptr: int, nbytes: int, first_row: int):
        """Row-strip shards: device_ptr receives frame row first_row onwards."""
        self._chk(self.lib.awsm_hip_bind_output_rows(self.ctx, device_ptr, nbytes, first_row), "bind_output_rows")

    def output_device_ptr(self) -> int:
        return self.lib.awsm_hip_output_device_ptr(self.ctx)

    # ---- readback ----
    def _vis_shape(self):
        return (self.height, self.width, 4) if getattr(self, "msaa", 0) == 4 else (self.height, self.width)

    def stream_handoff(self) -> int:
        """1: the overlapped pipeline hands frames between its streams through device-side flags, 0: through hipEvents (awsm_hip.h)."""
        self.lib.awsm_hip_stream_handoff.argtypes = [C.c_void_p]
        return int(self.lib.awsm_hip_stream_handoff(self.ctx))

    def read_gbuffer(self) -> np.ndarray:
        """(height, width, 6) f32: the reconstructed G-buffer texel per pixel (packed normal / tangent, barycentric), zeros where nothing was hit."""
        out = np.zeros((self.height, self.width, 6), dtype=np.float32)
        self.lib.awsm_hip_read_gbuffer.argtypes = [C.c_void_p, C.c_void_p]
        self._chk(self.lib.awsm_hip_read_gbuffer(self.ctx, out.ctypes.data_as(C.c_void_p)), "read_gbuffer")
        return out

    def read_visibility(self) -> np.ndarray:
        """Packed keys [H, W] (with MSAA x4: [H, W, 4], sample-minor)."""
        out = np.zeros(self._vis_shape(), dtype=np.uint64)
        self._chk(self.lib.awsm_hip_read_visibility(self.ctx, out.ctypes.data_as(C.c_void_p)), "read_visibility")
        return out

    def visibility_digest(self):
        """(sum key_i * (2 i + 1) mod 2^64, xor rotl(key_i, i mod 64)) of the last geometry pass's keys, computed on the device."""
        out = np.zeros(2, dtype=np.uint64)
        self._chk(self.lib.awsm_hip_visibility_digest(self.ctx, out.ctypes.data_as(C.c_void_p)), "visibility_digest")
        return int(out[0]), int(out[1])

    def read_visibility_unpacked(self):
        tri = np.zeros(self._vis_shape(), dtype=np.uint32)
        meta = np.zeros(self._vis_shape(), dtype=np.uint32)
        depth = np.zeros(self._vis_shape(), dtype=np.float32)
        self._chk(self.lib.awsm_hip_read_visibility_unpacked(self.ctx, tri.ctypes.data_as(C.c_void_p), meta.ctypes.data_as(C.c_void_p),
                                                             depth.ctypes.data_as(C.c_void_p)), "read_visibility_unpacked")
        return tri, meta, depth

    def read_opaque(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), dtype=np.uint16)
        self._chk(self.lib.awsm_hip_read_opaque(self.ctx, out.ctypes.data_as(C.c_void_p)), "read_opaque")
        return out

    def read_opaque_f32(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        self._chk(self.lib.awsm_hip_read_opaque_f32(self.ctx, out.ctypes.data_as(C.c_void_p)), "read_opaque_f32")
        return out

    def read_composite(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), dtype=np.uint16)
        self._chk(self.lib.awsm_hip_read_composite(self.ctx, out.ctypes.data_as(C.c_void_p)), "read_composite")
        return out

    def read_composite_f32(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        self._chk(self.lib.awsm_hip_read_composite_f32(self.ctx, out.ctypes.data_as(C.c_void_p)), "read_composite_f32")
        return out

    def bind_composite(self, device_ptr: Optional[int], nbytes: int = 0):
        self._chk(self.lib.awsm_hip_bind_composite(self.ctx, device_ptr, nbytes), "bind_composite")

    def read_transformed_forward(self, n_vertices: int):
        clip = np.zeros((max(1, n_vertices), 4), dtype=np.float32)
        nt = np.zeros((max(1, n_vertices), 8), dtype=np.float32)
        wpos = np.zeros((max(1, n_vertices), 4), dtype=np.float32)
        self._chk(self.lib.awsm_hip_read_transformed_forward(self.ctx, clip.ctypes.data_as(C.c_void_p), nt.ctypes.data_as(C.c_void_p),
                                                             wpos.ctypes.data_as(C.c_void_p), n_vertices), "read_transformed_forward")
        return clip, nt, wpos

    def read_transformed(self, n_vertices: int):
        clip = np.zeros((max(1, n_vertices), 4), dtype=np.float32)
        nt = np.zeros((max(1, n_vertices), 8), dtype=np.float32)
        self._chk(self.lib.awsm_hip_read_transformed(self.ctx, clip.ctypes.data_as(C.c_void_p), nt.ctypes.data_as(C.c_void_p), n_vertices), "read_transformed")
        return clip, nt

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cus, mem = C.c_uint32(), C.c_uint64()
        self._chk(self.lib.awsm_hip_device_info(self.ctx, name, 256, C.byref(cus), C.byref(mem)), "device_info")
        return {"name": name.value.decode(), "cu_count": cus.value, "hbm_bytes": mem.value}
