"""Byte-level known answers for the packers, derived BY HAND from the reference's Rust (not from oracle/host_mirror.py, which a second reader wrote from the
same sources as the C++ host): every word below is written out as a literal with the file:line of the `push_u32` / `write` / `extend_from_slice` that
produces it, and then compared with BOTH implementations — the C++ host layer (through the recording mock of the C-ABI: what would reach the device)
and the Python mirror the GPU parity tests feed the oracle with.  (VERDICT r3 "next" #7.  It stays "unpinned by the reference": the reference holds
no such vector; this removes the single-reader risk for the bytes.)

Paths are relative to /root/reference/crates/renderer/src/.
Inputs are this repo's Box (scenes.box_scene: face +Z first, corners (-.5,-.5,.5) (.5,-.5,.5) (.5,.5,.5) (-.5,.5,.5), normal (0,0,1), triangle 0 = indices
0 1 2, no TANGENT, no TEXCOORD; root node + mesh node; one material baseColorFactor (0.8, 0, 0, 1), metallicFactor 0) and a variant with TEXCOORD_0, one
16x16 base-colour texture and KHR_materials_ior.
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from awsm_renderer_amd import host as H
from awsm_renderer_amd import scenes
from awsm_renderer_amd.scene_desc import TextureRef
from oracle import scene_model as sm
from tests import helpers

MOCK_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mock")
MOCK = os.path.join(MOCK_DIR, "libmock_backend.so")

F_HALF, F_MHALF, F_ONE, F_08, F_145 = 0x3F000000, 0xBF000000, 0x3F800000, 0x3F4CCCCD, 0x3FB9999A      # 0.5, -0.5, 1.0, 0.8f, 1.45f as IEEE binary32


@pytest.fixture(scope="module")
def mock_lib():
    src = os.path.join(MOCK_DIR, "mock_backend.c")
    if not os.path.exists(MOCK) or os.path.getmtime(src) > os.path.getmtime(MOCK):
        subprocess.check_call(["gcc", "-O1", "-std=c11", "-fPIC", "-shared", "-o", MOCK, src])
    return MOCK


def words(*w):
    return struct.pack("<%dI" % len(w), *w)


def both(scene, mock_lib):
    """-> (C++ host mirrors, Python mirror bytes, the mesh key the host handed out, the draw list)"""
    r = H.Renderer(scene, backend_path=mock_lib, lut_rgba16f=np.zeros((4, 4, 4), dtype=np.uint16))
    r.render()
    model = helpers.build_model(scene)
    host_m = {w: r.host.mirror(w) for w in (sm.BUF_VIS_GEOM_DATA, sm.BUF_GEOM_META, sm.BUF_MATERIAL_META, sm.BUF_MATERIALS)}
    py_m = {w: bytes(model.mirrors()[w]) for w in host_m}
    key = r.keys.mesh_keys[0]
    draws = r.host.draw_list()
    r.close()
    return host_m, py_m, key, draws


def test_box_triangle_0_metas_and_material_words(mock_lib):
    host_m, py_m, mesh_key, draws = both(scenes.box_scene(64, 64), mock_lib)
    assert len(draws) == 1 and draws[0]["vis_data_off"] == 0 and draws[0]["geom_meta_off"] == 0 and draws[0]["tri_count"] == 12

    # ---- gltf/buffers/mesh/visibility.rs:35-165: 56 bytes per exploded vertex, three per triangle, FrontFace::Ccw keeps the order (:108-116) ----
    def vertex(px, py, pz, tri, b0, b1, orig):
        return words(px, py, pz,            # :137-139 position
                     tri,                   # :142 triangle_index as u32
                     b0, b1,                # :145-146 barycentric: BARYCENTRICS[corner] (:41-45) = (1,0) (0,1) (0,0)
                     0, 0, F_ONE,           # :149-151 normal (0, 0, 1)
                     0, 0, 0, F_ONE,        # :154-157 tangent: none in the mesh -> the default [0, 0, 0, 1] (:128-132)
                     orig)                  # :160 original vertex index
    tri0 = vertex(F_MHALF, F_MHALF, F_HALF, 0, F_ONE, 0, 0) + vertex(F_HALF, F_MHALF, F_HALF, 0, 0, F_ONE, 1) + vertex(F_HALF, F_HALF, F_HALF, 0, 0, 0, 2)
    assert len(tri0) == 168
    for name, m in (("C++ host", host_m), ("Python mirror", py_m)):
        assert m[sm.BUF_VIS_GEOM_DATA][:168] == tri0, name
        # triangle 1 = indices 0 2 3 (:106 `triangle_index` counts triangles, :160 keeps the ORIGINAL index): its second corner is vertex 2
        assert m[sm.BUF_VIS_GEOM_DATA][168 + 56: 168 + 112] == vertex(F_HALF, F_HALF, F_HALF, 1, 0, F_ONE, 2), name

    # ---- meshes/meta/geometry_meta.rs:44-113: 40 bytes ----
    hi, lo = (mesh_key >> 32) & 0xFFFFFFFF, mesh_key & 0xFFFFFFFF      # :69-73 KeyData::as_ffi split; the key is the one the host API returned
    geom = words(hi, lo,                    # :76-77
                 0, 0, 0,                   # :86-88 no morph
                 0, 0, 0,                   # :97-99 no skin
                 64,                        # :103 transforms.buffer_offset: the mesh hangs off the SECOND transform (root, then the mesh node), 64-byte slots (transforms.rs:68-72)
                 0)                         # :106-110 material_meta_buffers.offset(mesh_key): first mesh, slot 0
    # ---- meshes/meta/material_meta.rs:96-185: 68 bytes ----
    matm = words(hi, lo,                    # :130-131
                 0, 0, 0, 0,                # :148-151 no material morph (four words, although the comment says 20 bytes)
                 0,                         # :155 materials.buffer_offset: first material of the storage buffer
                 64,                        # :158 transform offset (as above)
                 36,                        # :160 normal-matrix offset: second slot of a 36-byte-stride buffer (transforms.rs:72,100-105)
                 0, 0,                      # :163-164 custom attribute indices / data offsets: first allocations
                 0,                         # :167 vertex_attribute_stride: the Box has no custom attribute
                 0,                         # :170-171 uv_sets_index
                 0, 0,                      # :174-176 uv sets, colour sets
                 0,                         # :179 visibility_geometry_data_offset: first allocation
                 0)                         # :182 is_hud
    assert len(geom) == 40 and len(matm) == 68
    # ---- materials/pbr.rs:258-589 through materials/writer.rs:64-99 ----
    skip = (0, 0, 0, 0, 0)                  # writer.rs:96-98 Value::SkipTexture = 20 zero bytes
    mat = words(1,                          # pbr.rs:266 MaterialShaderId::Pbr = 1 (materials.rs:75)
                0,                          # :268 alpha mode Opaque = 0 (materials.rs:268)
                0,                          # :269 alpha cutoff: none -> 0.0f32
                *skip,                      # :271-275 no base-colour texture
                F_08, 0, 0, F_ONE,          # :276-279 baseColorFactor (0.8, 0, 0, 1)
                *skip, 0, F_ONE,            # :281-287 no metallic-roughness texture; metallicFactor 0, roughnessFactor 1 (glTF default)
                *skip, F_ONE,               # :289-294 no normal texture; scale 1
                *skip, F_ONE,               # :296-301 no occlusion texture; strength 1
                *skip, 0, 0, 0,             # :303-310 no emissive texture; factor (0, 0, 0)
                0,                          # :312 debug bitmask
                *([0] * 12))                # :350-355 the twelve feature indices: no optional block -> all zero (:580-586 rewrites them in place)
    assert len(mat) == 52 * 4
    for name, m in (("C++ host", host_m), ("Python mirror", py_m)):
        assert m[sm.BUF_GEOM_META][:40] == geom, name
        assert m[sm.BUF_MATERIAL_META][:68] == matm, name
        assert m[sm.BUF_MATERIALS][:len(mat)] == mat, name


def test_textured_material_with_one_optional_block(mock_lib):
    sc = scenes.box_scene(64, 64)
    prim = sc.nodes[1].primitives[0]
    prim.uvs = [np.tile(np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float32), (6, 1))]
    sc.textures = [np.full((16, 16, 4), 200, dtype=np.uint8)]
    m = sc.materials[0]
    m.base_color_tex = TextureRef(0)
    m.ior = 1.45
    host_m, py_m, mesh_key, draws = both(sc, mock_lib)
    hi, lo = (mesh_key >> 32) & 0xFFFFFFFF, mesh_key & 0xFFFFFFFF
    skip = (0, 0, 0, 0, 0)
    tex = ((16 << 16) | 16,                 # writer.rs:136-144 size = height << 16 | width
           (0 << 12) | 0,                   # :146-153 layer << 12 | array index: the pool's first array, first layer
           (0 << 8) | 0,                    # :155-162 sampler index << 8 | uv set: the only sampler of the sorted set, TEXCOORD_0
           3 | (1 << 8) | (1 << 16),        # :164-184 flags (bit 0 present, bit 1 mipmaps: pool arrays are created with mipmap = true, renderer-core texture_pool.rs:172),
                                            #          address mode u = v = Repeat = 1 (writer.rs:56-63)
           0)                               # :187-188 texture transform offset: the identity slot, 0 (textures.rs:311-320)
    mat = words(1, 0, 0,                    # pbr.rs:266-269
                *tex,                       # :271-272
                F_08, 0, 0, F_ONE,          # :276-279
                *skip, 0, F_ONE,            # :281-287 (metallicFactor 0 from the Box material)
                *skip, F_ONE, *skip, F_ONE, *skip, 0, 0, 0,
                0,                          # :312 debug
                0, 0, 51, 0, 0, 0, 0, 0, 0, 0, 0, 0,      # :314-355 feature indices; ior is the third (:319,:335): current_index (:358-362) = 212 / 4 - 1 = 51 when the block is written (:374-377)
                F_145)                      # :376 the ior itself, word 52 (= index 51 behind the shader id)
    assert len(mat) == 53 * 4
    matm = words(hi, lo, 0, 0, 0, 0, 0, 64, 36, 0, 0,
                 8,                         # material_meta.rs:167 stride: one TEXCOORD set = 8 bytes
                 0,                         # :170-171 uv_sets_index: no COLOR_n in front of TEXCOORD_0
                 1, 0,                      # :174-176 one uv set, no colour set
                 0, 0)
    for name, mm in (("C++ host", host_m), ("Python mirror", py_m)):
        assert mm[sm.BUF_MATERIALS][:len(mat)] == mat, name
        assert mm[sm.BUF_MATERIAL_META][:68] == matm, name
