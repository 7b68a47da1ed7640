"""CPU tests of the C++ host layer through a recording mock of the C-ABI (tests/mock/mock_backend.c): the bytes that
reach the device, the write plan per frame, the draw list and the call order, all against the Python model of the
reference's host code (oracle/scene_model.py).  No GPU, no rendering."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from awsm_renderer_amd import host as H
from awsm_renderer_amd import scenes
from oracle import scene_model as sm
from tests import helpers

MOCK_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mock")
MOCK = os.path.join(MOCK_DIR, "libmock_backend.so")
OPS = {1: "create", 2: "write", 3: "resize", 4: "texture", 5: "sampler", 6: "env", 7: "geometry", 8: "opaque", 9: "frame_end", 10: "lut", 11: "shard", 12: "shard_bands", 13: "pick", 14: "generate_mips", 15: "transparent", 16: "stage_timers", 17: "hud_geometry", 18: "hud_transparent"}


@pytest.fixture(scope="module")
def mock():
    src = os.path.join(MOCK_DIR, "mock_backend.c")
    if not os.path.exists(MOCK) or os.path.getmtime(src) > os.path.getmtime(MOCK):
        subprocess.check_call(["gcc", "-O1", "-std=c11", "-fPIC", "-shared", "-o", MOCK, src])
    lib = C.CDLL(MOCK)
    lib.mock_log_count.restype = C.c_size_t
    lib.mock_log_count.argtypes = [C.c_void_p]
    lib.mock_log_get.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.mock_log_clear.argtypes = [C.c_void_p]
    lib.mock_buffer.restype = C.c_void_p
    lib.mock_buffer.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
    return lib


def log_of(mock, ctx):
    out = []
    for i in range(mock.mock_log_count(ctx)):
        op, which, a, b = C.c_int(), C.c_int(), C.c_uint64(), C.c_uint64()
        mock.mock_log_get(ctx, i, C.byref(op), C.byref(which), C.byref(a), C.byref(b))
        out.append((OPS[op.value], which.value, a.value, b.value))
    return out


def device_bytes(mock, ctx, which):
    n = C.c_size_t()
    p = mock.mock_buffer(ctx, which, C.byref(n))
    return C.string_at(p, n.value) if p else None


SCENES = {
    "box": lambda: scenes.box_scene(64, 64),
    "helmet": lambda: scenes.helmet_scene(64, 64, segments=16, rings=12, tex_size=16),
    "skinned_morph": lambda: scenes.skinned_morph_scene(64, 64, around=8, along=12, tex_size=16),
    "atrium": lambda: scenes.atrium_scene(96, 64, detail=0.125, tex_scale=1 / 64),
    "instanced": lambda: scenes.instanced_scene(96, 64),
}


@pytest.mark.parametrize("name", list(SCENES))
def test_mirrors_and_draw_list_match_the_model(name, mock):
    scene = SCENES[name]()
    r = H.Renderer(scene, backend_path=MOCK, lut_rgba16f=np.zeros((4, 4, 4), dtype=np.uint16))
    r.render()
    model = helpers.build_model(scene)
    ctx = r.host.device_ctx
    for which, data in model.mirrors().items():
        if which in (sm.BUF_LIGHTS, sm.BUF_LIGHTS_INFO):
            continue
        assert r.host.mirror(which) == bytes(data), f"mirror {which} differs"
        if which == sm.BUF_INSTANCES and name != "instanced":
            assert device_bytes(mock, ctx, which) is None      # written only once instancing is used (instances.rs:203-242: transform_gpu_dirty)
        elif which != sm.BUF_VIS_GEOM_INDEX:     # identity indices are never uploaded (redundant for a SW rasteriser)
            assert device_bytes(mock, ctx, which) == bytes(data), f"device copy of {which} differs from its mirror"
    assert device_bytes(mock, ctx, sm.BUF_LIGHTS)[:len(model.lights_bytes())] == model.lights_bytes()
    assert device_bytes(mock, ctx, sm.BUF_LIGHTS_INFO) == model.lights_info_bytes()
    want = [{k: v for k, v in d.items() if k != "mesh_key"} for d in model.collect_draws()]
    assert r.host.draw_list() == want
    r.close()


def _all_blocks_scene():
    """A box scene whose materials carry every optional block of the PBR word stream (pbr.rs:364-581), the four the shaders do not read
    (diffuse transmission, dispersion, anisotropy, iridescence) included, alone and all together."""
    import dataclasses
    from awsm_renderer_amd.scene_desc import MaterialDesc, NodeDesc, TextureRef
    sc = scenes.helmet_scene(64, 64, segments=8, rings=6, tex_size=16)
    T = lambda i: TextureRef(i)   # noqa: E731
    prim = sc.nodes[-1].primitives[0]
    extra = [
        MaterialDesc(diffuse_transmission={"tex": T(0), "factor": 0.4, "color_tex": T(1), "color_factor": (0.9, 0.5, 0.2)}),
        MaterialDesc(dispersion=0.35),
        MaterialDesc(anisotropy={"tex": T(2), "strength": 0.7, "rotation": 1.2}),
        MaterialDesc(iridescence={"tex": T(3), "factor": 0.8, "ior": 1.33, "thickness_tex": T(4), "thickness_min": 120.0, "thickness_max": 380.0}),
        MaterialDesc(vertex_color_set=None, emissive_strength=2.0, ior=1.45, specular={"factor": 0.8, "color_factor": (1, 0.9, 0.8)},
                     transmission={"factor": 0.0}, diffuse_transmission={"factor": 0.1}, volume={"thickness_factor": 0.2, "attenuation_distance": 3.0},
                     clearcoat={"factor": 0.5, "roughness_factor": 0.1}, sheen={"roughness_factor": 0.4, "color_factor": (0.1, 0.2, 0.3)},
                     dispersion=0.1, anisotropy={"strength": 0.2, "rotation": 0.3}, iridescence={"factor": 0.3}),
    ]
    for k, m in enumerate(extra):
        sc.materials.append(m)
        sc.nodes.append(NodeDesc(parent=0, translation=(0.1 * k, 0.0, 0.0), primitives=[dataclasses.replace(prim, material=len(sc.materials) - 1)]))
    return sc


def test_every_optional_pbr_block_reaches_the_materials_mirror(mock):
    """materials/pbr.rs:258-589: the Materials mirror of the C++ host equals the model's word stream, feature indices and block contents,
    for the blocks the shaders read and for the four they do not (diffuse_transmission 14 words, dispersion 1, anisotropy 7, iridescence 14)."""
    import struct
    scene = _all_blocks_scene()
    r = H.Renderer(scene, backend_path=MOCK, lut_rgba16f=np.zeros((4, 4, 4), dtype=np.uint16))
    r.render()
    model = helpers.build_model(scene)
    got, want = r.host.mirror(sm.BUF_MATERIALS), bytes(model.mirrors()[sm.BUF_MATERIALS])
    assert got == want
    assert device_bytes(mock, r.host.device_ctx, sm.BUF_MATERIALS) == want
    # the last material has all twelve feature indices set, in stream order, with the block sizes of the reference
    packer = sm.MaterialPacker(model.pool, scene.samplers, model.tex_transforms) if hasattr(model, "pool") else None
    words = None
    for off in range(0, len(want), 256):
        w = struct.unpack_from("<64I", want, off)
        if w[0] == 1 and all(w[40 + i] != 0 for i in range(12) if i != 0):       # PBR with every block but vertex colour
            words = w
    assert words is not None
    fi = list(words[40:52])
    sizes = {1: 1, 2: 1, 3: 14, 4: 6, 5: 14, 6: 10, 7: 18, 8: 14, 9: 1, 10: 7}
    for i, n in sizes.items():
        assert fi[i + 1] - fi[i] == n, (i, fi)
    r.close()


def test_write_gpu_order_and_frame_sequence(mock):
    """render.rs:73-97 then geometry -> opaque -> submit."""
    r = H.Renderer(scenes.skinned_morph_scene(64, 64, around=8, along=12, tex_size=16), backend_path=MOCK, lut_rgba16f=np.zeros((4, 4, 4), dtype=np.uint16))
    mock.mock_log_clear(r.host.device_ctx)
    r.render()
    log = log_of(mock, r.host.device_ctx)
    creates = [w for op, w, _, _ in log if op == "create"]
    assert creates == [sm.BUF_TRANSFORMS, sm.BUF_NORMAL_MATS, sm.BUF_MATERIALS, sm.BUF_LIGHTS, sm.BUF_LIGHTS_INFO, sm.BUF_SKIN_MATRICES,
                       sm.BUF_SKIN_INDEX_WEIGHTS, sm.BUF_MORPH_WEIGHTS, sm.BUF_MORPH_VALUES, sm.BUF_GEOM_META, sm.BUF_MATERIAL_META,
                       sm.BUF_TEXTURE_TRANSFORMS, sm.BUF_VIS_GEOM_DATA, sm.BUF_TRANSPARENCY_GEOM_DATA, sm.BUF_ATTR_DATA, sm.BUF_ATTR_INDEX, sm.BUF_CAMERA]   # meshes.rs:1240-1300
    tail = [op for op, _, _, _ in log if op in ("geometry", "opaque", "transparent", "frame_end")]
    assert tail == ["geometry", "opaque", "frame_end"]          # no transparent mesh in the scene: the composite is the opaque image
    assert [op for op, _, _, _ in log].index("geometry") > max(i for i, (op, _, _, _) in enumerate(log) if op in ("create", "write", "texture"))
    r.close()


def test_transparent_meshes_get_transparency_geometry_and_their_own_pass(mock):
    """gltf/buffers/mesh.rs:33-57 (visibility XOR transparency geometry by material), renderable.rs:78-90,131-135 (own list, back to
    front), render.rs:224-297 (the pass after the opaque pass)."""
    scene = scenes.transparent_scene(96, 64, tex_size=16)
    model = helpers.build_model(scene)
    r = H.Renderer(scene, backend_path=MOCK, lut_rgba16f=np.zeros((4, 4, 4), dtype=np.uint16))
    ctx = r.host.device_ctx
    mock.mock_log_clear(ctx)
    r.render()
    for which in (sm.BUF_MATERIALS, sm.BUF_VIS_GEOM_DATA, sm.BUF_TRANSPARENCY_GEOM_DATA, sm.BUF_ATTR_INDEX, sm.BUF_ATTR_DATA, sm.BUF_MATERIAL_META, sm.BUF_GEOM_META,
                  sm.BUF_MORPH_VALUES, sm.BUF_INSTANCES):
        assert r.host.mirror(which) == bytes(model.mirrors()[which]), f"mirror {which} differs"
        assert device_bytes(mock, ctx, which) == bytes(model.mirrors()[which])
    strip = lambda ds: [{k: v for k, v in d.items() if k != "mesh_key"} for d in ds]   # noqa: E731
    assert r.host.draw_list() == strip(model.collect_draws())
    want_tr = strip(model.collect_transparent_draws())
    assert len(want_tr) >= 10 and r.host.transparent_draw_list() == want_tr
    log = log_of(mock, ctx)
    assert [op for op, _, _, _ in log if op in ("geometry", "opaque", "transparent", "frame_end")] == ["geometry", "opaque", "transparent", "frame_end"]
    assert [a for op, _, a, _ in log if op == "transparent"] == [len(want_tr)]
    r.close()


def test_hud_meshes_carry_both_geometries_and_get_the_two_hud_passes(mock):
    """Mesh.hud (meshes/mesh.rs:28): both geometries (gltf/buffers/mesh.rs:37-39), is_hud in the MaterialMeshMeta (material_meta.rs:181-182), a list of
    its own sorted back to front (renderable.rs:77-90), drawn by the HUD geometry pass between the world's geometry and opaque passes and by the HUD
    transparent pass after the world's (render.rs:169-178,301-312)."""
    scene = scenes.hud_scene(96, 64, tex_size=16)
    model = helpers.build_model(scene)
    r = H.Renderer(scene, backend_path=MOCK, lut_rgba16f=np.zeros((4, 4, 4), dtype=np.uint16))
    ctx = r.host.device_ctx
    mock.mock_log_clear(ctx)
    r.render()
    for which in (sm.BUF_VIS_GEOM_DATA, sm.BUF_TRANSPARENCY_GEOM_DATA, sm.BUF_ATTR_INDEX, sm.BUF_ATTR_DATA, sm.BUF_MATERIAL_META, sm.BUF_GEOM_META):
        assert r.host.mirror(which) == bytes(model.mirrors()[which]), f"mirror {which} differs"
        assert device_bytes(mock, ctx, which) == bytes(model.mirrors()[which])
    strip = lambda ds: [{k: v for k, v in d.items() if k != "mesh_key"} for d in ds]   # noqa: E731
    assert r.host.draw_list() == strip(model.collect_draws())
    assert r.host.transparent_draw_list() == strip(model.collect_transparent_draws())
    hud_g, hud_t = r.host.hud_draw_lists()
    assert len(hud_g) == 4 and hud_g == strip(model.hud_geometry_draws) and hud_t == strip(model.hud_transparent_draws)
    assert all(g["geom_meta_off"] == t["geom_meta_off"] for g, t in zip(hud_g, hud_t))       # the same meshes through both passes
    mm = model.mirrors()[sm.BUF_MATERIAL_META]
    gm = model.mirrors()[sm.BUF_GEOM_META]
    for d in hud_g:      # geometry meta -> material meta offset -> is_hud word
        mmo = int.from_bytes(gm[d["geom_meta_off"] + 36:d["geom_meta_off"] + 40], "little")
        assert int.from_bytes(mm[mmo + 64:mmo + 68], "little") == 1
    log = log_of(mock, ctx)
    order = [op for op, _, _, _ in log if op in ("geometry", "hud_geometry", "opaque", "transparent", "hud_transparent", "frame_end")]
    assert order == ["geometry", "hud_geometry", "opaque", "transparent", "hud_transparent", "frame_end"]
    assert [a for op, _, a, _ in log if op in ("hud_geometry", "hud_transparent")] == [4, 4]
    r.close()


def test_dirty_upload_semantics_across_frames(mock):
    scene = scenes.atrium_scene(96, 64, detail=0.125, tex_scale=1 / 64)
    r = H.Renderer(scene, backend_path=MOCK, lut_rgba16f=np.zeros((4, 4, 4), dtype=np.uint16))
    ctx = r.host.device_ctx
    r.render()
    first = r.host.upload_bytes_last_frame()
    assert first > 128 * 1024 * 1024          # first frame: every mirror is created at its pow2 capacity and fully written

    # nothing changed -> nothing is written
    mock.mock_log_clear(ctx)
    r.render()
    assert r.host.upload_bytes_last_frame() == 0
    assert [x for x in log_of(mock, ctx) if x[0] in ("create", "write")] == []

    # update_camera -> exactly one 512-byte write
    mock.mock_log_clear(ctx)
    r.host.camera_update(scene.view, scene.proj, scene.camera_position)
    r.render()
    assert [x for x in log_of(mock, ctx) if x[0] in ("create", "write")] == [("write", sm.BUF_CAMERA, 0, 512)]

    # move one node many times, one upload: its 64-B world matrix + 36-B normal matrix slots (+ children, none here)
    key = r.keys.node_keys[5]
    for i in range(10):
        r.host.transform_set_local(key, (0.1 * i, 0.0, 0.0), (0, 0, 0, 1), (1, 1, 1))
    r.host.update_transforms()
    mock.mock_log_clear(ctx)
    r.render()
    writes = [x for x in log_of(mock, ctx) if x[0] in ("create", "write")]
    t_off = None
    for op, which, a, b in writes:
        if which == sm.BUF_TRANSFORMS:
            t_off = a
            assert b == 64
        elif which == sm.BUF_NORMAL_MATS:
            assert b == 36
        else:
            raise AssertionError(f"unexpected upload {op} {which} {a} {b}")
    assert t_off is not None and len(writes) == 2
    # ... and the device copy equals the mirror
    assert device_bytes(mock, ctx, sm.BUF_TRANSFORMS) == r.host.mirror(sm.BUF_TRANSFORMS)
    w = r.host.transform_world(key)
    assert np.allclose(w[3][:3], [0.9, 0.0, 0.0])
    r.close()


def test_material_update_rewrites_only_its_block(mock):
    scene = scenes.helmet_scene(64, 64, segments=16, rings=12, tex_size=16)
    r = H.Renderer(scene, backend_path=MOCK, lut_rgba16f=np.zeros((4, 4, 4), dtype=np.uint16))
    ctx = r.host.device_ctx
    r.render()
    mk = r.keys.material_keys[0]
    m = scene.materials[0]
    m.roughness_factor = 0.25
    r.host.material_update(mk, H.material_struct(m, r.host, {}))
    mock.mock_log_clear(ctx)
    r.render()
    writes = [x for x in log_of(mock, ctx) if x[0] in ("create", "write")]
    assert writes == [("write", sm.BUF_MATERIALS, 0, 256)]          # in place, whole 256-B buddy block marked dirty
    mat = np.frombuffer(r.host.mirror(sm.BUF_MATERIALS), dtype=np.float32)
    assert mat[1 + 17] == np.float32(0.25)                          # header word 17 = roughness_factor (pbr_material.wgsl:141)
    r.close()


def test_growth_recreates_the_device_buffer_and_rewrites_everything(mock):
    scene = scenes.box_scene(32, 32)
    r = H.Renderer(scene, backend_path=MOCK, lut_rgba16f=np.zeros((4, 4, 4), dtype=np.uint16))
    ctx = r.host.device_ctx
    r.render()
    # 40 more transforms overflow the 32-slot initial capacity: max(required, cap) * 2 growth, new buffer, full upload
    for i in range(40):
        r.host.transform_insert((i, 0, 0), (0, 0, 0, 1), (1, 1, 1))
    r.host.update_transforms()
    mock.mock_log_clear(ctx)
    r.render()
    log = [x for x in log_of(mock, ctx) if x[1] == sm.BUF_TRANSFORMS and x[0] in ("create", "write")]
    assert log[0][0] == "create" and log[1] == ("write", sm.BUF_TRANSFORMS, 0, log[0][2])
    assert log[0][2] == len(r.host.mirror(sm.BUF_TRANSFORMS)) == 66 * 64     # 2 + 31 live slots -> required 33 -> 66 slots
    r.close()


def test_mesh_remove_frees_and_reuses_blocks(mock):
    scene = scenes.atrium_scene(96, 64, detail=0.125, tex_scale=1 / 64)
    r = H.Renderer(scene, backend_path=MOCK, lut_rgba16f=np.zeros((4, 4, 4), dtype=np.uint16))
    r.render()
    n0 = len(r.host.draw_list())
    victim = r.keys.mesh_keys[10]
    r.host.mesh_remove(victim)
    assert len(r.host.draw_list()) in (n0, n0 - 1)
    r.render()
    # re-inserting the primitive reuses the freed MeshKey slot (slotmap: same idx, version + 2) and a free buddy block
    prim = [p for n in scene.nodes for p in n.primitives][10]
    new_key = r.host.mesh_insert(prim, r.keys.node_keys[11], r.keys.material_keys[prim.material])
    assert (new_key & 0xFFFFFFFF) == (victim & 0xFFFFFFFF) and (new_key >> 32) == (victim >> 32) + 2
    r.host.update_transforms()
    draws = r.host.draw_list()
    assert len(draws) == n0
    offs = sorted((d["vis_data_off"], d["tri_count"] * 168) for d in draws)
    for (o0, s0), (o1, _) in zip(offs, offs[1:]):
        assert o0 + s0 <= o1, "visibility-geometry blocks overlap"
    r.render()
    for which in (sm.BUF_VIS_GEOM_DATA, sm.BUF_ATTR_DATA, sm.BUF_ATTR_INDEX, sm.BUF_GEOM_META, sm.BUF_MATERIAL_META):
        assert device_bytes(mock, r.host.device_ctx, which) == r.host.mirror(which)
    r.close()


def test_missing_backend_fails_loudly(tmp_path):
    with pytest.raises(FileNotFoundError):
        H.Host(backend_path=str(tmp_path / "nope.so"))
    bogus = tmp_path / "libbogus.so"
    src = tmp_path / "b.c"
    src.write_text("int not_the_abi(void){return 0;}\n")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(bogus), str(src)])
    with pytest.raises(H.HostError):
        H.Host(backend_path=str(bogus))          # library loads but lacks the awsm_hip_* symbols


def test_pick_and_band_sharding_pass_through(mock):
    """awsm_host_pick / awsm_host_set_shard_bands reach the backend with their arguments; a miss is None (PickResult::Miss)."""
    r = H.Renderer(scenes.box_scene(64, 64), backend_path=MOCK, lut_rgba16f=np.zeros((4, 4, 4), dtype=np.uint16))
    r.render()
    mock.mock_log_clear(r.host.device_ctx)
    assert r.host.pick(12, 34) is None
    r.host.set_shard_bands(4, 3, compact_output=True)
    r.host.set_render_timings(False)          # AwsmRendererLogging.render_timings -> awsm_hip_set_stage_timers
    r.host.set_render_timings(True)
    log = log_of(mock, r.host.device_ctx)
    assert ("pick", 0, 12, 34) in log and ("shard_bands", 1, 4, 3) in log
    assert [e for e in log if e[0] == "stage_timers"] == [("stage_timers", 0, 0, 0), ("stage_timers", 0, 1, 0)]
    r.close()
