"""glTF ingest: scenes written as .glb / .gltf by awsm_renderer_amd.gltf_export and read back by the native reader
(awsm_host_load_gltf) must populate the host exactly as the same SceneDesc does through the key API — same mirrors, same draw
lists — plus the reader's own conversions (normalised integer attributes, u8/u16 indices, strips and fans, matrix nodes,
generated normals and tangents, data URIs, external files)."""
import base64
import json
import os
import struct

import numpy as np
import pytest

from awsm_renderer_amd import gltf_export, scenes
from awsm_renderer_amd import host as H
from awsm_renderer_amd.scene_desc import MaterialDesc, NodeDesc, PrimitiveDesc, SceneDesc, TextureRef
from oracle import scene_model as sm
from tests import helpers

MOCK = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mock", "libmock_backend.so")

MIRRORS = (sm.BUF_TRANSFORMS, sm.BUF_NORMAL_MATS, sm.BUF_MATERIALS, sm.BUF_SKIN_MATRICES, sm.BUF_SKIN_INDEX_WEIGHTS, sm.BUF_MORPH_WEIGHTS, sm.BUF_MORPH_VALUES,
           sm.BUF_GEOM_META, sm.BUF_MATERIAL_META, sm.BUF_VIS_GEOM_DATA, sm.BUF_VIS_GEOM_INDEX, sm.BUF_ATTR_DATA, sm.BUF_ATTR_INDEX, sm.BUF_TEXTURE_TRANSFORMS,
           sm.BUF_INSTANCES, sm.BUF_TRANSPARENCY_GEOM_DATA)


def _host_from_file(path, scene):
    h = H.Host(MOCK)
    h.resize(scene.width, scene.height)
    h.set_ibl_mip_counts(scene.prefiltered_mip_count, scene.irradiance_mip_count)
    info = h.load_gltf(path)
    h.update_transforms()
    h.camera_update(scene.view, scene.proj, scene.camera_position)
    return h, info


def _host_from_desc(scene):
    h = H.Host(MOCK)
    h.resize(scene.width, scene.height)
    H.populate(h, scene)
    h.update_transforms()
    h.camera_update(scene.view, scene.proj, scene.camera_position)
    return h


def _same(a, b):
    for which in MIRRORS:
        assert a.mirror(which) == b.mirror(which), f"mirror {which} differs"
    assert a.draw_list() == b.draw_list()
    assert a.transparent_draw_list() == b.transparent_draw_list()
    assert a.texture_arrays() == b.texture_arrays() if hasattr(a, "texture_arrays") else True


SCENES = {
    "box": lambda: scenes.box_scene(64, 64),
    "helmet": lambda: scenes.helmet_scene(96, 64, segments=12, rings=8, tex_size=16),
    "skinned_morph": lambda: scenes.skinned_morph_scene(64, 64, around=8, along=12, tex_size=16),
    "zoo": lambda: scenes.material_zoo_scene(96, 64, tex_size=16),
    "transparent": lambda: scenes.transparent_scene(96, 64, tex_size=16),
    "instanced": lambda: scenes.instanced_scene(96, 64),
}


@pytest.mark.parametrize("name", sorted(SCENES))
@pytest.mark.parametrize("container", ["glb", "gltf", "gltf_data_uri"])
def test_round_trip_equals_direct_population(name, container, tmp_path):
    if container != "glb" and name not in ("helmet", "skinned_morph"):
        pytest.skip("one container variant per scene is enough beyond glb")
    scene = SCENES[name]()
    path = str(tmp_path / (name + (".glb" if container == "glb" else ".gltf")))
    if container == "glb":
        gltf_export.write_glb(scene, path)
    else:
        gltf_export.write_gltf(scene, path, data_uri=(container == "gltf_data_uri"))
    a, info = _host_from_file(path, scene)
    b = _host_from_desc(scene)
    _same(a, b)
    n_tris = sum(int(np.asarray(p.indices).reshape(-1, 3).shape[0]) for n in scene.nodes for p in n.primitives)
    assert info["triangles"] == n_tris and info["images"] == len(scene.textures) and info["lights"] == len(scene.lights)
    assert info["meshes"] == sum(len(n.primitives) for n in scene.nodes)
    # the same frame goes to the device: creates / writes / passes in the same order with the same bytes
    a.render(); b.render()
    a.close(); b.close()


def _write_raw_glb(path, doc, blob):
    js = json.dumps(doc).encode()
    js += b" " * ((4 - len(js) % 4) % 4)
    blob = blob + b"\0" * ((4 - len(blob) % 4) % 4)
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, 28 + len(js) + len(blob)))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(blob), 0x004E4942) + blob)


def _quad_doc(extra_prim=None, indices=None, mode=4):
    """A unit quad with interleaved (strided) f32 positions + normalised u16 UVs + normalised u8 colours and u8 indices."""
    pos = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], dtype=np.float32)
    uv = np.array([[0, 0], [65535, 0], [65535, 65535], [0, 65535]], dtype=np.uint16)
    col = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255]], dtype=np.uint8)
    inter = b"".join(pos[i].tobytes() + uv[i].tobytes() + col[i].tobytes() + b"\0" for i in range(4))     # stride 20
    idx = np.array(indices if indices is not None else [0, 1, 2, 0, 2, 3], dtype=np.uint8).tobytes()
    blob = inter + idx
    views = [{"buffer": 0, "byteOffset": 0, "byteLength": len(inter), "byteStride": 20}, {"buffer": 0, "byteOffset": len(inter), "byteLength": len(idx)}]
    acc = [{"bufferView": 0, "byteOffset": 0, "componentType": 5126, "count": 4, "type": "VEC3", "min": [0, 0, 0], "max": [1, 1, 0]},
           {"bufferView": 0, "byteOffset": 12, "componentType": 5123, "normalized": True, "count": 4, "type": "VEC2"},
           {"bufferView": 0, "byteOffset": 16, "componentType": 5121, "normalized": True, "count": 4, "type": "VEC3"},
           {"bufferView": 1, "componentType": 5121, "count": len(idx), "type": "SCALAR"}]
    prim = {"attributes": {"POSITION": 0, "TEXCOORD_0": 1, "COLOR_0": 2}, "indices": 3, "mode": mode}
    if extra_prim:
        prim.update(extra_prim)
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}],
           "nodes": [{"matrix": [0, 2, 0, 0, -2, 0, 0, 0, 0, 0, 2, 0, 1, 2, 3, 1], "mesh": 0}],       # scale 2, 90 degrees about z, translation (1, 2, 3)
           "meshes": [{"primitives": [prim]}], "bufferViews": views, "accessors": acc, "buffers": [{"byteLength": len(blob)}]}
    return doc, blob


def test_reader_conversions(tmp_path):
    """Strided + normalised integer attributes, u8 indices, a matrix node, default material, generated normals."""
    doc, blob = _quad_doc()
    path = str(tmp_path / "quad.glb")
    _write_raw_glb(path, doc, blob)
    h = H.Host(MOCK)
    h.resize(32, 32)
    info = h.load_gltf(path)
    assert info["triangles"] == 2 and info["meshes"] == 1 and info["materials"] == 1
    h.update_transforms()
    attr = np.frombuffer(h.mirror(sm.BUF_ATTR_DATA)[:4 * 6 * 4], dtype=np.float32).reshape(4, 6)       # COLOR_0 rgba, TEXCOORD_0
    assert np.array_equal(attr[:, :4], np.array([[1, 0, 0, 1], [0, 1, 0, 1], [0, 0, 1, 1], [1, 1, 1, 1]], dtype=np.float32))
    assert np.array_equal(attr[:, 4:], np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float32))
    assert np.array_equal(np.frombuffer(h.mirror(sm.BUF_ATTR_INDEX)[:24], dtype=np.uint32), [0, 1, 2, 0, 2, 3])
    vis = np.frombuffer(h.mirror(sm.BUF_VIS_GEOM_DATA)[:6 * 56], dtype=np.float32).reshape(6, 14)
    assert np.allclose(vis[:, 6:9], [0, 0, 1])                                   # normals computed from the faces (buffers/normals.rs)
    slots = np.frombuffer(h.mirror(sm.BUF_TRANSFORMS), dtype=np.float32).reshape(-1, 4, 4)             # one of the slots holds the node's world matrix
    want = np.array([[0, 2, 0, 0], [-2, 0, 0, 0], [0, 0, 2, 0], [1, 2, 3, 1]], dtype=np.float32)         # matrix -> TRS -> matrix (transforms.rs)
    assert any(np.allclose(m, want, atol=1e-6) for m in slots)
    h.close()


@pytest.mark.parametrize("mode,indices,want", [(5, [0, 1, 3, 2], [0, 1, 3, 1, 2, 3]), (6, [0, 1, 2, 3], [0, 1, 2, 0, 2, 3])])
def test_strips_and_fans_become_lists(mode, indices, want, tmp_path):
    doc, blob = _quad_doc(indices=indices, mode=mode)
    path = str(tmp_path / "q.glb")
    _write_raw_glb(path, doc, blob)
    h = H.Host(MOCK)
    h.resize(32, 32)
    h.load_gltf(path)
    assert np.array_equal(np.frombuffer(h.mirror(sm.BUF_ATTR_INDEX)[:24], dtype=np.uint32), want)      # buffers/index.rs:146-201
    h.close()


def test_tangents_are_generated_for_normal_mapped_primitives(tmp_path):
    """ensure_tangents: a primitive without TANGENT whose material has a normal map gets per-vertex tangents — unit length,
    orthogonal to the normal, pointing along +u, handedness +1 for a right-handed UV layout."""
    sc = scenes.helmet_scene(64, 64, segments=16, rings=12, tex_size=16)
    prim = [p for n in sc.nodes for p in n.primitives][0]
    given = np.asarray(prim.tangents, dtype=np.float32).copy()
    prim.tangents = None
    path = str(tmp_path / "nt.glb")
    gltf_export.write_glb(sc, path)
    h = H.Host(MOCK)
    h.resize(64, 64)
    info = h.load_gltf(path)
    assert info["generated_tangents"] == 1
    T = int(np.asarray(prim.indices).reshape(-1, 3).shape[0])
    vis = np.frombuffer(h.mirror(sm.BUF_VIS_GEOM_DATA)[:T * 3 * 56], dtype=np.float32).reshape(T * 3, 14)
    n, t, w = vis[:, 6:9], vis[:, 9:12], vis[:, 12]
    orig = vis[:, 13].view(np.uint32)
    assert np.allclose(np.linalg.norm(t, axis=1), 1.0, atol=1e-5)
    assert np.abs((n * t).sum(axis=1)).max() < 1e-4
    assert set(np.unique(w)) <= {-1.0, 1.0}
    cos = (t * given[orig, :3]).sum(axis=1)                                       # against the analytic tangents of the parametric surface
    assert np.median(cos) > 0.95 and (w == given[orig, 3]).mean() > 0.95
    h.close()


def test_sparse_and_zero_filled_accessors(tmp_path):
    """gltf/buffers/accessor.rs:14-66: a sparse block substitutes elements of the (repacked) view data; an accessor without a bufferView
    starts as zeros.  Here: POSITION is dense + sparse (two vertices moved, u16 indices), TEXCOORD_0 is zero-filled + sparse (u8 indices),
    and the morph target — the way sample assets use sparse accessors — is zero-filled + sparse."""
    doc, blob = _quad_doc()
    sp_idx16 = np.array([1, 3], dtype=np.uint16).tobytes()
    sp_pos = np.array([[2, 0, 0], [0, 3, 0]], dtype=np.float32).tobytes()
    sp_idx8 = np.array([2], dtype=np.uint8).tobytes() + b"\0\0\0"
    sp_uv = np.array([[0.25, 0.75]], dtype=np.float32).tobytes()
    sp_dpos = np.array([[0, 0, 1], [0, 0, 2]], dtype=np.float32).tobytes()
    base = len(blob) + (4 - len(blob) % 4) % 4
    blob = blob + b"\0" * (base - len(blob))
    extra = sp_idx16 + sp_pos + sp_idx8 + sp_uv + sp_dpos
    o = [base, base + 4, base + 4 + 24, base + 4 + 24 + 4, base + 4 + 24 + 4 + 8]
    doc["bufferViews"] += [{"buffer": 0, "byteOffset": o[0], "byteLength": 4}, {"buffer": 0, "byteOffset": o[1], "byteLength": 24},
                           {"buffer": 0, "byteOffset": o[2], "byteLength": 1}, {"buffer": 0, "byteOffset": o[3], "byteLength": 8},
                           {"buffer": 0, "byteOffset": o[4], "byteLength": 24}]
    doc["accessors"][0]["sparse"] = {"count": 2, "indices": {"bufferView": 2, "componentType": 5123}, "values": {"bufferView": 3}}
    doc["accessors"][0]["max"] = [2, 3, 0]
    doc["accessors"][1] = {"componentType": 5126, "count": 4, "type": "VEC2", "sparse": {"count": 1, "indices": {"bufferView": 4, "componentType": 5121}, "values": {"bufferView": 5}}}
    doc["accessors"].append({"componentType": 5126, "count": 4, "type": "VEC3", "sparse": {"count": 2, "indices": {"bufferView": 2, "componentType": 5123}, "values": {"bufferView": 6}}})
    doc["meshes"][0]["primitives"][0]["targets"] = [{"POSITION": 4}]
    doc["meshes"][0]["weights"] = [0.5]
    doc["buffers"][0]["byteLength"] = len(blob) + len(extra)
    path = str(tmp_path / "sparse.glb")
    _write_raw_glb(path, doc, blob + extra)
    h = H.Host(MOCK)
    h.resize(32, 32)
    info = h.load_gltf(path)
    assert info["triangles"] == 2
    attr = np.frombuffer(h.mirror(sm.BUF_ATTR_DATA)[:4 * 6 * 4], dtype=np.float32).reshape(4, 6)
    assert np.array_equal(attr[:, 4:], np.array([[0, 0], [0, 0], [0.25, 0.75], [0, 0]], dtype=np.float32))            # zeros, one element substituted
    vis = np.frombuffer(h.mirror(sm.BUF_VIS_GEOM_DATA)[:6 * 56], dtype=np.float32).reshape(6, 14)
    want_pos = np.array([[0, 0, 0], [2, 0, 0], [1, 1, 0], [0, 3, 0]], dtype=np.float32)                                # vertices 1 and 3 replaced
    assert np.array_equal(vis[:, :3], want_pos[[0, 1, 2, 0, 2, 3]])
    morph = np.frombuffer(h.mirror(sm.BUF_MORPH_VALUES)[:4 * 40], dtype=np.float32).reshape(4, 10)                     # 10 floats / target / vertex
    assert np.array_equal(morph[:, :3], np.array([[0, 0, 0], [0, 0, 1], [0, 0, 0], [0, 0, 2]], dtype=np.float32))
    h.close()
    # a sparse index beyond the accessor, and sparse data beyond its view, are refused
    for mutate, msg in ((lambda d: d["accessors"][0]["sparse"].__setitem__("count", 9), "sparse"),
                        (lambda d: d["bufferViews"][2].__setitem__("byteLength", 2), "sparse data exceeds")):
        d2 = json.loads(json.dumps(doc))
        mutate(d2)
        _write_raw_glb(path, d2, blob + extra)
        h = H.Host(MOCK)
        h.resize(32, 32)
        with pytest.raises(H.HostError, match=msg):
            h.load_gltf(path)
        h.close()


def test_ext_mesh_gpu_instancing(tmp_path):
    """gltf/populate/extensions/instancing.rs: per-node TRANSLATION / ROTATION / SCALE accessors -> the instance-transform mirror of the
    node's meshes; a missing attribute is identity, integer rotations are cast without normalisation (the reference's `as f32`)."""
    doc, blob = _quad_doc()
    tr = np.array([[0, 0, 0], [3, 0, 0], [0, 4, 0]], dtype=np.float32).tobytes()
    ro = np.array([[0, 0, 0, 1], [0, 0, 1, 0], [0, 1, 0, 0]], dtype=np.int8).tobytes()          # i8, declared normalized: still cast raw
    base = len(blob) + (4 - len(blob) % 4) % 4
    blob = blob + b"\0" * (base - len(blob)) + tr + ro
    doc["bufferViews"] += [{"buffer": 0, "byteOffset": base, "byteLength": 36}, {"buffer": 0, "byteOffset": base + 36, "byteLength": 12}]
    doc["accessors"] += [{"bufferView": 2, "componentType": 5126, "count": 3, "type": "VEC3"}, {"bufferView": 3, "componentType": 5120, "normalized": True, "count": 3, "type": "VEC4"}]
    doc["nodes"][0]["extensions"] = {"EXT_mesh_gpu_instancing": {"attributes": {"TRANSLATION": 4, "ROTATION": 5}}}
    doc["extensionsUsed"] = ["EXT_mesh_gpu_instancing"]
    doc["buffers"][0]["byteLength"] = len(blob)
    path = str(tmp_path / "inst.glb")
    _write_raw_glb(path, doc, blob)
    h = H.Host(MOCK)
    h.resize(32, 32)
    h.load_gltf(path)
    h.update_transforms()
    h.camera_update(scenes.look_at_rh((0, 0, 30), (0, 0, 0)), scenes.perspective_rh(1.0, 1.0, 0.1, 100.0), (0, 0, 30))
    h.render()
    inst = np.frombuffer(h.mirror(sm.BUF_INSTANCES)[:3 * 64], dtype=np.float32).reshape(3, 4, 4)
    def trs(t, q):       # glam Mat4::from_scale_rotation_translation with unit scale, column-major [col][row]
        x, y, z, w = q
        r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y), 0], [2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x), 0],
                      [2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y), 0], [t[0], t[1], t[2], 1]], dtype=np.float32)
        return r
    assert np.allclose(inst[0], trs((0, 0, 0), (0, 0, 0, 1))) and np.allclose(inst[1], trs((3, 0, 0), (0, 0, 1, 0))) and np.allclose(inst[2], trs((0, 4, 0), (0, 1, 0, 0)))
    dl = h.draw_list()
    assert len(dl) == 1 and dl[0]["inst_count"] == 3
    h.close()
    doc["accessors"][4]["componentType"] = 5123            # translation must be f32
    _write_raw_glb(path, doc, blob)
    h = H.Host(MOCK)
    h.resize(32, 32)
    with pytest.raises(H.HostError, match="Vec3F32"):
        h.load_gltf(path)
    h.close()


def test_hostile_offsets_and_counts_are_refused(tmp_path):
    """Negative, fractional or enormous byteOffset / byteStride / byteLength / count values must end in an error message, not in a read
    outside the buffer (they used to be cast to size_t and wrap around the bounds check)."""
    cases = [
        (lambda d: d["bufferViews"][0].__setitem__("byteOffset", -4096), "non-negative"),
        (lambda d: d["bufferViews"][0].__setitem__("byteStride", 2048), "byteStride"),
        (lambda d: d["bufferViews"][0].__setitem__("byteOffset", 1.8e19), "non-negative|beyond|exceeds"),
        (lambda d: d["bufferViews"][0].__setitem__("byteLength", 10**9), "exceeds"),
        (lambda d: d["accessors"][0].__setitem__("byteOffset", 2**40), "exceeds"),
        (lambda d: d["accessors"][0].__setitem__("count", 2**31), "exceeds"),
        (lambda d: d["accessors"][0].__setitem__("count", -3), "non-negative"),
        (lambda d: d["accessors"][0].__setitem__("count", 2.5), "non-negative"),
        (lambda d: d["accessors"][3].__setitem__("byteOffset", 7), "exceeds"),
        (lambda d: d["bufferViews"][0].__setitem__("byteLength", 30), "exceeds"),          # accessor runs past the view although the buffer is long enough
        (lambda d: d["accessors"][0].__setitem__("bufferView", 1e300), "bufferView"),
        (lambda d: d["bufferViews"][1].__setitem__("buffer", -1), "buffer missing"),
    ]
    for mutate, msg in cases:
        doc, blob = _quad_doc()
        mutate(doc)
        path = str(tmp_path / "h.glb")
        _write_raw_glb(path, doc, blob)
        h = H.Host(MOCK)
        h.resize(32, 32)
        with pytest.raises(H.HostError, match=msg):
            h.load_gltf(path)
        h.close()
    # the image bufferView path has the same checks
    doc, blob = _quad_doc()
    doc["bufferViews"].append({"buffer": 0, "byteOffset": 8, "byteLength": 2**33})
    doc["images"] = [{"bufferView": 2, "mimeType": "image/png"}]
    doc["textures"] = [{"source": 0}]
    doc["materials"] = [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}]
    doc["meshes"][0]["primitives"][0]["material"] = 0
    path = str(tmp_path / "i.glb")
    _write_raw_glb(path, doc, blob)
    h = H.Host(MOCK)
    h.resize(32, 32)
    with pytest.raises(H.HostError, match="exceeds"):
        h.load_gltf(path)
    h.close()


def test_unsupported_inputs_fail_loudly(tmp_path):
    doc, blob = _quad_doc()
    jpeg = bytes([0xFF, 0xD8, 0xFF, 0xE0]) + bytes(32)
    doc["images"] = [{"uri": "data:image/jpeg;base64," + base64.b64encode(jpeg).decode()}]
    doc["textures"] = [{"source": 0}]
    doc["materials"] = [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}]
    path = str(tmp_path / "j.glb")
    _write_raw_glb(path, doc, blob)
    h = H.Host(MOCK)
    h.resize(32, 32)
    with pytest.raises(H.HostError, match="JPEG"):
        h.load_gltf(path)
    doc, blob = _quad_doc(mode=1)
    _write_raw_glb(path, doc, blob)
    with pytest.raises(H.HostError, match="mode 1"):
        h.load_gltf(path)
    doc, blob = _quad_doc()
    doc["extensionsRequired"] = ["KHR_draco_mesh_compression"]
    _write_raw_glb(path, doc, blob)
    with pytest.raises(H.HostError, match="KHR_draco_mesh_compression"):
        h.load_gltf(path)
    with pytest.raises(H.HostError, match="cannot read"):
        h.load_gltf(str(tmp_path / "missing.glb"))
    with open(path, "wb") as f:
        f.write(b"{ not json")
    with pytest.raises(H.HostError, match="JSON"):
        h.load_gltf(path)
    h.close()


# ------------------------------------------------------------------------------------------------ image decoders

def _test_image(w, h, seed=3, smooth=True):
    rng = np.random.default_rng(seed)
    if smooth:
        y, x = np.mgrid[0:h, 0:w].astype(np.float32)
        img = np.stack([127 + 100 * np.sin(x / 9.0) * np.cos(y / 13.0), 127 + 90 * np.cos(x / 17.0 + y / 11.0), 127 + 110 * np.sin((x + y) / 23.0), np.full_like(x, 255)], axis=-1)
        return img.clip(0, 255).astype(np.uint8)
    return rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)


@pytest.mark.parametrize("mode,size", [("RGBA", (37, 23)), ("RGB", (64, 40)), ("L", (19, 31)), ("LA", (16, 16)), ("P", (33, 17)), ("I;16", (20, 12)), ("1", (29, 9))])
def test_png_decoder_matches_pil(mode, size):
    import io
    from PIL import Image
    w, h = size
    src = Image.fromarray(_test_image(w, h, smooth=False), "RGBA")
    if mode == "I;16":
        im = Image.fromarray((np.arange(w * h, dtype=np.uint32).reshape(h, w) * 257 % 65536).astype(np.uint16))
    elif mode == "P":
        im = src.convert("RGB").quantize(colors=16)
    elif mode == "1":
        im = src.convert("L").point(lambda v: 255 if v > 127 else 0).convert("1")
    else:
        im = src.convert(mode)
    buf = io.BytesIO()
    im.save(buf, format="PNG")
    got = H.decode_image(buf.getvalue())
    if mode == "I;16":
        want = (np.asarray(im).astype(np.uint32) >> 8).astype(np.uint8)               # 16 -> 8 bits: the high byte
        assert np.array_equal(got[..., 0], want) and np.array_equal(got[..., 1], want) and (got[..., 3] == 255).all()
    else:
        assert np.array_equal(got, np.asarray(im.convert("RGBA")))


@pytest.mark.parametrize("subsampling,quality,size,gray", [(0, 92, (64, 48), False), (2, 85, (80, 56), False), (1, 90, (37, 29), False), (0, 80, (40, 40), True)])
def test_baseline_jpeg_decoder_is_close_to_libjpeg(subsampling, quality, size, gray):
    """Lossy format, decoders differ in IDCT rounding and chroma upsampling (replication here, triangle filter in libjpeg): close, not equal."""
    import io
    from PIL import Image
    w, h = size
    src = Image.fromarray(_test_image(w, h)[..., :3], "RGB")
    if gray:
        src = src.convert("L")
    buf = io.BytesIO()
    kw = {} if gray else {"subsampling": subsampling}
    src.save(buf, format="JPEG", quality=quality, **kw)
    got = H.decode_image(buf.getvalue()).astype(np.int32)
    want = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGBA")).astype(np.int32)
    assert got.shape == want.shape and (got[..., 3] == 255).all()
    d = np.abs(got[..., :3] - want[..., :3])
    assert d.mean() < (1.0 if subsampling == 0 else 3.0) and np.percentile(d, 99) <= (3 if subsampling == 0 else 16), (d.mean(), d.max())


def test_jpeg_restart_intervals_and_refusals():
    import io
    from PIL import Image
    src = Image.fromarray(_test_image(72, 40)[..., :3], "RGB")
    a, b = io.BytesIO(), io.BytesIO()
    src.save(a, format="JPEG", quality=90, subsampling=0)
    src.save(b, format="JPEG", quality=90, subsampling=0, restart_marker_blocks=3)
    assert b"\xff\xdd" in b.getvalue()
    assert np.array_equal(H.decode_image(a.getvalue()), H.decode_image(b.getvalue()))
    with pytest.raises(H.HostError, match="not supported"):
        H.decode_image(b"GIF89a" + bytes(32))


@pytest.mark.parametrize("subsampling,quality,size,gray,restart", [(0, 90, (72, 40), False, 0), (2, 85, (83, 57), False, 0), (1, 75, (37, 29), False, 0),
                                                                     (0, 95, (40, 40), True, 0), (2, 60, (130, 70), False, 2), (0, 100, (16, 16), False, 0)])
def test_progressive_jpeg_decodes_to_the_same_pixels_as_its_baseline_twin(subsampling, quality, size, gray, restart):
    """SOF2 (spectral selection + successive approximation, DC / AC first and refining scans, end-of-band runs, non-interleaved AC scans)
    carries the same quantised coefficients as the baseline encoding of the same picture at the same quality: the two must decode to
    identical pixels, and stay as close to libjpeg as the baseline decoder is."""
    import io
    from PIL import Image
    w, h = size
    src = Image.fromarray(_test_image(w, h)[..., :3], "RGB")
    if gray:
        src = src.convert("L")
    kw = {} if gray else {"subsampling": subsampling}
    if restart:
        kw["restart_marker_rows"] = restart
    base, prog = io.BytesIO(), io.BytesIO()
    src.save(base, format="JPEG", quality=quality, **kw)
    src.save(prog, format="JPEG", quality=quality, progressive=True, **kw)
    assert b"\xff\xc2" in prog.getvalue() and b"\xff\xc2" not in base.getvalue()
    got = H.decode_image(prog.getvalue())
    assert np.array_equal(got, H.decode_image(base.getvalue()))
    want = np.asarray(Image.open(io.BytesIO(prog.getvalue())).convert("RGBA")).astype(np.int32)
    d = np.abs(got.astype(np.int32)[..., :3] - want[..., :3])
    assert d.mean() < (1.0 if subsampling == 0 else 3.5), d.mean()
