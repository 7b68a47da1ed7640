"""SceneDesc -> glTF 2.0 (.glb, or .gltf with an external .bin and PNG files).

The synthetic scenes of this repo written as ordinary glTF files, so that the native reader (awsm_host_load_gltf,
awsm-renderer_amd/host/gltf.cpp) can be tested end to end without network access to the usual sample models: a scene loaded
from the file must leave the same mirrors as the same SceneDesc populated through the key API.  Values that glTF cannot carry
(the reference's texture-transform origin, exact light vectors, GPU instances, the vertex-colour set of a material) travel in
`extras`; everything else is plain glTF core + KHR material extensions.
"""
from __future__ import annotations

import base64
import io
import json
import os
import struct
from typing import Dict, List, Optional, Tuple

import numpy as np

from .scene_desc import MaterialDesc, SceneDesc, TextureRef

F = np.float32


def _f(x) -> float:
    """JSON number that reads back to the same f32"""
    return float(np.float32(x))


def _fl(v) -> List[float]:
    return [_f(x) for x in v]


class _Bin:
    def __init__(self):
        self.data = bytearray()
        self.views: List[dict] = []
        self.accessors: List[dict] = []

    def add(self, arr: np.ndarray, ctype: int, gtype: str, minmax: bool = False, target: Optional[int] = None) -> int:
        arr = np.ascontiguousarray(arr)
        while len(self.data) % 4:
            self.data.append(0)
        view = {"buffer": 0, "byteOffset": len(self.data), "byteLength": arr.nbytes}
        if target is not None:
            view["target"] = target
        self.data += arr.tobytes()
        self.views.append(view)
        acc = {"bufferView": len(self.views) - 1, "componentType": ctype, "count": int(arr.shape[0]), "type": gtype}
        if minmax:
            acc["min"] = _fl(arr.min(axis=0)); acc["max"] = _fl(arr.max(axis=0))
        self.accessors.append(acc)
        return len(self.accessors) - 1

    def add_blob(self, blob: bytes) -> int:
        while len(self.data) % 4:
            self.data.append(0)
        self.views.append({"buffer": 0, "byteOffset": len(self.data), "byteLength": len(blob)})
        self.data += blob
        return len(self.views) - 1


def _png(rgba: np.ndarray) -> bytes:
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(rgba, dtype=np.uint8), "RGBA").save(buf, format="PNG", compress_level=1)
    return buf.getvalue()


_WRAP = {0: 33071, 1: 10497, 2: 33648}


def _sampler(s: dict) -> dict:
    mag = 9729 if s.get("mag_filter", 1) else 9728
    mn, mip = s.get("min_filter", 1), s.get("mipmap_filter", 1)
    min_filter = {(0, 0): 9984, (1, 0): 9985, (0, 1): 9986, (1, 1): 9987}[(1 if mn else 0, 1 if mip else 0)]
    return {"wrapS": _WRAP[s.get("address_mode_u", 1)], "wrapT": _WRAP[s.get("address_mode_v", 1)], "magFilter": mag, "minFilter": min_filter,
            "extras": {"max_anisotropy": int(s.get("max_anisotropy", 1))}}


def scene_to_gltf(scene: SceneDesc, embed_images: bool = True) -> Tuple[dict, bytes, List[bytes]]:
    """Returns (document, binary buffer, PNG blobs).  With embed_images the PNGs are bufferViews of the binary buffer."""
    b = _Bin()
    doc: dict = {"asset": {"version": "2.0", "generator": "awsm-renderer_amd gltf_export"}, "scene": 0}
    used_ext = set()
    pngs = [_png(t) for t in scene.textures]
    # ---- textures = (image, sampler) pairs, ordered so that samplers first appear in scene.samplers order ----
    pairs: List[Tuple[int, int]] = []

    def collect(ref: Optional[TextureRef]):
        if ref is not None and 0 <= ref.texture < len(scene.textures) and 0 <= ref.sampler < len(scene.samplers) and (ref.sampler, ref.texture) not in pairs:
            pairs.append((ref.sampler, ref.texture))

    def material_refs(m: MaterialDesc):
        refs = [m.base_color_tex, m.metallic_roughness_tex, m.normal_tex, m.occlusion_tex, m.emissive_tex]
        for blk, keys in ((m.specular, ("tex", "color_tex")), (m.transmission, ("tex",)), (m.volume, ("thickness_tex",)),
                          (m.clearcoat, ("tex", "roughness_tex", "normal_tex")), (m.sheen, ("roughness_tex", "color_tex"))):
            if blk:
                refs += [blk.get(k) for k in keys]
        return refs

    for m in scene.materials:
        for r in material_refs(m):
            collect(r)
    for si in range(len(scene.samplers)):          # a sampler no material uses still takes its id
        if not any(p[0] == si for p in pairs) and scene.textures:
            pairs.append((si, 0))
    pairs.sort()
    tex_index = {p: i for i, p in enumerate(pairs)}
    if scene.samplers:
        doc["samplers"] = [_sampler(s) for s in scene.samplers]
    if pairs:
        doc["textures"] = [{"sampler": s, "source": t} for s, t in pairs]

    def texinfo(ref: Optional[TextureRef], extra: Optional[dict] = None) -> Optional[dict]:
        if ref is None:
            return None
        valid = 0 <= ref.texture < len(scene.textures) and 0 <= ref.sampler < len(scene.samplers)
        info = {"index": tex_index[(ref.sampler, ref.texture)] if valid else 1_000_000}      # dangling -> the reader skips the texture
        if ref.uv_index:
            info["texCoord"] = int(ref.uv_index)
        if ref.transform:
            t = ref.transform
            used_ext.add("KHR_texture_transform")
            info["extensions"] = {"KHR_texture_transform": {"offset": _fl(t.get("offset", (0, 0))), "rotation": _f(t.get("rotation", 0.0)), "scale": _fl(t.get("scale", (1, 1))),
                                                            "extras": {"origin": _fl(t.get("origin", (0, 0)))}}}
        if extra:
            info.update(extra)
        return info

    def put(d: dict, key: str, v):
        if v is not None:
            d[key] = v

    mats = []
    for m in scene.materials:
        g: dict = {"pbrMetallicRoughness": {"baseColorFactor": _fl(m.base_color_factor), "metallicFactor": _f(m.metallic_factor), "roughnessFactor": _f(m.roughness_factor)},
                   "emissiveFactor": _fl(m.emissive_factor), "doubleSided": bool(m.double_sided)}
        put(g["pbrMetallicRoughness"], "baseColorTexture", texinfo(m.base_color_tex))
        put(g["pbrMetallicRoughness"], "metallicRoughnessTexture", texinfo(m.metallic_roughness_tex))
        put(g, "normalTexture", texinfo(m.normal_tex, {"scale": _f(m.normal_scale)}))
        put(g, "occlusionTexture", texinfo(m.occlusion_tex, {"strength": _f(m.occlusion_strength)}))
        put(g, "emissiveTexture", texinfo(m.emissive_tex))
        if m.alpha_mode != "opaque":
            g["alphaMode"] = m.alpha_mode.upper()
            if m.alpha_mode == "mask":
                g["alphaCutoff"] = _f(m.alpha_cutoff)
        extras, ext = {}, {}
        if m.normal_tex is None and m.normal_scale != 1.0:
            extras["normal_scale"] = _f(m.normal_scale)
        if m.occlusion_tex is None and m.occlusion_strength != 1.0:
            extras["occlusion_strength"] = _f(m.occlusion_strength)
        if m.debug_bitmask:
            extras["debug_bitmask"] = int(m.debug_bitmask)
        if m.vertex_color_set is not None:
            extras["vertex_color_set"] = int(m.vertex_color_set)
        if m.kind == "unlit":
            ext["KHR_materials_unlit"] = {}
        if m.emissive_strength is not None:
            ext["KHR_materials_emissive_strength"] = {"emissiveStrength": _f(m.emissive_strength)}
        if m.ior is not None:
            ext["KHR_materials_ior"] = {"ior": _f(m.ior)}
        if m.specular is not None:
            s = m.specular
            e = {"specularFactor": _f(s.get("factor", 1.0)), "specularColorFactor": _fl(s.get("color_factor", (1, 1, 1)))}
            put(e, "specularTexture", texinfo(s.get("tex"))); put(e, "specularColorTexture", texinfo(s.get("color_tex")))
            ext["KHR_materials_specular"] = e
        if m.transmission is not None:
            s = m.transmission
            e = {"transmissionFactor": _f(s.get("factor", 0.0))}
            put(e, "transmissionTexture", texinfo(s.get("tex")))
            ext["KHR_materials_transmission"] = e
        if m.volume is not None:
            s = m.volume
            e = {"thicknessFactor": _f(s.get("thickness_factor", 0.0)), "attenuationDistance": _f(s.get("attenuation_distance", 0.0)),
                 "attenuationColor": _fl(s.get("attenuation_color", (1, 1, 1)))}
            put(e, "thicknessTexture", texinfo(s.get("thickness_tex")))
            ext["KHR_materials_volume"] = e
        if m.clearcoat is not None:
            s = m.clearcoat
            e = {"clearcoatFactor": _f(s.get("factor", 0.0)), "clearcoatRoughnessFactor": _f(s.get("roughness_factor", 0.0))}
            put(e, "clearcoatTexture", texinfo(s.get("tex"))); put(e, "clearcoatRoughnessTexture", texinfo(s.get("roughness_tex")))
            put(e, "clearcoatNormalTexture", texinfo(s.get("normal_tex"), {"scale": _f(s.get("normal_scale", 1.0))}))
            if s.get("normal_tex") is None and s.get("normal_scale", 1.0) != 1.0:
                e["extras"] = {"normal_scale": _f(s.get("normal_scale", 1.0))}
            ext["KHR_materials_clearcoat"] = e
        if m.sheen is not None:
            s = m.sheen
            e = {"sheenColorFactor": _fl(s.get("color_factor", (0, 0, 0))), "sheenRoughnessFactor": _f(s.get("roughness_factor", 0.0))}
            put(e, "sheenRoughnessTexture", texinfo(s.get("roughness_tex"))); put(e, "sheenColorTexture", texinfo(s.get("color_tex")))
            ext["KHR_materials_sheen"] = e
        if extras:
            g["extras"] = extras
        if ext:
            g["extensions"] = ext
            used_ext.update(ext.keys())
        mats.append(g)
    if mats:
        doc["materials"] = mats

    # ---- nodes / meshes / skins ----
    children: Dict[Optional[int], List[int]] = {}
    for i, n in enumerate(scene.nodes):
        children.setdefault(n.parent, []).append(i)
    nodes, meshes = [], []
    for i, n in enumerate(scene.nodes):
        g = {"translation": _fl(n.translation), "rotation": _fl(n.rotation), "scale": _fl(n.scale)}
        if children.get(i):
            g["children"] = children[i]
        if n.primitives:
            prims = []
            for p in n.primitives:
                attrs = {"POSITION": b.add(np.asarray(p.positions, dtype=F), 5126, "VEC3", minmax=True, target=34962),
                         "NORMAL": b.add(np.asarray(p.normals, dtype=F), 5126, "VEC3", target=34962)}
                if p.tangents is not None:
                    attrs["TANGENT"] = b.add(np.asarray(p.tangents, dtype=F), 5126, "VEC4", target=34962)
                for k, uv in enumerate(p.uvs):
                    attrs[f"TEXCOORD_{k}"] = b.add(np.asarray(uv, dtype=F), 5126, "VEC2", target=34962)
                for k, c in enumerate(p.colors):
                    attrs[f"COLOR_{k}"] = b.add(np.asarray(c, dtype=F), 5126, "VEC4", target=34962)
                for k, (j, w) in enumerate(zip(p.joints, p.weights)):
                    attrs[f"JOINTS_{k}"] = b.add(np.asarray(j, dtype=np.uint16), 5123, "VEC4", target=34962)
                    attrs[f"WEIGHTS_{k}"] = b.add(np.asarray(w, dtype=F), 5126, "VEC4", target=34962)
                gp = {"attributes": attrs, "indices": b.add(np.asarray(p.indices, dtype=np.uint32).reshape(-1), 5125, "SCALAR", target=34963), "material": int(p.material), "mode": 4}
                if p.morph_targets:
                    gp["targets"] = []
                    for t in p.morph_targets:
                        gt = {}
                        for key, name in (("positions", "POSITION"), ("normals", "NORMAL"), ("tangents", "TANGENT")):
                            if t.get(key) is not None:
                                gt[name] = b.add(np.asarray(t[key], dtype=F), 5126, "VEC3", minmax=(name == "POSITION"))
                        gp["targets"].append(gt)
                ex = {}
                if p.animated_morph_weights is not None:
                    ex["animated_morph_weights"] = _fl(p.animated_morph_weights)
                if p.instances is not None:
                    ex["instances"] = [_fl(list(t) + list(r) + list(s)) for (t, r, s) in p.instances]
                if ex:
                    gp["extras"] = ex
                prims.append(gp)
            gm = {"primitives": prims}
            w = n.primitives[0].morph_weights
            if n.primitives[0].morph_targets:
                gm["weights"] = _fl(w if w is not None else np.zeros(len(n.primitives[0].morph_targets)))
            meshes.append(gm)
            g["mesh"] = len(meshes) - 1
            if n.skin is not None:
                g["skin"] = int(n.skin)
        nodes.append(g)
    if scene.skins:
        doc["skins"] = [{"joints": [int(j) for j in sk.joints],
                         "inverseBindMatrices": b.add(np.asarray(sk.inverse_bind, dtype=F).reshape(len(sk.joints), 16), 5126, "MAT4")} for sk in scene.skins]
    scene_nodes = list(children.get(None, []))
    # ---- lights: KHR_lights_punctual on nodes outside the scene graph; the exact vectors travel in extras ----
    if scene.lights:
        used_ext.add("KHR_lights_punctual")
        lights = []
        for l in scene.lights:
            g = {"type": l["kind"], "color": _fl(l.get("color", (1, 1, 1))), "intensity": _f(l.get("intensity", 1.0)),
                 "extras": {"position": _fl(l.get("position", (0, 0, 0))), "direction": _fl(l.get("direction", (0, 0, -1))),
                            "inner_angle": _f(l.get("inner_angle", 0.0)), "outer_angle": _f(l.get("outer_angle", 0.0))}}
            if l.get("range", 0.0):
                g["range"] = _f(l["range"])
            if l["kind"] == "spot":
                g["spot"] = {}
            lights.append(g)
            nodes.append({"name": f"light{len(lights) - 1}", "extensions": {"KHR_lights_punctual": {"light": len(lights) - 1}}})
        doc["extensions"] = {"KHR_lights_punctual": {"lights": lights}}
    doc["nodes"] = nodes
    if meshes:
        doc["meshes"] = meshes
    doc["scenes"] = [{"nodes": scene_nodes}]
    # ---- images ----
    if pngs:
        if embed_images:
            doc["images"] = [{"bufferView": b.add_blob(png), "mimeType": "image/png"} for png in pngs]
        else:
            doc["images"] = [{"uri": f"image{i}.png"} for i in range(len(pngs))]
    doc["bufferViews"] = b.views
    doc["accessors"] = b.accessors
    doc["buffers"] = [{"byteLength": len(b.data)}]
    if used_ext:
        doc["extensionsUsed"] = sorted(used_ext)
    return doc, bytes(b.data), pngs


def write_glb(scene: SceneDesc, path: str) -> None:
    doc, blob, _ = scene_to_gltf(scene, embed_images=True)
    js = json.dumps(doc, separators=(",", ":")).encode("utf-8")
    js += b" " * ((4 - len(js) % 4) % 4)
    blob += b"\0" * ((4 - len(blob) % 4) % 4)
    total = 12 + 8 + len(js) + 8 + len(blob)
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, total))
        f.write(struct.pack("<II", len(js), 0x4E4F534A)); f.write(js)
        f.write(struct.pack("<II", len(blob), 0x004E4942)); f.write(blob)


def write_gltf(scene: SceneDesc, path: str, data_uri: bool = False) -> None:
    """`path`.gltf next to `<stem>.bin` and imageN.png files; with data_uri the buffer is a base64 data URI instead."""
    doc, blob, pngs = scene_to_gltf(scene, embed_images=False)
    d = os.path.dirname(os.path.abspath(path))
    stem = os.path.splitext(os.path.basename(path))[0]
    if data_uri:
        doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(blob).decode("ascii")
    else:
        doc["buffers"][0]["uri"] = stem + ".bin"
        with open(os.path.join(d, stem + ".bin"), "wb") as f:
            f.write(blob)
    for i, png in enumerate(pngs):
        with open(os.path.join(d, f"image{i}.png"), "wb") as f:
            f.write(png)
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
