"""Run by tests/test_sanitizers_cpu.py in a child process with the AddressSanitizer runtime preloaded: drives the ASan + UBSan build of the
C++ host layer (glTF reader included) through population, rendering against the mock backend, and a few hundred corrupted glTF files."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from awsm_renderer_amd import gltf_export, scenes  # noqa: E402
from awsm_renderer_amd import host as H  # noqa: E402

MOCK = os.path.join(ROOT, "tests", "mock", "libmock_backend.so")
tmp = sys.argv[1]


def frame(scene):
    r = H.Renderer(scene, backend_path=MOCK, lut_rgba16f=np.zeros((4, 4, 4), dtype=np.uint16))
    r.render()
    r.host.transform_set_local(r.keys.node_keys[1], (0.1, 0.2, 0.3), (0, 0, 0, 1), (1, 1, 1))
    r.update()
    r.render()
    r.close()


frame(scenes.skinned_morph_scene(64, 64, around=8, along=12, tex_size=16))
frame(scenes.material_zoo_scene(96, 64, tex_size=16))
frame(scenes.transparent_scene(96, 64, tex_size=16))
frame(scenes.instanced_scene(96, 64))

# glTF reader: valid files, then corrupted ones (bit flips, truncations): every outcome but a crash / sanitizer report is fine
sc = scenes.helmet_scene(64, 64, segments=8, rings=6, tex_size=8)
path = os.path.join(tmp, "h.glb")
gltf_export.write_glb(sc, path)
good = open(path, "rb").read()
h = H.Host(MOCK)
h.resize(64, 64)
print("loaded", h.load_gltf(path))
h.close()
rng = np.random.default_rng(12345)
ok = bad = 0
for case in range(300):
    data = bytearray(good)
    kind = case % 3
    if kind == 0:      # flip bytes inside the JSON chunk
        n = int.from_bytes(data[12:16], "little")
        for _ in range(rng.integers(1, 6)):
            data[20 + int(rng.integers(0, n))] = int(rng.integers(32, 127))
    elif kind == 1:    # flip bytes anywhere
        for _ in range(rng.integers(1, 16)):
            data[int(rng.integers(0, len(data)))] = int(rng.integers(0, 256))
    else:              # truncate
        data = data[:int(rng.integers(12, len(data)))]
    p = os.path.join(tmp, "c.glb")
    open(p, "wb").write(bytes(data))
    h = H.Host(MOCK)
    h.resize(64, 64)
    try:
        h.load_gltf(p)
        ok += 1
    except H.HostError:
        bad += 1
    h.close()
print("corrupted files: loaded", ok, "refused", bad)
# hostile numbers in an otherwise well-formed document: every byteOffset / byteStride / byteLength / count / index replaced, one at a time,
# by negative, fractional, huge and just-too-large values (a wrapped bounds check would read outside the heap buffer: ASan sees it)
import json
import struct
jlen = int.from_bytes(good[12:16], "little")
doc0, rest = json.loads(good[20:20 + jlen]), good[20 + jlen:]
sites = []
for key in ("bufferViews", "accessors", "images"):
    for i, obj in enumerate(doc0.get(key, [])[:4]):
        for field in ("byteOffset", "byteStride", "byteLength", "count", "bufferView", "buffer"):
            if field in obj or field in ("byteOffset", "byteStride"):
                sites.append((key, i, field))
for i, acc in enumerate(doc0.get("accessors", [])[:2]):
    sites.append(("sparse", i, "count"))
n_ok = n_bad = 0
for (key, i, field) in sites:
    for value in (-4096, 2.5, 2048, 2 ** 32 + 4, 1.8e19):
        d = json.loads(json.dumps(doc0))
        if key == "sparse":
            d["accessors"][i]["sparse"] = {"count": value, "indices": {"bufferView": 0, "componentType": 5125}, "values": {"bufferView": 0}}
        else:
            d[key][i][field] = value
        js = json.dumps(d).encode()
        js += b" " * ((4 - len(js) % 4) % 4)
        blob = struct.pack("<4sII", b"glTF", 2, 20 + len(js) + len(rest)) + struct.pack("<II", len(js), 0x4E4F534A) + js + rest
        p = os.path.join(tmp, "n.glb")
        open(p, "wb").write(blob)
        h = H.Host(MOCK)
        h.resize(64, 64)
        try:
            h.load_gltf(p)
            n_ok += 1
        except H.HostError:
            n_bad += 1
        h.close()
print("hostile numbers: loaded", n_ok, "refused", n_bad)
# image decoders on garbage
for case in range(200):
    blob = bytes(rng.integers(0, 256, size=int(rng.integers(8, 400)), dtype=np.uint8))
    blob = (b"\x89PNG\r\n\x1a\n" if case % 2 else b"\xff\xd8\xff") + blob
    try:
        H.decode_image(blob)
    except H.HostError:
        pass
# real JPEG / PNG streams (baseline and progressive; every scan type of jpeg.hpp) with bit flips and truncations
try:
    import io
    from PIL import Image
    yy, xx = np.mgrid[0:40, 0:56]
    pic = np.stack([(xx * 4) % 256, (yy * 6) % 256, (xx * yy) % 256], axis=-1).astype(np.uint8)
    streams = []
    for kw in ({"format": "JPEG", "quality": 85, "subsampling": 2}, {"format": "JPEG", "quality": 85, "subsampling": 2, "progressive": True},
               {"format": "JPEG", "quality": 95, "subsampling": 0, "progressive": True, "restart_marker_rows": 1}, {"format": "PNG"}):
        b = io.BytesIO()
        Image.fromarray(pic, "RGB").save(b, **kw)
        streams.append(b.getvalue())
    n_ok = n_bad = 0
    for case in range(600):
        blob = bytearray(streams[case % len(streams)])
        for _ in range(int(rng.integers(1, 6))):
            blob[int(rng.integers(2, len(blob)))] ^= 1 << int(rng.integers(0, 8))
        if case % 5 == 0:
            blob = blob[:int(rng.integers(4, len(blob)))]
        try:
            H.decode_image(bytes(blob))
            n_ok += 1
        except H.HostError:
            n_bad += 1
    print("corrupted images: decoded", n_ok, "refused", n_bad)
except ImportError:
    print("corrupted images: PIL not importable here, skipped")
print("SANITIZE_OK")
