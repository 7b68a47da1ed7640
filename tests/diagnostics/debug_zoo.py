import sys; sys.path.insert(0, '.')
import numpy as np
from awsm_renderer_amd import scenes
from tests import helpers
from oracle import oracle_lib
lut = oracle_lib.brdf_lut(64, 64)
sc = scenes.material_zoo_scene(640, 360)
m = helpers.build_model(sc)
orc = helpers.oracle_frame(m, lut)
dev, st = helpers.hip_frame(m, lut)
res = helpers.compare_frames(orc, dev)
print(res)
f32 = dev.read_opaque_f32()
diff = np.abs(f32[..., :3].astype(np.float64) - orc.rgba32f[..., :3]).max(axis=-1)
tri, meta, depth = orc.unpack_visibility()
bad = diff > 1e-4
print("bad pixels", bad.sum())
metas = np.unique(meta[bad], return_counts=True)
print("by material meta offset:", dict(zip(metas[0].tolist(), metas[1].tolist())))
draws = m.collect_draws()
import struct
for off in metas[0][:16]:
    mm = m.material_meta.raw[off:off+68]
    mat_off = struct.unpack_from("<I", mm, 24)[0]
    print("meta", off, "material offset", mat_off, "-> material index", [i for i,k in m.material_keys_by_index.items() if m.materials.offset(k)==mat_off])
ys, xs = np.nonzero(bad)
for y, x in list(zip(ys, xs))[:5]:
    print((y, x), "gpu", f32[y, x], "cpu", orc.rgba32f[y, x])
