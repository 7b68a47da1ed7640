"""Debug aid: print the pixels of an MSAA frame whose shaded colour is out of tolerance, with their keys."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import helpers
from awsm_renderer_amd import scenes
from oracle import oracle_lib
lut = oracle_lib.brdf_lut(16, 16)
sc = scenes.material_zoo_scene(300, 200) if len(sys.argv) < 2 else eval(sys.argv[1])
model = helpers.build_model(sc)
orc = helpers.oracle_frame(model, lut, msaa=4)
dev, stats = helpers.hip_frame(model, lut, msaa=4)
f32 = dev.read_opaque_f32()
ref = orc.rgba32f.astype(np.float64)
diff = np.abs(f32.astype(np.float64) - ref)
bound = 1e-4 * np.maximum(1.0, np.abs(ref))
bad = np.argwhere((diff[..., :3] > bound[..., :3]).any(axis=-1))
keys = dev.read_visibility()
for y, x in bad[:10]:
    print("pixel", x, y, "dev", f32[y, x], "ref", orc.rgba32f[y, x])
    for s in range(4):
        k = int(keys[y, x, s])
        print("   sample", s, "rank", 0xFFFFFFFF - (k & 0xFFFFFFFF) if k != 0xFFFFFFFFFFFFFFFF else None, "depth", np.uint32(k >> 32).view(np.float32) if k != 0xFFFFFFFFFFFFFFFF else None)
dev.close()
