import sys; sys.path.insert(0, '.')
import numpy as np
from awsm_renderer_amd import scenes
from tests import helpers
from oracle import oracle_lib
lut = oracle_lib.brdf_lut(16, 16)
sc = scenes.skinned_morph_scene(480, 270, around=32, along=96, tex_size=64)
m = helpers.build_model(sc)
orc = helpers.oracle_frame(m, lut)
dev, st = helpers.hip_frame(m, lut)
clip, nt = dev.read_transformed(orc.n_verts)
bad = np.nonzero((nt.view(np.uint32) != orc.nt.view(np.uint32)).any(axis=1))[0]
print("n bad", len(bad), "of", orc.n_verts, "first", bad[:10])
cols = (nt.view(np.uint32) != orc.nt.view(np.uint32)).sum(axis=0)
print("per column mismatches", cols)
for i in bad[:6]:
    print(i, "gpu", nt[i], "\n   cpu", orc.nt[i], "\n   ulp", (nt[i].view(np.int32) - orc.nt[i].view(np.int32)))
