import sys; sys.path.insert(0,'.')
import numpy as np
from awsm_renderer_amd import scenes
from tests import helpers
from oracle import oracle_lib
lut = oracle_lib.brdf_lut(16,16)
sc = scenes.atrium_scene(641, 363, detail=0.25, tex_scale=1 / 32)
model = helpers.build_model(sc)
dev, _ = helpers.hip_frame(model, lut, msaa=4)
full_f32, full_keys = dev.read_opaque_f32(), dev.read_visibility()
for rows in [(0,121),(121,250)]:
    dev.set_shard_rows(*rows)
    dev.geometry_pass(model.collect_draws()); dev.opaque_pass(); st = dev.frame_end()
    y0,y1=rows
    a=dev.read_opaque_f32()
    bad=(a[y0:y1].view(np.uint32)!=full_f32[y0:y1].view(np.uint32)).any(axis=2)
    ys,xs=np.nonzero(bad)
    print(rows,"bad px",bad.sum(),"rows",np.unique(ys+y0)[:20], "xs", xs[:10])
    if bad.sum():
        y,x=ys[0]+y0,xs[0]
        print(" first", (y,x), a[y,x], full_f32[y,x], "keys same", (dev.read_visibility()[y-1:y+2,x-1:x+2]==full_keys[y-1:y+2,x-1:x+2]).all(axis=2))
