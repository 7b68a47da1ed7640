import sys; sys.path.insert(0,'.')
import numpy as np
from awsm_renderer_amd import scenes
from awsm_renderer_amd.hip_backend import HipDevice
from tests import helpers
from oracle import oracle_lib
lut = oracle_lib.brdf_lut(16,16)
for name, sc in [("atrium", scenes.atrium_scene(641, 363, detail=0.25, tex_scale=1/32)), ("helmet", scenes.helmet_scene(320,180,segments=48,rings=36,tex_size=64)), ("skinned", scenes.skinned_morph_scene(320,200,around=16,along=24,tex_size=16))]:
    model = helpers.build_model(sc)
    orc = oracle_lib.frame_from_model(model, lut, msaa=4).transform().raster(8)
    dev = HipDevice(parity_tap=True)
    dev.resize(sc.width, sc.height, 4)
    dev.upload_mirrors(model.mirrors())
    dev.geometry_pass(model.collect_draws())
    st = dev.frame_end()
    keys = dev.read_visibility()
    print(name, keys.shape, "mismatch", int((keys != orc.keys).sum()), "covered samples", int((orc.keys != 0xFFFFFFFFFFFFFFFF).sum()), st["ms_raster"])
    dev.close()
