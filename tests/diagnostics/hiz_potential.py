"""How much of the raster kernel's work is spent on triangles that end up invisible in their tile?  Renders the 4K bench frame on the device, reads the
keys back and counts the (triangle, tile) pairs that win at least one pixel against the bin entries the frame produced.  Run on an MI355X:
    python tests/diagnostics/hiz_potential.py"""
import sys, numpy as np
sys.path.insert(0, '.')
from awsm_renderer_amd import scenes
from tests import helpers
from oracle import oracle_lib
sc = scenes.atrium_scene(3840, 2160)
model = helpers.build_model(sc)
lut = oracle_lib.brdf_lut(16, 16)
dev, stats = helpers.hip_frame(model, lut)
keys = dev.read_visibility()
H, W = keys.shape[:2]
rank = (0xFFFFFFFF - (keys & np.uint64(0xFFFFFFFF))).astype(np.int64)
hit = keys != np.uint64(0xFFFFFFFFFFFFFFFF)
ty, tx = np.mgrid[0:H, 0:W]
tile = (ty // 32) * ((W + 31) // 32) + (tx // 32)
pairs = np.unique(rank[hit] * 100000 + tile[hit])
print('bin entries', stats['bin_entries'], 'triangles binned', stats['triangles_binned'], '(triangle, tile) pairs that win a pixel', len(pairs), 'visible triangles', len(np.unique(rank[hit])))
