"""What fraction of a frame's (triangle, tile) bin entries could k_raster_tile drop at classification with a 4x4-block max-depth grid per tile, if it
processed a tile's list in rank order (front to back by mesh) in sub-batches of B and tested every later triangle's box against the blocks whose
pixels are already final?  CPU only (the oracle's keys); an upper bound of the scheme, not of occlusion culling.  python tests/diagnostics/hiz_sim.py 3840 2160"""
import sys, numpy as np, time
sys.path.insert(0, '.')
from awsm_renderer_amd import scenes
from tests import helpers
from oracle import oracle_lib
W, H = int(sys.argv[1]), int(sys.argv[2])
sc = scenes.atrium_scene(W, H)
model = helpers.build_model(sc)
lut = np.zeros((4, 4, 2), dtype=np.float16)
fr = oracle_lib.frame_from_model(model, lut)
t0 = time.time(); fr.transform().raster(8); print('oracle raster', round(time.time() - t0, 1), 's')
keys = fr.keys
hit = keys != np.uint64(0xFFFFFFFFFFFFFFFF)
rank_px = (0xFFFFFFFF - (keys & np.uint64(0xFFFFFFFF))).astype(np.int64)
depth_px = (keys >> np.uint64(32)).astype(np.uint32).view(np.float32).astype(np.float64)
depth_px = np.where(hit, depth_px, np.inf)
clip = fr.clip.reshape(-1, 3, 4).astype(np.float64)
w = clip[..., 3]
ok = (w > 1e-6).all(axis=1)
x = (clip[..., 0] / np.where(w > 1e-6, w, 1) * 0.5 + 0.5) * W; y = (1 - (clip[..., 1] / np.where(w > 1e-6, w, 1) * 0.5 + 0.5)) * H
z = clip[..., 2] / np.where(w > 1e-6, w, 1)
x0 = np.floor(x.min(1)).astype(int); x1 = np.ceil(x.max(1)).astype(int); y0 = np.floor(y.min(1)).astype(int); y1 = np.ceil(y.max(1)).astype(int)
zmin = z.min(1)
area2 = (x[:, 1] - x[:, 0]) * (y[:, 2] - y[:, 0]) - (x[:, 2] - x[:, 0]) * (y[:, 1] - y[:, 0])
onscreen = ok & (x1 >= 0) & (x0 < W) & (y1 >= 0) & (y0 < H) & (area2 != 0)
# (no back-face information here: the binned set is a subset; hidden-by-culling triangles inflate both numerator and denominator)
TX, TY = (W + 31) // 32, (H + 31) // 32
# block grid (4x4 px) of final depth max and of "latest winner rank" max per block
BH, BW = (H + 3) // 4, (W + 3) // 4
pad = np.full((BH * 4, BW * 4), np.inf); pad[:H, :W] = depth_px
padr = np.full((BH * 4, BW * 4), 1 << 40, dtype=np.int64); padr[:H, :W] = np.where(hit, rank_px, 1 << 40)
zblk = pad.reshape(BH, 4, BW, 4).max(axis=(1, 3))
rblk = padr.reshape(BH, 4, BW, 4).max(axis=(1, 3))      # the block is "complete" once every winner of its pixels has been processed
vis_tris = np.unique(rank_px[hit])
sign = np.sign(np.median(area2[vis_tris]))            # the facing the visible triangles have (the scene's materials cull back faces)
onscreen &= np.sign(area2) == sign
print('front-facing on-screen triangles', int(onscreen.sum()), 'visible', len(vis_tris))
tris = np.nonzero(onscreen)[0]
entries = 0; culled = {16: 0, 32: 0, 64: 0, 128: 0}; visible_pairs = 0
import collections
per_tile = collections.defaultdict(list)
for t in tris:
    for ty in range(max(y0[t], 0) // 32, min(y1[t], H - 1) // 32 + 1):
        for tx in range(max(x0[t], 0) // 32, min(x1[t], W - 1) // 32 + 1):
            per_tile[(ty, tx)].append(t)
for (ty, tx), lst in per_tile.items():
    lst = np.array(sorted(lst)); n = len(lst); entries += n
    for B in culled:
        for i in range(B, n):
            t = lst[i]; start = lst[(i // B) * B]        # first rank of this triangle's sub-batch: everything before it has been rasterised
            bx0 = max(x0[t], tx * 32) // 4; bx1 = min(x1[t], tx * 32 + 31, W - 1) // 4; by0 = max(y0[t], ty * 32) // 4; by1 = min(y1[t], ty * 32 + 31, H - 1) // 4
            zb = zblk[by0:by1 + 1, bx0:bx1 + 1]; rb = rblk[by0:by1 + 1, bx0:bx1 + 1]
            if ((rb < start) & (zb < zmin[t])).all(): culled[B] += 1
print('tiles', len(per_tile), 'entries (front-facing on-screen triangles)', entries, {B: round(c / entries, 3) for B, c in culled.items()})
