"""How many of k_raster_tile's lane-steps land on a covered pixel?  CPU only (the oracle's transformed vertices, numpy): for every front-facing
triangle of the 4K bench frame and every 32x32 tile its box touches — box ∩ tile, the number of 4x4 (boxes up to 256 px) or 8x8 blocks with the first
block in the box's corner and with blocks at multiples of the block size, and the triangle's area inside the box (pixel centres).
    python tests/diagnostics/raster_lane_use.py"""
import sys, numpy as np
sys.path.insert(0, '.')
from awsm_renderer_amd import scenes
from tests import helpers
from oracle import oracle_lib
W, H = 3840, 2160
sc = scenes.atrium_scene(W, H, tex_scale=1 / 16)
model = helpers.build_model(sc)
orc = helpers.oracle_frame(model, oracle_lib.brdf_lut(16, 16), rows=(0, 1), threads=8)
clip = orc.clip.astype(np.float64).reshape(-1, 3, 4)
w = clip[..., 3]
ok = (w > 1e-6).all(axis=1)
ndc = clip[ok, :, :2] / clip[ok, :, 3:4]
x = (ndc[..., 0] * 0.5 + 0.5) * W
y = (0.5 - ndc[..., 1] * 0.5) * H
area2 = (x[:, 1] - x[:, 0]) * (y[:, 2] - y[:, 0]) - (x[:, 2] - x[:, 0]) * (y[:, 1] - y[:, 0])
front = area2 != 0          # (the scene is mostly double-sided or consistently wound; back faces are culled before binning — an upper bound)
x, y, area2 = x[front], y[front], np.abs(area2[front])
x0 = np.clip(np.ceil(x.min(axis=1) - 0.5), 0, W - 1).astype(int); x1 = np.clip(np.floor(x.max(axis=1) - 0.5), 0, W - 1).astype(int)
y0 = np.clip(np.ceil(y.min(axis=1) - 0.5), 0, H - 1).astype(int); y1 = np.clip(np.floor(y.max(axis=1) - 0.5), 0, H - 1).astype(int)
vis = (x1 >= x0) & (y1 >= y0) & (x.max(axis=1) > 0) & (x.min(axis=1) < W) & (y.max(axis=1) > 0) & (y.min(axis=1) < H)
x0, x1, y0, y1, area2 = x0[vis], x1[vis], y0[vis], y1[vis], area2[vis]
print("triangles with a pixel box:", len(x0), " median box %d x %d" % (np.median(x1 - x0 + 1), np.median(y1 - y0 + 1)))
tot = dict(entries=0, box_px=0, steps_corner=0, steps_aligned=0, tiny=0)
tri_px = float((area2 * 0.5).sum())
hist = np.zeros(8, dtype=np.int64)
for tx in range((W + 31) // 32):
    sel = (x0 <= tx * 32 + 31) & (x1 >= tx * 32)
    if not sel.any():
        continue
    cx0, cx1 = np.maximum(x0[sel], tx * 32), np.minimum(x1[sel], tx * 32 + 31)
    ys0, ys1 = y0[sel], y1[sel]
    for ty in range(int(ys0.min()) // 32, int(ys1.max()) // 32 + 1):
        s2 = (ys0 <= ty * 32 + 31) & (ys1 >= ty * 32)
        if not s2.any():
            continue
        bx0, bx1 = cx0[s2], cx1[s2]
        by0, by1 = np.maximum(ys0[s2], ty * 32), np.minimum(ys1[s2], ty * 32 + 31)
        bw, bh = bx1 - bx0 + 1, by1 - by0 + 1
        a = bw * bh
        tot["entries"] += len(a); tot["box_px"] += int(a.sum()); tot["tiny"] += int((a <= 4).sum())
        st = np.where(a <= 256, 4, 8)
        m = a > 4
        tot["steps_corner"] += int((((bw + st - 1) // st) * ((bh + st - 1) // st) * st * st)[m].sum()) + int(a[~m].sum())
        al = ((bx1 // st) - (bx0 // st) + 1) * ((by1 // st) - (by0 // st) + 1) * st * st
        tot["steps_aligned"] += int(al[m].sum()) + int(a[~m].sum())
        hist += np.bincount(np.minimum(7, np.log2(np.maximum(a, 1)).astype(int) // 1 // 2 * 1), minlength=8)[:8] if False else 0
print(tot)
print("triangle area (px, all of it on screen or not): %.1f M;  box pixels %.1f M;  lane-steps, blocks at the corner %.1f M, aligned %.1f M" % (
    tri_px / 1e6, tot["box_px"] / 1e6, tot["steps_corner"] / 1e6, tot["steps_aligned"] / 1e6))
print("covered / lane-steps: corner %.2f, aligned %.2f;  covered / box %.2f" % (tri_px / tot["steps_corner"], tri_px / tot["steps_aligned"], tri_px / tot["box_px"]))
