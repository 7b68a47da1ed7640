"""How much of k_raster_tile's walk phase is lost to the lockstep of the four 16-lane groups of a wavefront?  CPU only.  Per tile and batch, the mid
triangles' block counts (4x4 blocks from the box's corner) are dealt to 16 groups the way the kernel does it (first triangle = group number, then a
shared counter), and a wavefront's cost per outer iteration is the nested maximum (rows, then columns) over its four groups.  Compared with: the
same deal with a flat block loop (cost = max of the groups' block counts), the flat loop over lists sorted by block count, and the ideal (sum / 16).
    python tests/diagnostics/raster_group_balance.py [batch]"""
import sys, heapq, numpy as np
sys.path.insert(0, '.')
from awsm_renderer_amd import scenes
from tests import helpers
from oracle import oracle_lib
W, H = 3840, 2160
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sc = scenes.atrium_scene(W, H, tex_scale=1 / 16)
model = helpers.build_model(sc)
orc = helpers.oracle_frame(model, oracle_lib.brdf_lut(16, 16), rows=(0, 1), threads=8)
clip = orc.clip.astype(np.float64).reshape(-1, 3, 4)
ok = (clip[..., 3] > 1e-6).all(axis=1)
idx = np.nonzero(ok)[0]
ndc = clip[ok, :, :2] / clip[ok, :, 3:4]
x = (ndc[..., 0] * 0.5 + 0.5) * W
y = (0.5 - ndc[..., 1] * 0.5) * H
area2 = (x[:, 1] - x[:, 0]) * (y[:, 2] - y[:, 0]) - (x[:, 2] - x[:, 0]) * (y[:, 1] - y[:, 0])
front = area2 < 0 if (area2 < 0).sum() > (area2 > 0).sum() else area2 > 0      # the majority winding is the front one (most materials are single-sided)
x, y = x[front], y[front]
x0 = np.clip(np.ceil(x.min(axis=1) - 0.5), 0, W - 1).astype(int); x1 = np.clip(np.floor(x.max(axis=1) - 0.5), 0, W - 1).astype(int)
y0 = np.clip(np.ceil(y.min(axis=1) - 0.5), 0, H - 1).astype(int); y1 = np.clip(np.floor(y.max(axis=1) - 0.5), 0, H - 1).astype(int)
vis = (x1 >= x0) & (y1 >= y0) & (x.max(axis=1) > 0) & (x.min(axis=1) < W) & (y.max(axis=1) > 0) & (y.min(axis=1) < H)
x0, x1, y0, y1 = x0[vis], x1[vis], y0[vis], y1[vis]
print("front-facing triangles with a pixel box:", len(x0))

def simulate(cols, rows, mode):
    """16 groups = 4 wavefronts x 4; returns the walk phase's length in block steps (the slowest wavefront)."""
    n = len(cols)
    if mode == "sorted":
        o = np.argsort(-(cols * rows), kind="stable"); cols, rows = cols[o], rows[o]
    nxt = 16
    cur = [[g + 4 * w if g + 4 * w < n else -1 for g in range(4)] for w in range(4)]      # tid >> 4 = group number 0..15: wave w holds groups 4w..4w+3
    cur = [[4 * w + g if 4 * w + g < n else -1 for g in range(4)] for w in range(4)]
    t = [0.0] * 4
    heap = [(0.0, w) for w in range(4)]
    heapq.heapify(heap)
    while heap:
        tw, w = heapq.heappop(heap)
        act = [j for j in cur[w] if j >= 0]
        if not act:
            t[w] = tw
            continue
        if mode == "nested":
            mr = max(rows[j] for j in act)
            cost = sum(max(cols[j] for j in act if rows[j] > r) for r in range(mr))
        else:
            cost = max(cols[j] * rows[j] for j in act)
        for g in range(4):
            if cur[w][g] >= 0:
                cur[w][g] = nxt if nxt < n else -1
                nxt += 1
        heapq.heappush(heap, (tw + cost + 1.0, w))      # + 1: fetching the next triangle
    return max(t)

res = dict(nested=0.0, flat=0.0, sorted=0.0, ideal=0.0, blocks=0)
tiles = 0
rng = np.random.default_rng(1)
for tx in range(0, (W + 31) // 32):
    sel = (x0 <= tx * 32 + 31) & (x1 >= tx * 32)
    if not sel.any():
        continue
    cx0, cx1 = np.maximum(x0[sel], tx * 32), np.minimum(x1[sel], tx * 32 + 31)
    ys0, ys1 = y0[sel], y1[sel]
    for ty in range(int(ys0.min()) // 32, int(ys1.max()) // 32 + 1):
        s2 = (ys0 <= ty * 32 + 31) & (ys1 >= ty * 32)
        if not s2.any():
            continue
        bw = cx1[s2] - cx0[s2] + 1
        bh = np.minimum(ys1[s2], ty * 32 + 31) - np.maximum(ys0[s2], ty * 32) + 1
        a = bw * bh
        tiles += 1
        for b0 in range(0, len(a), BATCH):
            m = (a[b0:b0 + BATCH] > 4) & (a[b0:b0 + BATCH] <= 256)
            if not m.any():
                continue
            cols, rows = ((bw[b0:b0 + BATCH][m] + 3) // 4), ((bh[b0:b0 + BATCH][m] + 3) // 4)
            for k in ("nested", "flat", "sorted"):
                res[k] += simulate(cols, rows, k)
            for cap in (2, 4, 8):      # every triangle in parts of whole block rows, at most `cap` blocks each (at least one row)
                pc, pr = [], []
                for c, r in zip(cols, rows):
                    per = max(1, cap // int(c))
                    for r0 in range(0, int(r), per):
                        pc.append(int(c)); pr.append(min(per, int(r) - r0))
                res["split%d" % cap] = res.get("split%d" % cap, 0.0) + simulate(np.array(pc), np.array(pr), "nested")
                res["items%d" % cap] = res.get("items%d" % cap, 0) + len(pc)
            res["tris"] = res.get("tris", 0) + len(cols)
            res["ideal"] += (cols * rows).sum() / 16.0
            res["blocks"] += int((cols * rows).sum())
print("tiles", tiles, "mid blocks %.2f M" % (res["blocks"] / 1e6))
print("walk-phase length in block steps, summed over tiles and batches: nested (now) %.0f k, flat loop %.0f k, flat + sorted by block count %.0f k, ideal %.0f k" % (
    res["nested"] / 1e3, res["flat"] / 1e3, res["sorted"] / 1e3, res["ideal"] / 1e3))
print("mid triangles %d;  in parts of whole block rows with at most 2 / 4 / 8 blocks: %.0f k / %.0f k / %.0f k steps (%d / %d / %d items; every item costs one step for its fetch)" % (
    res["tris"], res["split2"] / 1e3, res["split4"] / 1e3, res["split8"] / 1e3, res["items2"], res["items4"], res["items8"]))
