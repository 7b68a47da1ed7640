"""How stable is the sorted draw list under bench.py's camera orbit?  (CPU: the C++ host layer against the mock backend.)  The geometry cache keeps a draw's
outputs when it sits at the same place of the list as in the frame slot's previous frame (lag 2 with two slots).  Usage: python tools/draw_list_stability.py"""
import sys, math, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from awsm_renderer_amd import scenes
from awsm_renderer_amd import host as H
MOCK = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "mock", "libmock_backend.so")
W, Hh = 3840, 2160
scene = scenes.atrium_scene(W, Hh, detail=1.0, tex_scale=1/64)
r = H.Renderer(scene, backend_path=MOCK, lut_rgba16f=np.zeros((4, 4, 4), dtype=np.uint16))
r.render()
inv_view = np.linalg.inv(np.asarray(scene.view, dtype=np.float64).T)
eye0 = np.asarray(scene.camera_position, dtype=np.float64)
fwd = -inv_view[:3, 2]; right, up = inv_view[:3, 0], inv_view[:3, 1]
orbit_r = 0.02 * 30.0
lists = []
for i in range(260):
    a = 2.0 * math.pi * (i % 240) / 240.0
    eye = eye0 + orbit_r * (math.cos(a) * right + math.sin(a) * up)
    r.host.camera_update(scenes.look_at_rh(tuple(eye), tuple(eye + 30.0 * fwd)), scene.proj, tuple(eye))
    r.host.render()
    dl = r.host.draw_list()
    lists.append([(d["geom_meta_off"], d["vis_data_off"], d["tri_count"], d["flags"]) for d in dl])
def placed(l):
    out = {}; t = 0
    for i, d in enumerate(l):
        out[d] = (i, t); t += d[2]
    return out, t
for lag in (1, 2):
    same_tris = []; identical = 0; nd = []
    for i in range(lag + 10, len(lists)):
        a, ta = placed(lists[i]); b, tb = placed(lists[i - lag])
        s = sum(d[2] for d in lists[i] if d in b and b[d] == a[d])
        same_tris.append(s / ta); identical += lists[i] == lists[i - lag]; nd.append(len(lists[i]))
    print("lag", lag, "identical lists", identical, "of", len(same_tris), "mean frac tris in place", np.mean(same_tris), "min", np.min(same_tris), "draws", min(nd), max(nd))
    # same first_tri only (index may differ)
    st = []
    for i in range(lag + 10, len(lists)):
        a, ta = placed(lists[i]); b, tb = placed(lists[i - lag])
        st.append(sum(d[2] for d in lists[i] if d in b and b[d][1] == a[d][1]) / ta)
    print("   same first_tri only:", np.mean(st), np.min(st))
