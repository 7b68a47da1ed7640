#!/usr/bin/env python3
"""Render one of the synthetic scenes through the host layer + HIP kernels and save it as a PNG (needs an MI355X).

    python examples/render_png.py {box,helmet,skinned,atrium,zoo,instanced,transparent} out.png [--width W --height H --msaa 4 --mipmap]

The output image of the opaque pass is linear HDR RGBA16F; the PNG is Reinhard tone-mapped and gamma-encoded for viewing.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from awsm_renderer_amd import scenes                      # noqa: E402
from awsm_renderer_amd.hip_backend import HipDevice       # noqa: E402
from awsm_renderer_amd.host import Renderer               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene", choices=["box", "helmet", "skinned", "atrium", "zoo", "instanced", "transparent"])
    ap.add_argument("out")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--msaa", type=int, default=0, choices=(0, 4))
    ap.add_argument("--mipmap", action="store_true")
    ap.add_argument("--via-glb", action="store_true", help="write the scene to a .glb next to the output and render from the file (native glTF reader)")
    a = ap.parse_args()
    W, H = a.width, a.height
    sc = {"box": lambda: scenes.box_scene(W, H), "helmet": lambda: scenes.helmet_scene(W, H), "skinned": lambda: scenes.skinned_morph_scene(W, H),
          "atrium": lambda: scenes.atrium_scene(W, H, tex_scale=0.5), "zoo": lambda: scenes.material_zoo_scene(W, H),
          "instanced": lambda: scenes.instanced_scene(W, H), "transparent": lambda: scenes.transparent_scene(W, H, tex_size=256)}[a.scene]()
    gltf = None
    if a.via_glb:
        from awsm_renderer_amd import gltf_export
        gltf = os.path.splitext(a.out)[0] + ".glb"
        gltf_export.write_glb(sc, gltf)
    r = Renderer(sc, msaa=a.msaa, mipmap=a.mipmap, gltf=gltf)
    stats = r.render(sync=True)
    dev = HipDevice.from_ctx(r.host.device_ctx, W, H)
    final = dev.read_composite() if stats["forward_triangles"] else dev.read_opaque()      # the image after the transparent pass, when the scene has one
    img = final.view(np.float16).astype(np.float32)[..., :3]
    r.close()
    ldr = np.clip(img / (1.0 + img), 0.0, 1.0) ** (1.0 / 2.2)
    from PIL import Image
    Image.fromarray((ldr * 255.0 + 0.5).astype(np.uint8)).save(a.out)
    print(f"{a.out}: {W}x{H}, {stats['triangles_in']} triangles, {stats['covered_pixels']} covered pixels, "
          f"geometry {stats['ms_transform'] + stats['ms_bin'] + stats['ms_raster']:.3f} ms, opaque {stats['ms_shade']:.3f} ms, transparent {stats['ms_forward']:.3f} ms")


if __name__ == "__main__":
    main()
