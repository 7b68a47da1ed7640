#!/usr/bin/env python3
"""Writes tests/golden/oracle_digests.json: SHA-256 of the CPU oracle's outputs for small seeded scenes (see README.md).
Run from the repository root:  python tests/golden/make_oracle_digests.py"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from awsm_renderer_amd import scenes  # noqa: E402
from oracle import oracle_lib  # noqa: E402
from tests import helpers  # noqa: E402

CASES = {
    "box_64": (lambda: scenes.box_scene(64, 64), {}),
    "helmet_96x64": (lambda: scenes.helmet_scene(96, 64, segments=16, rings=12, tex_size=16), {}),
    "skinned_morph_80x64_msaa4": (lambda: scenes.skinned_morph_scene(80, 64, around=8, along=12, tex_size=16), {"msaa": 4}),
    "zoo_120x80_mips": (lambda: scenes.material_zoo_scene(120, 80, tex_size=16), {"mipmap": True}),
    "transparent_120x72": (lambda: scenes.transparent_scene(120, 72, tex_size=16), {"transparent": True}),
}


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def digests():
    lut = oracle_lib.brdf_lut(16, 16)
    out = {"brdf_lut_16": sha(lut)}
    for name, (make, kw) in CASES.items():
        model = helpers.build_model(make())
        fr = helpers.oracle_frame(model, lut, msaa=kw.get("msaa", 0), mipmap=kw.get("mipmap", False))
        d = {"clip": sha(fr.clip), "normal_tangent": sha(fr.nt), "keys": sha(fr.keys), "rgba16f": sha(fr.rgba16f)}
        if kw.get("mipmap"):
            d["mip_chain_0"] = sha(fr.mip_chains[0][0])
        if kw.get("transparent"):
            fr.forward(model.collect_transparent_draws())
            d["forward_clip"] = sha(fr.fwd_clip)
            d["composite16f"] = sha(fr.composite16f)
        out[name] = d
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_digests.json")
    with open(path, "w") as f:
        json.dump(digests(), f, indent=1, sort_keys=True)
    print("wrote", path)
