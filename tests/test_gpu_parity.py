"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same seeded scenes.

Bar (BASELINE.json north_star): triangle-id + depth bit-exact; shaded RGB within 1e-4 absolute (checked on the
f32 parity tap; the RGBA16F image must be within 1 f16 ulp of the oracle's, since 1e-4 is below half an f16 ulp
for values above 0.125)."""
import math

import numpy as np
import pytest

from awsm_renderer_amd import scenes
from tests import helpers

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-4


def _check(scene, lut, rows=(0, 0)):
    model = helpers.build_model(scene)
    orc = helpers.oracle_frame(model, lut, rows=rows)
    dev, stats = helpers.hip_frame(model, lut, rows=rows)
    r = helpers.compare_frames(orc, dev, rows=None if rows == (0, 0) else rows, rgb_tol=RGB_TOL)
    dev.close()
    assert r["clip_mismatch"] == 0 and r["nt_mismatch"] == 0, r
    assert r["key_mismatch"] == 0, r
    assert r["covered"] > 0, r
    assert stats["covered_pixels"] == r["covered"], (stats, r)
    assert r["rgb_over_tol"] == 0, r
    assert r["alpha_mismatch"] == 0, r
    assert r["f16_max_ulp"] <= 1, r
    return r, stats


def test_box(oracle_lut):
    _check(scenes.box_scene(256, 256), oracle_lut)


def test_helmet_small(oracle_lut):
    _check(scenes.helmet_scene(480, 270, segments=64, rings=48, tex_size=128), oracle_lut)


def test_skinned_morph_small(oracle_lut):
    _check(scenes.skinned_morph_scene(480, 270, around=32, along=96, tex_size=64), oracle_lut)


def test_atrium_small(oracle_lut):
    _check(scenes.atrium_scene(640, 360, detail=0.25, tex_scale=1 / 16), oracle_lut)


def test_atrium_odd_size_and_shard(oracle_lut):
    # width/height not multiples of the 32-px tile or the 16-px shade block; shard rows [64, 201)
    _check(scenes.atrium_scene(333, 201, detail=0.125, tex_scale=1 / 32), oracle_lut)
    _check(scenes.atrium_scene(333, 201, detail=0.125, tex_scale=1 / 32), oracle_lut, rows=(64, 201))


def test_empty_pipeline_is_skybox_only(oracle_lut):
    scene = scenes.box_scene(128, 96)
    scene.skybox_rgba = (0.25, 0.5, 0.75, 1.0)
    model = helpers.build_model(scene)
    dev, _ = helpers.hip_frame(model, oracle_lut, has_opaque=False)
    img = dev.read_opaque_f32()
    dev.close()
    assert np.all(img == np.array([0.25, 0.5, 0.75, 1.0], dtype=np.float32))
