"""(1) The reference's frustum known-answer tests restated one for one
(/root/reference/crates/renderer/src/frustum/tests.rs:19-88, 4 tests), against the Python restatement and the C++ host.
(2) write_buffer_with_dirty_ranges' plan (buffer/helpers.rs:124-220): the reference has no tests for it; these cases pin the
documented rules (32-range cap, 60 % threshold, sort + coalesce of overlapping/adjacent ranges) and py == cpp."""
import math

import numpy as np
import pytest

from oracle import host_mirror as hm
from tests.buffer_adapters import IMPLS


@pytest.fixture(params=["py", "cpp"])
def H(request):
    return IMPLS[request.param][2]


def translation(x, y, z):
    m = hm.mat4_identity()
    m[3][:3] = [x, y, z]
    return m


def cube(w, h):
    a = hm.Aabb.new_cube(w, h)
    return a.min, a.max


def transformed(H, mat, box):
    return H.aabb_transformed(mat, box[0], box[1])


def instance_union(H, base, base_world, instance_translations):
    lo = hi = None
    for t in instance_translations:
        world = hm.mat4_mul(base_world, hm.mat4_from_srt((1, 1, 1), (0, 0, 0, 1), t))
        mn, mx = transformed(H, world, base)
        lo = mn if lo is None else np.minimum(lo, mn)
        hi = mx if hi is None else np.maximum(hi, mx)
    return lo, hi


def test_perspective_frustum_culls_non_instanced(H):  # tests.rs:19
    vp = hm.mat4_mul(hm.perspective_rh(math.radians(90.0), 1.0, 1.0, 10.0), hm.mat4_identity())
    base = cube(1.0, 1.0)
    inside = transformed(H, translation(0, 0, -5), base)
    outside = transformed(H, translation(0, 0, 5), base)
    assert H.frustum_intersects(vp, *inside)
    assert not H.frustum_intersects(vp, *outside)


def test_perspective_frustum_culls_instanced_union(H):  # tests.rs:34
    vp = hm.mat4_mul(hm.perspective_rh(math.radians(60.0), 1.0, 1.0, 20.0), hm.mat4_identity())
    base = cube(1.0, 1.0)
    inst = [(0, 0, 0), (100, 0, 0)]
    assert H.frustum_intersects(vp, *instance_union(H, base, translation(0, 0, -5), inst))
    assert not H.frustum_intersects(vp, *instance_union(H, base, translation(0, 0, 5), inst))


def test_orthographic_frustum_culls_non_instanced(H):  # tests.rs:54
    vp = hm.mat4_mul(hm.orthographic_rh(-2, 2, -2, 2, 1, 10), hm.mat4_identity())
    base = cube(1.0, 1.0)
    assert H.frustum_intersects(vp, *transformed(H, translation(0, 0, -5), base))
    assert not H.frustum_intersects(vp, *transformed(H, translation(3, 0, -5), base))


def test_orthographic_frustum_culls_instanced_union(H):  # tests.rs:68
    vp = hm.mat4_mul(hm.orthographic_rh(-2, 2, -2, 2, 1, 10), hm.mat4_identity())
    base = cube(1.0, 1.0)
    assert H.frustum_intersects(vp, *instance_union(H, base, translation(0, 0, -5), [(0, 0, 0), (0, 5, 0)]))
    assert not H.frustum_intersects(vp, *instance_union(H, base, translation(0, 0, -5), [(5, 0, 0), (6, 0, 0)]))


# ------------------------------------------------------------------------------------------------ write plan
def test_write_plan_empty(H):
    assert H.write_plan(1024, []) == []
    assert H.write_plan(0, [(0, 4)]) == []


def test_write_plan_single_small_range(H):
    assert H.write_plan(1024, [(64, 16)]) == [(64, 16)]


def test_write_plan_threshold_is_60_percent_inclusive(H):
    assert H.write_plan(1000, [(0, 600)]) == [(0, 1000)]      # 600*100 >= 1000*60 -> full write
    assert H.write_plan(1000, [(0, 596)]) == [(0, 596)]
    assert H.write_plan(1000, [(0, 300), (500, 300)]) == [(0, 1000)]   # the sum counts, even when disjoint


def test_write_plan_more_than_32_ranges_is_a_full_write(H):
    ranges = [(i * 64, 4) for i in range(33)]
    assert H.write_plan(1 << 20, ranges) == [(0, 1 << 20)]
    assert len(H.write_plan(1 << 20, ranges[:32])) == 32


def test_write_plan_sorts_and_coalesces_overlapping_and_adjacent(H):
    got = H.write_plan(1 << 16, [(512, 64), (0, 16), (16, 16), (540, 100), (4096, 8)])
    assert got == [(0, 32), (512, 128), (4096, 8)]


def test_write_plan_py_equals_cpp_on_random_inputs():
    rng = np.random.default_rng(7)
    py, cpp = IMPLS["py"][2], IMPLS["cpp"][2]
    for _ in range(300):
        raw_len = int(rng.integers(1, 1 << 16)) * 4
        n = int(rng.integers(0, 40))
        ranges = []
        for _ in range(n):
            off = int(rng.integers(0, raw_len // 4)) * 4
            size = min(int(rng.integers(1, 64)) * 4, raw_len - off)
            if size:
                ranges.append((off, size))
        assert py.write_plan(raw_len, ranges) == cpp.write_plan(raw_len, ranges)
