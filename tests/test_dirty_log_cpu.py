"""CPU: the geometry cache's dirty log (awsm-renderer_amd/csrc/dirty_log.hpp, header-only; the C-ABI layer logs every awsm_hip_buffer_write / buffer_create in
it and hands k_deform_transform the ranges written since a frame slot's arrays were computed).  Driven through a small g++ program, and checked against a
model in Python: whatever the log answers must COVER every byte written after `since` to a buffer the vertex stage reads (it may cover more — merging,
bounding ranges — never less), "all" whenever a buffer was written whose every write invalidates every draw, and nothing of writes at or before `since`.
The writes themselves are what the reference's dirty-range writer produces (buffer/helpers.rs:124-220): 4-byte aligned ranges, merged, in offset order."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRANSFORMS, INSTANCES, GEOM_META, VIS_GEOM_DATA, MORPH_WEIGHTS, SKIN_MATRICES, MATERIAL_META = 0, 17, 10, 12, 8, 6, 11
ATTR_DATA, ATTR_INDEX, MORPH_VALUES, SKIN_INDEX_WEIGHTS = 14, 15, 9, 7
MATERIALS, LIGHTS, CAMERA, NORMAL_MATS = 2, 3, 5, 1
PRECISE = (TRANSFORMS, INSTANCES, GEOM_META, VIS_GEOM_DATA, MORPH_WEIGHTS, SKIN_MATRICES, MATERIAL_META)
GLOBAL = (ATTR_DATA, ATTR_INDEX, MORPH_VALUES, SKIN_INDEX_WEIGHTS)


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = tmp_path_factory.mktemp("dirty_log") / "driver"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-fsanitize=address,undefined", "-o", str(exe), os.path.join(ROOT, "tests", "native", "dirty_log_driver.cpp")], check=True)

    def run(script):
        p = subprocess.run([str(exe)], input=script, capture_output=True, text=True)
        assert p.returncode == 0 and not p.stderr, p.stderr[-2000:]
        return p.stdout.splitlines()
    return run


def parse(line):
    if line == "all":
        return None
    t = line.split()
    assert t[0] == "ok"
    v = list(map(int, t[2:]))
    assert len(v) == 3 * int(t[1])
    return [(v[i], v[i + 1], v[i + 2]) for i in range(0, len(v), 3)]


def test_ranges_merge_and_respect_the_sequence_number(driver):
    out = driver("\n".join([
        "w %d 640 704 1" % TRANSFORMS,          # one node's mat4
        "w %d 704 768 2" % TRANSFORMS,          # the next slot, next write: adjacent -> one range
        "w %d 4096 4160 3" % TRANSFORMS,        # far away: its own range
        "w %d 0 512 4" % CAMERA,                # ignored
        "w %d 256 296 5" % GEOM_META,
        "q 0 12", "q 2 12", "q 3 12", "q 5 12", "s",
        "p 3", "s", "q 0 12",
    ]))
    assert parse(out[0]) == [(TRANSFORMS, 640, 768), (TRANSFORMS, 4096, 4160), (GEOM_META, 256, 296)]
    # the merged range carries the newer number: a reader at 1 (who has seen the first write only) still gets it, whole
    assert parse(out[1]) == [(TRANSFORMS, 4096, 4160), (GEOM_META, 256, 296)]
    assert parse(out[2]) == [(GEOM_META, 256, 296)] and parse(out[3]) == []
    assert out[4] == "3" and out[5] == "1" and parse(out[6]) == [(GEOM_META, 256, 296)]


def test_global_buffers_and_overflow_mean_everything(driver):
    out = driver("\n".join(["w %d 0 64 1" % TRANSFORMS, "w %d 1024 2048 2" % ATTR_DATA, "q 0 12", "q 1 12", "q 2 12", "w %d 64 128 3" % TRANSFORMS, "q 2 12"]))
    assert parse(out[0]) is None and parse(out[1]) is None            # the attribute write is after both
    assert parse(out[2]) == [] and parse(out[3]) == [(TRANSFORMS, 64, 128)]
    # more ranges than the kernel's argument block holds: one bounding range per buffer; more buffers than that: everything
    many = ["w %d %d %d %d" % (TRANSFORMS, 1024 * i, 1024 * i + 64, i + 1) for i in range(20)] + ["q 0 12", "q 0 1"] + \
           ["w %d 0 40 21" % GEOM_META, "q 0 1", "q 0 2"]
    out = driver("\n".join(many))
    assert parse(out[0]) == [(TRANSFORMS, 0, 1024 * 19 + 64)] and parse(out[1]) == [(TRANSFORMS, 0, 1024 * 19 + 64)]
    assert parse(out[2]) is None and parse(out[3]) == [(TRANSFORMS, 0, 1024 * 19 + 64), (GEOM_META, 0, 40)]
    # the log itself is bounded: past its capacity everything before counts as written
    flood = ["w %d %d %d %d" % (VIS_GEOM_DATA, 4096 * i, 4096 * i + 168, i + 1) for i in range(600)] + ["s", "q 0 12", "q 599 12", "q 600 12"]
    out = driver("\n".join(flood))
    assert int(out[0]) < 512 and parse(out[1]) is None and parse(out[3]) == []


def test_random_writes_are_always_covered(driver):
    """2,000 random writes over all buffers, with readers at random sequence numbers and argument-block sizes: the answer covers every byte the model says
    was written after the reader's number to a buffer the vertex stage reads, and mentions no buffer that was not written after it."""
    rng = np.random.default_rng(5)
    bufs = list(PRECISE) + list(GLOBAL) + [MATERIALS, LIGHTS, CAMERA, NORMAL_MATS]
    writes, script = [], []
    for seq in range(1, 401):
        buf = int(rng.choice(bufs, p=np.array([6] * 7 + [0.15] * 4 + [2] * 4) / (42 + 0.6 + 8)))
        lo = int(rng.integers(0, 1 << 14)) * 4
        hi = lo + int(rng.integers(1, 64)) * 4
        if rng.random() < 0.3 and writes and writes[-1][0] == buf:      # the writer's next merged range of the same mirror: right behind the previous one
            lo = writes[-1][2]; hi = lo + 64
        writes.append((buf, lo, hi, seq))
        script.append("w %d %d %d %d" % (buf, lo, hi, seq))
        if seq % 7 == 0:
            since, mx = int(rng.integers(0, seq + 1)), int(rng.choice([1, 3, 12]))
            script.append("q %d %d" % (since, mx))
            writes.append(("q", since, mx, seq))
        if seq % 50 == 0:
            script.append("p %d" % (seq - 45))
            writes.append(("p", seq - 45, 0, seq))
    out = driver("\n".join(script))
    k, pruned_to = 0, 0
    for i, w in enumerate(writes):
        if w[0] == "p":
            pruned_to = max(pruned_to, w[1])
        if w[0] != "q":
            continue
        since, mx = w[1], w[2]
        got = parse(out[k]); k += 1
        if since < pruned_to:
            continue                      # (a reader older than a prune has been promised nothing: the C-ABI layer prunes only what every slot has seen)
        after = [x for x in writes[:i] if x[0] not in ("q", "p") and x[3] > since]
        if any(x[0] in GLOBAL for x in after):
            assert got is None, (w, got)
            continue
        need = [x for x in after if x[0] in PRECISE]
        if got is None:
            assert len({x[0] for x in need}) > mx or any(x[0] in GLOBAL for x in writes[:i] if x[0] not in ("q", "p") and x[3] > since), (w, need)
            continue
        assert len(got) <= mx
        for buf, lo, hi, _ in need:
            assert any(g[0] == buf and g[1] <= lo and hi <= g[2] for g in got), (w, (buf, lo, hi), got)
        assert {g[0] for g in got} <= {x[0] for x in need}, (w, got)
    assert k == len(out)
