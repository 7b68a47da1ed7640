"""tests/golden: the reference's own known-answer tables for the pinned allocators, and the oracle's digests (README.md there)."""
import importlib.util
import json
import os

import pytest

from tests.buffer_adapters import IMPLS

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def known():
    return json.load(open(os.path.join(GOLDEN, "reference_known_answers.json")))


@pytest.mark.parametrize("impl", ["py", "cpp"])
def test_reference_known_answer_tables(impl, known):
    DUB, DSB, h = IMPLS[impl]
    ds = known["dynamic_storage"]
    for n, want in ds["round_pow2"]["rows"]:
        assert h.round_pow2(n) == want, ds["round_pow2"]["cite"]
    for off, want in ds["offset_to_index_4_leaves"]["rows"]:
        assert h.offset_to_index(off, 4) == want, ds["offset_to_index_4_leaves"]["cite"]
    for idx, want in ds["index_to_offset_4_leaves"]["rows"]:
        assert h.index_to_offset(idx, 4) == want, ds["index_to_offset_4_leaves"]["cite"]
    for initial, want in ds["capacity_for_initial_bytes"]["rows"]:
        assert DSB(initial).capacity() == want, ds["capacity_for_initial_bytes"]["cite"]
    g = known["dynamic_uniform"]["growth"]
    u = DUB(g["initial_capacity"], g["byte_size"], g["aligned"])
    assert u.take_resize() in (None, -1)
    for k in range(1, g["keys_inserted"] + 1):
        u.update(k, bytes(g["byte_size"]))
    assert u.capacity() == g["capacity_after"] and u.size() == g["size_after"], g["cite"]
    assert u.take_resize() == g["gpu_needs_resize_after"] and u.take_resize() in (None, -1), g["cite"]


def test_oracle_matches_its_committed_digests():
    """The arithmetic contract did not move: the oracle reproduces the committed digests bit for bit (regenerate with
    tests/golden/make_oracle_digests.py when the contract is changed on purpose)."""
    spec = importlib.util.spec_from_file_location("make_oracle_digests", os.path.join(GOLDEN, "make_oracle_digests.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(GOLDEN, "oracle_digests.json")))
    assert mod.digests() == want
