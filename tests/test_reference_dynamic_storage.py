"""The reference's own known-answer tests for DynamicStorageBuffer, restated one for one
(/root/reference/crates/renderer/src/buffer/dynamic_storage.rs:411-1318, `mod test`, 32 tests) and run against BOTH
the pinned Python restatement (oracle/host_mirror.py) and the C++ host implementation (awsm-renderer_amd/host).
Each test carries the reference test's name and line."""
import pytest

from tests.buffer_adapters import IMPLS, KeyGen, create_keys

MIN_BLOCK = 256


@pytest.fixture(params=["py", "cpp"])
def impl(request):
    return IMPLS[request.param]


def new(impl, initial=1024, zero=0):
    return impl[1](initial, zero)


def is_pow2(n):
    return n > 0 and (n & (n - 1)) == 0


def test_new_buffer_initialization(impl):  # :434
    b = new(impl)
    assert b.capacity() == 1024
    assert all(x == 0 for x in b.raw())
    assert b.len() == 0
    assert b.tree_root() == 1024


def test_insert_single_item(impl):  # :451
    b = new(impl)
    _, k1, _, _ = create_keys()
    data = b"hello world test data"
    off = b.update(k1, data)
    assert b.contains(k1)
    assert off == 0
    assert b.raw()[off:off + len(data)] == data
    size = b.size_of(k1)
    assert is_pow2(size) and size >= MIN_BLOCK


def test_insert_multiple_items(impl):  # :477
    b = new(impl)
    _, k1, k2, _ = create_keys()
    d1, d2 = b"first data block", b"second data block with more content"
    o1, o2 = b.update(k1, d1), b.update(k2, d2)
    assert b.contains(k1) and b.contains(k2)
    assert o1 != o2
    assert b.raw()[o1:o1 + len(d1)] == d1 and b.raw()[o2:o2 + len(d2)] == d2


def test_update_existing_item_same_size(impl):  # :500
    b = new(impl)
    _, k1, _, _ = create_keys()
    o0 = b.update(k1, b"initial data content")
    s0 = b.size_of(k1)
    upd = b"updated data content"
    o1 = b.update(k1, upd)
    assert o0 == o1 and b.size_of(k1) == s0
    assert b.raw()[o1:o1 + len(upd)] == upd


def test_update_existing_item_larger_size(impl):  # :526
    b = new(impl)
    _, k1, _, _ = create_keys()
    b.update(k1, bytes([1]) * 10)
    large = bytes([2]) * 300
    o = b.update(k1, large)
    assert b.size_of(k1) >= 512
    assert b.raw()[o:o + 300] == large


def test_remove_item(impl):  # :550
    b = new(impl)
    _, k1, k2, _ = create_keys()
    o1 = b.update(k1, b"data one")
    b.update(k2, b"data two")
    s1 = b.size_of(k1)
    b.remove(k1)
    assert b.offset(k1) is None and b.size_of(k1) is None and not b.contains(k1)
    assert all(x == 0 for x in b.raw()[o1:o1 + s1])
    assert b.offset(k2) is not None


def test_buddy_allocation_reuse(impl):  # :580
    b = new(impl)
    _, k1, k2, k3 = create_keys()
    data = bytes([1]) * 100
    b.update(k1, data)
    b.update(k2, data)
    o1 = b.offset(k1)
    b.remove(k1)
    b.update(k3, data)
    assert b.offset(k3) == o1


def test_buffer_growth(impl):  # :604
    b = new(impl, 512)
    kg, _, _, _ = create_keys()
    large = bytes([42]) * 400
    k1, k2 = kg.insert(), kg.insert()
    b.update(k1, large)
    cap0 = b.capacity()
    b.update(k2, large)
    assert b.capacity() > cap0 and is_pow2(b.capacity())
    assert b.offset(k1) is not None and b.offset(k2) is not None


def test_gpu_resize_flag(impl):  # :634
    b = new(impl, 256)
    _, k1, k2, _ = create_keys()
    b.take_resize()
    b.update(k1, b"small")
    assert b.take_resize() is None
    b.update(k2, bytes([1]) * 200)
    assert b.take_resize() is not None
    assert b.take_resize() is None


def test_power_of_two_rounding(impl):  # :660
    b = new(impl)
    kg = KeyGen()
    for size in [1, 15, 16, 17, 100, 255, 256, 257, 500]:
        k = kg.insert()
        b.update(k, bytes([0xAA]) * size)
        a = b.size_of(k)
        assert is_pow2(a) and a >= size and a >= MIN_BLOCK


def test_buddy_coalescing(impl):  # :681
    b = new(impl)
    kg, _, _, _ = create_keys()
    k1, k2 = kg.insert(), kg.insert()
    data = bytes([1]) * MIN_BLOCK
    b.update(k1, data)
    b.update(k2, data)
    b.remove(k1)
    b.remove(k2)
    k3 = kg.insert()
    assert b.update(k3, bytes([2]) * (MIN_BLOCK * 2)) == 0


def test_update_with_unchecked(impl):  # :708
    b = new(impl)
    _, k1, _, _ = create_keys()
    b.update(k1, bytes(100))
    assert b.offset(k1) == 0 and b.size_of(k1) >= 100
    b.patch(k1, 0, b"TEST")
    o = b.offset(k1)
    assert b.raw()[o:o + 4] == b"TEST"


def test_update_with_unchecked_missing_key(impl):  # :733  #[should_panic(expected = "not found")]
    b = new(impl)
    _, k1, _, _ = create_keys()
    with pytest.raises(KeyError, match="not found"):
        b.patch(k1, 0, b"")


def test_zero_value_variants(impl):  # :743
    b1, b2 = new(impl, 512, 0), new(impl, 512, 0xFF)
    _, k1, k2, _ = create_keys()
    b1.update(k1, b"testdata")
    b2.update(k2, b"testdata")
    o1, s1, o2, s2 = b1.offset(k1), b1.size_of(k1), b2.offset(k2), b2.size_of(k2)
    b1.remove(k1)
    b2.remove(k2)
    assert all(x == 0 for x in b1.raw()[o1:o1 + s1])
    assert all(x == 0xFF for x in b2.raw()[o2:o2 + s2])


def test_large_scale_operations(impl):  # :775
    b = new(impl, 1024)
    kg = KeyGen()
    keys = []
    for i in range(50):
        k = kg.insert()
        keys.append(k)
        b.update(k, bytes([i % 256]) * (10 + (i * 7) % 200))
    raw = b.raw()
    for i, k in enumerate(keys):
        o = b.offset(k)
        assert o is not None and b.size_of(k) is not None
        size = 10 + (i * 7) % 200
        assert raw[o:o + size] == bytes([i % 256]) * size
    for i, k in enumerate(keys):
        if i % 2 == 0:
            b.remove(k)
    for i in range(100, 125):
        b.update(kg.insert(), bytes([i % 256]) * (15 + (i * 11) % 150))


def test_raw_slice_access(impl):  # :821
    b = new(impl)
    _, k1, _, _ = create_keys()
    assert len(b.raw()) == 1024
    data = b"test data content here"
    b.update(k1, data)
    o = b.offset(k1)
    assert b.raw()[o:o + len(data)] == data


def test_used_size_tracking(impl):  # :838
    b = new(impl)
    _, k1, k2, k3 = create_keys()
    assert b.used_size() == 0
    b.update(k1, bytes([1]) * 100)
    s1 = b.size_of(k1)
    assert b.used_size() == s1
    b.update(k2, bytes([2]) * 200)
    s2 = b.size_of(k2)
    assert b.used_size() == s1 + s2
    b.update(k3, bytes([3]) * 50)
    s3 = b.size_of(k3)
    assert b.used_size() == s1 + s2 + s3
    b.remove(k2)
    assert b.used_size() == s1 + s3


def test_minimum_block_size(impl):  # :864
    b = new(impl)
    _, k1, _, _ = create_keys()
    b.update(k1, b"x")
    assert b.size_of(k1) == MIN_BLOCK


def test_buddy_tree_operations(impl):  # :877
    b = new(impl, 1024)
    kg = KeyGen()
    k1, k2, k3 = kg.insert(), kg.insert(), kg.insert()
    b.update(k1, bytes([1]) * 100)
    b.update(k2, bytes([2]) * 200)
    b.remove(k1)
    b.update(k3, bytes([3]) * 150)
    o2, s2, o3, s3 = b.offset(k2), b.size_of(k2), b.offset(k3), b.size_of(k3)
    assert o3 + s3 <= o2 or o2 + s2 <= o3
    raw = b.raw()
    assert raw[o2:o2 + min(200, s2)] == bytes([2]) * min(200, s2)
    assert raw[o3:o3 + min(150, s3)] == bytes([3]) * min(150, s3)


def test_allocation_patterns(impl):  # :923
    b = new(impl, 2048)
    kg = KeyGen()
    keys = []
    for _ in range(4):
        k = kg.insert()
        keys.append(k)
        b.update(k, bytes([0xAA]) * MIN_BLOCK)
    b.remove(keys[0])
    b.remove(keys[2])
    assert b.update(kg.insert(), bytes([0xBB]) * (MIN_BLOCK * 2)) >= MIN_BLOCK * 4


def test_grow_with_existing_allocations(impl):  # :957
    b = new(impl, 512)
    kg, _, _, _ = create_keys()
    k1, k2 = kg.insert(), kg.insert()
    d1, d2 = bytes([0x11]) * 100, bytes([0x22]) * 150
    o1, o2 = b.update(k1, d1), b.update(k2, d2)
    b.update(kg.insert(), bytes([0x33]) * 400)
    assert b.offset(k1) == o1 and b.offset(k2) == o2
    assert b.raw()[o1:o1 + 100] == d1 and b.raw()[o2:o2 + 150] == d2


def test_initial_size_rounding(impl):  # :988
    assert new(impl, 1000).capacity() == 1024
    assert new(impl, 2000).capacity() == 2048
    assert new(impl, 10).capacity() == MIN_BLOCK


def test_offset_and_size_queries(impl):  # :1004
    b = new(impl)
    _, k1, k2, _ = create_keys()
    assert b.offset(k1) is None and b.size_of(k1) is None
    b.update(k1, bytes([1]) * 100)
    o1, s1 = b.offset(k1), b.size_of(k1)
    assert o1 == 0 and s1 >= 100 and is_pow2(s1)
    b.update(k2, bytes([2]) * 300)
    o2, s2 = b.offset(k2), b.size_of(k2)
    assert o1 != o2 and s2 >= 300 and is_pow2(s2)
    b.remove(k1)
    assert b.offset(k1) is None and b.size_of(k1) is None


def test_update_smaller_data_clears_tail(impl):  # :1042
    b = new(impl)
    _, k1, _, _ = create_keys()
    b.update(k1, bytes([0xAA]) * 200)
    o, s = b.offset(k1), b.size_of(k1)
    b.update(k1, bytes([0xBB]) * 50)
    assert b.offset(k1) == o and b.size_of(k1) == s
    raw = b.raw()
    assert raw[o:o + 50] == bytes([0xBB]) * 50
    assert all(x == 0 for x in raw[o + 50:o + s])


def test_helper_functions(impl):  # :1076
    h = impl[2]
    assert h.round_pow2(0) == MIN_BLOCK and h.round_pow2(1) == MIN_BLOCK and h.round_pow2(MIN_BLOCK) == MIN_BLOCK
    assert h.round_pow2(MIN_BLOCK + 1) == MIN_BLOCK * 2
    assert h.round_pow2(1000) == 1024 and h.round_pow2(1024) == 1024 and h.round_pow2(1025) == 2048
    leaves = 4
    assert [h.offset_to_index(MIN_BLOCK * i, leaves) for i in range(4)] == [3, 4, 5, 6]
    assert [h.index_to_offset(i, leaves) for i in (3, 4, 5, 6)] == [0, MIN_BLOCK, MIN_BLOCK * 2, MIN_BLOCK * 3]
    assert h.index_to_offset(0, leaves) == 0 and h.index_to_offset(1, leaves) == 0 and h.index_to_offset(2, leaves) == MIN_BLOCK * 2


def test_complex_allocation_deallocation_pattern(impl):  # :1107
    b = new(impl, 4096)
    kg = KeyGen()
    allocs = []
    for i in range(10):
        k = kg.insert()
        size = MIN_BLOCK * (1 << (i % 3))
        b.update(k, bytes([i % 256]) * size)
        allocs.append((k, size))
    for i in range(1, 10, 3):
        b.remove(allocs[i][0])
    for i in range(20, 25):
        size = MIN_BLOCK * (1 << (i % 2))
        o = b.update(kg.insert(), bytes([i % 256]) * size)
        assert b.raw()[o:o + size] == bytes([i % 256]) * size


def test_extreme_fragmentation_handling(impl):  # :1151
    b = new(impl, 8192)
    kg = KeyGen()
    n = 8192 // MIN_BLOCK
    keys = []
    for i in range(n):
        k = kg.insert()
        keys.append(k)
        b.update(k, bytes([i % 256]) * MIN_BLOCK)
    for i in range(0, n, 2):
        b.remove(keys[i])
    large = bytes([0xFF]) * (MIN_BLOCK * 4)
    o = b.update(kg.insert(), large)
    assert b.capacity() > 8192
    assert b.raw()[o:o + len(large)] == large


def test_new_utility_methods(impl):  # :1189
    b = new(impl)
    _, k1, k2, _ = create_keys()
    assert b.len() == 0
    b.update(k1, b"data1")
    assert b.len() == 1
    b.update(k2, b"data2_longer")
    assert b.len() == 2 and b.contains(k1) and b.contains(k2)
    assert b.capacity() == 1024
    b.remove(k1)
    assert b.len() == 1 and not b.contains(k1) and b.contains(k2)


def test_zero_sized_allocation(impl):  # :1223
    b = new(impl)
    _, k1, _, _ = create_keys()
    b.update(k1, b"")
    assert b.contains(k1) and b.size_of(k1) == MIN_BLOCK and b.offset(k1) == 0


def test_maximum_fragmentation_recovery(impl):  # :1236
    b = new(impl, 2048)
    kg = KeyGen()
    keys = []
    for i in range(8):
        k = kg.insert()
        keys.append(k)
        b.update(k, bytes([i]) * MIN_BLOCK)
    for i in range(0, 8, 2):
        b.remove(keys[i])
    kn = kg.insert()
    b.update(kn, bytes([0xFF]) * MIN_BLOCK)
    assert b.offset(kn) % (MIN_BLOCK * 2) == 0


def test_concurrent_like_access_pattern(impl):  # :1262
    b = new(impl)
    kg = KeyGen()
    ops = []
    for i in range(20):
        k = kg.insert()
        data = bytes([i % 256]) * (50 + (i * 17) % 200)
        b.update(k, data)
        ops.append((k, data))
        if i > 5 and i % 3 == 0:
            idx = (i - 5) // 2
            if idx < len(ops):
                b.remove(ops[idx][0])
    raw = b.raw()
    for k, data in ops:
        o = b.offset(k)
        if o is not None:
            assert raw[o:o + len(data)] == data


def test_growth_with_multiple_size_requirements(impl):  # :1296
    b = new(impl, 512)
    kg, _, _, _ = create_keys()
    k1 = kg.insert()
    huge = bytes([0x42]) * 2048
    b.update(k1, huge)
    assert b.capacity() >= 2048
    o = b.offset(k1)
    assert b.raw()[o:o + 2048] == huge


# ---- beyond the reference's tests: the dirty-range arithmetic its callers rely on (dynamic_storage.rs:196-211) ----
def test_dirty_ranges_are_whole_blocks_and_4_byte_aligned(impl):
    b = new(impl)
    _, k1, k2, _ = create_keys()
    b.update(k1, b"abc")                    # insert marks the whole 256-B block
    b.update(k2, bytes(300))                # 512-B block
    assert b.take_dirty() == [(0, 256), (512, 512)]
    b.update(k1, b"xy")                     # in-place update marks the whole old block
    b.remove(k2)
    assert b.take_dirty() == [(0, 256), (512, 512)]
    assert b.take_dirty() == []
