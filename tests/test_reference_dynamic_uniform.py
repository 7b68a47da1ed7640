"""The reference's own known-answer tests for DynamicUniformBuffer, restated one for one
(/root/reference/crates/renderer/src/buffer/dynamic_uniform.rs:291-1613, `mod test`, 31 tests) and run against BOTH
the pinned Python restatement and the C++ host implementation.  Each test carries the reference test's name + line."""
import pytest

from tests.buffer_adapters import IMPLS, KeyGen, create_keys


@pytest.fixture(params=["py", "cpp"])
def impl(request):
    return IMPLS[request.param]


def new(impl, cap=2, byte_size=16, aligned=32, zero=0):
    return impl[0](cap, byte_size, aligned, zero)


def test_new_buffer_initialization(impl):  # :316
    b = new(impl)
    assert b.size() == 64 and b.capacity() == 2 and b.next_slot() == 2
    assert len(b.free_slots()) == 2 and b.byte_size() == 16 and b.aligned() == 32
    assert b.free_slots() == [1, 0]
    assert b.len() == 0


def test_insert_single_item(impl):  # :334
    b = new(impl)
    _, k1, _, _ = create_keys()
    data = b"hello world 1234"
    b.update(k1, data)
    assert b.slot(k1) == 0 and b.free_slots() == [1]
    off = b.offset(k1)
    assert off == 0 and b.raw()[off:off + 16] == data


def test_insert_multiple_items(impl):  # :352
    b = new(impl)
    _, k1, k2, _ = create_keys()
    d1, d2 = b"data for key one", b"data for key two"
    b.update(k1, d1)
    b.update(k2, d2)
    assert b.slot(k1) is not None and b.slot(k2) is not None and len(b.free_slots()) == 0
    o1, o2 = b.offset(k1), b.offset(k2)
    assert o1 != o2 and b.raw()[o1:o1 + 16] == d1 and b.raw()[o2:o2 + 16] == d2


def test_buffer_growth(impl):  # :377
    b = new(impl)
    _, k1, k2, k3 = create_keys()
    b.update(k1, b"data one 1234567")
    b.update(k2, b"data two 1234567")
    s0 = b.size()
    b.update(k3, b"data three 12345")
    assert b.capacity() == 6 and b.size() == 192 and b.size() > s0
    assert all(b.offset(k) is not None for k in (k1, k2, k3))


def test_gpu_resize_flag(impl):  # :403
    b = new(impl)
    _, k1, k2, k3 = create_keys()
    assert b.take_resize() is None
    b.update(k1, b"test data 123456")
    assert b.take_resize() is None
    b.update(k2, b"more test data12")
    b.update(k3, b"even more data12")
    assert b.take_resize() == 192
    assert b.take_resize() is None


def test_update_existing_item(impl):  # :423
    b = new(impl)
    _, k1, _, _ = create_keys()
    b.update(k1, b"initial data1234")
    o0 = b.offset(k1)
    b.update(k1, b"updated data1234")
    o1 = b.offset(k1)
    assert o0 == o1 and b.raw()[o1:o1 + 16] == b"updated data1234"


def test_update_with_callback(impl):  # :445 — update_with(key, |offset, data|) expressed as two update_offset calls
    b = new(impl)
    _, k1, _, _ = create_keys()
    b.update_offset(k1, 0, b"test")
    off = b.offset(k1)
    b.update_offset(k1, 4, off.to_bytes(4, "little"))
    assert b.raw()[off:off + 4] == b"test" and b.raw()[off + 4:off + 8] == off.to_bytes(4, "little")


def test_update_offset(impl):  # :466
    b = new(impl)
    _, k1, _, _ = create_keys()
    b.update(k1, bytes(16))
    b.update_offset(k1, 8, b"partial")
    off = b.offset(k1)
    data = b.raw()[off:off + 16]
    assert data[0:8] == bytes(8) and data[8:15] == b"partial" and data[15] == 0


def test_remove_item(impl):  # :485
    b = new(impl)
    _, k1, k2, _ = create_keys()
    b.update(k1, b"data one 1234567")
    b.update(k2, b"data two 1234567")
    o1 = b.offset(k1)
    assert len(b.free_slots()) == 0
    b.remove(k1)
    assert b.offset(k1) is None and b.slot(k1) is None and len(b.free_slots()) == 1
    assert b.raw()[o1:o1 + 32] == bytes(32)
    assert b.offset(k2) is not None


def test_slot_reuse_after_removal(impl):  # :512
    b = new(impl)
    _, k1, k2, k3 = create_keys()
    b.update(k1, b"first item 12345")
    b.update(k2, b"second item 1234")
    o1 = b.offset(k1)
    b.remove(k1)
    b.update(k3, b"third item 12345")
    o3 = b.offset(k3)
    assert o1 == o3 and len(b.free_slots()) == 0 and b.raw()[o3:o3 + 16] == b"third item 12345"


def test_remove_nonexistent_key(impl):  # :535
    b = new(impl)
    _, k1, _, _ = create_keys()
    assert b.remove(k1) is False
    assert b.len() == 0 and len(b.free_slots()) == 2


def test_keys_iterator(impl):  # :548 (key membership expressed through slot())
    b = new(impl)
    _, k1, k2, k3 = create_keys()
    assert b.len() == 0
    b.update(k1, b"data1234567890ab")
    b.update(k2, b"more data 123456")
    assert b.len() == 2 and b.slot(k1) is not None and b.slot(k2) is not None and b.slot(k3) is None
    b.remove(k1)
    assert b.len() == 1 and b.slot(k2) is not None and b.slot(k1) is None


def test_zero_value_variants(impl):  # :575
    b1, b2 = impl[0](1, 8, 16, 0), impl[0](1, 8, 16, 0xFF)
    _, k1, k2, _ = create_keys()
    b1.update(k1, b"testdata")
    b2.update(k2, b"testdata")
    b1.remove(k1)
    b2.remove(k2)
    assert b1.raw()[0:16] == bytes(16) and b2.raw()[0:16] == bytes([0xFF]) * 16


def test_large_scale_operations(impl):  # :598
    b = new(impl, 8, 16, 32)
    kg = KeyGen()
    keys = []
    for i in range(100):
        k = kg.insert()
        keys.append(k)
        b.update(k, ("item_%03d_%08d" % (i, i * 12345)).encode()[:16].ljust(16, b"\0"))
    assert b.len() == 100
    for i, k in enumerate(keys):
        if i % 2 == 0:
            b.remove(k)
    assert b.len() == 50
    for i in range(200, 250):
        b.update(kg.insert(), ("new_item_%03d" % i).encode().ljust(16, b"\0"))
    assert b.len() == 100


def test_raw_slice_access(impl):  # :645
    b = new(impl)
    _, k1, _, _ = create_keys()
    assert b.raw() == bytes(64)
    b.update(k1, b"test data conten")
    off = b.offset(k1)
    assert b.raw()[off:off + 16] == b"test data conten"


def test_alignment_behavior(impl):  # :664
    b = impl[0](2, 10, 16, 0)
    _, k1, k2, _ = create_keys()
    b.update(k1, bytes([1]) * 10)
    b.update(k2, bytes([2]) * 10)
    o1, o2 = b.offset(k1), b.offset(k2)
    assert abs(o2 - o1) == 16
    assert b.raw()[o1:o1 + 10] == bytes([1]) * 10 and b.raw()[o2:o2 + 10] == bytes([2]) * 10


def test_resize_slot_allocation_correctness(impl):  # :693
    b = new(impl)
    kg = KeyGen()
    k1, k2 = kg.insert(), kg.insert()
    b.update(k1, b"data1_1234567890")
    b.update(k2, b"data2_1234567890")
    assert len(b.free_slots()) == 0 and b.next_slot() == 2 and b.capacity() == 2
    k3 = kg.insert()
    b.update(k3, b"data3_1234567890")
    assert b.capacity() == 6 and b.slot(k3) == 2 and 2 not in b.free_slots()
    assert sorted(b.free_slots()) == [3, 4, 5]
    k4 = kg.insert()
    b.update(k4, b"data4_1234567890")
    s4 = b.slot(k4)
    assert s4 != 2 and 3 <= s4 <= 5
    o3, o4 = b.offset(k3), b.offset(k4)
    assert b.raw()[o3:o3 + 16] == b"data3_1234567890" and b.raw()[o4:o4 + 16] == b"data4_1234567890"


def test_resize_with_required_slots_exceeding_capacity(impl):  # :760
    b = new(impl)
    kg = KeyGen()
    b.force_state(5)          # buffer.free_slots.clear(); buffer.next_slot = 5
    k1 = kg.insert()
    b.update(k1, b"capacity_test_12")
    assert b.capacity() == 12
    assert sorted(b.free_slots()) == list(range(6, 12))
    assert b.slot(k1) == 5 and 5 not in b.free_slots()


def test_resize_slot_range_allocation(impl):  # :808
    b = new(impl)
    kg = KeyGen()
    k1, k2 = kg.insert(), kg.insert()
    b.update(k1, b"data1_1234567890")
    b.update(k2, b"data2_1234567890")
    k3 = kg.insert()
    b.update(k3, b"data3_1234567890")
    assert b.slot(k3) == 2 and sorted(b.free_slots()) == [3, 4, 5] and 2 not in b.free_slots()
    k4 = kg.insert()
    b.update(k4, b"data4_1234567890")
    assert b.slot(k4) == 5          # pops from the end of free_slots
    for k in (k1, k2, k3, k4):
        assert b.slot(k) not in b.free_slots()


def test_resize_slot_assignment_consistency(impl):  # :857
    b = new(impl)
    kg = KeyGen()
    b.force_state(3)
    k1 = kg.insert()
    b.update(k1, b"test_slot_3_data")
    assert b.slot(k1) == 3 and 3 not in b.free_slots() and sorted(b.free_slots()) == [4, 5, 6, 7]
    k2 = kg.insert()
    b.update(k2, b"test_slot_7_data")
    assert b.slot(k2) == 7


def test_resize_semantic_correctness(impl):  # :906
    b = new(impl)
    kg = KeyGen()
    k1, k2 = kg.insert(), kg.insert()
    b.update(k1, b"data1_1234567890")
    b.update(k2, b"data2_1234567890")
    b.remove(k1)
    freed = 1 if b.slot(k2) == 0 else 0
    assert freed in b.free_slots()
    k3 = kg.insert()
    b.update(k3, b"data3_1234567890")
    assert b.slot(k3) == freed and b.capacity() == 2
    k4 = kg.insert()
    b.update(k4, b"data4_1234567890")
    assert b.capacity() == 6 and b.slot(k4) == 2 and sorted(b.free_slots()) == [3, 4, 5]


def test_offset_consistency_after_resize(impl):  # :960
    b = new(impl)
    kg = KeyGen()
    k1, k2 = kg.insert(), kg.insert()
    b.update(k1, b"data1___________")
    b.update(k2, b"data2___________")
    o1b, o2b, s1, s2 = b.offset(k1), b.offset(k2), b.slot(k1), b.slot(k2)
    assert o1b == s1 * 32 and o2b == s2 * 32
    k3 = kg.insert()
    b.update(k3, b"data3___________")
    assert b.slot(k1) == s1 and b.slot(k2) == s2
    o1, o2, o3 = b.offset(k1), b.offset(k2), b.offset(k3)
    assert o1 == b.slot(k1) * 32 and o2 == b.slot(k2) * 32 and o3 == b.slot(k3) * 32
    assert o1 == o1b and o2 == o2b and len({o1, o2, o3}) == 3
    raw = b.raw()
    assert raw[o1:o1 + 16] == b"data1___________" and raw[o2:o2 + 16] == b"data2___________" and raw[o3:o3 + 16] == b"data3___________"


def test_resize_next_slot_update(impl):  # :1053
    b = impl[0](1, 16, 32, 0)
    kg = KeyGen()
    keys = []
    for i in range(1, 6):
        k = kg.insert()
        keys.append(k)
        b.update(k, ("ITEM_%02d_________" % i).encode())
        if i == 2:
            assert b.next_slot() > max(b.free_slots() or [0])
    assert len({b.slot(k) for k in keys}) == 5
    raw = b.raw()
    for i, k in enumerate(keys):
        o = b.offset(k)
        assert raw[o:o + 16] == ("ITEM_%02d_________" % (i + 1)).encode()


def test_offset_after_multiple_resizes_and_removals(impl):  # :1175
    b = impl[0](1, 16, 32, 0)
    kg = KeyGen()
    items = []
    for i in range(8):
        k = kg.insert()
        data = ("item_%02d_data____" % i).encode()
        items.append((k, data))
        b.update(k, data)
        assert b.offset(k) == b.slot(k) * 32
        raw = b.raw()
        for pk, pd in items:
            assert b.offset(pk) == b.slot(pk) * 32 and raw[b.offset(pk):b.offset(pk) + 16] == pd
    removed = [items[1][0], items[3][0], items[5][0]]
    for k in removed:
        b.remove(k)
    raw = b.raw()
    for k, d in items:
        if k in removed:
            assert b.offset(k) is None
        else:
            assert b.offset(k) == b.slot(k) * 32 and raw[b.offset(k):b.offset(k) + 16] == d
    for i in range(8, 12):
        k = kg.insert()
        d = ("new_item_%02d____" % i).encode()[:16].ljust(16, b"_")
        b.update(k, d)
        assert b.offset(k) == b.slot(k) * 32 and b.raw()[b.offset(k):b.offset(k) + 16] == d


def test_offset_edge_cases_with_manual_state(impl):  # :1330
    b = impl[0](3, 16, 64, 0)
    kg = KeyGen()
    k1, k2 = kg.insert(), kg.insert()
    b.update(k1, b"test1___________")
    b.update(k2, b"test2___________")
    s1, s2 = b.slot(k1), b.slot(k2)
    assert b.offset(k1) == s1 * 64 and b.offset(k2) == s2 * 64
    b.force_state(10)
    k3 = kg.insert()
    b.update(k3, b"test3___________")
    assert b.slot(k3) == 10 and b.offset(k3) == 640
    assert b.offset(k1) == s1 * 64 and b.offset(k2) == s2 * 64
    assert len({b.offset(k1), b.offset(k2), b.offset(k3)}) == 3


def test_offset_boundary_values(impl):  # :1403
    b = impl[0](1, 1, 1, 0)
    kg = KeyGen()
    k1 = kg.insert()
    b.update(k1, bytes([0x42]))
    assert b.offset(k1) == 0
    k2 = kg.insert()
    b.update(k2, bytes([0x43]))
    assert b.offset(k1) == b.slot(k1) and b.offset(k2) == b.slot(k2)
    assert b.raw()[b.offset(k1)] == 0x42 and b.raw()[b.offset(k2)] == 0x43


def test_new_utility_methods(impl):  # :1439
    b = new(impl)
    _, k1, k2, _ = create_keys()
    assert b.len() == 0
    b.update(k1, b"data1___________")
    assert b.len() == 1
    b.update(k2, b"data2___________")
    assert b.len() == 2 and b.capacity() == 2 and len(b.free_slots()) == 0
    b.remove(k1)
    assert b.len() == 1 and b.slot(k1) is None and b.slot(k2) is not None and len(b.free_slots()) == 1


def test_update_panics_on_oversized_data(impl):  # :1470  #[should_panic]
    b = impl[0](1, 10, 16, 0)
    _, k1, _, _ = create_keys()
    with pytest.raises(ValueError):
        b.update(k1, bytes(11))


def test_zero_capacity_initialization(impl):  # :1483
    b = impl[0](0, 16, 32, 0)
    assert b.capacity() == 0 and b.size() == 0 and b.len() == 0


def test_concurrent_operations_simulation(impl):  # :1494
    b = new(impl)
    kg = KeyGen()
    keys = []
    for i in range(10):
        k = kg.insert()
        keys.append(k)
        b.update(k, ("data_%03d________" % i).encode())
    for i in range(0, 10, 3):
        b.remove(keys[i])
    for i, k in enumerate(keys):
        if i % 3 != 0:
            b.update(k, ("updt_%03d________" % i).encode())
    for i in range(10, 15):
        b.update(kg.insert(), ("new__%03d________" % i).encode())
    raw = b.raw()
    for i, k in enumerate(keys):
        if i % 3 != 0:
            o = b.offset(k)
            assert raw[o:o + 16] == ("updt_%03d________" % i).encode()


def test_no_dangling_slots_after_resize(impl):  # :1546
    b = new(impl)
    kg = KeyGen()
    k1, k2 = kg.insert(), kg.insert()
    b.update(k1, b"data1___________")
    b.update(k2, b"data2___________")
    k3 = kg.insert()
    b.update(k3, b"data3___________")
    allk = [k1, k2, k3]
    for i in range(4, 7):
        k = kg.insert()
        allk.append(k)
        b.update(k, ("data%d___________" % i).encode())
    assert len(b.free_slots()) == 0
    k7 = kg.insert()
    allk.append(k7)
    b.update(k7, b"data7___________")
    for i in range(8, 20):
        k = kg.insert()
        allk.append(k)
        b.update(k, ("data%d__________" % i).encode()[:16])
    max_slot = max(b.slot(k) for k in allk)
    assert b.len() / (max_slot + 1) > 0.5


# ---- beyond the reference's tests: only byte_size bytes of a slot are dirty on update, the full slot on remove ----
def test_dirty_ranges(impl):
    b = new(impl)
    _, k1, k2, _ = create_keys()
    b.update(k1, bytes(16))
    b.update(k2, bytes(16))
    assert b.take_dirty() == [(0, 16), (32, 16)]
    b.remove(k1)
    assert b.take_dirty() == [(0, 32)]
