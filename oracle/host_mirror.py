"""
host_mirror.py — TEST INFRASTRUCTURE ONLY (parity oracle for the C++ host layer).

Pure-Python restatement of the reference's CPU-side GPU mirrors and of the small pieces of third-party
arithmetic they rely on.  Paths relative to /root/reference/crates/renderer/src/ :

  buffer/dynamic_storage.rs:39-409    DynamicStorageBuffer (buddy allocator, MIN_BLOCK = 256)
  buffer/dynamic_uniform.rs:40-289    DynamicUniformBuffer (fixed slots + free list)
  buffer/helpers.rs:124-220           write_buffer_with_dirty_ranges (32-range cap, 60 % threshold, coalescing)
  frustum.rs:42-89, bounds.rs:38-61   Frustum::from_view_projection / intersects_aabb, Aabb::transform
  slotmap 1.1.1 (Cargo.lock:1029-1031; absent from /root/reference): key = (version << 32) | idx, slot 0 is a
      sentinel so the first key is idx 1 / version 1, free list is LIFO, a reused slot's version grows by 2.
  glam 0.31.0 (Cargo.lock:552-555; absent from /root/reference), scalar code path: Mat4::from_scale_rotation_
      translation, mul_mat4, inverse, determinant, transform_point3 (NO perspective divide), perspective_rh,
      orthographic_rh (both 0..1 depth), look_at_rh.

PINNED: the two buffer classes, the write plan and the frustum are checked against the reference's own unit tests
restated in tests/test_host_mirror_reference_cases.py (63 buffer tests + 4 frustum tests).
UNPINNED: the slotmap / glam restatements (published algorithms restated from memory of the pinned versions).
"""
from __future__ import annotations

import math
import struct
from typing import Dict, List, Optional, Tuple

import numpy as np

F = np.float32

# ------------------------------------------------------------------------------------------------ slotmap


class SlotMap:
    """slotmap::SlotMap / DenseSlotMap key allocation (values are kept in insertion order like DenseSlotMap)."""

    def __init__(self):
        self._slots: List[List[int]] = [[0, 0]]  # [version, next_free]; slot 0 = sentinel
        self._free_head = 1
        self._dense_keys: List[Tuple[int, int]] = []
        self._values: Dict[Tuple[int, int], object] = {}

    def insert(self, value=None) -> Tuple[int, int]:
        idx = self._free_head
        if idx < len(self._slots):
            slot = self._slots[idx]
            self._free_head = slot[1]
            slot[0] |= 1  # occupied versions are odd
        else:
            self._slots.append([1, 0])
            self._free_head = len(self._slots)
        key = (idx, self._slots[idx][0])
        self._dense_keys.append(key)
        self._values[key] = value
        return key

    def remove(self, key) -> bool:
        if key not in self._values:
            return False
        idx, _ = key
        slot = self._slots[idx]
        slot[0] += 1  # even = vacant
        slot[1] = self._free_head
        self._free_head = idx
        # DenseSlotMap: swap_remove
        i = self._dense_keys.index(key)
        self._dense_keys[i] = self._dense_keys[-1]
        self._dense_keys.pop()
        del self._values[key]
        return True

    def contains(self, key) -> bool:
        return key in self._values

    def get(self, key):
        return self._values.get(key)

    def set(self, key, value):
        self._values[key] = value

    def keys(self):
        return list(self._dense_keys)

    def items(self):
        return [(k, self._values[k]) for k in self._dense_keys]

    def __len__(self):
        return len(self._dense_keys)


def key_as_ffi(key: Tuple[int, int]) -> int:
    idx, version = key
    return (version << 32) | idx


# ------------------------------------------------------------------------------------------------ buffers

MIN_BLOCK = 256


def round_pow2(n: int) -> int:
    p = 1
    while p < n:
        p <<= 1
    return max(p, MIN_BLOCK)


def index_to_offset(idx: int, leaves: int) -> int:
    while idx < leaves - 1:
        idx = idx * 2 + 1
    return (idx + 1 - leaves) * MIN_BLOCK


def offset_to_index(off: int, leaves: int) -> int:
    return leaves - 1 + off // MIN_BLOCK


def _mark_dirty(ranges: List[Tuple[int, int]], raw_len: int, offset: int, size: int):
    if size == 0 or raw_len == 0 or offset >= raw_len:
        return
    start = offset & ~3
    end = min(offset + size, raw_len)
    end = min((end + 3) & ~3, raw_len)
    if start < end:
        ranges.append((start, end - start))


class DynamicStorageBuffer:
    """buffer/dynamic_storage.rs:39-409."""

    def __init__(self, initial_bytes: int, zero: int = 0):
        orig = initial_bytes
        cap = round_pow2(max(initial_bytes, MIN_BLOCK))
        self.zero = zero
        self.raw = bytearray([zero]) * cap
        self.dirty_ranges: List[Tuple[int, int]] = []
        self.slots: Dict[object, Tuple[int, int]] = {}
        self.gpu_needs_resize = cap != orig
        self._init_tree(cap)

    def _init_tree(self, cap: int):
        leaves = cap // MIN_BLOCK
        self.tree = [0] * (2 * leaves - 1)
        # init_full: node at depth d holds cap >> d
        size, start, count = cap, 0, 1
        while True:
            for i in range(start, start + count):
                self.tree[i] = size
            if size <= MIN_BLOCK:
                break
            start, count, size = start + count, count * 2, size // 2

    def _fix_parents(self, idx: int):
        while idx != 0:
            parent = (idx - 1) >> 1
            left = parent * 2 + 1
            new_val = max(self.tree[left], self.tree[left + 1])
            if self.tree[parent] == new_val:
                break
            self.tree[parent] = new_val
            idx = parent

    def _alloc(self, req: int) -> Optional[int]:
        if req > self.tree[0]:
            return None
        idx, size = 0, len(self.raw)
        while size > req:
            left = idx * 2 + 1
            idx = left if self.tree[left] >= req else left + 1
            size //= 2
        self.tree[idx] = 0
        self._fix_parents(idx)
        return index_to_offset(idx, len(self.raw) // MIN_BLOCK)

    def _free(self, offset: int, size: int):
        leaves = len(self.raw) // MIN_BLOCK
        idx, blk = offset_to_index(offset, leaves), MIN_BLOCK
        while blk < size:
            idx = (idx - 1) >> 1
            blk <<= 1
        self.tree[idx] = blk
        while idx != 0:
            parent = (idx - 1) >> 1
            left = parent * 2 + 1
            right = left + 1
            merged = self.tree[left] == blk and self.tree[right] == blk
            new_val = (blk << 1) if merged else max(self.tree[left], self.tree[right])
            if self.tree[parent] == new_val:
                break
            self.tree[parent] = new_val
            if merged:
                idx = parent
                blk <<= 1
            else:
                break

    def _grow(self, min_extra: int):
        old_cap = len(self.raw)
        new_cap = old_cap * 2
        while new_cap - old_cap < min_extra:
            new_cap *= 2
        self.raw.extend(bytearray([self.zero]) * (new_cap - old_cap))
        self.gpu_needs_resize = True
        self._init_tree(new_cap)
        leaves = new_cap // MIN_BLOCK
        for offset, size in self.slots.values():
            idx, sz = offset_to_index(offset, leaves), MIN_BLOCK
            while sz < size:
                idx = (idx - 1) >> 1
                sz <<= 1
            self.tree[idx] = 0
            self._fix_parents(idx)

    def _insert(self, key, data: bytes) -> int:
        req = round_pow2(max(len(data), MIN_BLOCK))
        off = self._alloc(req)
        if off is None:
            self._grow(max(req, len(self.raw)))
            off = self._alloc(req)
            assert off is not None, "allocation after grow must succeed"
        self.raw[off:off + len(data)] = data
        self.slots[key] = (off, req)
        _mark_dirty(self.dirty_ranges, len(self.raw), off, req)
        return off

    def update(self, key, data: bytes) -> int:
        if key in self.slots:
            off, old = self.slots[key]
            if len(data) <= old:
                self.raw[off:off + len(data)] = data
                if len(data) < old:
                    self.raw[off + len(data):off + old] = bytearray([self.zero]) * (old - len(data))
                _mark_dirty(self.dirty_ranges, len(self.raw), off, old)
                return off
            self.remove(key)
        return self._insert(key, data)

    def update_with_unchecked(self, key, fn):
        if key not in self.slots:
            raise KeyError(f"Key {key} not found in DynamicBuddyBuffer")
        off, size = self.slots[key]
        view = bytearray(self.raw[off:off + size])
        fn(off, view)
        self.raw[off:off + size] = view
        _mark_dirty(self.dirty_ranges, len(self.raw), off, size)

    def remove(self, key):
        if key in self.slots:
            off, size = self.slots.pop(key)
            self.raw[off:off + size] = bytearray([self.zero]) * size
            _mark_dirty(self.dirty_ranges, len(self.raw), off, size)
            self._free(off, size)

    def used_size(self):
        return sum(s for _, s in self.slots.values())

    def offset(self, key):
        return self.slots[key][0] if key in self.slots else None

    def size(self, key):
        return self.slots[key][1] if key in self.slots else None

    def capacity(self):
        return len(self.raw)

    def take_dirty_ranges(self):
        r, self.dirty_ranges = self.dirty_ranges, []
        return r

    def clear_dirty_ranges(self):
        self.dirty_ranges = []

    def take_gpu_needs_resize(self):
        size = len(self.raw) if self.gpu_needs_resize else None
        self.gpu_needs_resize = False
        return size


class DynamicUniformBuffer:
    """buffer/dynamic_uniform.rs:40-289."""

    def __init__(self, initial_capacity: int, byte_size: int, aligned_slice_size: Optional[int] = None, zero: int = 0):
        self.aligned = aligned_slice_size if aligned_slice_size is not None else byte_size
        self.byte_size = byte_size
        self.zero = zero
        self.raw = bytearray([zero]) * (initial_capacity * self.aligned)
        self.dirty_ranges: List[Tuple[int, int]] = []
        self.gpu_needs_resize = False
        self.slot_indices: Dict[object, int] = {}
        self.free_slots = list(range(initial_capacity - 1, -1, -1))
        self.capacity_slots = initial_capacity
        self.next_slot = initial_capacity

    def _resize(self, required_slots: int):
        new_cap = max(required_slots, self.capacity_slots) * 2
        self.raw.extend(bytearray([self.zero]) * (new_cap * self.aligned - len(self.raw)))
        self.free_slots.extend(range(required_slots, new_cap))
        self.next_slot = new_cap
        self.capacity_slots = new_cap
        self.gpu_needs_resize = True

    def update_with(self, key, fn):
        slot = self.slot_indices.get(key)
        if slot is None:
            if self.free_slots:
                slot = self.free_slots.pop()
            else:
                slot = self.next_slot
                if (slot + 1) * self.aligned > len(self.raw):
                    self._resize(slot + 1)
                self.next_slot += 1
            self.slot_indices[key] = slot
        off = slot * self.aligned
        view = bytearray(self.raw[off:off + self.byte_size])
        fn(off, view)
        assert len(view) == self.byte_size
        self.raw[off:off + self.byte_size] = view
        _mark_dirty(self.dirty_ranges, len(self.raw), off, self.byte_size)

    def update(self, key, values: bytes):
        if len(values) > self.byte_size:
            raise ValueError("values exceed byte_size")

        def fn(_, data):
            data[:len(values)] = values

        self.update_with(key, fn)

    def update_offset(self, key, offset: int, values: bytes):
        def fn(_, data):
            if offset + len(values) > len(data):
                raise ValueError("out of slot")
            data[offset:offset + len(values)] = values

        self.update_with(key, fn)

    def remove(self, key) -> bool:
        slot = self.slot_indices.pop(key, None)
        if slot is None:
            return False
        self.free_slots.append(slot)
        off = slot * self.aligned
        self.raw[off:off + self.aligned] = bytearray([self.zero]) * self.aligned
        _mark_dirty(self.dirty_ranges, len(self.raw), off, self.aligned)
        return True

    def offset(self, key):
        slot = self.slot_indices.get(key)
        return None if slot is None else slot * self.aligned

    def slot_index(self, key):
        return self.slot_indices.get(key)

    def size(self):
        return len(self.raw)

    def take_dirty_ranges(self):
        r, self.dirty_ranges = self.dirty_ranges, []
        return r

    def clear_dirty_ranges(self):
        self.dirty_ranges = []

    def take_gpu_needs_resize(self):
        size = len(self.raw) if self.gpu_needs_resize else None
        self.gpu_needs_resize = False
        return size


DIRTY_RANGE_FULL_WRITE_THRESHOLD_PERCENT = 60
DIRTY_RANGE_MAX_RANGES = 32


def coalesce_ranges(ranges):
    if not ranges:
        return ranges
    merged = []
    cur_start, cur_end = ranges[0][0], ranges[0][0] + ranges[0][1]
    for start, size in ranges[1:]:
        end = start + size
        if start <= cur_end:
            cur_end = max(cur_end, end)
        else:
            merged.append((cur_start, cur_end - cur_start))
            cur_start, cur_end = start, end
    merged.append((cur_start, cur_end - cur_start))
    return merged


def write_plan(raw_len: int, ranges, threshold_percent=DIRTY_RANGE_FULL_WRITE_THRESHOLD_PERCENT, max_ranges=DIRTY_RANGE_MAX_RANGES):
    """buffer/helpers.rs:138-196: the list of (offset, size) writeBuffer calls; [(0, raw_len)] = full write."""
    if raw_len == 0 or not ranges:
        return []
    if len(ranges) > max_ranges:
        return [(0, raw_len)]
    dirty = sum(s for _, s in ranges)
    if dirty * 100 >= raw_len * threshold_percent:
        return [(0, raw_len)]
    if len(ranges) > 1:
        ranges = coalesce_ranges(sorted(ranges, key=lambda r: r[0]))
    out = []
    for off, size in ranges:
        if size == 0:
            continue
        end = min(off + size, raw_len)
        if end > off:
            out.append((off, end - off))
    return out


# ------------------------------------------------------------------------------------------------ glam (f32)


def v3(x, y, z):
    return np.array([x, y, z], dtype=F)


def mat4_from_srt(scale, rot, trans) -> np.ndarray:
    """Mat4::from_scale_rotation_translation; returns column-major 4x4 as m[col][row]."""
    x, y, z, w = (F(v) for v in rot)
    x2, y2, z2 = x + x, y + y, z + z
    xx, xy, xz = x * x2, x * y2, x * z2
    yy, yz, zz = y * y2, y * z2, z * z2
    wx, wy, wz = w * x2, w * y2, w * z2
    one = F(1.0)
    xa = np.array([one - (yy + zz), xy + wz, xz - wy, 0], dtype=F)
    ya = np.array([xy - wz, one - (xx + zz), yz + wx, 0], dtype=F)
    za = np.array([xz + wy, yz - wx, one - (xx + yy), 0], dtype=F)
    m = np.zeros((4, 4), dtype=F)
    m[0] = xa * F(scale[0])
    m[1] = ya * F(scale[1])
    m[2] = za * F(scale[2])
    m[3] = np.array([trans[0], trans[1], trans[2], 1.0], dtype=F)
    return m


def mat4_mul_vec4(m, v):
    res = m[0] * F(v[0])
    res = res + m[1] * F(v[1])
    res = res + m[2] * F(v[2])
    res = res + m[3] * F(v[3])
    return res.astype(F)


def mat4_mul(a, b):
    out = np.zeros((4, 4), dtype=F)
    for c in range(4):
        out[c] = mat4_mul_vec4(a, b[c])
    return out


def mat4_identity():
    return np.eye(4, dtype=F)


def mat4_transpose(m):
    return np.ascontiguousarray(m.T)


def mat4_inverse(m):
    m00, m01, m02, m03 = m[0]
    m10, m11, m12, m13 = m[1]
    m20, m21, m22, m23 = m[2]
    m30, m31, m32, m33 = m[3]
    coef00 = m22 * m33 - m32 * m23
    coef02 = m12 * m33 - m32 * m13
    coef03 = m12 * m23 - m22 * m13
    coef04 = m21 * m33 - m31 * m23
    coef06 = m11 * m33 - m31 * m13
    coef07 = m11 * m23 - m21 * m13
    coef08 = m21 * m32 - m31 * m22
    coef10 = m11 * m32 - m31 * m12
    coef11 = m11 * m22 - m21 * m12
    coef12 = m20 * m33 - m30 * m23
    coef14 = m10 * m33 - m30 * m13
    coef15 = m10 * m23 - m20 * m13
    coef16 = m20 * m32 - m30 * m22
    coef18 = m10 * m32 - m30 * m12
    coef19 = m10 * m22 - m20 * m12
    coef20 = m20 * m31 - m30 * m21
    coef22 = m10 * m31 - m30 * m11
    coef23 = m10 * m21 - m20 * m11
    A = lambda *v: np.array(v, dtype=F)  # noqa: E731
    fac0, fac1, fac2 = A(coef00, coef00, coef02, coef03), A(coef04, coef04, coef06, coef07), A(coef08, coef08, coef10, coef11)
    fac3, fac4, fac5 = A(coef12, coef12, coef14, coef15), A(coef16, coef16, coef18, coef19), A(coef20, coef20, coef22, coef23)
    vec0, vec1, vec2, vec3_ = A(m10, m00, m00, m00), A(m11, m01, m01, m01), A(m12, m02, m02, m02), A(m13, m03, m03, m03)
    inv0 = vec1 * fac0 - vec2 * fac1 + vec3_ * fac2
    inv1 = vec0 * fac0 - vec2 * fac3 + vec3_ * fac4
    inv2 = vec0 * fac1 - vec1 * fac3 + vec3_ * fac5
    inv3 = vec0 * fac2 - vec1 * fac4 + vec2 * fac5
    sign_a, sign_b = A(1, -1, 1, -1), A(-1, 1, -1, 1)
    inverse = np.stack([inv0 * sign_a, inv1 * sign_b, inv2 * sign_a, inv3 * sign_b]).astype(F)
    col0 = A(inverse[0][0], inverse[1][0], inverse[2][0], inverse[3][0])
    dot0 = m[0] * col0
    dot1 = ((dot0[0] + dot0[1]) + dot0[2]) + dot0[3]
    rcp_det = F(1.0) / dot1
    return (inverse * rcp_det).astype(F)


def mat4_determinant(m):
    m00, m01, m02, m03 = m[0]
    m10, m11, m12, m13 = m[1]
    m20, m21, m22, m23 = m[2]
    m30, m31, m32, m33 = m[3]
    a2323 = m22 * m33 - m23 * m32
    a1323 = m21 * m33 - m23 * m31
    a1223 = m21 * m32 - m22 * m31
    a0323 = m20 * m33 - m23 * m30
    a0223 = m20 * m32 - m22 * m30
    a0123 = m20 * m31 - m21 * m30
    return (m00 * (m11 * a2323 - m12 * a1323 + m13 * a1223) - m01 * (m10 * a2323 - m12 * a0323 + m13 * a0223)
            + m02 * (m10 * a1323 - m11 * a0323 + m13 * a0123) - m03 * (m10 * a1223 - m11 * a0223 + m12 * a0123))


def mat4_transform_point3(m, p):
    res = m[0] * F(p[0])
    res = m[1] * F(p[1]) + res
    res = m[2] * F(p[2]) + res
    res = m[3] + res
    return res[:3].astype(F)


def perspective_rh(fov_y, aspect, z_near, z_far):
    half = F(0.5) * F(fov_y)
    sin_fov, cos_fov = F(math.sin(float(half))), F(math.cos(float(half)))
    h = cos_fov / sin_fov
    w = h / F(aspect)
    r = F(z_far) / (F(z_near) - F(z_far))
    m = np.zeros((4, 4), dtype=F)
    m[0][0] = w
    m[1][1] = h
    m[2][2] = r
    m[2][3] = -1.0
    m[3][2] = r * F(z_near)
    return m


def orthographic_rh(left, right, bottom, top, near, far):
    rcp_w, rcp_h = F(1.0) / (F(right) - F(left)), F(1.0) / (F(top) - F(bottom))
    r = F(1.0) / (F(near) - F(far))
    m = np.zeros((4, 4), dtype=F)
    m[0][0] = rcp_w + rcp_w
    m[1][1] = rcp_h + rcp_h
    m[2][2] = r
    m[3] = np.array([-(F(left) + F(right)) * rcp_w, -(F(top) + F(bottom)) * rcp_h, r * F(near), 1.0], dtype=F)
    return m


def _normalize3(v):
    v = v.astype(F)
    length = np.sqrt(((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]).astype(F)).astype(F)
    return (v / length).astype(F)


def _dot3(a, b):
    return F((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2])


def _cross3(a, b):
    return np.array([a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]], dtype=F)


def look_at_rh(eye, center, up):
    eye, center, up = (np.asarray(x, dtype=F) for x in (eye, center, up))
    f = _normalize3(center - eye)
    s = _normalize3(_cross3(f, up))
    u = _cross3(s, f)
    m = np.zeros((4, 4), dtype=F)
    m[0] = [s[0], u[0], -f[0], 0]
    m[1] = [s[1], u[1], -f[1], 0]
    m[2] = [s[2], u[2], -f[2], 0]
    m[3] = [-_dot3(eye, s), -_dot3(eye, u), _dot3(eye, f), 1]
    return m


# ------------------------------------------------------------------------------------------------ bounds / frustum


class Aabb:
    """bounds.rs:7-75."""

    def __init__(self, mn, mx):
        self.min = np.asarray(mn, dtype=F).copy()
        self.max = np.asarray(mx, dtype=F).copy()

    @staticmethod
    def new_cube(width, height):
        w, h = F(width), F(height)
        return Aabb([-w / 2, -h / 2, -w / 2], [w / 2, h / 2, w / 2])

    def extend(self, other):
        self.min = np.minimum(self.min, other.min)
        self.max = np.maximum(self.max, other.max)

    def transformed(self, mat):
        mn, mx = self.min, self.max
        corners = [(mn[0], mn[1], mn[2]), (mx[0], mn[1], mn[2]), (mn[0], mx[1], mn[2]), (mx[0], mx[1], mn[2]),
                   (mn[0], mn[1], mx[2]), (mx[0], mn[1], mx[2]), (mn[0], mx[1], mx[2]), (mx[0], mx[1], mx[2])]
        first = mat4_transform_point3(mat, corners[0])
        lo, hi = first.copy(), first.copy()
        for c in corners[1:]:
            t = mat4_transform_point3(mat, c)
            lo, hi = np.minimum(lo, t), np.maximum(hi, t)
        return Aabb(lo, hi)


class Frustum:
    """frustum.rs:42-89 (right-handed view-projection, WebGPU 0..1 depth)."""

    def __init__(self, view_projection):
        vp = view_projection
        rows = [np.array([vp[0][r], vp[1][r], vp[2][r], vp[3][r]], dtype=F) for r in range(4)]
        raw = [rows[3] + rows[0], rows[3] - rows[0], rows[3] + rows[1], rows[3] - rows[1], rows[2], rows[3] - rows[2]]
        self.planes = []
        for row in raw:
            n = row[:3].astype(F)
            d = F(row[3])
            length = np.sqrt(((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]).astype(F)).astype(F)
            if length > 0:
                n, d = (n / length).astype(F), F(d / length)
            self.planes.append((n, d))

    def intersects_aabb(self, aabb: Aabb) -> bool:
        for n, d in self.planes:
            p = np.array([aabb.max[i] if n[i] >= 0 else aabb.min[i] for i in range(3)], dtype=F)
            if _dot3(n, p) + d < 0:
                return False
        return True


def f32_bytes(arr) -> bytes:
    return np.asarray(arr, dtype=F).tobytes()


def u32_bytes(*vals) -> bytes:
    return struct.pack("<%dI" % len(vals), *vals)
