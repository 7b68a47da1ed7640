// buffers.hpp — CPU-side GPU mirrors of the reference, re-implemented for the C++ host layer.
//
//   SlotKey / SlotMap            slotmap 1.1.1 key semantics (key = version << 32 | idx, first key idx 1 version 1,
//                                LIFO free list, dense insertion-ordered iteration with swap-remove like DenseSlotMap)
//   DynamicUniformBuffer         /root/reference/crates/renderer/src/buffer/dynamic_uniform.rs:40-289
//   DynamicStorageBuffer         /root/reference/crates/renderer/src/buffer/dynamic_storage.rs:39-409
//   write_plan                   /root/reference/crates/renderer/src/buffer/helpers.rs:124-220
//
// Behaviour (offsets, growth sizes, dirty ranges, resize flag) is pinned by the reference's own unit tests,
// restated in tests/test_buffers_reference_cases.py and run against this implementation through the C API.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <utility>
#include <vector>

namespace awsm_host {

using SlotKey = uint64_t;   // KeyData::as_ffi(): (version << 32) | idx ; 0 is never a valid key
inline uint32_t key_idx(SlotKey k) { return (uint32_t)(k & 0xFFFFFFFFull); }
inline uint32_t key_version(SlotKey k) { return (uint32_t)(k >> 32); }

template <typename T>
class SlotMap {
  public:
    SlotMap() { slots_.push_back({0u, 0u, 0u}); }
    SlotKey insert(T value) {
        uint32_t idx = free_head_;
        if (idx < slots_.size()) {
            free_head_ = slots_[idx].next_free;
            slots_[idx].version |= 1u;
        } else {
            slots_.push_back({1u, 0u, 0u});
            free_head_ = (uint32_t)slots_.size();
        }
        slots_[idx].dense = (uint32_t)dense_keys_.size();
        SlotKey k = ((uint64_t)slots_[idx].version << 32) | idx;
        dense_keys_.push_back(k);
        dense_vals_.push_back(std::move(value));
        return k;
    }
    bool contains(SlotKey k) const {
        uint32_t i = key_idx(k);
        return i != 0 && i < slots_.size() && slots_[i].version == key_version(k) && (slots_[i].version & 1u);
    }
    T* get(SlotKey k) { return contains(k) ? &dense_vals_[slots_[key_idx(k)].dense] : nullptr; }
    const T* get(SlotKey k) const { return contains(k) ? &dense_vals_[slots_[key_idx(k)].dense] : nullptr; }
    bool remove(SlotKey k) {
        if (!contains(k)) return false;
        uint32_t i = key_idx(k);
        uint32_t d = slots_[i].dense;
        slots_[i].version += 1u;
        slots_[i].next_free = free_head_;
        free_head_ = i;
        uint32_t last = (uint32_t)dense_keys_.size() - 1;
        if (d != last) {   // swap_remove
            dense_keys_[d] = dense_keys_[last];
            dense_vals_[d] = std::move(dense_vals_[last]);
            slots_[key_idx(dense_keys_[d])].dense = d;
        }
        dense_keys_.pop_back();
        dense_vals_.pop_back();
        return true;
    }
    size_t size() const { return dense_keys_.size(); }
    const std::vector<SlotKey>& keys() const { return dense_keys_; }
    std::vector<T>& values() { return dense_vals_; }
    const std::vector<T>& values() const { return dense_vals_; }

  private:
    struct Slot { uint32_t version, next_free, dense; };
    std::vector<Slot> slots_;
    uint32_t free_head_ = 1;
    std::vector<SlotKey> dense_keys_;
    std::vector<T> dense_vals_;
};

using DirtyRange = std::pair<size_t, size_t>;   // (offset, size)

inline void mark_dirty_range(std::vector<DirtyRange>& ranges, size_t raw_len, size_t offset, size_t size) {
    if (size == 0 || raw_len == 0 || offset >= raw_len) return;
    size_t start = offset & ~size_t(3);
    size_t end = std::min(offset + size, raw_len);
    end = std::min((end + 3) & ~size_t(3), raw_len);   // WebGPU writeBuffer: 4-byte aligned
    if (start < end) ranges.push_back({start, end - start});
}

// ---------------------------------------------------------------------------------------------------------
class DynamicUniformBuffer {
  public:
    DynamicUniformBuffer(size_t initial_capacity, size_t byte_size, size_t aligned_slice_size = 0, uint8_t zero = 0)
        : byte_size_(byte_size), aligned_(aligned_slice_size ? aligned_slice_size : byte_size), zero_(zero),
          capacity_slots_(initial_capacity), next_slot_(initial_capacity) {
        raw_.assign(initial_capacity * aligned_, zero);
        for (size_t i = initial_capacity; i-- > 0;) free_slots_.push_back(i);   // reversed so slot 0 is used first
    }
    // returns false if values exceed byte_size (the reference panics)
    bool update(SlotKey key, const uint8_t* values, size_t len) {
        if (len > byte_size_) return false;
        size_t off = slot_for(key) * aligned_;
        if (len) memcpy(raw_.data() + off, values, len);
        mark_dirty_range(dirty_, raw_.size(), off, byte_size_);
        return true;
    }
    bool update_offset(SlotKey key, size_t offset, const uint8_t* values, size_t len) {
        if (offset + len > byte_size_) return false;
        size_t off = slot_for(key) * aligned_;
        if (len) memcpy(raw_.data() + off + offset, values, len);
        mark_dirty_range(dirty_, raw_.size(), off, byte_size_);
        return true;
    }
    bool remove(SlotKey key) {
        auto it = slot_indices_.find(key);
        if (it == slot_indices_.end()) return false;
        size_t slot = it->second;
        slot_indices_.erase(it);
        free_slots_.push_back(slot);
        size_t off = slot * aligned_;
        std::fill(raw_.begin() + off, raw_.begin() + off + aligned_, zero_);
        mark_dirty_range(dirty_, raw_.size(), off, aligned_);
        return true;
    }
    bool contains(SlotKey key) const { return slot_indices_.count(key) != 0; }
    long long offset(SlotKey key) const { auto it = slot_indices_.find(key); return it == slot_indices_.end() ? -1 : (long long)(it->second * aligned_); }
    long long slot_index(SlotKey key) const { auto it = slot_indices_.find(key); return it == slot_indices_.end() ? -1 : (long long)it->second; }
    size_t size() const { return raw_.size(); }
    size_t len() const { return slot_indices_.size(); }
    size_t capacity() const { return capacity_slots_; }
    size_t next_slot() const { return next_slot_; }
    size_t byte_size() const { return byte_size_; }
    size_t aligned_slice_size() const { return aligned_; }
    const std::vector<size_t>& free_slots() const { return free_slots_; }
    const std::vector<uint8_t>& raw() const { return raw_; }
    std::vector<DirtyRange> take_dirty_ranges() { std::vector<DirtyRange> r; r.swap(dirty_); return r; }
    void clear_dirty_ranges() { dirty_.clear(); }
    long long take_gpu_needs_resize() { long long s = needs_resize_ ? (long long)raw_.size() : -1; needs_resize_ = false; return s; }
    // test hooks mirroring the reference tests that poke private state (dynamic_uniform.rs:769-771,868-870,1382-1384)
    void test_force_state(size_t next_slot) { free_slots_.clear(); next_slot_ = next_slot; }

  private:
    size_t slot_for(SlotKey key) {
        auto it = slot_indices_.find(key);
        if (it != slot_indices_.end()) return it->second;
        size_t slot;
        if (!free_slots_.empty()) { slot = free_slots_.back(); free_slots_.pop_back(); }
        else {
            slot = next_slot_;
            if ((slot + 1) * aligned_ > raw_.size()) resize(slot + 1);
            next_slot_ += 1;
        }
        slot_indices_[key] = slot;
        return slot;
    }
    void resize(size_t required_slots) {
        size_t new_cap = std::max(required_slots, capacity_slots_) * 2;
        raw_.resize(new_cap * aligned_, zero_);
        for (size_t s = required_slots; s < new_cap; s++) free_slots_.push_back(s);
        next_slot_ = new_cap;
        capacity_slots_ = new_cap;
        needs_resize_ = true;
    }
    size_t byte_size_, aligned_;
    uint8_t zero_;
    std::vector<uint8_t> raw_;
    std::vector<DirtyRange> dirty_;
    bool needs_resize_ = false;
    std::unordered_map<SlotKey, size_t> slot_indices_;
    std::vector<size_t> free_slots_;
    size_t capacity_slots_, next_slot_;
};

// ---------------------------------------------------------------------------------------------------------
class DynamicStorageBuffer {
  public:
    static constexpr size_t MIN_BLOCK = 256;
    static size_t round_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return std::max(p, MIN_BLOCK); }
    static size_t index_to_offset(size_t idx, size_t leaves) { while (idx < leaves - 1) idx = idx * 2 + 1; return (idx + 1 - leaves) * MIN_BLOCK; }
    static size_t offset_to_index(size_t off, size_t leaves) { return leaves - 1 + off / MIN_BLOCK; }

    explicit DynamicStorageBuffer(size_t initial_bytes, uint8_t zero = 0) : zero_(zero) {
        size_t cap = round_pow2(std::max(initial_bytes, MIN_BLOCK));
        raw_.assign(cap, zero);
        needs_resize_ = cap != initial_bytes;
        init_tree(cap);
    }
    size_t update(SlotKey key, const uint8_t* bytes, size_t len) {
        auto it = slots_.find(key);
        if (it != slots_.end()) {
            size_t off = it->second.first, old = it->second.second;
            if (len <= old) {
                if (len) memcpy(raw_.data() + off, bytes, len);
                if (len < old) std::fill(raw_.begin() + off + len, raw_.begin() + off + old, zero_);
                mark_dirty_range(dirty_, raw_.size(), off, old);
                return off;
            }
            remove(key);
        }
        return insert(key, bytes, len);
    }
    // f(offset, block_ptr, block_size); returns false if the key is missing (the reference panics)
    template <typename Fn>
    bool update_with_unchecked(SlotKey key, Fn f) {
        auto it = slots_.find(key);
        if (it == slots_.end()) return false;
        f(it->second.first, raw_.data() + it->second.first, it->second.second);
        mark_dirty_range(dirty_, raw_.size(), it->second.first, it->second.second);
        return true;
    }
    void remove(SlotKey key) {
        auto it = slots_.find(key);
        if (it == slots_.end()) return;
        size_t off = it->second.first, size = it->second.second;
        slots_.erase(it);
        std::fill(raw_.begin() + off, raw_.begin() + off + size, zero_);
        mark_dirty_range(dirty_, raw_.size(), off, size);
        free_block(off, size);
    }
    bool contains(SlotKey key) const { return slots_.count(key) != 0; }
    long long offset(SlotKey key) const { auto it = slots_.find(key); return it == slots_.end() ? -1 : (long long)it->second.first; }
    long long size_of(SlotKey key) const { auto it = slots_.find(key); return it == slots_.end() ? -1 : (long long)it->second.second; }
    size_t used_size() const { size_t s = 0; for (auto& kv : slots_) s += kv.second.second; return s; }
    size_t len() const { return slots_.size(); }
    size_t capacity() const { return raw_.size(); }
    size_t tree_root() const { return tree_[0]; }
    const std::vector<uint8_t>& raw() const { return raw_; }
    std::vector<DirtyRange> take_dirty_ranges() { std::vector<DirtyRange> r; r.swap(dirty_); return r; }
    void clear_dirty_ranges() { dirty_.clear(); }
    long long take_gpu_needs_resize() { long long s = needs_resize_ ? (long long)raw_.size() : -1; needs_resize_ = false; return s; }

  private:
    void init_tree(size_t cap) {
        size_t leaves = cap / MIN_BLOCK;
        tree_.assign(2 * leaves - 1, 0);
        size_t size = cap, start = 0, count = 1;
        for (;;) {
            for (size_t i = start; i < start + count; i++) tree_[i] = size;
            if (size <= MIN_BLOCK) break;
            start += count; count *= 2; size /= 2;
        }
    }
    void fix_parents(size_t idx) {
        while (idx != 0) {
            size_t parent = (idx - 1) >> 1, left = parent * 2 + 1;
            size_t nv = std::max(tree_[left], tree_[left + 1]);
            if (tree_[parent] == nv) break;
            tree_[parent] = nv;
            idx = parent;
        }
    }
    bool alloc(size_t req, size_t* out) {
        if (req > tree_[0]) return false;
        size_t idx = 0, size = raw_.size();
        while (size > req) {
            size_t left = idx * 2 + 1;
            idx = tree_[left] >= req ? left : left + 1;   // first fit, left first
            size /= 2;
        }
        tree_[idx] = 0;
        fix_parents(idx);
        *out = index_to_offset(idx, raw_.size() / MIN_BLOCK);
        return true;
    }
    void free_block(size_t offset, size_t size) {
        size_t leaves = raw_.size() / MIN_BLOCK;
        size_t idx = offset_to_index(offset, leaves), blk = MIN_BLOCK;
        while (blk < size) { idx = (idx - 1) >> 1; blk <<= 1; }
        tree_[idx] = blk;
        while (idx != 0) {
            size_t parent = (idx - 1) >> 1, left = parent * 2 + 1, right = left + 1;
            bool merged = tree_[left] == blk && tree_[right] == blk;
            size_t nv = merged ? (blk << 1) : std::max(tree_[left], tree_[right]);
            if (tree_[parent] == nv) break;
            tree_[parent] = nv;
            if (merged) { idx = parent; blk <<= 1; } else break;
        }
    }
    void grow(size_t min_extra) {
        size_t old_cap = raw_.size(), new_cap = old_cap * 2;
        while (new_cap - old_cap < min_extra) new_cap *= 2;
        raw_.resize(new_cap, zero_);
        needs_resize_ = true;
        init_tree(new_cap);
        size_t leaves = new_cap / MIN_BLOCK;
        for (auto& kv : slots_) {   // re-mark existing allocations as used
            size_t idx = offset_to_index(kv.second.first, leaves), sz = MIN_BLOCK;
            while (sz < kv.second.second) { idx = (idx - 1) >> 1; sz <<= 1; }
            tree_[idx] = 0;
            fix_parents(idx);
        }
    }
    size_t insert(SlotKey key, const uint8_t* bytes, size_t len) {
        size_t req = round_pow2(std::max(len, MIN_BLOCK));
        size_t off;
        if (!alloc(req, &off)) {
            grow(std::max(req, raw_.size()));
            alloc(req, &off);
        }
        if (len) memcpy(raw_.data() + off, bytes, len);
        slots_[key] = {off, req};
        mark_dirty_range(dirty_, raw_.size(), off, req);
        return off;
    }
    uint8_t zero_;
    std::vector<uint8_t> raw_;
    std::vector<DirtyRange> dirty_;
    std::vector<size_t> tree_;
    std::unordered_map<SlotKey, std::pair<size_t, size_t>> slots_;
    bool needs_resize_ = false;
};

// helpers.rs:124-220.  Returns the writeBuffer calls to issue; a single (0, raw_len) entry is the full write.
inline std::vector<DirtyRange> write_plan(size_t raw_len, std::vector<DirtyRange> ranges, uint64_t threshold_percent = 60, size_t max_ranges = 32) {
    std::vector<DirtyRange> out;
    if (raw_len == 0 || ranges.empty()) return out;
    if (ranges.size() > max_ranges) { out.push_back({0, raw_len}); return out; }
    uint64_t dirty = 0;
    for (auto& r : ranges) dirty += r.second;
    if (dirty * 100 >= (uint64_t)raw_len * threshold_percent) { out.push_back({0, raw_len}); return out; }
    if (ranges.size() > 1) {
        std::sort(ranges.begin(), ranges.end(), [](const DirtyRange& a, const DirtyRange& b) { return a.first < b.first; });
        std::vector<DirtyRange> merged;
        size_t cs = ranges[0].first, ce = cs + ranges[0].second;
        for (size_t i = 1; i < ranges.size(); i++) {
            size_t s = ranges[i].first, e = s + ranges[i].second;
            if (s <= ce) ce = std::max(ce, e);
            else { merged.push_back({cs, ce - cs}); cs = s; ce = e; }
        }
        merged.push_back({cs, ce - cs});
        ranges.swap(merged);
    }
    for (auto& r : ranges) {
        if (r.second == 0) continue;
        size_t end = std::min(r.first + r.second, raw_len);
        if (end > r.first) out.push_back({r.first, end - r.first});
    }
    return out;
}

}  // namespace awsm_host
