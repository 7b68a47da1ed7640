// dirty_log.hpp — what was written to the buffers the vertex stage reads, and when (geometry cache).  Host-side, no HIP: the C-ABI
// layer (awsm_hip.cpp) logs every awsm_hip_buffer_write / buffer_create here, and a frame slot's geometry pass asks for the ranges written since the
// slot's arrays were last computed.  Header-only so that tests/test_dirty_log_cpu.py can drive it with g++ on a box without a GPU.
//
// The producer of these writes is the reference's dirty propagation (crates/renderer/src/transforms.rs:390-435 marks a node and its children dirty;
// buffer/helpers.rs:124-220 turns the dirty ranges of a mirror into merged writes): one write per merged range, a whole-buffer write after a resize.
#pragma once
#include <stdint.h>
#include <algorithm>
#include <vector>

namespace awsm {

struct DirtyRange { uint32_t buf, lo, hi; uint64_t seq; };      // bytes [lo, hi) of buffer `buf`, written by the write with sequence number `seq`

class DirtyLog {
public:
    enum Kind { kIgnored, kPrecise, kGlobal };
    // precise: the kernel resolves the range against each draw's own blocks; global: static per-mesh data whose blocks' extents the kernel does not
    // know — every draw is recomputed once; ignored: nothing the vertex stage reads (the camera is per frame by design).  Indices are AwsmBuf values.
    static Kind kind_of(uint32_t buf) {
        switch (buf) {
        case 14: case 15: case 9: case 7: return kGlobal;      // ATTR_DATA, ATTR_INDEX, MORPH_VALUES, SKIN_INDEX_WEIGHTS
        case 0: case 17: case 10: case 12: case 8: case 6: case 11: return kPrecise;      // TRANSFORMS, INSTANCES, GEOM_META, VIS_GEOM_DATA, MORPH_WEIGHTS, SKIN_MATRICES, MATERIAL_META
        default: return kIgnored;
        }
    }
    static constexpr size_t kMaxEntries = 512;

    // a write of bytes [lo, hi) to `buf` that has just received sequence number `seq` (strictly increasing over all writes)
    void log(uint32_t buf, size_t lo, size_t hi, uint64_t seq) {
        const Kind k = kind_of(buf);
        if (k == kIgnored) return;
        if (k == kGlobal) { all_dirty_seq_ = seq; return; }
        const uint32_t l = (uint32_t)std::min<size_t>(lo, 0xFFFFFFFFu), h = (uint32_t)std::min<size_t>(hi, 0xFFFFFFFFu);
        if (!entries_.empty()) {
            DirtyRange& last = entries_.back();
            if (last.buf == buf && last.seq + 1 >= seq && l <= last.hi && last.lo <= h) {      // consecutive writes that touch: one range, under the newer number (whoever saw the older one sees it)
                last.lo = std::min(last.lo, l); last.hi = std::max(last.hi, h); last.seq = seq; return;
            }
        }
        if (entries_.size() >= kMaxEntries) { all_dirty_seq_ = seq; entries_.clear(); return; }      // too much to remember: everything counts as written
        entries_.push_back({buf, l, h, seq});
    }
    // everything up to this sequence number counts as "all written"
    uint64_t all_dirty_seq() const { return all_dirty_seq_; }
    // The ranges written after `since`, merged per buffer (overlapping or adjacent ones), at most `max_ranges`: with more, one bounding range per buffer;
    // still more: false (the caller recomputes everything).  Also false when something global was written after `since`.
    bool ranges_since(uint64_t since, size_t max_ranges, std::vector<DirtyRange>& out) const {
        out.clear();
        if (all_dirty_seq_ > since) return false;
        for (const DirtyRange& d : entries_) if (d.seq > since) out.push_back(d);
        std::sort(out.begin(), out.end(), [](const DirtyRange& x, const DirtyRange& y) { return x.buf != y.buf ? x.buf < y.buf : x.lo < y.lo; });
        auto merge = [&](bool whole_buffer) {
            std::vector<DirtyRange> m;
            for (const DirtyRange& d : out) {
                if (!m.empty() && m.back().buf == d.buf && (whole_buffer || d.lo <= m.back().hi)) { m.back().hi = std::max(m.back().hi, d.hi); m.back().seq = std::max(m.back().seq, d.seq); }
                else m.push_back(d);
            }
            out.swap(m);
        };
        merge(false);
        if (out.size() > max_ranges) merge(true);
        if (out.size() > max_ranges) { out.clear(); return false; }
        return true;
    }
    // entries every reader has seen (sequence numbers up to `oldest`) are of no further use
    void prune(uint64_t oldest) {
        size_t keep = 0;
        for (const DirtyRange& d : entries_) if (d.seq > oldest) entries_[keep++] = d;
        entries_.resize(keep);
    }
    size_t size() const { return entries_.size(); }

private:
    std::vector<DirtyRange> entries_;
    uint64_t all_dirty_seq_ = 0;
};

}  // namespace awsm
