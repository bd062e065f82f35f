// gv_dirty_ranges.hpp — itemised dirty slot ranges of a bound pool (host-only, no HIP: unit-tested on the CPU tier under
// ASan / UBSan, tests/cpp/dirty_ranges_test.cpp).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace gv {

// Itemised dirty marks of a pool (setPosition on scattered entities, transform.hpp:74-104): kept as disjoint
// ranges instead of one covering range, so a handful of moved entities at opposite ends of a 10 M pool re-mirror a
// handful of slots, not everything in between. Overlapping and adjacent ranges are merged; beyond kMax ranges, ranges
// closer than a growing `gap` are merged too until they fit.
struct DirtyRanges {
    struct R {
        uint32_t lo, hi;
    };
    std::vector<R> items;
    static constexpr size_t kMax = 16384;
    bool any() const { return !items.empty(); }
    void clear() { items.clear(); }
    void add(uint32_t first, uint32_t count)
    {
        if (count == 0)
            return;
        const uint32_t hi = (uint32_t)std::min<uint64_t>((uint64_t)first + count, UINT32_MAX);  // saturating, see DirtyRange
        if (!items.empty() && first <= items.back().hi && hi >= items.back().lo) {  // extends / overlaps the last mark
            items.back().lo = std::min(items.back().lo, first);
            items.back().hi = std::max(items.back().hi, hi);
            return;
        }
        items.push_back({first, hi});
        if (items.size() > 4 * kMax)
            normalise(UINT32_MAX, 0);
    }
    // sorted, clamped to [0, limit), merged
    void normalise(uint32_t limit, uint32_t gap)
    {
        for (auto& r : items)
            r.hi = std::min(r.hi, limit);
        items.erase(std::remove_if(items.begin(), items.end(), [](const R& r) { return r.lo >= r.hi; }), items.end());
        std::sort(items.begin(), items.end(), [](const R& a, const R& b) { return a.lo < b.lo; });
        for (;;) {
            std::vector<R> merged;
            for (const R& r : items) {
                if (!merged.empty() && (uint64_t)r.lo <= (uint64_t)merged.back().hi + gap)
                    merged.back().hi = std::max(merged.back().hi, r.hi);
                else
                    merged.push_back(r);
            }
            items.swap(merged);
            if (items.size() <= kMax || gap >= (1u << 30))
                break;
            gap = gap ? gap * 2 : 1;
        }
    }
    uint64_t total() const
    {
        uint64_t t = 0;
        for (const R& r : items)
            t += r.hi - r.lo;
        return t;
    }
};

}  // namespace gv
