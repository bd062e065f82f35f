// dirty_ranges_test.cpp — gv::DirtyRanges (garden_amd/csrc/gv_dirty_ranges.hpp) against a bitmap model: random marks
// (overlapping, adjacent, empty, wrapping first + count), normalise with and without a gap, the kMax collapse. The
// invariant the mirror relies on: after normalise(limit, gap) the ranges are sorted, disjoint, inside [0, limit), cover
// EVERY marked slot below the limit, and with gap == 0 cover nothing else. Built with -fsanitize=address,undefined.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../garden_amd/csrc/gv_dirty_ranges.hpp"

static uint32_t rnd(uint64_t& s)
{
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(s >> 33);
}

int main()
{
    uint64_t seed = 12345;
    for (int round = 0; round < 400; round++) {
        const uint32_t limit = 1000 + rnd(seed) % 60000;
        std::vector<uint8_t> marked(limit, 0);
        gv::DirtyRanges d;
        const int marks = 1 + (int)(rnd(seed) % (round % 7 == 0 ? 40000 : 300));
        for (int k = 0; k < marks; k++) {
            uint32_t first = rnd(seed) % (limit + 50), count = rnd(seed) % 40;
            if (rnd(seed) % 50 == 0)
                count = 0xFFFFFFFFu - (rnd(seed) % 3);  // first + count wraps: must saturate, not vanish
            if (rnd(seed) % 5 == 0 && !d.items.empty())
                first = d.items.back().hi;  // adjacent to the previous mark
            d.add(first, count);
            for (uint64_t i = first; i < (uint64_t)first + count && i < limit; i++)
                marked[i] = 1;
        }
        const uint32_t gap = round % 3 == 0 ? 0u : rnd(seed) % 64;
        d.normalise(limit, gap);
        if (d.items.size() > gv::DirtyRanges::kMax) {
            printf("{\"ok\": false, \"why\": \"%zu ranges after normalise\"}\n", d.items.size());
            return 1;
        }
        std::vector<uint8_t> covered(limit, 0);
        uint32_t prev_hi = 0;
        bool first_range = true;
        uint64_t total = 0;
        for (const auto& r : d.items) {
            if (r.lo >= r.hi || r.hi > limit || (!first_range && r.lo <= prev_hi && gap == 0 && r.lo < prev_hi)) {
                printf("{\"ok\": false, \"why\": \"bad range [%u, %u) limit %u\"}\n", r.lo, r.hi, limit);
                return 1;
            }
            if (!first_range && r.lo < prev_hi) {
                printf("{\"ok\": false, \"why\": \"ranges overlap\"}\n");
                return 1;
            }
            for (uint32_t i = r.lo; i < r.hi; i++)
                covered[i] = 1;
            total += r.hi - r.lo;
            prev_hi = r.hi;
            first_range = false;
        }
        if (total != d.total()) {
            printf("{\"ok\": false, \"why\": \"total()\"}\n");
            return 1;
        }
        const bool exact = gap == 0 && (size_t)marks <= gv::DirtyRanges::kMax;  // (a collapse beyond kMax may widen ranges)
        for (uint32_t i = 0; i < limit; i++)
            if ((marked[i] && !covered[i]) || (exact && covered[i] && !marked[i])) {
                printf("{\"ok\": false, \"why\": \"slot %u marked %d covered %d (round %d)\"}\n", i, marked[i], covered[i], round);
                return 1;
            }
    }
    printf("{\"ok\": true}\n");
    return 0;
}
