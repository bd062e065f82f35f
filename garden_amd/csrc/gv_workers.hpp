// gv_workers.hpp — persistent host worker threads for the library's O(N) host passes (AoS -> mirror gathers, the
// mirror order build, the isVisible write-back). The reference runs its loops on a long-lived ThreadPool
// (source/thread-pool.cpp:173-200: contiguous ranges, the caller takes part); spawning threads per call instead cost
// ~1.2 ms per tick at 10^5 entities (16 threads, twice), more than everything else in the tick together.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <thread>

namespace gv {

// Runs job(part) for part = 0 .. parts-1 on the parked workers plus the calling thread; returns when all are done.
// One run at a time per process (contexts on different threads queue up behind each other).
void run_parts(uint32_t parts, const std::function<void(uint32_t)>& job);

// how many parts a pass over `count` items is worth: 1 below 128 Ki items (waking the workers costs ~60 us on the
// 256-thread host, about what one thread needs for a 10^5-item pass)
inline uint32_t worker_parts(size_t count)
{
    static const uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
    constexpr size_t floor = (size_t)1 << 17;
    // 16 threads: measured on the 256-thread box, 64 made the 10 M gather slower (60 vs 34 ms)
    return count < floor ? 1u : std::min(hw, 16u);
}

// fn(lo, hi) over contiguous sub-ranges of [first, first + count)  (ThreadPool::addItems, thread-pool.cpp:180-194).
// floor_items: the pass is worth the workers from that many items up (default: worker_parts' 128 Ki — a pass whose items each
// miss the cache, like the strided isVisible write-back, pays for the wake-up much earlier)
template <typename F>
void parallel_ranges(uint32_t first, uint32_t count, F&& fn, uint32_t floor_items = 0)
{
    const uint32_t parts = floor_items && count >= floor_items ? worker_parts((size_t)1 << 30) : worker_parts(count);
    if (parts == 1) {
        fn(first, first + count);
        return;
    }
    const uint32_t per = (count + parts - 1) / parts;
    run_parts(parts, [&](uint32_t t) {
        const uint32_t lo = first + std::min(count, per * t), hi = first + std::min(count, per * (t + 1));
        if (lo < hi)
            fn(lo, hi);
    });
}

}  // namespace gv
