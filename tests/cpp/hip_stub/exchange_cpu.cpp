// TEST-ONLY (see hip/hip_runtime.h): host stand-ins for the launch functions the exchange step's HOST half depends on
// for its decisions — the shard copy and the row headers that size later frames (garden_amd/csrc/gv_exchange.cpp) — so that
// gv_exchange_visible / gv_exchange_shards run for real under the sanitizers with several ranks (threads) over
// tests/cpp/rccl_stub. "Device" memory is host memory here. Never linked into the product library.
#include <algorithm>
#include <atomic>

#include "gv_kernels.hpp"

namespace gv {

hipError_t launch_copy_idx(const uint32_t* src, const uint32_t* count, uint32_t* dst, uint32_t capacity, uint32_t base,
                           const uint32_t* map, hipStream_t)
{
    const uint32_t n = std::min(*count, capacity);
    for (uint32_t i = 0; i < n; i++)
        dst[i] = (map ? map[src[i]] : src[i]) + base;
    return hipSuccess;
}

hipError_t launch_copy_shard(const uint32_t* src, const uint32_t* count, uint32_t* dst, uint32_t capacity, uint32_t base,
                             const uint32_t* map, hipStream_t)
{
    const uint32_t total = *count, n = std::min(total, capacity);
    dst[0] = total;
    for (uint32_t i = 0; i < n; i++)
        dst[1 + i] = (map ? map[src[i]] : src[i]) + base;
    return hipSuccess;
}

hipError_t launch_copy_shard_batch(const ShardItem* items, uint32_t n, uint32_t, uint32_t* dst, hipStream_t)
{
    uint32_t at = 1u + n, total = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t c = std::min(*items[i].count, items[i].capacity);
        dst[1u + i] = c;
        for (uint32_t k = 0; k < c; k++)
            dst[at + k] = (items[i].map ? items[i].map[items[i].src[k]] : items[i].src[k]) + items[i].base;
        at += c;
        total += c;
    }
    dst[0] = n + total;
    return hipSuccess;
}

hipError_t launch_exchange_headers(const uint32_t* rows, uint32_t row_words, uint32_t world, uint32_t hdr_words, uint32_t* host_words, uint32_t seq,
                                   hipStream_t)
{
    for (uint32_t r = 0; r < world; r++)
        for (uint32_t w = 0; w < hdr_words; w++)
            host_words[1u + (size_t)r * hdr_words + w] = rows[(size_t)r * row_words + w];
    std::atomic_thread_fence(std::memory_order_release);
    host_words[0] = seq;
    return hipSuccess;
}

hipError_t launch_peer_scatter(const uint32_t* shard, uint32_t cap_words, const PeerRows& rows, uint32_t world, hipStream_t)
{
    const uint32_t words = std::min(1u + shard[0], cap_words);
    for (uint32_t r = 0; r < world; r++)
        std::copy(shard, shard + words, rows.dst[r]);
    return hipSuccess;
}

}  // namespace gv
