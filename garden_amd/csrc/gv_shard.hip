// gv_shard.hip — exchange-shard encodings that are not on the visibility path itself (the index-list shard lives with the
// cull kernels' other outputs in gv_cull.hip).
#include "gv_device.hpp"

namespace gv {

// The visible list of a view as a BIT per mirror entry behind its count: dst[0] = draw_count, bit (e & 31) of dst[1 + (e >> 5)]
// = mirror entry e is visible. That is the cull kernel's own output (one 64-bit ballot word per wave, little-endian = two of
// these words), so the shard is a 1.6 MB copy per 12.5 M entries instead of a scatter: turning the bits into POOL-slot order
// on the device was measured first — 2.6 M atomicOr into random words: +126 us per frame, the byte scatter of the fetch path
// 176 us — and dropped; a consumer maps entry -> pool slot with the owner's table (gv_pool_mirror_slots), which only
// changes when the mirror is rebuilt. `bytes` (mirror order) replaces `ballots` for the one-launch cull + emit of small
// pools, which produces no ballot words.
__global__ __launch_bounds__(256) void mask_shard_kernel(const uint32_t* __restrict__ ballots, const uint8_t* __restrict__ bytes,
                                                         const uint32_t* __restrict__ count, uint32_t entries, uint32_t* __restrict__ dst,
                                                         uint32_t words)
{
    if (blockIdx.x == 0 && threadIdx.x == 0)
        dst[0] = *count;
    const uint32_t live_words = (entries + 31u) / 32u;
    if (ballots) {
        for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < words; w += gridDim.x * blockDim.x)
            dst[1 + w] = w < live_words ? ballots[w] : 0u;
        return;
    }
    // from the isVisible bytes: lane l of a wave owns entry 64 * k + l of the wave's k-th round
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t pair = wave; pair * 2u < words; pair += waves) {  // wave-uniform
        const uint32_t e = pair * 64u + lane;
        const unsigned long long word = __ballot(e < entries && bytes[e] != 0);
        if (lane == 0) {
            dst[1 + pair * 2u] = (uint32_t)word;
            if (pair * 2u + 1u < words)
                dst[2 + pair * 2u] = (uint32_t)(word >> 32);
        }
    }
}

hipError_t launch_mask_shard(const unsigned long long* ballots, const uint8_t* bytes, const uint32_t* count, uint32_t entries, uint32_t* dst,
                             uint32_t words, hipStream_t stream)
{
    const uint32_t blocks = std::max(1u, std::min(1024u, (words + 255u) / 256u));
    hipLaunchKernelGGL(mask_shard_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const uint32_t*>(ballots), bytes, count, entries,
                       dst, words);
    return hipGetLastError();
}

// The count header of every gathered row -> pinned host memory, then the frame's sequence number behind a system-scope
// release: what gv_exchange_visible sizes the next frames' shards from (gv_exchange.cpp). One wave.
__global__ __launch_bounds__(64) void exchange_headers_kernel(const uint32_t* __restrict__ rows, uint32_t row_words, uint32_t world,
                                                             uint32_t* __restrict__ host_words, uint32_t seq)
{
    for (uint32_t r = threadIdx.x; r < world; r += 64)
        __hip_atomic_store(host_words + r, rows[(size_t)r * row_words], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(host_words + world, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_exchange_headers(const uint32_t* rows, uint32_t row_words, uint32_t world, uint32_t* host_words, uint32_t seq,
                                   hipStream_t stream)
{
    hipLaunchKernelGGL(exchange_headers_kernel, dim3(1), dim3(64), 0, stream, rows, row_words, world, host_words, seq);
    return hipGetLastError();
}

}  // namespace gv
