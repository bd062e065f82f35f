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

// gv_exchange_views: ALL the lists of a frame into one shard — dst = [n + total, c_0 .. c_{n-1}, list 0, list 1 ...], every list
// as copy_shard_kernel writes it (index map, base). blockIdx.y = item; each workgroup derives its list's place from the counts in
// front of it (n <= 128 words: one wave).
__global__ __launch_bounds__(256) void copy_shard_batch_kernel(const ShardItem* __restrict__ items, uint32_t n, uint32_t* __restrict__ dst)
{
    __shared__ uint32_t s_offset, s_count;
    const uint32_t i = blockIdx.y;
    if (threadIdx.x < 64) {
        uint32_t before = 0, total = 0;
        for (uint32_t k = threadIdx.x; k < n; k += 64) {
            const uint32_t c = min(*items[k].count, items[k].capacity);
            total += c;
            before += k < i ? c : 0u;
        }
        for (int off = 32; off; off >>= 1) {
            before += __shfl_down(before, off);
            total += __shfl_down(total, off);
        }
        if (threadIdx.x == 0) {
            s_offset = 1u + n + before;
            s_count = min(*items[i].count, items[i].capacity);
            if (blockIdx.x == 0) {
                dst[1u + i] = s_count;
                if (i == 0)
                    dst[0] = n + total;
            }
        }
    }
    __syncthreads();
    const uint32_t* __restrict__ src = items[i].src;
    const uint32_t* __restrict__ map = items[i].map;
    const uint32_t base = items[i].base, count = s_count;
    uint32_t* __restrict__ out = dst + s_offset;
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < count; j += gridDim.x * 256)
        out[j] = (map ? map[src[j]] : src[j]) + base;
}

hipError_t launch_copy_shard_batch(const ShardItem* device_items, uint32_t n, uint32_t widest, uint32_t* dst, hipStream_t stream)
{
    const uint32_t blocks = std::max(1u, std::min(2048u, (widest + 255u) / 256u));
    hipLaunchKernelGGL(copy_shard_batch_kernel, dim3(blocks, n), dim3(256), 0, stream, device_items, n, dst);
    return hipGetLastError();
}

// The leading `hdr_words` words of every gathered row (the count header, and behind it the per-list counts of a gv_exchange_views
// frame) -> pinned host memory, then the frame's sequence number behind a system-scope release: what gv_exchange_visible sizes the
// next frames' shards from (gv_exchange.cpp). host_words[0] = seq, host_words[1 + r * hdr_words + w] = word w of row r.
__global__ __launch_bounds__(256) void exchange_headers_kernel(const uint32_t* __restrict__ rows, uint32_t row_words, uint32_t world, uint32_t hdr_words,
                                                              uint32_t* __restrict__ host_words, uint32_t seq)
{
    for (uint32_t k = threadIdx.x; k < world * hdr_words; k += 256) {
        const uint32_t r = k / hdr_words, w = k - r * hdr_words;
        __hip_atomic_store(host_words + 1u + k, rows[(size_t)r * row_words + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(host_words, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_exchange_headers(const uint32_t* rows, uint32_t row_words, uint32_t world, uint32_t hdr_words, uint32_t* host_words, uint32_t seq,
                                   hipStream_t stream)
{
    hipLaunchKernelGGL(exchange_headers_kernel, dim3(1), dim3(256), 0, stream, rows, row_words, world, hdr_words, host_words, seq);
    return hipGetLastError();
}

// GV_EXCHANGE_PEER (gv_exchange_init_peers): what an all-gatherv is on a fully connected node whose devices one process holds —
// every rank stores its list into its row of every rank's rows. The length is read from the shard's own header (1 + shard[0] words:
// the count, a batched frame's table, the lists), so only what the lists really hold crosses a link and the host predicts
// nothing. 16-byte stores; a workgroup's piece goes to all destinations before the next piece (every link busy at once, the piece
// read once). Shard and rows are 16-byte aligned (hipMalloc; row strides are multiples of 4 words).
__global__ __launch_bounds__(256) void peer_scatter_kernel(const uint32_t* __restrict__ shard, uint32_t cap_words, PeerRows rows, uint32_t world)
{
    const uint32_t words = min(1u + shard[0], cap_words), quads = words >> 2;
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(shard);
    for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < quads; q += gridDim.x * 256) {
        const uint4 v = src[q];
        for (uint32_t r = 0; r < world; r++)
            reinterpret_cast<uint4*>(rows.dst[r])[q] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (words & 3u)) {
        const uint32_t w = (quads << 2) + threadIdx.x, v = shard[w];
        for (uint32_t r = 0; r < world; r++)
            rows.dst[r][w] = v;
    }
}

hipError_t launch_peer_scatter(const uint32_t* shard, uint32_t cap_words, const PeerRows& rows, uint32_t world, hipStream_t stream)
{
    // (the list's length is the device's knowledge: the grid covers the capacity at 16 quads per lane, workgroups beyond the list leave at once)
    const uint32_t blocks = std::max(1u, std::min(1024u, (cap_words / 4u + 256u * 16u - 1u) / (256u * 16u)));
    hipLaunchKernelGGL(peer_scatter_kernel, dim3(blocks), dim3(256), 0, stream, shard, cap_words, rows, world);
    return hipGetLastError();
}

}  // namespace gv
