// gv_shard.hip — exchange-shard encodings that are not on the visibility path itself (the index-list shard lives with the
// cull kernels' other outputs in gv_cull.hip).
#include "gv_device.hpp"

namespace gv {

// The visible list of a view as a BIT per pool slot behind its count: dst[0] = draw_count, bit (s & 31) of dst[1 + (s >> 5)] =
// slot s is visible. dst[1 .. 1 + words) must be zero on entry (the caller clears it in stream order). Same information as
// the index list; 1/32 of a word per slot whatever the view, where the list costs a word per VISIBLE slot — the smaller
// encoding above ~3 % visibility, and a fixed size, which is what an all-gather wants.
__global__ __launch_bounds__(256) void mask_shard_kernel(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ count,
                                                         uint32_t* __restrict__ dst, uint32_t words)
{
    const uint32_t n = *count;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        dst[0] = n;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const uint32_t slot = idx[j], word = slot >> 5;
        if (word < words)
            atomicOr(&dst[1 + word], 1u << (slot & 31u));
    }
}

hipError_t launch_mask_shard(const uint32_t* idx, const uint32_t* count, uint32_t* dst, uint32_t words, uint32_t capacity, hipStream_t stream)
{
    hipError_t rc = hipMemsetAsync(dst + 1, 0, (size_t)words * sizeof(uint32_t), stream);
    if (rc != hipSuccess)
        return rc;
    const uint32_t blocks = std::max(1u, std::min(2048u, (capacity + 255u) / 256u));
    hipLaunchKernelGGL(mask_shard_kernel, dim3(blocks), dim3(256), 0, stream, idx, count, dst, words);
    return hipGetLastError();
}

}  // namespace gv
