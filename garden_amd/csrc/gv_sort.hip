// gv_sort.hip — gv_sort: sortMeshes (source/system/render/mesh.cpp:265-328) on the device.
#include "gv_device.hpp"

namespace gv {

// ------------------------------------------------------------------------------------------------
// sortMeshes (mesh.cpp:265-328): order the compact records by distanceSq — ascending for unsorted buffers
// (front to back, operator< at render/mesh.hpp:196), descending for the sorted / translucent ones (:204).
// Stable LSD radix sort on a 32-bit order-preserving key, 8 bits per pass; ties keep ascending slot order
// (std::sort in the reference is unstable, so any tie order is within its contract).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kSortTile = 4096;  // keys per workgroup per pass (16 sub-tiles of 256)

__global__ __launch_bounds__(256) void sort_keys_kernel(const float* __restrict__ dist, const uint32_t* __restrict__ count,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                        uint32_t descending)
{
    const uint32_t n = *count;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const uint32_t u = __float_as_uint(dist[j]);
        uint32_t k = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);  // ascending float order == ascending key order
        if (descending)
            k = ~k;
        keys[j] = k;
        vals[j] = j;
    }
}

// Launch geometry is sized for the pool capacity (known on the host); the record count lives on the device, so
// every kernel derives the live tile count from it and surplus workgroups exit at once.
// per-workgroup digit histogram, bin-major with a fixed stride: hist[bin * stride + tile]
__global__ __launch_bounds__(256) void radix_hist_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ count,
                                                         uint32_t* __restrict__ hist, uint32_t shift, uint32_t stride)
{
    const uint32_t n = *count;
    const uint32_t lo = blockIdx.x * kSortTile;
    if (lo >= n)
        return;
    __shared__ uint32_t bins[256];
    bins[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t hi = min(lo + kSortTile, n);
    for (uint32_t j = lo + threadIdx.x; j < hi; j += 256)
        atomicAdd(&bins[(keys[j] >> shift) & 255u], 1u);
    __syncthreads();
    hist[threadIdx.x * stride + blockIdx.x] = bins[threadIdx.x];
}

// one workgroup per digit: exclusive scan of that digit's per-tile counts in place + the digit's total
__global__ __launch_bounds__(256) void radix_bin_scan_kernel(uint32_t* __restrict__ hist, uint32_t* __restrict__ bin_total,
                                                             const uint32_t* __restrict__ count, uint32_t stride)
{
    __shared__ uint32_t wave_sum[4];
    const uint32_t tiles = (*count + kSortTile - 1) / kSortTile;
    uint32_t* row = hist + (size_t)blockIdx.x * stride;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < tiles; base += 256) {
        const uint32_t idx = base + threadIdx.x;
        const uint32_t v = idx < tiles ? row[idx] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d)
                incl += up;
        }
        if (lane == 63)
            wave_sum[wave] = incl;
        __syncthreads();
        uint32_t wave_prefix = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; w++) {
            wave_prefix += w < wave ? wave_sum[w] : 0u;
            all += wave_sum[w];
        }
        if (idx < tiles)
            row[idx] = carry + wave_prefix + incl - v;
        carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0)
        bin_total[blockIdx.x] = carry;
}

// stable scatter: sub-tiles of 256 keys in order; rank inside a wave by digit matching (8 ballots), across waves
// and sub-tiles through LDS counters. Digit d of this tile starts at (sum of lower digits' totals) + hist[d][tile].
__global__ __launch_bounds__(256) void radix_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                            const uint32_t* __restrict__ vals_in,
                                                            uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                            const uint32_t* __restrict__ count,
                                                            const uint32_t* __restrict__ hist,
                                                            const uint32_t* __restrict__ bin_total, uint32_t shift,
                                                            uint32_t stride)
{
    const uint32_t n = *count;
    const uint32_t lo = blockIdx.x * kSortTile;
    if (lo >= n)
        return;
    __shared__ uint32_t base[256];        // next output position of each digit for this workgroup
    __shared__ uint32_t wcount[4][256];   // per-wave digit counts of the current sub-tile
    __shared__ uint32_t wave_sum[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    {  // exclusive scan of the 256 digit totals, one digit per lane
        const uint32_t v = bin_total[threadIdx.x];
        uint32_t incl = v;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d)
                incl += up;
        }
        if (lane == 63)
            wave_sum[wave] = incl;
        __syncthreads();
        uint32_t wave_prefix = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; w++)
            wave_prefix += w < wave ? wave_sum[w] : 0u;
        base[threadIdx.x] = wave_prefix + incl - v + hist[threadIdx.x * stride + blockIdx.x];
    }
#pragma unroll
    for (uint32_t w = 0; w < 4; w++)
        wcount[w][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t hi = min(lo + kSortTile, n);
    for (uint32_t t = lo; t < hi; t += 256) {  // uniform trip count
        const uint32_t j = t + threadIdx.x;
        const bool valid = j < hi;
        const uint32_t key = valid ? keys_in[j] : 0u, val = valid ? vals_in[j] : 0u;
        const uint32_t d = (key >> shift) & 255u;
        unsigned long long peer = __ballot(valid);
#pragma unroll
        for (uint32_t b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peer &= bit ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peer & ((1ull << lane) - 1ull));
        const bool leader = valid && rank == 0;
        if (leader)
            wcount[wave][d] = (uint32_t)__popcll(peer);
        __syncthreads();
        if (valid) {
            uint32_t pos = base[d] + rank;
#pragma unroll
            for (uint32_t w = 0; w < 4; w++)
                pos += w < wave ? wcount[w][d] : 0u;
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __syncthreads();
        if (leader) {
            atomicAdd(&base[d], (uint32_t)__popcll(peer));
            wcount[wave][d] = 0;
        }
        __syncthreads();
    }
}

// permute the 56-byte records by the sorted positions
__global__ __launch_bounds__(256) void sort_gather_kernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ count,
                                                          const uint32_t* __restrict__ idx_in, const float* __restrict__ model_in,
                                                          const float* __restrict__ dist_in, uint32_t* __restrict__ idx_out,
                                                          float* __restrict__ model_out, float* __restrict__ dist_out)
{
    const uint32_t n = *count;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const uint32_t src = order[j];
        idx_out[j] = idx_in[src];
        dist_out[j] = dist_in[src];
        const float4* sm = reinterpret_cast<const float4*>(model_in + (size_t)src * 12);
        float4* dm = reinterpret_cast<float4*>(model_out + (size_t)j * 12);
        const float4 m0 = sm[0], m1 = sm[1], m2 = sm[2];
        dm[0] = m0;
        dm[1] = m1;
        dm[2] = m2;
    }
}

// Small pools (up to kSmallSort records possible): ONE launch instead of fourteen — a tick of an engine-sized scene
// (10^4 entities) is launch-bound. Rank sort: the position of record i is the number of records that order before
// it, (key, emission index) compared as a pair, so the result is the stable order the radix passes produce. A
// workgroup owns 64 records; its four waves each count over a quarter of the keys (staged in LDS, read as uniform
// 16-byte broadcasts) and the partial counts meet in LDS. n^2 / 4 comparisons per wave, all CUs busy, no
// inter-workgroup step: ~6 us at 2 k records where a one-workgroup bitonic network took 42 us.
constexpr uint32_t kSmallSort = kSmallSortMaxSlots;

// records of [jlo, jhi) (multiples of 4) that order before record i with key ki. WHERE: 0 = every j is below the
// workgroup's records (ties count), 2 = every j is above them (ties do not), 1 = overlapping (compare the pair)
template <int WHERE>
__device__ __forceinline__ uint32_t count_before(const uint32_t* key, uint32_t jlo, uint32_t jhi, uint32_t ki, uint32_t i)
{
    uint32_t before = 0;
#pragma unroll 4
    for (uint32_t j = jlo; j < jhi; j += 4) {
        const uint4 k = *reinterpret_cast<const uint4*>(key + j);  // uniform address: one LDS broadcast per 4 keys
        if (WHERE == 0)
            before += (k.x <= ki) + (k.y <= ki) + (k.z <= ki) + (k.w <= ki);
        else if (WHERE == 2)
            before += (k.x < ki) + (k.y < ki) + (k.z < ki) + (k.w < ki);
        else
            before += ((k.x < ki) | ((k.x == ki) & (j < i))) + ((k.y < ki) | ((k.y == ki) & (j + 1 < i))) +
                      ((k.z < ki) | ((k.z == ki) & (j + 2 < i))) + ((k.w < ki) | ((k.w == ki) & (j + 3 < i)));
    }
    return before;
}

__device__ __forceinline__ void sort_small_block(const SortBuffers& b, uint32_t capacity, uint32_t descending, uint32_t block)
{
    extern __shared__ uint32_t key[];  // order-preserving keys of all n records, padded to a multiple of 4
    __shared__ uint32_t partial[4][64];
    const uint32_t n = min(*b.count, capacity);
    const uint32_t i0 = block * 64;
    if (i0 >= n)
        return;
    const uint32_t n4 = (n + 3u) & ~3u;
    for (uint32_t j = threadIdx.x; j < n4; j += 256) {
        uint32_t k = 0xFFFFFFFFu;  // padding: never counted (its index is beyond every record's)
        if (j < n) {
            const uint32_t u = __float_as_uint(b.dist_in[j]);
            k = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
            if (descending)
                k = ~k;
        }
        key[j] = k;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, part = threadIdx.x >> 6;
    const uint32_t i = i0 + lane;
    const uint32_t ki = key[min(i, n4 - 1)];
    const uint32_t per = ((n4 >> 2) + 3u) & ~3u;  // keys per wave, a multiple of 4
    const uint32_t jlo = min(part * per, n4), jhi = min(jlo + per, n4);
    const uint32_t own_lo = min(max(i0 & ~3u, jlo), jhi), own_hi = min(max((i0 + 64u + 3u) & ~3u, jlo), jhi);
    const uint32_t before = count_before<0>(key, jlo, own_lo, ki, i) + count_before<1>(key, own_lo, own_hi, ki, i) +
                            count_before<2>(key, own_hi, jhi, ki, i);
    partial[part][lane] = before;
    __syncthreads();
    if (part != 0 || i >= n)
        return;
    const uint32_t rank = partial[0][lane] + partial[1][lane] + partial[2][lane] + partial[3][lane];
    b.idx_out[rank] = b.idx_in[i];
    b.dist_out[rank] = b.dist_in[i];
    const float4* sm = reinterpret_cast<const float4*>(b.model_in + (size_t)i * 12);
    float4* dm = reinterpret_cast<float4*>(b.model_out + (size_t)rank * 12);
    const float4 m0 = sm[0], m1 = sm[1], m2 = sm[2];
    dm[0] = m0;
    dm[1] = m1;
    dm[2] = m2;
}

__global__ __launch_bounds__(256) void sort_small_kernel(const SortBuffers b, uint32_t capacity, uint32_t descending)
{
    sort_small_block(b, capacity, descending, blockIdx.x);
}

// several views of one small pool (main camera + shadow passes) in one launch: blockIdx.y picks the view
__global__ __launch_bounds__(256) void sort_small_batch_kernel(const SortBatch batch, uint32_t capacity)
{
    sort_small_block(batch.view[blockIdx.y], capacity, batch.descending[blockIdx.y], blockIdx.x);
}

hipError_t launch_sort(const SortBuffers& b, uint32_t capacity, bool descending, hipStream_t stream)
{
    if (capacity == 0)
        return hipSuccess;
    if (capacity <= kSmallSort) {
        const uint32_t lds = ((capacity + 3u) & ~3u) * 4;  // 64 KB of keys at the limit, beside the 1 KB of partial counts
        static const hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void*>(sort_small_kernel),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, kSmallSort * 4);
        if (raised != hipSuccess)
            return raised;
        hipLaunchKernelGGL(sort_small_kernel, dim3((capacity + 63) / 64), dim3(256), lds, stream, b, capacity, descending ? 1u : 0u);
        return hipGetLastError();
    }
    const uint32_t stride = (capacity + kSortTile - 1) / kSortTile;  // tiles at full capacity = hist row stride
    const uint32_t wide = min((capacity + 255u) / 256u, 4096u);
    hipLaunchKernelGGL(sort_keys_kernel, dim3(wide), dim3(256), 0, stream, b.dist_in, b.count, b.keys[0], b.vals[0],
                       descending ? 1u : 0u);
    for (uint32_t pass = 0; pass < 4; pass++) {
        const uint32_t src = pass & 1u, dst = src ^ 1u;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(stride), dim3(256), 0, stream, b.keys[src], b.count, b.hist, pass * 8, stride);
        hipLaunchKernelGGL(radix_bin_scan_kernel, dim3(256), dim3(256), 0, stream, b.hist, b.bin_total, b.count, stride);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(stride), dim3(256), 0, stream, b.keys[src], b.vals[src], b.keys[dst],
                           b.vals[dst], b.count, b.hist, b.bin_total, pass * 8, stride);
    }
    // 4 passes: the sorted order ends in vals[0]
    hipLaunchKernelGGL(sort_gather_kernel, dim3(wide), dim3(256), 0, stream, b.vals[0], b.count, b.idx_in, b.model_in, b.dist_in,
                       b.idx_out, b.model_out, b.dist_out);
    return hipGetLastError();
}

hipError_t launch_sort_small_batch(const SortBatch& batch, uint32_t views, uint32_t capacity, hipStream_t stream)
{
    if (capacity == 0 || views == 0)
        return hipSuccess;
    const uint32_t lds = ((capacity + 3u) & ~3u) * 4;
    static const hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void*>(sort_small_batch_kernel),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, kSmallSort * 4);
    if (raised != hipSuccess)
        return raised;
    hipLaunchKernelGGL(sort_small_batch_kernel, dim3((capacity + 63) / 64, views), dim3(256), lds, stream, batch, capacity);
    return hipGetLastError();
}

}  // namespace gv
