// Dev tool: what read bandwidth does this box deliver to (a) ONE contiguous stream, (b) five separate streams with the cull
// kernel's element sizes (16 + 8 + 32 + 8 + 1 B per entry), (c) the same bytes laid out as tiles of 256 entries
// ([16 B x 256][8 B x 256][32 B x 256][8 B x 256][1 B x 256] = 16 640 B per tile, one workgroup per tile)?
//   hipcc --offload-arch=gfx950 -O3 tools/read_probe.hip -o /tmp/read_probe && /tmp/read_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef float f32x2n __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 nt16(const float4* p)
{
    const f32x4n v = __builtin_nontemporal_load(reinterpret_cast<const f32x4n*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float2 nt8(const float2* p)
{
    const f32x2n v = __builtin_nontemporal_load(reinterpret_cast<const f32x2n*>(p));
    return make_float2(v.x, v.y);
}

__global__ __launch_bounds__(256) void one_stream(const float4* __restrict__ a, size_t quads, float* sink)
{
    float acc = 0;
    const size_t base = (size_t)blockIdx.x * 1040;  // 16 640 B per workgroup, as a tile
    for (uint32_t q = threadIdx.x; q < 1040; q += 256)
        if (base + q < quads) { const float4 v = nt16(a + base + q); acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) *sink = acc;
}

__global__ __launch_bounds__(256) void five_streams(const float4* __restrict__ a, const float2* __restrict__ b, const float4* __restrict__ ab,
                                                    const float2* __restrict__ c, const unsigned char* __restrict__ f, uint32_t n, float* sink)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float acc = 0;
    if (i < n) {
        const float4 va = nt16(a + i); const float2 vb = nt8(b + i);
        const float4 x0 = nt16(ab + 2 * (size_t)i), x1 = nt16(ab + 2 * (size_t)i + 1);
        const float2 vc = nt8(c + i);
        acc = va.x + va.w + vb.x + vb.y + x0.x + x0.w + x1.x + x1.w + vc.x + vc.y + (float)f[i];
    }
    if (acc == 12345.678f) *sink = acc;
}

// MASK: which of the five streams are read (bit 0 a, 1 b, 2 ab, 3 c, 4 flags)
template <int MASK>
__global__ __launch_bounds__(256) void some_streams(const float4* __restrict__ a, const float2* __restrict__ b, const float4* __restrict__ ab,
                                                    const float2* __restrict__ c, const unsigned char* __restrict__ f, uint32_t n, float* sink)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float acc = 0;
    if (i < n) {
        if (MASK & 1) { const float4 v = nt16(a + i); acc += v.x + v.w; }
        if (MASK & 2) { const float2 v = nt8(b + i); acc += v.x + v.y; }
        if (MASK & 4) { const float4 x0 = nt16(ab + 2 * (size_t)i), x1 = nt16(ab + 2 * (size_t)i + 1); acc += x0.x + x0.w + x1.x + x1.w; }
        if (MASK & 8) { const float2 v = nt8(c + i); acc += v.x + v.y; }
        if (MASK & 16) acc += (float)f[i];
    }
    if (acc == 12345.678f) *sink = acc;
}

// the five streams + the cull kernel's outputs: one isVisible byte per entry, one ballot word per wave
template <bool BYTES, bool WORDS>
__global__ __launch_bounds__(256) void five_streams_out(const float4* __restrict__ a, const float2* __restrict__ b, const float4* __restrict__ ab,
                                                        const float2* __restrict__ c, const unsigned long long* __restrict__ bits, uint32_t n,
                                                        unsigned char* __restrict__ vis, unsigned long long* __restrict__ mask)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float acc = 0;
    if (i < n) {
        const float4 va = nt16(a + i); const float2 vb = nt8(b + i);
        const float4 x0 = nt16(ab + 2 * (size_t)i), x1 = nt16(ab + 2 * (size_t)i + 1);
        const float2 vc = nt8(c + i);
        const unsigned long long w = bits[i >> 6];
        acc = va.x + va.w + vb.x + vb.y + x0.x + x0.w + x1.x + x1.w + vc.x + vc.y + (float)((w >> (i & 63)) & 1);
    }
    const bool visible = acc > 1.0f;
    if (BYTES && i < n)
        vis[i] = visible ? 1 : 0;
    const unsigned long long word = __ballot(visible);
    if (WORDS && (threadIdx.x & 63) == 0)
        mask[i >> 6] = word;
}

// ballot words combined per workgroup: 4 lanes store the 4 words of the tile as one 32-byte piece
template <bool ATOMIC, bool NT = false>
__global__ __launch_bounds__(256) void five_streams_words4(const float4* __restrict__ a, const float2* __restrict__ b, const float4* __restrict__ ab,
                                                           const float2* __restrict__ c, const unsigned long long* __restrict__ bits, uint32_t n,
                                                           unsigned long long* __restrict__ mask, uint32_t* __restrict__ counts)
{
    __shared__ unsigned long long w4[4];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float acc = 0;
    if (i < n) {
        const float4 va = nt16(a + i); const float2 vb = nt8(b + i);
        const float4 x0 = nt16(ab + 2 * (size_t)i), x1 = nt16(ab + 2 * (size_t)i + 1);
        const float2 vc = nt8(c + i);
        const unsigned long long w = bits[i >> 6];
        acc = va.x + va.w + vb.x + vb.y + x0.x + x0.w + x1.x + x1.w + vc.x + vc.y + (float)((w >> (i & 63)) & 1);
    }
    const unsigned long long word = __ballot(acc > 1.0f);
    if ((threadIdx.x & 63) == 0)
        w4[threadIdx.x >> 6] = word;
    __syncthreads();
    if (threadIdx.x < 4) {
        if (NT)
            __builtin_nontemporal_store(w4[threadIdx.x], &mask[(size_t)blockIdx.x * 4 + threadIdx.x]);
        else
            mask[(size_t)blockIdx.x * 4 + threadIdx.x] = w4[threadIdx.x];
    }
    if (ATOMIC && threadIdx.x == 0) {
        const uint32_t total = (uint32_t)(__popcll(w4[0]) + __popcll(w4[1]) + __popcll(w4[2]) + __popcll(w4[3]));
        if (total)
            atomicAdd(&counts[blockIdx.x / 16], total);
    }
}

// the five streams with PER entries per lane (tiles of 256 * PER entries): more loads in flight per lane, fewer workgroups
template <int PER>
__global__ __launch_bounds__(256) void five_streams_ilp(const float4* __restrict__ a, const float2* __restrict__ b, const float4* __restrict__ ab,
                                                        const float2* __restrict__ c, const unsigned long long* __restrict__ bits, uint32_t n,
                                                        unsigned long long* __restrict__ mask)
{
    float acc[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t i = (blockIdx.x * PER + k) * 256 + threadIdx.x;
        acc[k] = 0;
        if (i < n) {
            const float4 va = nt16(a + i); const float2 vb = nt8(b + i);
            const float4 x0 = nt16(ab + 2 * (size_t)i), x1 = nt16(ab + 2 * (size_t)i + 1);
            const float2 vc = nt8(c + i);
            const unsigned long long w = bits[i >> 6];
            acc[k] = va.x + va.w + vb.x + vb.y + x0.x + x0.w + x1.x + x1.w + vc.x + vc.y + (float)((w >> (i & 63)) & 1);
        }
    }
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t i = (blockIdx.x * PER + k) * 256 + threadIdx.x;
        const unsigned long long word = __ballot(acc[k] > 1.0f);
        if ((threadIdx.x & 63) == 0 && i < n)
            mask[i >> 6] = word;
    }
}

__global__ __launch_bounds__(256) void tiled(const unsigned char* __restrict__ tiles, uint32_t ntiles, float* sink)
{
    const unsigned char* t = tiles + (size_t)blockIdx.x * 16640;
    const uint32_t k = threadIdx.x;
    float acc = 0;
    if (blockIdx.x < ntiles) {
        const float4 va = nt16(reinterpret_cast<const float4*>(t) + k);
        const float2 vb = nt8(reinterpret_cast<const float2*>(t + 4096) + k);
        const float4 x0 = nt16(reinterpret_cast<const float4*>(t + 6144) + 2 * k), x1 = nt16(reinterpret_cast<const float4*>(t + 6144) + 2 * k + 1);
        const float2 vc = nt8(reinterpret_cast<const float2*>(t + 14336) + k);
        acc = va.x + va.w + vb.x + vb.y + x0.x + x0.w + x1.x + x1.w + vc.x + vc.y + (float)t[16384 + k];
    }
    if (acc == 12345.678f) *sink = acc;
}

// TILES consecutive 256-entry tiles per workgroup, one after the other (same registers), their ballot words stored together
template <int TILES>
__global__ __launch_bounds__(256) void five_streams_tiles(const float4* __restrict__ a, const float2* __restrict__ b, const float4* __restrict__ ab,
                                                          const float2* __restrict__ c, const unsigned long long* __restrict__ bits, uint32_t n,
                                                          unsigned long long* __restrict__ mask)
{
    __shared__ unsigned long long words[TILES * 4];
    for (int k = 0; k < TILES; k++) {
        const uint32_t i = (blockIdx.x * TILES + k) * 256 + threadIdx.x;
        float acc = 0;
        if (i < n) {
            const float4 va = nt16(a + i); const float2 vb = nt8(b + i);
            const float4 x0 = nt16(ab + 2 * (size_t)i), x1 = nt16(ab + 2 * (size_t)i + 1);
            const float2 vc = nt8(c + i);
            const unsigned long long w = bits[i >> 6];
            acc = va.x + va.w + vb.x + vb.y + x0.x + x0.w + x1.x + x1.w + vc.x + vc.y + (float)((w >> (i & 63)) & 1);
        }
        const unsigned long long word = __ballot(acc > 1.0f);
        if ((threadIdx.x & 63) == 0)
            words[k * 4 + (threadIdx.x >> 6)] = word;
    }
    __syncthreads();
    if (threadIdx.x < TILES * 4)
        mask[(size_t)blockIdx.x * TILES * 4 + threadIdx.x] = words[threadIdx.x];
}

// the tiled layout with the cull kernel's remaining output (ballot words, one 32-byte store per workgroup)
__global__ __launch_bounds__(256) void tiled_words(const unsigned char* __restrict__ tiles, uint32_t ntiles, unsigned long long* __restrict__ mask)
{
    __shared__ unsigned long long w4[4];
    const unsigned char* t = tiles + (size_t)blockIdx.x * 16640;
    const uint32_t k = threadIdx.x;
    float acc = 0;
    if (blockIdx.x < ntiles) {
        const float4 va = nt16(reinterpret_cast<const float4*>(t) + k);
        const float2 vb = nt8(reinterpret_cast<const float2*>(t + 4096) + k);
        const float4 x0 = nt16(reinterpret_cast<const float4*>(t + 6144) + 2 * k), x1 = nt16(reinterpret_cast<const float4*>(t + 6144) + 2 * k + 1);
        const float2 vc = nt8(reinterpret_cast<const float2*>(t + 14336) + k);
        const unsigned long long w = reinterpret_cast<const unsigned long long*>(t + 16384)[k >> 6];
        acc = va.x + va.w + vb.x + vb.y + x0.x + x0.w + x1.x + x1.w + vc.x + vc.y + (float)((w >> (k & 63)) & 1);
    }
    const unsigned long long word = __ballot(acc > 1.0f);
    if ((k & 63) == 0)
        w4[k >> 6] = word;
    __syncthreads();
    if (k < 4)
        mask[(size_t)blockIdx.x * 4 + k] = w4[k];
}

int main()
{
    const uint32_t n = 10'000'000, ntiles = (n + 255) / 256;
    const size_t bytes = (size_t)ntiles * 16640;
    unsigned char* buf; float* sink;
    hipMalloc(&buf, bytes + 4096); hipMalloc(&sink, 4); hipMemset(buf, 0, bytes);
    // the five streams, each its own allocation like the mirror
    float4 *a, *ab; float2 *b, *c; unsigned char* f;
    hipMalloc(&a, (size_t)n * 16); hipMalloc(&b, (size_t)n * 8); hipMalloc(&ab, (size_t)n * 32); hipMalloc(&c, (size_t)n * 8); hipMalloc(&f, n);
    hipMemset(a, 0, (size_t)n * 16); hipMemset(b, 0, (size_t)n * 8); hipMemset(ab, 0, (size_t)n * 32); hipMemset(c, 0, (size_t)n * 8); hipMemset(f, 0, n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch, double gb) {
        std::vector<float> ms;
        for (int r = 0; r < 15; r++) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float t; hipEventElapsedTime(&t, e0, e1); ms.push_back(t); }
        std::sort(ms.begin(), ms.end());
        printf("%-44s %7.1f us  %.2f TB/s\n", name, ms[7] * 1000, gb / (ms[7] * 1e-3) / 1e3);
    };
    const double gb = 65.0 * n / 1e9;
    time("one contiguous stream (650 MB)", [&] { hipLaunchKernelGGL(one_stream, dim3(ntiles), dim3(256), 0, 0, (const float4*)buf, bytes / 16, sink); }, bytes / 1e9);
    time("five streams, one allocation each", [&] { hipLaunchKernelGGL(five_streams, dim3(ntiles), dim3(256), 0, 0, a, b, ab, c, f, n, sink); }, gb);
#define SOME(MASK, BYTES, NAME) time(NAME, [&] { hipLaunchKernelGGL(some_streams<MASK>, dim3(ntiles), dim3(256), 0, 0, a, b, ab, c, f, n, sink); }, BYTES * n / 1e9)
    SOME(15, 64.0, "  without the flag bytes (64 B)");
    SOME(29, 57.0, "  without b (57 B)");
    SOME(23, 57.0, "  without c (57 B)");
    SOME(27, 33.0, "  without ab (33 B)");
    SOME(30, 49.0, "  without a (49 B)");
    SOME(5, 48.0, "  a + ab only (48 B)");
    SOME(4, 32.0, "  ab only (32 B)");
    {  // the five streams back to back in ONE allocation, and the same with odd page offsets between them
        unsigned char* one; hipMalloc(&one, (size_t)n * 65 + (1u << 24)); hipMemset(one, 0, (size_t)n * 65 + (1u << 24));
        for (size_t pad : {(size_t)0, (size_t)4096 * 7, (size_t)4096 * 131 + 256}) {
            unsigned char* p = one;
            const float4* a2 = (const float4*)p; p += (size_t)n * 16 + pad;
            const float2* b2 = (const float2*)p; p += (size_t)n * 8 + pad;
            const float4* ab2 = (const float4*)p; p += (size_t)n * 32 + pad;
            const float2* c2 = (const float2*)p; p += (size_t)n * 8 + pad;
            const unsigned char* f2 = p;
            char name[96]; snprintf(name, sizeof(name), "five streams in one allocation, gaps %zu B", pad);
            time(name, [&] { hipLaunchKernelGGL(five_streams, dim3(ntiles), dim3(256), 0, 0, a2, b2, ab2, c2, f2, n, sink); }, gb);
        }
    }
    {
        unsigned char* vis; unsigned long long *mask, *bits;
        hipMalloc(&vis, n); hipMalloc(&mask, (size_t)n / 8 + 64); hipMalloc(&bits, (size_t)n / 8 + 64); hipMemset(bits, 0xFF, (size_t)n / 8 + 64);
        time("five streams (bits instead of flag bytes), no outputs", [&] { hipLaunchKernelGGL((five_streams_out<false, false>), dim3(ntiles), dim3(256), 0, 0, a, b, ab, c, bits, n, vis, mask); }, 64.125 * n / 1e9);
        time("  + a ballot word per wave", [&] { hipLaunchKernelGGL((five_streams_out<false, true>), dim3(ntiles), dim3(256), 0, 0, a, b, ab, c, bits, n, vis, mask); }, 64.25 * n / 1e9);
        uint32_t* counts; hipMalloc(&counts, (ntiles / 16 + 1) * 4); hipMemset(counts, 0, (ntiles / 16 + 1) * 4);
        time("  + the 4 ballot words of a workgroup as one 32-byte store", [&] { hipLaunchKernelGGL(five_streams_words4<false>, dim3(ntiles), dim3(256), 0, 0, a, b, ab, c, bits, n, mask, counts); }, 64.25 * n / 1e9);
        time("  + the 4 ballot words, nontemporal store", [&] { hipLaunchKernelGGL((five_streams_words4<false, true>), dim3(ntiles), dim3(256), 0, 0, a, b, ab, c, bits, n, mask, counts); }, 64.25 * n / 1e9);
        time("  + those words and one atomicAdd per workgroup (chunk counts)", [&] { hipLaunchKernelGGL(five_streams_words4<true>, dim3(ntiles), dim3(256), 0, 0, a, b, ab, c, bits, n, mask, counts); }, 64.25 * n / 1e9);
        time("  + an isVisible byte per entry", [&] { hipLaunchKernelGGL((five_streams_out<true, false>), dim3(ntiles), dim3(256), 0, 0, a, b, ab, c, bits, n, vis, mask); }, 65.125 * n / 1e9);
        time("  + both (the cull kernel's outputs)", [&] { hipLaunchKernelGGL((five_streams_out<true, true>), dim3(ntiles), dim3(256), 0, 0, a, b, ab, c, bits, n, vis, mask); }, 65.25 * n / 1e9);
    }
    {
        unsigned long long *mask2, *bits2;
        hipMalloc(&mask2, (size_t)n / 8 + 64); hipMalloc(&bits2, (size_t)n / 8 + 64); hipMemset(bits2, 0xFF, (size_t)n / 8 + 64);
        for (uint32_t m : {n, 1000000u}) {
            const uint32_t t1 = (m + 255) / 256;
            char name[96];
            snprintf(name, sizeof(name), "%u entries, 1 per lane + ballot words", m);
            time(name, [&] { hipLaunchKernelGGL(five_streams_ilp<1>, dim3(t1), dim3(256), 0, 0, a, b, ab, c, bits2, m, mask2); }, 64.25 * m / 1e9);
            snprintf(name, sizeof(name), "%u entries, 2 per lane", m);
            time(name, [&] { hipLaunchKernelGGL(five_streams_ilp<2>, dim3((t1 + 1) / 2), dim3(256), 0, 0, a, b, ab, c, bits2, m, mask2); }, 64.25 * m / 1e9);
            snprintf(name, sizeof(name), "%u entries, 4 per lane", m);
            time(name, [&] { hipLaunchKernelGGL(five_streams_ilp<4>, dim3((t1 + 3) / 4), dim3(256), 0, 0, a, b, ab, c, bits2, m, mask2); }, 64.25 * m / 1e9);
        }
    }
    {
        unsigned long long *mask4, *bits4;
        hipMalloc(&mask4, (size_t)n / 8 + 4096); hipMalloc(&bits4, (size_t)n / 8 + 4096); hipMemset(bits4, 0xFF, (size_t)n / 8 + 4096);
        time("five streams, 1 tile per workgroup, 32-byte word store", [&] { hipLaunchKernelGGL(five_streams_tiles<1>, dim3(ntiles), dim3(256), 0, 0, a, b, ab, c, bits4, n, mask4); }, 64.25 * n / 1e9);
        time("five streams, 2 tiles per workgroup, 64-byte word store", [&] { hipLaunchKernelGGL(five_streams_tiles<2>, dim3((ntiles + 1) / 2), dim3(256), 0, 0, a, b, ab, c, bits4, n, mask4); }, 64.25 * n / 1e9);
        time("five streams, 4 tiles per workgroup, 128-byte word store", [&] { hipLaunchKernelGGL(five_streams_tiles<4>, dim3((ntiles + 3) / 4), dim3(256), 0, 0, a, b, ab, c, bits4, n, mask4); }, 64.25 * n / 1e9);
        time("five streams, 16 tiles per workgroup, 512-byte word store", [&] { hipLaunchKernelGGL(five_streams_tiles<16>, dim3((ntiles + 15) / 16), dim3(256), 0, 0, a, b, ab, c, bits4, n, mask4); }, 64.25 * n / 1e9);
    }
    {
        unsigned long long* mask3; hipMalloc(&mask3, (size_t)n / 8 + 64);
        hipMemset(buf, 0xFF, 4096);
        time("tiles of 256 entries + the ballot words", [&] { hipLaunchKernelGGL(tiled_words, dim3(ntiles), dim3(256), 0, 0, buf, ntiles, mask3); }, 64.25 * n / 1e9);
    }
    time("tiles of 256 entries (16 640 B each)", [&] { hipLaunchKernelGGL(tiled, dim3(ntiles), dim3(256), 0, 0, buf, ntiles, sink); }, gb);
    return 0;
}
