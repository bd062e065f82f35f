// Dev tool: moving N 48-byte records through a random permutation — gather (random reads, sequential writes) against
// scatter (sequential reads, random writes), three lanes per record (one float4 each) in both.
//   hipcc --offload-arch=gfx950 -O3 tools/permute_probe.hip -o /tmp/permute_probe && /tmp/permute_probe [records]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

template <bool SCATTER, int UNROLL>
__global__ __launch_bounds__(256) void permute(const float4* __restrict__ src, float4* __restrict__ dst, const uint32_t* __restrict__ perm, uint32_t n)
{
    const uint32_t per_block = 4096 * 3;  // a tile of 4096 records like the sort's last pass
    const size_t base = (size_t)blockIdx.x * per_block;
#pragma unroll UNROLL
    for (uint32_t q = threadIdx.x; q < per_block; q += 256) {
        const size_t g = base + q;
        const uint32_t t = (uint32_t)(g / 3), part = (uint32_t)(g - (size_t)t * 3);
        if (t < n) {
            const uint32_t p = perm[t];
            if (SCATTER)
                dst[(size_t)p * 3 + part] = src[g];
            else
                dst[g] = src[(size_t)p * 3 + part];
        }
    }
}

// gather whose SOURCE rows sit at a 64-byte stride (48 used): every record is one aligned 64-byte piece of one 128-byte line, where a
// 48-byte stride has 3 of 8 records straddle two lines. The destination stays a packed 48-byte row.
template <int UNROLL>
__global__ __launch_bounds__(256) void gather_padded(const float4* __restrict__ src, float4* __restrict__ dst, const uint32_t* __restrict__ perm, uint32_t n)
{
    const uint32_t per_block = 4096 * 3;
    const size_t base = (size_t)blockIdx.x * per_block;
#pragma unroll UNROLL
    for (uint32_t q = threadIdx.x; q < per_block; q += 256) {
        const size_t g = base + q;
        const uint32_t t = (uint32_t)(g / 3), part = (uint32_t)(g - (size_t)t * 3);
        if (t < n)
            dst[g] = src[(size_t)perm[t] * 4 + part];
    }
}

// the same with the 4-byte pool slot beside the model (the sort's last pass moves both): gather = idx_out[t] = idx_in[perm[t]]
// (a random 4-byte read: a whole sector fetched), scatter = idx_out[perm[t]] = idx_in[t] (a random 4-byte write: nothing fetched)
template <bool SCATTER, int UNROLL>
__global__ __launch_bounds__(256) void permute_idx(const float4* __restrict__ src, float4* __restrict__ dst, const uint32_t* __restrict__ perm,
                                                   const uint32_t* __restrict__ idx_in, uint32_t* __restrict__ idx_out, uint32_t n)
{
    const uint32_t per_block = 4096 * 3;
    const size_t base = (size_t)blockIdx.x * per_block;
    for (uint32_t t = blockIdx.x * 4096 + threadIdx.x; t < min(n, (blockIdx.x + 1) * 4096u); t += 256) {
        const uint32_t p = perm[t];
        if (SCATTER)
            idx_out[p] = idx_in[t];
        else
            idx_out[t] = idx_in[p];
    }
#pragma unroll UNROLL
    for (uint32_t q = threadIdx.x; q < per_block; q += 256) {
        const size_t g = base + q;
        const uint32_t t = (uint32_t)(g / 3), part = (uint32_t)(g - (size_t)t * 3);
        if (t < n) {
            const uint32_t p = perm[t];
            if (SCATTER)
                dst[(size_t)p * 3 + part] = src[g];
            else
                dst[g] = src[(size_t)p * 3 + part];
        }
    }
}

int main(int argc, char** argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 2124723u;
    std::vector<uint32_t> perm(n);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(5);
    std::shuffle(perm.begin(), perm.end(), rng);
    float4 *src, *dst; uint32_t* dperm;
    hipMalloc(&src, (size_t)n * 48); hipMalloc(&dst, (size_t)n * 48); hipMalloc(&dperm, (size_t)n * 4);
    hipMemset(src, 1, (size_t)n * 48);
    hipMemcpy(dperm, perm.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const uint32_t blocks = (n + 4095) / 4096;
    auto run = [&](const char* name, auto kernel) {
        std::vector<float> ms;
        for (int rep = 0; rep < 9; rep++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, src, dst, dperm, n);
            hipEventRecord(b); hipEventSynchronize(b);
            float t; hipEventElapsedTime(&t, a, b); ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-34s %7.1f us  (%.2f TB/s of the 96 useful bytes per record)\n", name, ms[4] * 1000, (double)n * 96 / (ms[4] * 1e-3) / 1e12);
    };
    printf("%u records of 48 bytes, random permutation\n", n);
    run("gather, 4 loads in flight", permute<false, 4>);
    run("gather, 16 loads in flight", permute<false, 16>);
    run("gather, 48 loads in flight", permute<false, 48>);
    run("scatter, unroll 4", permute<true, 4>);
    run("scatter, unroll 16", permute<true, 16>);
    {   // padded source rows (64-byte stride)
        float4* wide;
        hipMalloc(&wide, (size_t)n * 64); hipMemset(wide, 1, (size_t)n * 64);
        float4* keep = src; src = wide;
        run("gather, source rows 64-B stride", gather_padded<4>);
        src = keep; hipFree(wide);
    }
    for (uint32_t window : {4096u, 16384u, 65536u, 262144u}) {  // a permutation that stays inside windows of that many records
        std::vector<uint32_t> local(n);
        std::iota(local.begin(), local.end(), 0u);
        for (uint32_t lo = 0; lo < n; lo += window)
            std::shuffle(local.begin() + lo, local.begin() + std::min(n, lo + window), rng);
        hipMemcpy(dperm, local.data(), (size_t)n * 4, hipMemcpyHostToDevice);
        char name[64]; snprintf(name, sizeof name, "gather inside windows of %u", window);
        run(name, permute<false, 4>);
    }
    hipMemcpy(dperm, perm.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    uint32_t *idx_in, *idx_out;
    hipMalloc(&idx_in, (size_t)n * 4); hipMalloc(&idx_out, (size_t)n * 4);
    hipMemset(idx_in, 1, (size_t)n * 4);
    auto run_idx = [&](const char* name, auto kernel) {
        std::vector<float> ms;
        for (int rep = 0; rep < 9; rep++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, src, dst, dperm, idx_in, idx_out, n);
            hipEventRecord(b); hipEventSynchronize(b);
            float t; hipEventElapsedTime(&t, a, b); ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-34s %7.1f us\n", name, ms[4] * 1000);
    };
    run_idx("gather + 4-byte slot gather", permute_idx<false, 4>);
    run_idx("scatter + 4-byte slot scatter", permute_idx<true, 4>);
    return 0;
}
