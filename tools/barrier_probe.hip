// Dev tool: cost of a grid-wide barrier between persistent workgroups on gfx950, per variant.
//   hipcc --offload-arch=gfx950 -O3 tools/barrier_probe.hip -o /tmp/barrier_probe && /tmp/barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

template <int MODE>
__device__ __forceinline__ void grid_barrier(uint32_t* counter, uint32_t target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE == 0)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((int32_t)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0)
            __builtin_amdgcn_s_sleep(1);
        if (MODE == 0)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// every workgroup writes a slice, barrier, reads the neighbour's slice (so the fences have something to do)
template <int MODE>
__global__ __launch_bounds__(256) void probe(uint32_t* counter, uint32_t base, uint32_t rounds, uint32_t* data, uint32_t* out)
{
    uint32_t arrived = base, acc = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        if (MODE == 2)
            __hip_atomic_store(&data[blockIdx.x * 256 + threadIdx.x], r + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            data[blockIdx.x * 256 + threadIdx.x] = r + blockIdx.x;
        grid_barrier<MODE>(counter, arrived += gridDim.x);
        const uint32_t nb = (blockIdx.x + 1) % gridDim.x;
        if (MODE == 2)
            acc += __hip_atomic_load(&data[nb * 256 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (r + nb);
        else
            acc += data[nb * 256 + threadIdx.x] - (r + nb);
        grid_barrier<MODE>(counter, arrived += gridDim.x);
    }
    if (acc)
        atomicAdd(out, 1u);  // stale reads
}

__global__ void empty_kernel() {}

int main()
{
    uint32_t *counter, *data, *out;
    hipMalloc(&counter, 4); hipMalloc(&data, 256 * 256 * 4); hipMalloc(&out, 4);
    hipMemset(counter, 0, 4); hipMemset(out, 0, 4);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    uint32_t base = 0;
    const uint32_t rounds = 50;
    for (uint32_t grid : {16u, 40u, 64u, 157u, 256u}) {
        for (int mode = 0; mode < 3; mode++) {
            std::vector<float> ms;
            for (int rep = 0; rep < 7; rep++) {
                hipEventRecord(a, s);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, s, counter, base, rounds, data, out);
                if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, s, counter, base, rounds, data, out);
                if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, s, counter, base, rounds, data, out);
                hipEventRecord(b, s);
                hipEventSynchronize(b);
                base += grid * rounds * 2;
                float t; hipEventElapsedTime(&t, a, b); ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            uint32_t stale; hipMemcpy(&stale, out, 4, hipMemcpyDeviceToHost); hipMemset(out, 0, 4);
            printf("grid %3u mode %d (%s): %.2f us per barrier (median kernel %.1f us), workgroups with stale reads %u\n", grid, mode,
                   mode == 0 ? "agent fences" : mode == 1 ? "no fences, plain data" : "no fences, agent-scope atomic data", ms[3] * 1000 / (rounds * 2),
                   ms[3] * 1000, stale);
        }
    }
    // back-to-back empty launches for scale
    std::vector<float> ms;
    for (int rep = 0; rep < 7; rep++) {
        hipEventRecord(a, s);
        for (int k = 0; k < 100; k++) hipLaunchKernelGGL(empty_kernel, dim3(40), dim3(256), 0, s);
        hipEventRecord(b, s); hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("empty kernel, 100 back to back: %.2f us each\n", ms[3] * 10);
    return 0;
}
