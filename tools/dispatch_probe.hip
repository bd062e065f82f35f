// Dev tool: what a launch of many short workgroups costs before any of them does anything — kernel time (hipEvents around
// 20 back-to-back launches, divided) for grids of 256-lane workgroups that (a) return at once, (b) load one word and
// store one word, (c) do that behind a workgroup barrier. The emit launch of a 10 M-entry pool is 9768 such workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/dispatch_probe.hip -o /tmp/dispatch_probe && /tmp/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ in, unsigned* __restrict__ out)
{
    if (MODE == 0)
        return;
    __shared__ unsigned s[4];
    const unsigned v = in[blockIdx.x * 256 + threadIdx.x];
    if (MODE == 2) {
        if ((threadIdx.x & 63) == 0)
            s[threadIdx.x >> 6] = v;
        __syncthreads();
        out[blockIdx.x * 256 + threadIdx.x] = v + s[0] + s[1] + s[2] + s[3];
    } else {
        out[blockIdx.x * 256 + threadIdx.x] = v;
    }
}

int main()
{
    const unsigned max_blocks = 1u << 16;
    unsigned *in, *out;
    hipMalloc(&in, (size_t)max_blocks * 256 * 4); hipMalloc(&out, (size_t)max_blocks * 256 * 4);
    hipMemset(in, 0, (size_t)max_blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto kernel, unsigned blocks) {
        std::vector<float> us;
        for (int rep = 0; rep < 7; rep++) {
            hipEventRecord(a);
            for (int i = 0; i < 20; i++)
                hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, in, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float t; hipEventElapsedTime(&t, a, b); us.push_back(t * 1000 / 20);
        }
        std::sort(us.begin(), us.end());
        printf("%-28s %6u workgroups: %6.2f us per launch\n", name, blocks, us[3]);
    };
    for (unsigned blocks : {256u, 2442u, 9768u, 39072u}) {
        run("return at once", k<0>, blocks);
        run("load + store", k<1>, blocks);
        run("load, barrier, store", k<2>, blocks);
    }
    return 0;
}
