// Dev probe: do hipGraphs shorten a short dependent kernel chain on this stack? Three ~10 us kernels back to back,
// 2000 iterations: plain stream launches vs one captured graph launched per iteration (vs the graph with per-launch
// kernel-parameter updates, which the visibility pass would need because the view changes every frame).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void work(float* p, int n, float k)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * k + 1.0f;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    const int n = 1 << 22;  // 16 MB: ~6 us per kernel
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemset(d, 0, n * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int iters = 2000;
    auto run = [&](auto&& body, const char* name) {
        for (int i = 0; i < 50; i++) body(i);
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iters; i++) body(i);
        hipStreamSynchronize(s);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
        printf("%-34s %7.2f us / iteration\n", name, us);
    };
    run([&](int i) { for (int k = 0; k < 3; k++) hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, s, d, n, 1.0f + i * 1e-9f); }, "3 stream launches");
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int k = 0; k < 3; k++) hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, s, d, n, 1.0f);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    run([&](int) { hipGraphLaunch(ge, s); }, "1 graph launch (3 kernel nodes)");
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn); CK(hipGraphGetNodes(g, nodes.data(), &nn));
    run([&](int i) {
        float k = 1.0f + i * 1e-9f; int nv = n; float* dp = d;
        void* args[3] = {&dp, &nv, &k};
        for (size_t q = 0; q < nn; q++) {
            hipKernelNodeParams p{};
            p.func = (void*)work; p.gridDim = dim3(n / 256); p.blockDim = dim3(256); p.kernelParams = args;
            hipGraphExecKernelNodeSetParams(ge, nodes[q], &p);
        }
        hipGraphLaunch(ge, s);
    }, "graph + 3 param updates per launch");
    return 0;
}
