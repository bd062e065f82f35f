// tools/event_probe.hip — what does a hipEvent bracket around ONE kernel measure, against the kernel's own duration (rocprofv3)?
//   hipcc --offload-arch=gfx950 -O3 tools/event_probe.hip -o /tmp/event_probe
//   rocprofv3 --kernel-trace --stats -d gpurun_out/event_probe -- /tmp/event_probe
// Three ways, each over 200 launches of a ~100 us streaming kernel that follows a short kernel on the same stream (as the cull
// follows the pyramid build): A = hipEventRecord before and after the launch (what KernelTimer does); B = start / stop events
// attached to the dispatch (hipExtLaunchKernelGGL); C = untimed (the rocprofv3 trace is the reference for all three: the three
// phases launch three instantiations of the same kernel, so --stats lists them apart).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int PHASE>
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ in, float* __restrict__ out, size_t n4)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f)
        out[0] = acc;
}
__global__ void short_kernel(float* out) { if (threadIdx.x == 9999) out[1] = 1.f; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    const size_t bytes = 720ull << 20, n4 = bytes / 16;
    float4* in; float* out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(in, 0, bytes));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int N = 200;
    std::vector<hipEvent_t> a0(N), a1(N), b0(N), b1(N);
    for (int i = 0; i < N; i++) { CK(hipEventCreate(&a0[i])); CK(hipEventCreate(&a1[i])); CK(hipEventCreate(&b0[i])); CK(hipEventCreate(&b1[i])); }
    const dim3 grid(256 * 16), block(256);
    for (int i = 0; i < 3000; i++)  // clocks
        hipLaunchKernelGGL(stream_kernel<3>, grid, block, 0, s, in, out, n4 / 8);
    CK(hipStreamSynchronize(s));
    for (int i = 0; i < N; i++) {  // A: two records
        hipLaunchKernelGGL(short_kernel, dim3(64), dim3(64), 0, s, out);
        CK(hipEventRecord(a0[i], s));
        hipLaunchKernelGGL(stream_kernel<0>, grid, block, 0, s, in, out, n4);
        CK(hipEventRecord(a1[i], s));
        hipLaunchKernelGGL(short_kernel, dim3(64), dim3(64), 0, s, out);
    }
    CK(hipStreamSynchronize(s));
    for (int i = 0; i < N; i++) {  // B: events attached to the dispatch
        hipLaunchKernelGGL(short_kernel, dim3(64), dim3(64), 0, s, out);
        hipExtLaunchKernelGGL(stream_kernel<1>, grid, block, 0, s, b0[i], b1[i], 0, in, out, n4);
        hipLaunchKernelGGL(short_kernel, dim3(64), dim3(64), 0, s, out);
    }
    CK(hipStreamSynchronize(s));
    for (int i = 0; i < N; i++) {  // C: untimed
        hipLaunchKernelGGL(short_kernel, dim3(64), dim3(64), 0, s, out);
        hipLaunchKernelGGL(stream_kernel<2>, grid, block, 0, s, in, out, n4);
        hipLaunchKernelGGL(short_kernel, dim3(64), dim3(64), 0, s, out);
    }
    CK(hipStreamSynchronize(s));
    auto summary = [&](const char* what, std::vector<float>& v) {
        std::sort(v.begin(), v.end());
        double sum = 0; for (float x : v) sum += x;
        std::printf("%-58s mean %8.2f  median %8.2f  min %8.2f us\n", what, sum / v.size() * 1e3, v[v.size() / 2] * 1e3, v[0] * 1e3);
    };
    std::vector<float> A(N), B(N), Bs(N);
    for (int i = 0; i < N; i++) {
        CK(hipEventElapsedTime(&A[i], a0[i], a1[i]));
        CK(hipEventElapsedTime(&B[i], b0[i], b1[i]));
        if (hipEventElapsedTime(&Bs[i], b1[i], b1[i]) != hipSuccess) Bs[i] = -1.f;
    }
    summary("A  record / launch / record            stream_kernel<0>", A);
    summary("B  hipExtLaunchKernelGGL(start, stop)  stream_kernel<1>", B);
    summary("B' elapsed(stop, stop) of the same     stream_kernel<1>", Bs);
    std::printf("C  untimed                             stream_kernel<2>   (see the rocprofv3 stats for all three)\n");
    return 0;
}
