// Dev probe: would interleaving the two float4 transform streams (pos|sx and quat) into one 32-byte record stream cost
// the streaming cull kernel anything? (It would save one sector per visible record in emit.) Read-only kernels over
// 10 M entries: five SoA streams as today vs {32-byte ab records + the three others}.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f4 ld4(const f4* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ f2 ld2(const f2* p) { return __builtin_nontemporal_load(p); }
__global__ __launch_bounds__(256) void soa(const f4* a, const f4* b, const f2* c, const f4* ma, const f2* mb, float* sink, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const f4 x = ld4(a + i), y = ld4(b + i), m = ld4(ma + i);
    const f2 z = ld2(c + i), w = ld2(mb + i);
    const float s = x.x + x.y + x.z + x.w + y.x + y.y + y.z + y.w + z.x + z.y + m.x + m.y + m.z + m.w + w.x + w.y;
    if (s == 12345.678f) sink[i] = s;
}
__global__ __launch_bounds__(256) void interleaved(const f4* ab, const f2* c, const f4* ma, const f2* mb, float* sink, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const f4 x = ld4(ab + 2 * (size_t)i), y = ld4(ab + 2 * (size_t)i + 1), m = ld4(ma + i);
    const f2 z = ld2(c + i), w = ld2(mb + i);
    const float s = x.x + x.y + x.z + x.w + y.x + y.y + y.z + y.w + z.x + z.y + m.x + m.y + m.z + m.w + w.x + w.y;
    if (s == 12345.678f) sink[i] = s;
}
// 40-byte transform records {pos|sx, quat, sy|sz} and 24-byte mesh records {min|max.x, max.yz}: 8-byte aligned only
__global__ __launch_bounds__(256) void packed(const char* xf40, const char* mesh24, float* sink, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const char* t = xf40 + (size_t)i * 40;
    const char* q = mesh24 + (size_t)i * 24;
    f4 x, y, m; f2 z, w;
    __builtin_memcpy(&x, t, 16); __builtin_memcpy(&y, t + 16, 16); __builtin_memcpy(&z, t + 32, 8);
    __builtin_memcpy(&m, q, 16); __builtin_memcpy(&w, q + 16, 8);
    const float s = x.x + x.y + x.z + x.w + y.x + y.y + y.z + y.w + z.x + z.y + m.x + m.y + m.z + m.w + w.x + w.y;
    if (s == 12345.678f) sink[i] = s;
}
__global__ __launch_bounds__(256) void packed_xf_only(const char* xf40, const f4* ma, const f2* mb, float* sink, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const char* t = xf40 + (size_t)i * 40;
    f4 x, y; f2 z;
    __builtin_memcpy(&x, t, 16); __builtin_memcpy(&y, t + 16, 16); __builtin_memcpy(&z, t + 32, 8);
    const f4 m = ld4(ma + i); const f2 w = ld2(mb + i);
    const float s = x.x + x.y + x.z + x.w + y.x + y.y + y.z + y.w + z.x + z.y + m.x + m.y + m.z + m.w + w.x + w.y;
    if (s == 12345.678f) sink[i] = s;
}
int main()
{
    const uint32_t n = 10000000;
    f4 *a, *b, *ab, *ma; f2 *c, *mb; float* sink;
    CK(hipMalloc(&a, (size_t)n * 16)); CK(hipMalloc(&b, (size_t)n * 16)); CK(hipMalloc(&ab, (size_t)n * 32));
    CK(hipMalloc(&ma, (size_t)n * 16)); CK(hipMalloc(&c, (size_t)n * 8)); CK(hipMalloc(&mb, (size_t)n * 8)); CK(hipMalloc(&sink, (size_t)n * 4));
    CK(hipMemset(a, 0, (size_t)n * 16)); CK(hipMemset(b, 0, (size_t)n * 16)); CK(hipMemset(ab, 0, (size_t)n * 32));
    CK(hipMemset(ma, 0, (size_t)n * 16)); CK(hipMemset(c, 0, (size_t)n * 8)); CK(hipMemset(mb, 0, (size_t)n * 8));
    char *xf40, *mesh24;
    CK(hipMalloc(&xf40, (size_t)n * 40 + 64)); CK(hipMalloc(&mesh24, (size_t)n * 24 + 64));
    CK(hipMemset(xf40, 0, (size_t)n * 40 + 64)); CK(hipMemset(mesh24, 0, (size_t)n * 24 + 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto&& launch) {
        std::vector<float> ts;
        for (int r = 0; r < 25; r++) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-36s median %7.1f us -> %5.0f GB/s (64 B/entry)\n", name, ts[12] * 1e3, n * 64.0 / (ts[12] * 1e-3) / 1e9);
    };
    const dim3 g((n + 255) / 256), blk(256);
    for (int rep = 0; rep < 2; rep++) {
        timeit("five SoA streams", [&] { hipLaunchKernelGGL(soa, g, blk, 0, 0, a, b, c, ma, mb, sink, n); });
        timeit("ab interleaved (32-B records) + 3", [&] { hipLaunchKernelGGL(interleaved, g, blk, 0, 0, ab, c, ma, mb, sink, n); });
        timeit("xf 40-B records + mesh SoA", [&] { hipLaunchKernelGGL(packed_xf_only, g, blk, 0, 0, xf40, ma, mb, sink, n); });
        timeit("xf 40-B records + mesh 24-B records", [&] { hipLaunchKernelGGL(packed, g, blk, 0, 0, xf40, mesh24, sink, n); });
    }
    return 0;
}
