// kbench.hip — DEV TOOL (not product, not shipped): isolates what bounds the cull kernel on MI355X.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/kbench.hip -o tools/kbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <functional>
#include <algorithm>
#include "../garden_amd/csrc/gv_device_math.hpp"
using namespace gv;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Args {
    const float4* ma; const float3* mb; const float4* xa; const float4* xb; const float3* xc;
    const float4* mb4; const float4* xc4;
    uint8_t* vis; unsigned long long* mask; float* sink; uint32_t n; uint32_t per_xcd; uint32_t nblocks;
    float planes[6][4]; float cam[3];
};

template <int MODE, int BLOCK, bool XCD, bool NT, bool PAD>
__global__ __launch_bounds__(BLOCK) void k(const Args a)
{
    uint32_t lb = blockIdx.x;
    if (XCD) { lb = (blockIdx.x & 7u) * a.per_xcd + (blockIdx.x >> 3); if (lb >= a.nblocks) return; }
    const uint32_t i = lb * BLOCK + threadIdx.x;
    if (i >= a.n) return;
    float4 ma, xa, xb; float3 mb, xc;
    if (NT) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 t0 = __builtin_nontemporal_load((const f4*)&a.ma[i]), t1 = __builtin_nontemporal_load((const f4*)&a.xa[i]), t2 = __builtin_nontemporal_load((const f4*)&a.xb[i]);
        ma = make_float4(t0.x, t0.y, t0.z, t0.w); xa = make_float4(t1.x, t1.y, t1.z, t1.w); xb = make_float4(t2.x, t2.y, t2.z, t2.w);
    } else { ma = a.ma[i]; xa = a.xa[i]; xb = a.xb[i]; }
    if (PAD) { float4 t = a.mb4[i]; mb = make_float3(t.x, t.y, t.z); float4 u = a.xc4[i]; xc = make_float3(u.x, u.y, u.z); }
    else { mb = a.mb[i]; xc = a.xc[i]; }
    if (MODE == 0) {  // read only
        float s = ma.x + ma.y + ma.z + ma.w + mb.x + mb.y + mb.z + xa.x + xa.y + xa.z + xa.w + xb.x + xb.y + xb.z + xb.w + xc.x + xc.y + xc.z;
        if (s == 12345.678f) a.sink[i] = s;
        return;
    }
    const Mat34 local = calc_model(xa.x, xa.y, xa.z, xb.x, xb.y, xb.z, xb.w, xa.w, xc.x, xc.y);
    const Mat34 m = translated(local, a.cam[0], a.cam[1], a.cam[2]);
    Corners c;
    aabb_corners(m, ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, c);
    bool behind = false;
#pragma unroll
    for (int p = 0; p < 5; p++)
        behind = behind || all_behind_plane(c, a.planes[p][0], a.planes[p][1], a.planes[p][2], a.planes[p][3]);
    const bool visible = !behind && (__float_as_uint(mb.z) & 1u);
    if (MODE == 1) { if (visible && ma.x == 12345.678f) a.sink[i] = 1.0f; return; }
    a.vis[i] = visible;
    const unsigned long long w = __ballot(visible);
    if ((threadIdx.x & 63) == 0) a.mask[i >> 6] = w;
}

// 2 slots per lane, all loads issued first
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k2(const Args a)
{
    const uint32_t base = blockIdx.x * BLOCK * 2 + threadIdx.x;
    float4 ma[2], xa[2], xb[2]; float3 mb[2], xc[2];
#pragma unroll
    for (int s = 0; s < 2; s++) { const uint32_t i = min(base + s * BLOCK, a.n - 1); ma[s] = a.ma[i]; mb[s] = a.mb[i]; xa[s] = a.xa[i]; xb[s] = a.xb[i]; xc[s] = a.xc[i]; }
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const uint32_t i = base + s * BLOCK;
        const Mat34 local = calc_model(xa[s].x, xa[s].y, xa[s].z, xb[s].x, xb[s].y, xb[s].z, xb[s].w, xa[s].w, xc[s].x, xc[s].y);
        const Mat34 m = translated(local, a.cam[0], a.cam[1], a.cam[2]);
        Corners c;
        aabb_corners(m, ma[s].x, ma[s].y, ma[s].z, ma[s].w, mb[s].x, mb[s].y, c);
        bool behind = false;
#pragma unroll
        for (int p = 0; p < 5; p++)
            behind = behind || all_behind_plane(c, a.planes[p][0], a.planes[p][1], a.planes[p][2], a.planes[p][3]);
        const bool visible = !behind && (__float_as_uint(mb[s].z) & 1u) && i < a.n;
        if (i < a.n) a.vis[i] = visible;
        const unsigned long long w = __ballot(visible);
        if ((threadIdx.x & 63) == 0 && i < a.n) a.mask[i >> 6] = w;
    }
}

// grid-stride persistent: each block loops over tiles, prefetching the next tile's loads
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void kp(const Args a)
{
    const uint32_t ntiles = (a.n + BLOCK - 1) / BLOCK;
    uint32_t t = blockIdx.x;
    if (t >= ntiles) return;
    uint32_t i = min(t * BLOCK + threadIdx.x, a.n - 1);
    float4 ma = a.ma[i], xa = a.xa[i], xb = a.xb[i]; float3 mb = a.mb[i], xc = a.xc[i];
    for (; t < ntiles; t += gridDim.x) {
        const uint32_t tn = t + gridDim.x;
        const uint32_t in = min(tn * BLOCK + threadIdx.x, a.n - 1);
        float4 nma = ma, nxa = xa, nxb = xb; float3 nmb = mb, nxc = xc;
        if (tn < ntiles) { nma = a.ma[in]; nmb = a.mb[in]; nxa = a.xa[in]; nxb = a.xb[in]; nxc = a.xc[in]; }
        const uint32_t cur = t * BLOCK + threadIdx.x;
        const Mat34 local = calc_model(xa.x, xa.y, xa.z, xb.x, xb.y, xb.z, xb.w, xa.w, xc.x, xc.y);
        const Mat34 m = translated(local, a.cam[0], a.cam[1], a.cam[2]);
        Corners c;
        aabb_corners(m, ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, c);
        bool behind = false;
#pragma unroll
        for (int p = 0; p < 5; p++)
            behind = behind || all_behind_plane(c, a.planes[p][0], a.planes[p][1], a.planes[p][2], a.planes[p][3]);
        const bool visible = !behind && (__float_as_uint(mb.z) & 1u) && cur < a.n;
        if (cur < a.n) a.vis[cur] = visible;
        const unsigned long long w = __ballot(visible);
        if ((threadIdx.x & 63) == 0 && cur < a.n) a.mask[cur >> 6] = w;
        ma = nma; mb = nmb; xa = nxa; xb = nxb; xc = nxc;
    }
}

__global__ void copy_k(const float4* __restrict__ src, float4* __restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void read_k(const float4* __restrict__ src, float* sink, size_t n)
{
    float s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = src[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) sink[0] = s;
}

int main(int argc, char** argv)
{
    const uint32_t n = argc > 1 ? atoi(argv[1]) : 10000000;
    std::vector<float> h((size_t)n * 4);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    Args a{};
    float4 *ma, *xa, *xb, *mb4, *xc4; float3 *mb, *xc;
    CK(hipMalloc(&ma, (size_t)n * 16)); CK(hipMalloc(&xa, (size_t)n * 16)); CK(hipMalloc(&xb, (size_t)n * 16));
    CK(hipMalloc(&mb4, (size_t)n * 16)); CK(hipMalloc(&xc4, (size_t)n * 16));
    CK(hipMalloc(&mb, (size_t)n * 12)); CK(hipMalloc(&xc, (size_t)n * 12));
    CK(hipMalloc(&a.vis, n)); CK(hipMalloc(&a.mask, (size_t)(n / 64 + 1) * 8)); CK(hipMalloc(&a.sink, (size_t)n * 4));
    for (auto p : {(void*)ma, (void*)xa, (void*)xb, (void*)mb4, (void*)xc4}) CK(hipMemcpy(p, h.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    // scale positions up so that ~20% are visible-ish; flags bit set
    std::vector<float> hx((size_t)n * 4);
    for (size_t i = 0; i < n; i++) { hx[i*4] = h[i*4] * 1000; hx[i*4+1] = h[i*4+1] * 1000; hx[i*4+2] = h[i*4+2] * 1000; hx[i*4+3] = 1.0f; }
    CK(hipMemcpy(xa, hx.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    std::vector<float> h3((size_t)n * 3);
    for (size_t i = 0; i < n; i++) { h3[i*3] = 1.0f; h3[i*3+1] = 1.0f; uint32_t f = 1; memcpy(&h3[i*3+2], &f, 4); }
    CK(hipMemcpy(mb, h3.data(), (size_t)n * 12, hipMemcpyHostToDevice)); CK(hipMemcpy(xc, h3.data(), (size_t)n * 12, hipMemcpyHostToDevice));
    a.ma = ma; a.mb = mb; a.xa = xa; a.xb = xb; a.xc = xc; a.mb4 = mb4; a.xc4 = xc4; a.n = n;
    const float s = 0.70710678f;
    float pl[6][4] = {{s,0,s,0},{-s,0,s,0},{0,-s,s,0},{0,s,s,0},{0,0,1,-0.01f},{0,0,0,0}};
    memcpy(a.planes, pl, sizeof(pl));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double bytes, std::function<void()> fn) {
        for (int i = 0; i < 5; i++) fn();
        CK(hipStreamSynchronize(st));
        std::vector<float> ts;
        for (int r = 0; r < 15; r++) { CK(hipEventRecord(e0, st)); fn(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms); }
        std::sort(ts.begin(), ts.end());
        printf("%-44s median %8.1f us  min %8.1f us  -> %6.0f GB/s\n", name, ts[7] * 1e3, ts[0] * 1e3, bytes / (ts[7] * 1e-3) / 1e9);
    };
    const double rd = (double)n * 72, rdw = rd + n * 1.125, rdpad = (double)n * 80;
    auto grid = [&](int block, bool xcd) { a.nblocks = (n + block - 1) / block; a.per_xcd = (a.nblocks + 7) / 8; return dim3(xcd ? a.per_xcd * 8 : a.nblocks); };
    timeit("copy float4 (n*16 B rd + wr), 2048x256", (double)n * 32, [&] { hipLaunchKernelGGL(copy_k, dim3(2048), dim3(256), 0, st, ma, mb4, (size_t)n); });
    timeit("read float4 stream (n*16 B), 2048x256", (double)n * 16, [&] { hipLaunchKernelGGL(read_k, dim3(2048), dim3(256), 0, st, ma, a.sink, (size_t)n); });
#define RUN(NAME, MODE, BLOCK, XCD, NT, PAD, BYTES) { dim3 g = grid(BLOCK, XCD); timeit(NAME, BYTES, [&] { hipLaunchKernelGGL((k<MODE, BLOCK, XCD, NT, PAD>), g, dim3(BLOCK), 0, st, a); }); }
    RUN("read5 only, 256, linear", 0, 256, false, false, false, rd)
    RUN("read5 only, 256, xcd", 0, 256, true, false, false, rd)
    RUN("read5 only, 256, linear, nt", 0, 256, false, true, false, rd)
    RUN("read5 only, 256, linear, padded float4", 0, 256, false, false, true, rdpad)
    RUN("read5 only, 512, linear", 0, 512, false, false, false, rd)
    RUN("read5 only, 1024, linear", 0, 1024, false, false, false, rd)
    RUN("read5+compute, 256, linear", 1, 256, false, false, false, rd)
    RUN("read5+compute, 256, xcd", 1, 256, true, false, false, rd)
    RUN("read5+compute, 256, linear, padded", 1, 256, false, false, true, rdpad)
    RUN("read5+compute+write, 256, linear", 2, 256, false, false, false, rdw)
    RUN("read5+compute+write, 256, xcd", 2, 256, true, false, false, rdw)
    RUN("read5+compute+write, 256, linear, nt", 2, 256, false, true, false, rdw)
    RUN("read5+compute+write, 512, linear", 2, 512, false, false, false, rdw)
    RUN("read5+compute+write, 1024, linear", 2, 1024, false, false, false, rdw)
    { dim3 g((n + 511) / 512); timeit("2 slots/lane, 256", rdw, [&] { hipLaunchKernelGGL(k2<256>, g, dim3(256), 0, st, a); }); }
    for (int gsz : {2048, 4096, 8192}) { char nm[64]; snprintf(nm, 64, "persistent prefetch, 256 x %d", gsz); timeit(nm, rdw, [&] { hipLaunchKernelGGL(kp<256>, dim3(gsz), dim3(256), 0, st, a); }); }
    return 0;
}
