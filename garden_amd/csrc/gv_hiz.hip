// gv_hiz.hip — Hi-Z pyramid: HizRenderSystem::downsampleHiz (source/system/render/hiz.cpp:104-167) with the reduction
// rule of shaders/hiz.frag:23-63.
#include "gv_device.hpp"
#include "gv_hiz_kernels.hpp"
#include "gv_hiz_device.hpp"

namespace gv {

// ------------------------------------------------------------------------------------------------
// Hi-Z pyramid
// ------------------------------------------------------------------------------------------------
// F16: the pyramid is an RG16F image (GV_CONFIG_HIZ_RG16F): pairs are packed binary16, written with the min rounded toward
// -inf and the max toward +inf. Only a reduction of the fp32 depth image rounds; halfs reduce to halfs.
template <bool F16>
__device__ __forceinline__ float2 hiz_src(const float* d, const float2* p, uint32_t sw, uint32_t x, uint32_t y)
{
    if (d) {  // HIZ_VARIANT_FIRST: (d, d)  hiz.frag:57-60
        const float v = d[(size_t)y * sw + x];
        return make_float2(v, v);
    }
    if (F16)
        return unpack_rg16f(reinterpret_cast<const uint32_t*>(p)[(size_t)y * sw + x]);
    return p[(size_t)y * sw + x];
}
// One destination texel of a level from the level before it; any size (hiz.frag:27-56 with the odd-size branches).
// SRC(x, y) -> (min, max) of the source level's texel.
template <class SRC>
__device__ __forceinline__ float2 hiz_level_texel_from(SRC src, uint32_t sw, uint32_t sh, uint32_t px, uint32_t py, uint32_t rule)
{
    const bool odd_x = (sw & 1u) != 0, odd_y = (sh & 1u) != 0;
    const uint32_t x0 = 2 * px, y0 = 2 * py;
    const uint32_t x1 = min(x0 + 1, sw - 1), y1 = min(y0 + 1, sh - 1);
    const uint32_t x2 = min(x0 + 2, sw - 1), y2 = min(y0 + 2, sh - 1);
    float2 mm = src(x0, y0);
    hiz_acc(mm, src(x1, y0));
    hiz_acc(mm, src(x0, y1));
    hiz_acc(mm, src(x1, y1));
    if (odd_x) {  // hiz.frag:36-41
        hiz_acc(mm, src(x2, y1));
        hiz_acc(mm, src(x2, y0));
        if (odd_y)  // hiz.frag:43-47
            hiz_acc(mm, src(x2, y2));
    }
    if (odd_y) {  // hiz.frag:49-55 reads gather components .y/.z = (2p.x+1, 2p.y+2), (2p.x+1, 2p.y+1)
        hiz_acc(mm, src(x1, y2));
        if (rule == 1u)  // GV_HIZ_RULE_CONSERVATIVE: the whole extra row
            hiz_acc(mm, src(x0, y2));
    }
    return mm;
}
template <bool F16>
__device__ __forceinline__ float2 hiz_level_texel(const float* __restrict__ src_depth, const float2* __restrict__ src_pairs, uint32_t sw,
                                                  uint32_t sh, uint32_t px, uint32_t py, uint32_t rule)
{
    return hiz_level_texel_from([&](uint32_t x, uint32_t y) { return hiz_src<F16>(src_depth, src_pairs, sw, x, y); }, sw, sh, px, py, rule);
}

// The same texel with the source read two columns at a time (one 8- or 16-byte load per row instead of two scalar ones: the
// scalar form touches every other word per instruction and is bound by the texture addresser, not by memory). Needs
// sw >= 2 && sh >= 2: then no coordinate of hiz.frag's footprint is clamped. Same accumulation order as above (the order
// decides which of +0 / -0 and whether a NaN survives).
struct __attribute__((aligned(4))) HizDepth2 { float a, b; };
struct __attribute__((aligned(4))) HizHalf2 { uint32_t a, b; };
struct __attribute__((aligned(8))) HizPair2 { float2 a, b; };
template <bool F16>
__device__ __forceinline__ void hiz_src2(const float* d, const float2* p, uint32_t sw, uint32_t x, uint32_t y, float2& a, float2& b)
{
    const size_t at = (size_t)y * sw + x;
    if (d) {
        const HizDepth2 v = *reinterpret_cast<const HizDepth2*>(d + at);
        a = make_float2(v.a, v.a);
        b = make_float2(v.b, v.b);
    } else if (F16) {
        const HizHalf2 v = *reinterpret_cast<const HizHalf2*>(reinterpret_cast<const uint32_t*>(p) + at);
        a = unpack_rg16f(v.a);
        b = unpack_rg16f(v.b);
    } else {
        const HizPair2 v = *reinterpret_cast<const HizPair2*>(p + at);
        a = v.a;
        b = v.b;
    }
}
template <bool F16>
__device__ __forceinline__ float2 hiz_level_texel_rows(const float* __restrict__ src_depth, const float2* __restrict__ src_pairs, uint32_t sw,
                                                       uint32_t sh, uint32_t px, uint32_t py, uint32_t rule)
{
    const bool odd_x = (sw & 1u) != 0, odd_y = (sh & 1u) != 0;
    const uint32_t x0 = 2 * px, y0 = 2 * py;
    float2 v00, v10, v01, v11, v20 = {}, v21 = {}, v02 = {}, v12 = {}, v22 = {};
    hiz_src2<F16>(src_depth, src_pairs, sw, x0, y0, v00, v10);
    hiz_src2<F16>(src_depth, src_pairs, sw, x0, y0 + 1, v01, v11);
    if (odd_x) {
        v20 = hiz_src<F16>(src_depth, src_pairs, sw, x0 + 2, y0);
        v21 = hiz_src<F16>(src_depth, src_pairs, sw, x0 + 2, y0 + 1);
    }
    if (odd_y) {
        hiz_src2<F16>(src_depth, src_pairs, sw, x0, y0 + 2, v02, v12);
        if (odd_x)
            v22 = hiz_src<F16>(src_depth, src_pairs, sw, x0 + 2, y0 + 2);
    }
    float2 mm = v00;
    hiz_acc(mm, v10);
    hiz_acc(mm, v01);
    hiz_acc(mm, v11);
    if (odd_x) {  // hiz.frag:36-41
        hiz_acc(mm, v21);
        hiz_acc(mm, v20);
        if (odd_y)  // hiz.frag:43-47
            hiz_acc(mm, v22);
    }
    if (odd_y) {  // hiz.frag:49-55
        hiz_acc(mm, v12);
        if (rule == 1u)
            hiz_acc(mm, v02);
    }
    (void)sh;
    return mm;
}

// One destination texel per lane.
template <bool F16>
__global__ __launch_bounds__(256) void hiz_level_kernel(const float* __restrict__ src_depth,
                                                        const float2* __restrict__ src_pairs,
                                                        float2* __restrict__ dst, uint32_t sw, uint32_t sh, uint32_t dw,
                                                        uint32_t dh, uint32_t rule)
{
    const uint32_t px = blockIdx.x * 64 + (threadIdx.x & 63u);
    const uint32_t py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= dw || py >= dh)
        return;
    hiz_store<F16>(dst, (size_t)py * dw + px, hiz_level_texel<F16>(src_depth, src_pairs, sw, sh, px, py, rule));
}

// The small levels of a pyramid of ANY size in one launch: one workgroup walks levels [first, first + count) in turn. Every
// level is written to the pyramid AND kept in LDS, where the next one reads it (two 64 KB halves in turn): a level costs a
// barrier instead of a launch or a round trip through L2. Frame sizes are rarely divisible by 64 (1920 x 1080, 2560 x 1440,
// 3840 x 2160: none is), so without this a build is one launch per level, most of them a few hundred texels.
template <bool F16>
__global__ __launch_bounds__(1024) void hiz_tail_kernel(const HizTailArgs a)
{
    __shared__ float2 level[2][kHizTailTexels];
    for (uint32_t l = 0; l < a.count; l++) {
        const uint32_t k = a.first + l;
        const uint32_t sw = a.w[k - 1], sh = a.h[k - 1], dw = a.w[k], dh = a.h[k];
        float2* dst = F16 ? reinterpret_cast<float2*>(reinterpret_cast<uint32_t*>(a.mips) + a.offset[k]) : a.mips + a.offset[k];
        float2* keep = level[l & 1u];
        const float2* prev = level[(l & 1u) ^ 1u];
        if (l == 0) {  // from memory: all of a lane's loads first, then its stores (the stores would fence the next texel's loads)
            const float* src_depth = k == 1 ? a.depth : nullptr;
            const float2* src_pairs = k == 1 ? nullptr
                                             : (F16 ? reinterpret_cast<const float2*>(reinterpret_cast<const uint32_t*>(a.mips) + a.offset[k - 1])
                                                    : a.mips + a.offset[k - 1]);
            float2 mine[kHizTailTexels / 1024];
#pragma unroll
            for (uint32_t u = 0; u < kHizTailTexels / 1024; u++) {
                const uint32_t t = threadIdx.x + u * 1024;
                if (t < dw * dh) {
                    mine[u] = sw >= 2 && sh >= 2 ? hiz_level_texel_rows<F16>(src_depth, src_pairs, sw, sh, t % dw, t / dw, a.rule)
                                                 : hiz_level_texel<F16>(src_depth, src_pairs, sw, sh, t % dw, t / dw, a.rule);
                    if (F16 && k == 1)  // the one place a value leaves fp32: what the texel holds is what the next level reduces
                        mine[u] = unpack_rg16f(pack_rg16f(mine[u]));
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < kHizTailTexels / 1024; u++) {
                const uint32_t t = threadIdx.x + u * 1024;
                if (t < dw * dh) {
                    hiz_store<F16>(dst, t, mine[u]);
                    keep[t] = mine[u];
                }
            }
        } else {
            for (uint32_t t = threadIdx.x; t < dw * dh; t += 1024) {
                const float2 mm = hiz_level_texel_from([&](uint32_t x, uint32_t y) { return prev[y * sw + x]; }, sw, sh, t % dw, t / dw, a.rule);
                hiz_store<F16>(dst, t, mm);
                keep[t] = mm;
            }
        }
        __syncthreads();
    }
}

// Three levels of ANY size in one launch: a workgroup owns 32 x 32 texels of level k+1, 16 x 16 of level k+2 and 8 x 8 of
// level k+3. The odd-size rule (hiz.frag:36-55) makes a texel reach one source column / row further, so the workgroup
// computes a rim beside what it owns — 37 x 37 of level k+1 and 18 x 18 of level k+2, kept in LDS, never stored — instead
// of waiting for its neighbours: 1.34x the level-k+1 arithmetic, no second and third launch, no round trip of the two
// intermediate levels through L2. Frame sizes are rarely divisible by 64 (1920 x 1080, 2560 x 1440, 3840 x 2160: none is),
// so this, not hiz_fused_kernel, is what an engine's pyramid build runs.
constexpr uint32_t kF3Own1 = 32, kF3Rim1 = 37, kF3Own2 = 16, kF3Rim2 = 18, kF3Own3 = 8;
template <bool F16>
__global__ __launch_bounds__(256) void hiz_fused3_kernel(const HizFused3Args a)
{
    __shared__ float2 l1[kF3Rim1][kF3Rim1 + 1];
    __shared__ float2 l2[kF3Rim2][kF3Rim2 + 1];
    const uint32_t sw = a.w[0], sh = a.h[0], w1 = a.w[1], h1 = a.h[1], w2 = a.w[2], h2 = a.h[2], w3 = a.w[3], h3 = a.h[3];
    // level k+1: [x1, x1 + cw1) x [y1, y1 + ch1), all loads of a lane's texels first
    const uint32_t x1 = blockIdx.x * kF3Own1, y1 = blockIdx.y * kF3Own1;
    const uint32_t cw1 = min(kF3Rim1, w1 - x1), ch1 = min(kF3Rim1, h1 - y1);
    constexpr uint32_t kPerLane = (kF3Rim1 * kF3Rim1 + 255u) / 256u;
    float2 mine[kPerLane];
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {
        const uint32_t t = threadIdx.x + u * 256u;
        const uint32_t lx = t % kF3Rim1, ly = t / kF3Rim1;
        if (lx < cw1 && ly < ch1) {
            float2 mm = hiz_level_texel_rows<F16>(a.depth, a.src_pairs, sw, sh, x1 + lx, y1 + ly, a.rule);
            if (F16 && a.depth)  // the one place a value leaves fp32: what the texel holds is what the next level reduces
                mm = unpack_rg16f(pack_rg16f(mm));
            mine[u] = mm;
        }
    }
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {
        const uint32_t t = threadIdx.x + u * 256u;
        const uint32_t lx = t % kF3Rim1, ly = t / kF3Rim1;
        if (lx < cw1 && ly < ch1) {
            l1[ly][lx] = mine[u];
            if (lx < kF3Own1 && ly < kF3Own1)
                hiz_store<F16>(a.dst[0], (size_t)(y1 + ly) * w1 + x1 + lx, mine[u]);
        }
    }
    __syncthreads();
    // level k+2 from LDS: [x2, x2 + cw2) x [y2, y2 + ch2)
    const uint32_t x2 = blockIdx.x * kF3Own2, y2 = blockIdx.y * kF3Own2;
    const uint32_t cw2 = x2 < w2 ? min(kF3Rim2, w2 - x2) : 0u, ch2 = y2 < h2 ? min(kF3Rim2, h2 - y2) : 0u;
    for (uint32_t t = threadIdx.x; t < kF3Rim2 * kF3Rim2; t += 256u) {
        const uint32_t lx = t % kF3Rim2, ly = t / kF3Rim2;
        if (lx >= cw2 || ly >= ch2)
            continue;
        const float2 mm = hiz_level_texel_from([&](uint32_t x, uint32_t y) { return l1[y - y1][x - x1]; }, w1, h1, x2 + lx, y2 + ly, a.rule);
        l2[ly][lx] = mm;
        if (lx < kF3Own2 && ly < kF3Own2)
            hiz_store<F16>(a.dst[1], (size_t)(y2 + ly) * w2 + x2 + lx, mm);
    }
    __syncthreads();
    // level k+3 from LDS
    const uint32_t x3 = blockIdx.x * kF3Own3, y3 = blockIdx.y * kF3Own3;
    const uint32_t cw3 = x3 < w3 ? min(kF3Own3, w3 - x3) : 0u, ch3 = y3 < h3 ? min(kF3Own3, h3 - y3) : 0u;
    const uint32_t lx = threadIdx.x % kF3Own3, ly = threadIdx.x / kF3Own3;
    if (lx < cw3 && ly < ch3) {
        const float2 mm = hiz_level_texel_from([&](uint32_t x, uint32_t y) { return l2[y - y2][x - x2]; }, w2, h2, x3 + lx, y3 + ly, a.rule);
        hiz_store<F16>(a.dst[2], (size_t)(y3 + ly) * w3 + x3 + lx, mm);
    }
}

hipError_t launch_hiz_fused3(const HizFused3Args& args, bool rg16f, hipStream_t stream)
{
    const dim3 grid((args.w[1] + kF3Own1 - 1) / kF3Own1, (args.h[1] + kF3Own1 - 1) / kF3Own1);
    if (rg16f)
        hipLaunchKernelGGL(hiz_fused3_kernel<true>, grid, dim3(256), 0, stream, args);
    else
        hipLaunchKernelGGL(hiz_fused3_kernel<false>, grid, dim3(256), 0, stream, args);
    return hipGetLastError();
}

// Four levels per launch (HizFused4Args): level k+1 in LDS with a rim of 11 (75 x 75 for the 64 x 64 the workgroup owns), then
// 37 x 37, 18 x 18 and the 8 x 8 of level k+4. 512 lanes: twelve level-k+1 texels each.
constexpr uint32_t kF4Threads = 512;
constexpr uint32_t kF4Own1 = 64, kF4Rim1 = 75, kF4Own2 = 32, kF4Rim2 = 37, kF4Own3 = 16, kF4Rim3 = 18, kF4Own4 = 8;
template <bool F16>
__global__ __launch_bounds__(kF4Threads) void hiz_fused4_kernel(const HizFused4Args a)
{
    __shared__ float2 l1[kF4Rim1][kF4Rim1 + 1];
    __shared__ float2 l2[kF4Rim2][kF4Rim2 + 1];
    __shared__ float2 l3[kF4Rim3][kF4Rim3 + 1];
    const uint32_t sw = a.w[0], sh = a.h[0], w1 = a.w[1], h1 = a.h[1], w2 = a.w[2], h2 = a.h[2], w3 = a.w[3], h3 = a.h[3], w4 = a.w[4], h4 = a.h[4];
    const uint32_t x1 = blockIdx.x * kF4Own1, y1 = blockIdx.y * kF4Own1;
    const uint32_t cw1 = min(kF4Rim1, w1 - x1), ch1 = min(kF4Rim1, h1 - y1);
    constexpr uint32_t kPerLane = (kF4Rim1 * kF4Rim1 + kF4Threads - 1u) / kF4Threads;
    constexpr uint32_t kBatch = 4;  // texels whose loads are in flight together
    for (uint32_t u0 = 0; u0 < kPerLane; u0 += kBatch) {
        float2 mine[kBatch];
#pragma unroll
        for (uint32_t u = 0; u < kBatch; u++) {
            const uint32_t t = threadIdx.x + (u0 + u) * kF4Threads;
            const uint32_t lx = t % kF4Rim1, ly = t / kF4Rim1;
            if (lx < cw1 && ly < ch1) {
                float2 mm = hiz_level_texel_rows<F16>(a.depth, a.src_pairs, sw, sh, x1 + lx, y1 + ly, a.rule);
                if (F16 && a.depth)  // the one place a value leaves fp32: what the texel holds is what the next level reduces
                    mm = unpack_rg16f(pack_rg16f(mm));
                mine[u] = mm;
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < kBatch; u++) {
            const uint32_t t = threadIdx.x + (u0 + u) * kF4Threads;
            const uint32_t lx = t % kF4Rim1, ly = t / kF4Rim1;
            if (lx < cw1 && ly < ch1) {
                l1[ly][lx] = mine[u];
                if (lx < kF4Own1 && ly < kF4Own1)
                    hiz_store<F16>(a.dst[0], (size_t)(y1 + ly) * w1 + x1 + lx, mine[u]);
            }
        }
    }
    __syncthreads();
    const uint32_t x2 = blockIdx.x * kF4Own2, y2 = blockIdx.y * kF4Own2;
    const uint32_t cw2 = x2 < w2 ? min(kF4Rim2, w2 - x2) : 0u, ch2 = y2 < h2 ? min(kF4Rim2, h2 - y2) : 0u;
    for (uint32_t t = threadIdx.x; t < kF4Rim2 * kF4Rim2; t += kF4Threads) {
        const uint32_t lx = t % kF4Rim2, ly = t / kF4Rim2;
        if (lx >= cw2 || ly >= ch2)
            continue;
        const float2 mm = hiz_level_texel_from([&](uint32_t x, uint32_t y) { return l1[y - y1][x - x1]; }, w1, h1, x2 + lx, y2 + ly, a.rule);
        l2[ly][lx] = mm;
        if (lx < kF4Own2 && ly < kF4Own2)
            hiz_store<F16>(a.dst[1], (size_t)(y2 + ly) * w2 + x2 + lx, mm);
    }
    __syncthreads();
    const uint32_t x3 = blockIdx.x * kF4Own3, y3 = blockIdx.y * kF4Own3;
    const uint32_t cw3 = x3 < w3 ? min(kF4Rim3, w3 - x3) : 0u, ch3 = y3 < h3 ? min(kF4Rim3, h3 - y3) : 0u;
    if (threadIdx.x < kF4Rim3 * kF4Rim3) {
        const uint32_t lx = threadIdx.x % kF4Rim3, ly = threadIdx.x / kF4Rim3;
        if (lx < cw3 && ly < ch3) {
            const float2 mm = hiz_level_texel_from([&](uint32_t x, uint32_t y) { return l2[y - y2][x - x2]; }, w2, h2, x3 + lx, y3 + ly, a.rule);
            l3[ly][lx] = mm;
            if (lx < kF4Own3 && ly < kF4Own3)
                hiz_store<F16>(a.dst[2], (size_t)(y3 + ly) * w3 + x3 + lx, mm);
        }
    }
    __syncthreads();
    const uint32_t x4 = blockIdx.x * kF4Own4, y4 = blockIdx.y * kF4Own4;
    const uint32_t cw4 = x4 < w4 ? min(kF4Own4, w4 - x4) : 0u, ch4 = y4 < h4 ? min(kF4Own4, h4 - y4) : 0u;
    const uint32_t lx = threadIdx.x % kF4Own4, ly = threadIdx.x / kF4Own4;
    if (lx < cw4 && ly < ch4) {
        const float2 mm = hiz_level_texel_from([&](uint32_t x, uint32_t y) { return l3[y - y3][x - x3]; }, w3, h3, x4 + lx, y4 + ly, a.rule);
        hiz_store<F16>(a.dst[3], (size_t)(y4 + ly) * w4 + x4 + lx, mm);
    }
}

hipError_t launch_hiz_fused4(const HizFused4Args& args, bool rg16f, hipStream_t stream)
{
    const dim3 grid((args.w[1] + kF4Own1 - 1) / kF4Own1, (args.h[1] + kF4Own1 - 1) / kF4Own1);
    if (rg16f)
        hipLaunchKernelGGL(hiz_fused4_kernel<true>, grid, dim3(kF4Threads), 0, stream, args);
    else
        hipLaunchKernelGGL(hiz_fused4_kernel<false>, grid, dim3(kF4Threads), 0, stream, args);
    return hipGetLastError();
}

hipError_t launch_hiz_tail(const HizTailArgs& args, bool rg16f, hipStream_t stream)
{
    if (args.count == 0)
        return hipSuccess;
    if (rg16f)
        hipLaunchKernelGGL(hiz_tail_kernel<true>, dim3(1), dim3(1024), 0, stream, args);
    else
        hipLaunchKernelGGL(hiz_tail_kernel<false>, dim3(1), dim3(1024), 0, stream, args);
    return hipGetLastError();
}

hipError_t launch_hiz_level(const float* src_depth, const float2* src_pairs, float2* dst, uint32_t sw, uint32_t sh,
                            uint32_t dw, uint32_t dh, uint32_t rule, bool rg16f, hipStream_t stream)
{
    const dim3 grid((dw + 63) / 64, (dh + 3) / 4);
    if (rg16f)
        hipLaunchKernelGGL(hiz_level_kernel<true>, grid, dim3(256), 0, stream, src_depth, src_pairs, dst, sw, sh, dw, dh, rule);
    else
        hipLaunchKernelGGL(hiz_level_kernel<false>, grid, dim3(256), 0, stream, src_depth, src_pairs, dst, sw, sh, dw, dh, rule);
    return hipGetLastError();
}

// Fused: one workgroup reduces a 64x64 source tile to 32^2, 16^2, 8^2, 4^2, 2^2 and 1 texel — six
// levels in one pass, the source read once, intermediate levels staged in LDS instead of re-read
// from HBM (the reference re-reads every mip in its own render pass, hiz.cpp:155-164).
template <bool PAIRS, bool F16>
__global__ __launch_bounds__(256) void hiz_fused_kernel(const float* __restrict__ src_depth,
                                                        const float2* __restrict__ src_pairs, const HizFusedDst dst,
                                                        uint32_t sw, uint32_t sh)
{
    hiz_fused_tile<PAIRS, F16>(src_depth, src_pairs, dst, sw, sh, blockIdx.x, blockIdx.y);
}

// The same from the depth image with TWO tiles per workgroup, side by side: all eight row loads of a lane are in flight before the
// first tile is reduced, so the second tile's bytes travel while the first goes through its four barriers — twice the bytes in
// flight per resident workgroup, half as many workgroups (one round of them over the GPU at 4096 x 4096 instead of two). Round 6,
// same box, 300 back-to-back builds at 4096 x 4096: one tile per workgroup 17.1 us, two 15.6, four 16.8 (profiles/withdrawn.md 42).
template <bool F16, uint32_t TILES>
__global__ __launch_bounds__(256) void hiz_fused_depth2_kernel(const float* __restrict__ src_depth, const HizFusedDst dst, uint32_t sw)
{
    float4 rows[TILES][4];
#pragma unroll
    for (uint32_t t = 0; t < TILES; t++)
        hiz_load_depth_rows(src_depth, sw, TILES * blockIdx.x + t, blockIdx.y, rows[t]);
#pragma unroll
    for (uint32_t t = 0; t < TILES; t++)
        hiz_reduce_depth_rows<F16>(rows[t], dst, sw, TILES * blockIdx.x + t, blockIdx.y);
}

hipError_t launch_hiz_fused(const float* src_depth, const float2* src_pairs, const HizFusedDst& dst, uint32_t sw,
                            uint32_t sh, bool rg16f, hipStream_t stream)
{
    if (src_depth && sw % 128 == 0 && (sw / 128) * (sh / 64) >= 1024) {  // (large images: enough pairs of tiles to fill the GPU)
        const dim3 pairs(sw / 128, sh / 64);
        if (rg16f)
            hipLaunchKernelGGL((hiz_fused_depth2_kernel<true, 2>), pairs, dim3(256), 0, stream, src_depth, dst, sw);
        else
            hipLaunchKernelGGL((hiz_fused_depth2_kernel<false, 2>), pairs, dim3(256), 0, stream, src_depth, dst, sw);
        return hipGetLastError();
    }
    const dim3 grid(sw / 64, sh / 64);
    if (src_depth && rg16f)
        hipLaunchKernelGGL((hiz_fused_kernel<false, true>), grid, dim3(256), 0, stream, src_depth, src_pairs, dst, sw, sh);
    else if (src_depth)
        hipLaunchKernelGGL((hiz_fused_kernel<false, false>), grid, dim3(256), 0, stream, src_depth, src_pairs, dst, sw, sh);
    else if (rg16f)
        hipLaunchKernelGGL((hiz_fused_kernel<true, true>), grid, dim3(256), 0, stream, src_depth, src_pairs, dst, sw, sh);
    else
        hipLaunchKernelGGL((hiz_fused_kernel<true, false>), grid, dim3(256), 0, stream, src_depth, src_pairs, dst, sw, sh);
    return hipGetLastError();
}

}  // namespace gv
