// gv_hiz_device.hpp — the fused six-level reduction of one 64 x 64 source tile as a device function (hiz_fused_kernel,
// gv_hiz.hip). HizRenderSystem::downsampleHiz (source/system/render/hiz.cpp:104-167), shaders/hiz.frag:23-63.
#pragma once
#include "gv_device.hpp"
#include "gv_hiz_kernels.hpp"

namespace gv {

template <bool F16>
__device__ __forceinline__ void hiz_store(float2* level, size_t at, float2 mm)
{
    if (F16)
        reinterpret_cast<uint32_t*>(level)[at] = pack_rg16f(mm);
    else
        level[at] = mm;
}
__device__ __forceinline__ void hiz_acc(float2& mm, float2 t)
{
    mm.x = t.x < mm.x ? t.x : mm.x;  // MIN_DEPTH  depth.gsl:30-31
    mm.y = t.y > mm.y ? t.y : mm.y;  // MAX_DEPTH  depth.gsl:32-33
}

// Fused 6-level reduction of the 64 x 64 source tile (tile_x, tile_y) through LDS (256 lanes); dst.level[l] = level (src + 1 + l).
// hiz_fused_reduce: from the lane's 4 x 4 source texels (mn / mx) on; FIRST: the source was the fp32 depth image.
template <bool FIRST, bool F16>
__device__ __forceinline__ void hiz_fused_reduce(const float (&mn)[4][4], const float (&mx)[4][4], const HizFusedDst& dst, uint32_t sw, uint32_t tile_x,
                                                 uint32_t tile_y)
{
    __shared__ float2 lds16[16][17];
    __shared__ float2 lds8[8][9];
    __shared__ float2 lds4[4][5];
    __shared__ float2 lds2[2][3];
    const uint32_t tx = threadIdx.x & 15u, ty = threadIdx.x >> 4;
    const uint32_t ox = tile_x * 64, oy = tile_y * 64;
    constexpr bool PAIRS = !FIRST;
    // level +1: 2x2 texels per lane
    float2 q[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            float2 mm = make_float2(mn[2 * a][2 * b], mx[2 * a][2 * b]);
            hiz_acc(mm, make_float2(mn[2 * a][2 * b + 1], mx[2 * a][2 * b + 1]));
            hiz_acc(mm, make_float2(mn[2 * a + 1][2 * b], mx[2 * a + 1][2 * b]));
            hiz_acc(mm, make_float2(mn[2 * a + 1][2 * b + 1], mx[2 * a + 1][2 * b + 1]));
            if (F16 && !PAIRS)  // the one place a value leaves fp32: from here on every level reduces representable halfs
                mm = unpack_rg16f(pack_rg16f(mm));
            q[a][b] = mm;
        }
    const uint32_t w1 = sw >> 1;
    if (dst.level[0]) {  // null: the level stays virtual (queries reduce the source themselves)
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const size_t at = (size_t)(oy / 2 + 2 * ty + a) * w1 + ox / 2 + 2 * tx;
            if (F16)
                *reinterpret_cast<uint2*>(reinterpret_cast<uint32_t*>(dst.level[0]) + at) = make_uint2(pack_rg16f(q[a][0]), pack_rg16f(q[a][1]));
            else
                *reinterpret_cast<float4*>(dst.level[0] + at) = make_float4(q[a][0].x, q[a][0].y, q[a][1].x, q[a][1].y);
        }
    }
    // level +2: one texel per lane
    float2 m2 = q[0][0];
    hiz_acc(m2, q[0][1]);
    hiz_acc(m2, q[1][0]);
    hiz_acc(m2, q[1][1]);
    hiz_store<F16>(dst.level[1], (size_t)(oy / 4 + ty) * (sw >> 2) + ox / 4 + tx, m2);
    lds16[ty][tx] = m2;
    __syncthreads();
    if (threadIdx.x < 64) {  // level +3: 8x8
        const uint32_t x = threadIdx.x & 7u, y = threadIdx.x >> 3;
        float2 mm = lds16[2 * y][2 * x];
        hiz_acc(mm, lds16[2 * y][2 * x + 1]);
        hiz_acc(mm, lds16[2 * y + 1][2 * x]);
        hiz_acc(mm, lds16[2 * y + 1][2 * x + 1]);
        hiz_store<F16>(dst.level[2], (size_t)(oy / 8 + y) * (sw >> 3) + ox / 8 + x, mm);
        lds8[y][x] = mm;
    }
    __syncthreads();
    if (threadIdx.x < 16) {  // level +4: 4x4
        const uint32_t x = threadIdx.x & 3u, y = threadIdx.x >> 2;
        float2 mm = lds8[2 * y][2 * x];
        hiz_acc(mm, lds8[2 * y][2 * x + 1]);
        hiz_acc(mm, lds8[2 * y + 1][2 * x]);
        hiz_acc(mm, lds8[2 * y + 1][2 * x + 1]);
        hiz_store<F16>(dst.level[3], (size_t)(oy / 16 + y) * (sw >> 4) + ox / 16 + x, mm);
        lds4[y][x] = mm;
    }
    __syncthreads();
    if (threadIdx.x < 4) {  // level +5: 2x2
        const uint32_t x = threadIdx.x & 1u, y = threadIdx.x >> 1;
        float2 mm = lds4[2 * y][2 * x];
        hiz_acc(mm, lds4[2 * y][2 * x + 1]);
        hiz_acc(mm, lds4[2 * y + 1][2 * x]);
        hiz_acc(mm, lds4[2 * y + 1][2 * x + 1]);
        hiz_store<F16>(dst.level[4], (size_t)(oy / 32 + y) * (sw >> 5) + ox / 32 + x, mm);
        lds2[y][x] = mm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // level +6: 1 texel
        float2 mm = lds2[0][0];
        hiz_acc(mm, lds2[0][1]);
        hiz_acc(mm, lds2[1][0]);
        hiz_acc(mm, lds2[1][1]);
        hiz_store<F16>(dst.level[5], (size_t)(oy / 64) * (sw >> 6) + ox / 64, mm);
    }
}

// the lane's 4 x 4 source texels of tile (tile_x, tile_y): four 16-byte loads from the depth image
__device__ __forceinline__ void hiz_load_depth_rows(const float* __restrict__ src_depth, uint32_t sw, uint32_t tile_x, uint32_t tile_y, float4 (&rows)[4])
{
    const uint32_t px = tile_x * 64 + 4 * (threadIdx.x & 15u), py = tile_y * 64 + 4 * (threadIdx.x >> 4);
#pragma unroll
    for (int r = 0; r < 4; r++)
        rows[r] = stream_load(reinterpret_cast<const float4*>(src_depth + (size_t)(py + r) * sw + px));
}
template <bool F16>
__device__ __forceinline__ void hiz_reduce_depth_rows(const float4 (&rows)[4], const HizFusedDst& dst, uint32_t sw, uint32_t tile_x, uint32_t tile_y)
{
    float mn[4][4], mx[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        mn[r][0] = mx[r][0] = rows[r].x; mn[r][1] = mx[r][1] = rows[r].y;
        mn[r][2] = mx[r][2] = rows[r].z; mn[r][3] = mx[r][3] = rows[r].w;
    }
    hiz_fused_reduce<true, F16>(mn, mx, dst, sw, tile_x, tile_y);
}

template <bool PAIRS, bool F16>
__device__ __forceinline__ void hiz_fused_tile(const float* __restrict__ src_depth, const float2* __restrict__ src_pairs, const HizFusedDst& dst,
                                               uint32_t sw, uint32_t sh, uint32_t tile_x, uint32_t tile_y)
{
    (void)sh;
    if (!PAIRS) {
        float4 rows[4];
        hiz_load_depth_rows(src_depth, sw, tile_x, tile_y, rows);
        hiz_reduce_depth_rows<F16>(rows, dst, sw, tile_x, tile_y);
        return;
    }
    const uint32_t px = tile_x * 64 + 4 * (threadIdx.x & 15u), py = tile_y * 64 + 4 * (threadIdx.x >> 4);
    float mn[4][4], mx[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (F16) {
            const uint4 t = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint32_t*>(src_pairs) + (size_t)(py + r) * sw + px);
            const float2 t0 = unpack_rg16f(t.x), t1 = unpack_rg16f(t.y), t2 = unpack_rg16f(t.z), t3 = unpack_rg16f(t.w);
            mn[r][0] = t0.x; mx[r][0] = t0.y; mn[r][1] = t1.x; mx[r][1] = t1.y;
            mn[r][2] = t2.x; mx[r][2] = t2.y; mn[r][3] = t3.x; mx[r][3] = t3.y;
        } else {
            const float4* row = reinterpret_cast<const float4*>(src_pairs + (size_t)(py + r) * sw + px);
            const float4 lo = row[0], hi = row[1];
            mn[r][0] = lo.x; mx[r][0] = lo.y; mn[r][1] = lo.z; mx[r][1] = lo.w;
            mn[r][2] = hi.x; mx[r][2] = hi.y; mn[r][3] = hi.z; mx[r][3] = hi.w;
        }
    }
    hiz_fused_reduce<false, F16>(mn, mx, dst, sw, tile_x, tile_y);
}

}  // namespace gv
