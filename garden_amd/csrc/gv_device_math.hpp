// gv_device_math.hpp — gfx950 device-side arithmetic of the visibility pass.
//
// The reference's math library (cfnptr/math) is an empty submodule in the checkout, so the
// operation order is fixed here and in DESIGN.md §4 "Canonical arithmetic": every multiply that feeds
// an add is an explicit fma, the 4x4 product is a k = 0..3 fma chain from +0 (the order a
// v_mfma_f32_4x4x1_16b_f32 chain with a zero C operand produces), and nothing uses rcp/rsq
// approximations. Compile with -ffp-contract=off -fno-fast-math.
//
// The cull kernel is VALU-issue-bound when written with scalar v_fma_f32 (one wave64 instruction per
// ~4 cycles per SIMD, profiles/r01a_*), so the corner / plane / product arithmetic is written on
// 2-wide vectors: hipcc lowers __builtin_elementwise_fma on float2 to v_pk_fma_f32 (two IEEE fmas per
// lane per instruction, SGPR operands broadcast through op_sel). Each element sees exactly the scalar
// sequence of the oracle, so the bits do not change.
//
// Call sites being replaced (reference paths):
//   math::calcModel        include/garden/system/transform.hpp:199,207,224
//   f32x4x4 operator*      include/garden/system/transform.hpp:209
//   math::translate        include/garden/system/transform.hpp:211,213
//   isBehindFrustum        include/garden/system/render/mesh.hpp:145
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gv {

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float a) { return v2f{a, a}; }
// NaN-propagating maximum (v_maximum3_f32): "all d < 0" == "maximum(d...) < 0" including the NaN case
// (a NaN distance makes the comparison false, exactly like the oracle's !(d < 0)).
__device__ __forceinline__ float max_nan(float a, float b) { return __builtin_elementwise_maximum(a, b); }

// Affine 3x4 part of a column-major f32x4x4 whose bottom row is (0,0,0,1): columns c0..c3, rows x,y,z.
struct Mat34 {
    float c0x, c0y, c0z;
    float c1x, c1y, c1z;
    float c2x, c2y, c2z;
    float c3x, c3y, c3z;
};

// T * R * S from position, unit quaternion (xyzw) and scale.
__device__ __forceinline__ Mat34 calc_model(float px, float py, float pz, float qx, float qy, float qz,
                                            float qw, float sx, float sy, float sz)
{
    const float x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
    const float zz = qz * z2, yy = qy * y2;
    const float wx = qw * x2, wy = qw * y2, wz = qw * z2;
    const float r00 = 1.0f - fmaf(qy, y2, zz);
    const float r11 = 1.0f - fmaf(qx, x2, zz);
    const float r22 = 1.0f - fmaf(qx, x2, yy);
    // (r10, r01) = fma(x, y2, (+wz, -wz)) etc.: one packed fma per off-diagonal pair
    const v2f p_xy = pk_fma(splat(qx), splat(y2), v2f{wz, -wz});
    const v2f p_xz = pk_fma(splat(qx), splat(z2), v2f{-wy, wy});
    const v2f p_yz = pk_fma(splat(qy), splat(z2), v2f{wx, -wx});
    const float r10 = p_xy.x, r01 = p_xy.y;
    const float r20 = p_xz.x, r02 = p_xz.y;
    const float r21 = p_yz.x, r12 = p_yz.y;
    Mat34 m;
    m.c0x = r00 * sx; m.c0y = r10 * sx; m.c0z = r20 * sx;
    m.c1x = r01 * sy; m.c1y = r11 * sy; m.c1z = r21 * sy;
    m.c2x = r02 * sz; m.c2y = r12 * sz; m.c2z = r22 * sz;
    m.c3x = px; m.c3y = py; m.c3z = pz;
    return m;
}

// One output column of a * b, rows x,y as a packed pair and row z scalar: per element the fma chain
// over k = 0..3 from +0; b3 is the bottom-row element of b's column (0 for c0..c2, 1 for c3).
__device__ __forceinline__ void mul_column(const Mat34& a, float b0, float b1, float b2, float b3, float& ox, float& oy,
                                           float& oz)
{
    v2f acc = pk_fma(v2f{a.c0x, a.c0y}, splat(b0), splat(0.0f));
    acc = pk_fma(v2f{a.c1x, a.c1y}, splat(b1), acc);
    acc = pk_fma(v2f{a.c2x, a.c2y}, splat(b2), acc);
    acc = pk_fma(v2f{a.c3x, a.c3y}, splat(b3), acc);
    float z = fmaf(a.c0z, b0, 0.0f);
    z = fmaf(a.c1z, b1, z);
    z = fmaf(a.c2z, b2, z);
    z = fmaf(a.c3z, b3, z);
    ox = acc.x;
    oy = acc.y;
    oz = z;
}

// parentModel * model, rows 0..2 (row 3 of both operands is exactly (0,0,0,1) and stays so).
__device__ __forceinline__ Mat34 mul_affine(const Mat34& a, const Mat34& b)
{
    Mat34 r;
    mul_column(a, b.c0x, b.c0y, b.c0z, 0.0f, r.c0x, r.c0y, r.c0z);
    mul_column(a, b.c1x, b.c1y, b.c1z, 0.0f, r.c1x, r.c1y, r.c1z);
    mul_column(a, b.c2x, b.c2y, b.c2z, 0.0f, r.c2x, r.c2y, r.c2z);
    mul_column(a, b.c3x, b.c3y, b.c3z, 1.0f, r.c3x, r.c3y, r.c3z);
    return r;
}

// math::translate(-cameraPosition, model): c3.xyz - cam (transform.hpp:211,213). Returns a fresh value so the
// matrix never lives in memory (in-place field updates made LLVM keep c3 in scratch/LDS).
__device__ __forceinline__ Mat34 translated(const Mat34& a, float cx, float cy, float cz)
{
    Mat34 r;
    r.c0x = a.c0x; r.c0y = a.c0y; r.c0z = a.c0z;
    r.c1x = a.c1x; r.c1y = a.c1y; r.c1z = a.c1z;
    r.c2x = a.c2x; r.c2y = a.c2y; r.c2z = a.c2z;
    r.c3x = a.c3x - cx;
    r.c3y = a.c3y - cy;
    r.c3z = a.c3z - cz;
    return r;
}

// The 8 local corners (bit0 -> x, bit1 -> y, bit2 -> z selects max) through the model, as 4 packed pairs
// per coordinate: pair j holds corners (2j, 2j+1), i.e. (x = min, x = max) for one (y, z) choice.
// Per row: t_z = fma(c2, z, c3); t_yz = fma(c1, y, t_z); p = fma(c0, x, t_yz) — the same bits as
// evaluating each corner on its own.
struct Corners {
    v2f x[4], y[4], z[4];
};
__device__ __forceinline__ void corner_row(float c0, float c1, float c2, float c3, v2f xs, v2f ys, v2f zs, v2f (&out)[4])
{
    const v2f tz = pk_fma(splat(c2), zs, splat(c3));          // (z = min, z = max)
    const v2f ty0 = pk_fma(splat(c1), ys, splat(tz.x));        // z = min: (y = min, y = max)
    const v2f ty1 = pk_fma(splat(c1), ys, splat(tz.y));        // z = max
    out[0] = pk_fma(splat(c0), xs, splat(ty0.x));              // corners 0,1: y min, z min
    out[1] = pk_fma(splat(c0), xs, splat(ty0.y));              // corners 2,3: y max, z min
    out[2] = pk_fma(splat(c0), xs, splat(ty1.x));              // corners 4,5: y min, z max
    out[3] = pk_fma(splat(c0), xs, splat(ty1.y));              // corners 6,7: y max, z max
}
__device__ __forceinline__ void aabb_corners(const Mat34& m, float mnx, float mny, float mnz, float mxx, float mxy,
                                             float mxz, Corners& c)
{
    const v2f xs = {mnx, mxx}, ys = {mny, mxy}, zs = {mnz, mxz};
    corner_row(m.c0x, m.c1x, m.c2x, m.c3x, xs, ys, zs, c.x);
    corner_row(m.c0y, m.c1y, m.c2y, m.c3y, xs, ys, zs, c.y);
    corner_row(m.c0z, m.c1z, m.c2z, m.c3z, xs, ys, zs, c.z);
}

// isBehindFrustum for one plane: all 8 signed distances d = fma(nx,px, fma(ny,py, fma(nz,pz, nw))) < 0.
__device__ __forceinline__ bool all_behind_plane(const Corners& c, float nx, float ny, float nz, float nw)
{
    v2f d[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
        d[j] = pk_fma(splat(nx), c.x[j], pk_fma(splat(ny), c.y[j], pk_fma(splat(nz), c.z[j], splat(nw))));
    const float m = max_nan(max_nan(max_nan(d[0].x, d[0].y), max_nan(d[1].x, d[1].y)),
                            max_nan(max_nan(d[2].x, d[2].y), max_nan(d[3].x, d[3].y)));
    return m < 0.0f;
}

// RG16F pyramid texels (GV_CONFIG_HIZ_RG16F; HizRenderSystem::bufferFormat, render/hiz.hpp:41): binary16 with the min rounded
// toward -inf and the max toward +inf, so a stored pair still bounds every depth it covers. Same function as the oracle's
// gvo_half_directed (checked against every binary16 value; tests/test_gpu_cull.py compares the two on every half and its
// float neighbours). NaN stays NaN (quiet), +-0 keep their sign, overflow goes to +-inf or +-65504 by direction.
__device__ __forceinline__ uint32_t half_directed(float f, bool up)
{
    // nearest-even by the hardware conversion (v_cvt_f16_f32; subnormal halfs included), then one step in the wanted
    // direction when nearest went the other way. A step is +-1 on the encoding: away from zero when the direction grows the
    // magnitude (0x7BFF + 1 = inf), toward zero otherwise (inf - 1 = 65504; -0 steps to the smallest negative half).
    const uint32_t sign = __float_as_uint(f) >> 31;
    if (f != f)
        return (sign << 15) | 0x7E00u;
    uint32_t h = (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)f);
    const float back = (float)__builtin_bit_cast(_Float16, (unsigned short)h);
    const bool grow = up != (sign != 0u);
    if (up ? back < f : back > f)
        h = grow ? h + 1u : h - 1u;
    return h;
}
// exact (binary16 is a subset of binary32; subnormals included: v_cvt_f32_f16 with the kernels' default denormal mode)
__device__ __forceinline__ float half_to_float(uint32_t half_bits)
{
    return (float)__builtin_bit_cast(_Float16, (unsigned short)half_bits);
}
// (min, max) -> one RG16F texel (min in the low half) and back
__device__ __forceinline__ uint32_t pack_rg16f(float2 mm)
{
    return half_directed(mm.x, false) | (half_directed(mm.y, true) << 16);
}
__device__ __forceinline__ float2 unpack_rg16f(uint32_t texel)
{
    return make_float2(half_to_float(texel & 0xFFFFu), half_to_float(texel >> 16));
}

}  // namespace gv
