// gv_device_math.hpp — gfx950 device-side arithmetic of the visibility pass.
//
// The reference's math library (cfnptr/math) is an empty submodule in the checkout, so the
// operation order is fixed here and in DESIGN.md §"Canonical arithmetic": every multiply that feeds
// an add is an explicit fmaf(), the 4x4 product is a k = 0..3 fmaf chain from +0 (the order a
// v_mfma_f32_4x4x1_16b_f32 chain with a zero C operand produces), and nothing uses rcp/rsq
// approximations. Compile with -ffp-contract=off -fno-fast-math.
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
    const float r10 = fmaf(qx, y2, wz), r01 = fmaf(qx, y2, -wz);
    const float r20 = fmaf(qx, z2, -wy), r02 = fmaf(qx, z2, wy);
    const float r21 = fmaf(qy, z2, wx), r12 = fmaf(qy, z2, -wx);
    Mat34 m;
    m.c0x = r00 * sx; m.c0y = r10 * sx; m.c0z = r20 * sx;
    m.c1x = r01 * sy; m.c1y = r11 * sy; m.c1z = r21 * sy;
    m.c2x = r02 * sz; m.c2y = r12 * sz; m.c2z = r22 * sz;
    m.c3x = px; m.c3y = py; m.c3z = pz;
    return m;
}

// One output element of a * b: fmaf chain over k = 0..3 from +0; (b3) is the bottom-row element of
// b's column (0 for c0..c2, 1 for c3), a3 the element of a's translation column.
__device__ __forceinline__ float mul_elem(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3)
{
    float acc = fmaf(a0, b0, 0.0f);
    acc = fmaf(a1, b1, acc);
    acc = fmaf(a2, b2, acc);
    acc = fmaf(a3, b3, acc);
    return acc;
}

// parentModel * model, rows 0..2 (row 3 of both operands is exactly (0,0,0,1) and stays so).
__device__ __forceinline__ Mat34 mul_affine(const Mat34& a, const Mat34& b)
{
    Mat34 r;
    r.c0x = mul_elem(a.c0x, a.c1x, a.c2x, a.c3x, b.c0x, b.c0y, b.c0z, 0.0f);
    r.c0y = mul_elem(a.c0y, a.c1y, a.c2y, a.c3y, b.c0x, b.c0y, b.c0z, 0.0f);
    r.c0z = mul_elem(a.c0z, a.c1z, a.c2z, a.c3z, b.c0x, b.c0y, b.c0z, 0.0f);
    r.c1x = mul_elem(a.c0x, a.c1x, a.c2x, a.c3x, b.c1x, b.c1y, b.c1z, 0.0f);
    r.c1y = mul_elem(a.c0y, a.c1y, a.c2y, a.c3y, b.c1x, b.c1y, b.c1z, 0.0f);
    r.c1z = mul_elem(a.c0z, a.c1z, a.c2z, a.c3z, b.c1x, b.c1y, b.c1z, 0.0f);
    r.c2x = mul_elem(a.c0x, a.c1x, a.c2x, a.c3x, b.c2x, b.c2y, b.c2z, 0.0f);
    r.c2y = mul_elem(a.c0y, a.c1y, a.c2y, a.c3y, b.c2x, b.c2y, b.c2z, 0.0f);
    r.c2z = mul_elem(a.c0z, a.c1z, a.c2z, a.c3z, b.c2x, b.c2y, b.c2z, 0.0f);
    r.c3x = mul_elem(a.c0x, a.c1x, a.c2x, a.c3x, b.c3x, b.c3y, b.c3z, 1.0f);
    r.c3y = mul_elem(a.c0y, a.c1y, a.c2y, a.c3y, b.c3x, b.c3y, b.c3z, 1.0f);
    r.c3z = mul_elem(a.c0z, a.c1z, a.c2z, a.c3z, b.c3x, b.c3y, b.c3z, 1.0f);
    return r;
}

// The 8 local corners (bit0 -> x, bit1 -> y, bit2 -> z selects max) through the model, sharing the
// partial sums: per row t_z = fma(c2, z, c3); t_yz = fma(c1, y, t_z); p = fma(c0, x, t_yz) — the same
// bits as evaluating each corner on its own.
struct Corners {
    float x[8], y[8], z[8];
};
__device__ __forceinline__ void corner_row(float c0, float c1, float c2, float c3, float mnx, float mny, float mnz,
                                           float mxx, float mxy, float mxz, float (&out)[8])
{
    const float tz0 = fmaf(c2, mnz, c3), tz1 = fmaf(c2, mxz, c3);
    const float t00 = fmaf(c1, mny, tz0), t10 = fmaf(c1, mxy, tz0);
    const float t01 = fmaf(c1, mny, tz1), t11 = fmaf(c1, mxy, tz1);
    out[0] = fmaf(c0, mnx, t00); out[1] = fmaf(c0, mxx, t00);
    out[2] = fmaf(c0, mnx, t10); out[3] = fmaf(c0, mxx, t10);
    out[4] = fmaf(c0, mnx, t01); out[5] = fmaf(c0, mxx, t01);
    out[6] = fmaf(c0, mnx, t11); out[7] = fmaf(c0, mxx, t11);
}
__device__ __forceinline__ void aabb_corners(const Mat34& m, float mnx, float mny, float mnz, float mxx, float mxy,
                                             float mxz, Corners& c)
{
    corner_row(m.c0x, m.c1x, m.c2x, m.c3x, mnx, mny, mnz, mxx, mxy, mxz, c.x);
    corner_row(m.c0y, m.c1y, m.c2y, m.c3y, mnx, mny, mnz, mxx, mxy, mxz, c.y);
    corner_row(m.c0z, m.c1z, m.c2z, m.c3z, mnx, mny, mnz, mxx, mxy, mxz, c.z);
}

}  // namespace gv
