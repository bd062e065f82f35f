// csm_lite.hpp — the other side of the shadow passes: what CsmRenderSystem::prepareShadowRender hands to
// prepareMeshes for cascade `passIndex` (source/system/render/csm.cpp:262-305, 308-325): the light's viewProj and
// cameraOffset computed from the main camera's view, the light direction and the cascade's depth slice.
// the batched multi-view cull needs this on the host. The reference's math module is absent
// (SURVEY.md F1), so the helpers below are build-defined in the conventions the rest of the path uses (column-major,
// column vectors, +Z forward, reversed Z, Y flipped by the projection as calcPerspProjInfRevZ does in
// garden_amd/scene.py); parity unpinned like the rest. Host-only, header-only, no device code.
#pragma once
#include "garden_host.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>

namespace garden {
namespace csm {

struct Vec3 {
    double x, y, z;
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Vec3 normalized(Vec3 a)
{
    const double l = std::sqrt(dot(a, a));
    return l > 0 ? a * (1.0 / l) : Vec3{0, 0, 1};
}

// 4x4, column-major (m[4 * c + r]), in double: this is per-frame camera maths, rounded to fp32 once at the end
struct Mat4 {
    double m[16];
};
inline Mat4 fromF32(const f32x4x4& a)
{
    Mat4 r;
    for (int k = 0; k < 16; k++)
        r.m[k] = a.m[k];
    return r;
}
inline f32x4x4 toF32(const Mat4& a)
{
    f32x4x4 r;
    for (int k = 0; k < 16; k++)
        r.m[k] = (float)a.m[k];
    return r;
}
inline Mat4 mul(const Mat4& a, const Mat4& b)
{
    Mat4 r;
    for (int c = 0; c < 4; c++)
        for (int row = 0; row < 4; row++) {
            double s = 0;
            for (int k = 0; k < 4; k++)
                s += a.m[4 * k + row] * b.m[4 * c + k];
            r.m[4 * c + row] = s;
        }
    return r;
}
inline void transform(const Mat4& a, const double (&v)[4], double (&out)[4])
{
    for (int row = 0; row < 4; row++)
        out[row] = a.m[row] * v[0] + a.m[4 + row] * v[1] + a.m[8 + row] * v[2] + a.m[12 + row] * v[3];
}
inline Vec3 transformPoint(const Mat4& a, Vec3 p)
{
    const double v[4] = {p.x, p.y, p.z, 1.0};
    double o[4];
    transform(a, v, o);
    return {o[0], o[1], o[2]};
}
// inverse4x4 by Gauss-Jordan with partial pivoting; a singular input yields the identity
inline Mat4 inverse(const Mat4& a)
{
    double w[4][8];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            w[r][c] = a.m[4 * c + r];
            w[r][4 + c] = r == c ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; col++) {
        int pivot = col;
        for (int r = col + 1; r < 4; r++)
            if (std::fabs(w[r][col]) > std::fabs(w[pivot][col]))
                pivot = r;
        if (w[pivot][col] == 0.0) {
            Mat4 id{};
            id.m[0] = id.m[5] = id.m[10] = id.m[15] = 1.0;
            return id;
        }
        if (pivot != col)
            for (int c = 0; c < 8; c++)
                std::swap(w[pivot][c], w[col][c]);
        const double inv = 1.0 / w[col][col];
        for (int c = 0; c < 8; c++)
            w[col][c] *= inv;
        for (int r = 0; r < 4; r++)
            if (r != col && w[r][col] != 0.0) {
                const double f = w[r][col];
                for (int c = 0; c < 8; c++)
                    w[r][c] -= f * w[col][c];
            }
    }
    Mat4 out;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++)
            out.m[4 * c + r] = w[r][4 + c];
    return out;
}

// calcPerspProjRevZ: finite reversed-Z perspective, depth 1 at nearPlane and 0 at farPlane, w = view-space z
inline Mat4 perspRevZ(double fieldOfView, double aspectRatio, double nearPlane, double farPlane)
{
    const double f = 1.0 / std::tan(fieldOfView * 0.5);
    Mat4 p{};
    p.m[0] = f / aspectRatio;
    p.m[5] = -f;
    p.m[10] = -nearPlane / (farPlane - nearPlane);
    p.m[11] = 1.0;
    p.m[14] = nearPlane * farPlane / (farPlane - nearPlane);
    return p;
}
// calcOrthoProjRevZ(width(min,max), height(min,max), depth(min,max)): the box to x, y in [-1, 1] (y flipped like the
// perspective), depth 1 at depth.min and 0 at depth.max
inline Mat4 orthoRevZ(double x0, double x1, double y0, double y1, double z0, double z1)
{
    Mat4 p{};
    p.m[0] = 2.0 / (x1 - x0);
    p.m[5] = -2.0 / (y1 - y0);
    p.m[10] = -1.0 / (z1 - z0);
    p.m[12] = -(x1 + x0) / (x1 - x0);
    p.m[13] = (y1 + y0) / (y1 - y0);
    p.m[14] = z1 / (z1 - z0);
    p.m[15] = 1.0;
    return p;
}
// lookAt(eye, center): +Z looks from eye towards center, world +Y up (a light straight up or down falls back to +X)
inline Mat4 lookAt(Vec3 eye, Vec3 center)
{
    const Vec3 f = normalized(center - eye);
    Vec3 up{0, 1, 0};
    if (std::fabs(dot(f, up)) > 0.999)
        up = {1, 0, 0};
    const Vec3 s = normalized(cross(up, f)), u = cross(f, s);
    Mat4 v{};
    v.m[0] = s.x; v.m[4] = s.y; v.m[8] = s.z; v.m[12] = -dot(s, eye);
    v.m[1] = u.x; v.m[5] = u.y; v.m[9] = u.z; v.m[13] = -dot(u, eye);
    v.m[2] = f.x; v.m[6] = f.y; v.m[10] = f.z; v.m[14] = -dot(f, eye);
    v.m[15] = 1.0;
    return v;
}

struct Cascade {
    f32x4x4 viewProj;
    f32x4 cameraOffset;
};

// calcLightViewProj, csm.cpp:262-305. `view` is the camera view with its translation zeroed (graphics.cpp:201), so
// everything here is camera-relative like the rest of the path.
inline Cascade calcLightViewProj(const f32x4x4& view, f32x4 lightDir, float fieldOfView, float aspectRatio, float nearPlane,
                                 float farPlane, float zCoeff, uint32_t shadowMapSize)
{
    const Mat4 invViewProj = inverse(mul(perspRevZ(fieldOfView, aspectRatio, nearPlane, farPlane), fromF32(view)));
    Vec3 corners[8];
    int n = 0;
    for (int z = 0; z < 2; z++)          // csm.cpp:269-279: the slice's 8 corners, un-projected
        for (int y = 0; y < 2; y++)
            for (int x = 0; x < 2; x++) {
                const double ndc[4] = {x * 2.0 - 1.0, y * 2.0 - 1.0, (double)z, 1.0};
                double c[4];
                transform(invViewProj, ndc, c);
                corners[n++] = {c[0] / c[3], c[1] / c[3], c[2] / c[3]};
            }
    Vec3 center{0, 0, 0};
    for (const Vec3& c : corners)
        center = center + c;
    center = center * (1.0 / 8.0);       // :281-284
    const Vec3 dir{lightDir.x, lightDir.y, lightDir.z};
    const Mat4 lightView = lookAt(center - dir, center);  // :286
    Vec3 lo{DBL_MAX, DBL_MAX, DBL_MAX}, hi{-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (const Vec3& c : corners) {      // :289-294
        const Vec3 t = transformPoint(lightView, c);
        lo = {std::min(lo.x, t.x), std::min(lo.y, t.y), std::min(lo.z, t.z)};
        hi = {std::max(hi.x, t.x), std::max(hi.y, t.y), std::max(hi.z, t.z)};
    }
    lo.z = lo.z < 0.0 ? lo.z * zCoeff : lo.z / zCoeff;   // :296-297: pull the light's near / far apart so casters
    hi.z = hi.z < 0.0 ? hi.z / zCoeff : hi.z * zCoeff;   // outside the slice still land in the map
    const double unitsPerTexel = (hi.x - lo.x) / (double)shadowMapSize;  // :299-304: snap to texels
    Vec3 lightCameraPos = transformPoint(lightView, center);
    lightCameraPos.x = std::floor(lightCameraPos.x / unitsPerTexel) * unitsPerTexel;
    lightCameraPos.z = std::floor(lightCameraPos.z / unitsPerTexel) * unitsPerTexel;
    const Vec3 snapped = transformPoint(inverse(lightView), lightCameraPos);
    const Mat4 stabilized = lookAt(snapped - dir, snapped);
    Cascade out;
    const Vec3 offset = (dir * lo.z + center) * -1.0;  // :306
    out.cameraOffset = f32x4((float)offset.x, (float)offset.y, (float)offset.z, 0.0f);
    out.viewProj = toF32(mul(orthoRevZ(lo.x, hi.x, lo.y, hi.y, lo.z, hi.z), stabilized));  // :307-309
    return out;
}

// prepareShadowRender's slice selection, csm.cpp:318-324: cascade i covers [distance * splits[i-1], distance * splits[i]]
// (the camera's own near plane for i = 0, `distance` for the last one)
inline Cascade cascade(uint32_t passIndex, uint32_t cascadeCount, const float* cascadeSplits, float distance, const f32x4x4& view,
                       f32x4 lightDir, float fieldOfView, float aspectRatio, float cameraNearPlane, float zCoeff,
                       uint32_t shadowMapSize)
{
    const float nearPlane = passIndex > 0 ? distance * cascadeSplits[passIndex - 1] : cameraNearPlane;
    const float farPlane = passIndex + 1 < cascadeCount ? distance * cascadeSplits[passIndex] : distance;
    return calcLightViewProj(view, lightDir, fieldOfView, aspectRatio, nearPlane, farPlane, zCoeff, shadowMapSize);
}

}  // namespace csm
}  // namespace garden
