// gv_device.hpp — device-side helpers shared by the kernel files: stream loads / stores, the transform record and its
// parent-chain model (transform.hpp:197-214), the per-entity filter chain + frustum test of the cull
// (mesh.cpp:140-157, render/mesh.hpp:142-146) and the build-defined Hi-Z occlusion query.
#pragma once
#include "gv_kernels.hpp"

#include <algorithm>

#include "gv_device_math.hpp"

namespace gv {


// ------------------------------------------------------------------------------------------------
// shared device helpers
// ------------------------------------------------------------------------------------------------
// The mirror streams are read once per frame: nontemporal loads (no L2/MALL allocation priority) measured
// +20 % on this access pattern (round-1 probe tools/kbench.hip, in the history: 6.1 -> 7.1 TB/s). Ancestor re-reads use plain loads.
typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef float f32x2n __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 stream_load(const float4* p)
{
    const f32x4n v = __builtin_nontemporal_load(reinterpret_cast<const f32x4n*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float2 stream_load(const float2* p)
{
    const f32x2n v = __builtin_nontemporal_load(reinterpret_cast<const f32x2n*>(p));
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ uint32_t stream_load(const uint32_t* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stream_store(float4* p, float4 v)
{
    __builtin_nontemporal_store(f32x4n{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4n*>(p));
}
__device__ __forceinline__ void stream_store(float* p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ uint32_t stream_load(const uint8_t* p) { return __builtin_nontemporal_load(p); }

// one transform entry: TRS + flag bits
struct XfRecord {
    float4 a, b;
    float2 c;
    uint32_t flags;
};
__device__ __forceinline__ XfRecord load_xf(const TransformMirror& xf, uint32_t s)
{
    XfRecord r;
    r.a = xf.ab[s].a;
    r.b = xf.ab[s].b;
    r.c = xf.c[s];
    r.flags = xf.flags[s];
    return r;
}
__device__ __forceinline__ XfRecord gather_xf(const TransformMirror& xf, uint32_t s, bool with_flags)
{
    XfRecord r;
    r.a = xf.ab[s].a;
    r.b = xf.ab[s].b;
    r.c = xf.c[s];
    r.flags = with_flags ? xf.flags[s] : 0u;
    return r;
}
__device__ __forceinline__ XfRecord stream_xf(const TransformMirror& xf, uint32_t s)  // the once-per-frame read
{
    XfRecord r;
    r.a = stream_load(&xf.ab[s].a);
    r.b = stream_load(&xf.ab[s].b);
    r.c = stream_load(&xf.c[s]);
    r.flags = stream_load(&xf.flags[s]);
    return r;
}
__device__ __forceinline__ Mat34 local_model(const XfRecord& r)
{
    return calc_model(r.a.x, r.a.y, r.a.z, r.b.x, r.b.y, r.b.z, r.b.w, r.a.w, r.c.x, r.c.y);
}

// transform.hpp:197-214: model = calcModel(self); while (parent) model = calcModel(parent) * model.
// `m` is the already-built self model of entry `s`; the parent stream is only touched when the pool has chains.
__device__ __forceinline__ Mat34 chain_model(const TransformMirror& xf, Mat34 m, uint32_t s, uint32_t flags)
{
    if (xf.max_depth != 0 && (flags & kXfWithAncestors)) {
        uint32_t p = xf.parent[s];
        for (uint32_t d = 0; d < xf.max_depth && p != kSlotNone; d++) {
            const XfRecord pr = load_xf(xf, p);
            m = mul_affine(local_model(pr), m);
            p = xf.parent[p];
        }
    }
    return m;
}

// Workgroup -> slot range is linear. An XCD-contiguous remap (each XCD's L2 owning one eighth of the slot
// range) was measured with flat and hierarchical scenes, random and Morton slot order: no effect (the streams
// have no inter-workgroup reuse and ancestor lines are shared through the Infinity Cache anyway).
__device__ __forceinline__ float hiz_min_texel(const HizDevice& hz, uint32_t level, uint32_t lw, uint32_t x, uint32_t y)
{
    if (level == 0)
        return hz.depth[(size_t)y * hz.width + x];
    if (level == 1 && hz.level1_virtual) {
        // Level 1 is the biggest level to write (half of all pyramid bytes) and the least read: with even sizes its
        // texel is just the 2x2 reduction of the depth image, in the build's own order (hiz.frag:29-33, MIN_DEPTH).
        const float2* row0 = reinterpret_cast<const float2*>(hz.depth + (size_t)(2 * y) * hz.width + 2 * x);
        const float2* row1 = reinterpret_cast<const float2*>(hz.depth + (size_t)(2 * y + 1) * hz.width + 2 * x);
        const float2 a = *row0, b = *row1;
        float m = a.x;
        m = a.y < m ? a.y : m;
        m = b.x < m ? b.x : m;
        m = b.y < m ? b.y : m;
        return hz.rg16f ? half_to_float(half_directed(m, false)) : m;  // what the stored texel would hold
    }
    const uint64_t at = hz.mip_offset[level] + (uint64_t)y * lw + x;
    if (hz.rg16f)  // uniform
        return half_to_float(reinterpret_cast<const uint32_t*>(hz.mips)[at] & 0xFFFFu);
    return hz.mips[at].x;
}
__device__ __forceinline__ float2 hiz_pair(const HizDevice& hz, uint64_t at)
{
    if (hz.rg16f)  // uniform
        return unpack_rg16f(reinterpret_cast<const uint32_t*>(hz.mips)[at]);
    return hz.mips[at];
}

__device__ __forceinline__ float clamp01(float a)
{
    return a > 0.0f ? (a < 1.0f ? a : 1.0f) : 0.0f;
}

constexpr uint32_t kHizCoarseStep = 4;  // early-accept level = query level + 4 (+3..+5 measured equal, +1/+2 slower)

// Build-defined occlusion query (SURVEY.md §8a-7'; the reference has none). Returns true if occluded.
__device__ __forceinline__ bool hiz_occluded(const HizDevice& hz, const float (&vp)[16], const Corners& c)
{
    float u[8], v[8], zc[8];
    bool bounded = true;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const v2f clx = pk_fma(splat(vp[0]), c.x[j], pk_fma(splat(vp[4]), c.y[j], pk_fma(splat(vp[8]), c.z[j], splat(vp[12]))));
        const v2f cly = pk_fma(splat(vp[1]), c.x[j], pk_fma(splat(vp[5]), c.y[j], pk_fma(splat(vp[9]), c.z[j], splat(vp[13]))));
        const v2f clz = pk_fma(splat(vp[2]), c.x[j], pk_fma(splat(vp[6]), c.y[j], pk_fma(splat(vp[10]), c.z[j], splat(vp[14]))));
        const v2f clw = pk_fma(splat(vp[3]), c.x[j], pk_fma(splat(vp[7]), c.y[j], pk_fma(splat(vp[11]), c.z[j], splat(vp[15]))));
        bounded = bounded && (clw.x > 0.0f) && (clw.y > 0.0f);
        // IEEE-correct division (-fhip-fp32-correctly-rounded-divide-sqrt), one per corner
        const v2f rcp = {1.0f / clw.x, 1.0f / clw.y};
        const v2f uu = pk_fma(clx * rcp, splat(0.5f), splat(0.5f));
        const v2f vv = pk_fma(cly * rcp, splat(0.5f), splat(0.5f));
        const v2f zz = clz * rcp;
        u[2 * j] = uu.x; u[2 * j + 1] = uu.y;
        v[2 * j] = vv.x; v[2 * j + 1] = vv.y;
        zc[2 * j] = zz.x; zc[2 * j + 1] = zz.y;
    }
    if (!bounded)
        return false;
    // IEEE minNum/maxNum reductions (v_min3_f32 / v_max3_f32), as the oracle's fminf/fmaxf
    const float umin0 = fminf(fminf(fminf(u[0], u[1]), fminf(u[2], u[3])), fminf(fminf(u[4], u[5]), fminf(u[6], u[7])));
    const float umax0 = fmaxf(fmaxf(fmaxf(u[0], u[1]), fmaxf(u[2], u[3])), fmaxf(fmaxf(u[4], u[5]), fmaxf(u[6], u[7])));
    const float vmin0 = fminf(fminf(fminf(v[0], v[1]), fminf(v[2], v[3])), fminf(fminf(v[4], v[5]), fminf(v[6], v[7])));
    const float vmax0 = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
    const float znear = fmaxf(fmaxf(fmaxf(zc[0], zc[1]), fmaxf(zc[2], zc[3])), fmaxf(fmaxf(zc[4], zc[5]), fmaxf(zc[6], zc[7])));
    float umin = umin0, umax = umax0, vmin = vmin0, vmax = vmax0;
    umin = clamp01(umin);
    umax = clamp01(umax);
    vmin = clamp01(vmin);
    vmax = clamp01(vmax);
    const int W = (int)hz.width, H = (int)hz.height;
    int ix0 = (int)(umin * (float)W), ix1 = (int)(umax * (float)W);
    int iy0 = (int)(vmin * (float)H), iy1 = (int)(vmax * (float)H);
    ix0 = min(ix0, W - 1);
    ix1 = min(ix1, W - 1);
    iy0 = min(iy0, H - 1);
    iy1 = min(iy1, H - 1);
    // Smallest level at which the pixel rect touches <= 2x2 texels. The oracle walks levels upward; per axis
    // the condition (i1 >> L) - (i0 >> L) <= 1 is monotone in L and first holds at floor(log2(n)) or one above
    // (n = i1 - i0 >= 2), so the level is max over the axes of that closed form.
    auto axis_level = [](int i0, int i1) -> uint32_t {
        const int n = i1 - i0;
        if (n <= 1)
            return 0u;
        const uint32_t l = 31u - (uint32_t)__clz(n);
        return ((i1 >> l) - (i0 >> l)) <= 1 ? l : l + 1u;
    };
    const uint32_t level = min(max(axis_level(ix0, ix1), axis_level(iy0, iy1)), hz.mip_count - 1u);
    // Exact early decisions from a coarse, cache-resident level (nested pyramids only; the answer is unchanged either
    // way). Its <= 4 texels cover a superset of the fine footprint, so their min is <= zFar and their max is >= every
    // fine texel, zFar included:
    //   zNear <  min(coarse)  =>  zNear < zFar   : occluded  — most occluded boxes end here,
    //   zNear >= max(coarse)  =>  zNear >= zFar  : visible   — boxes in front of everything around them end here,
    // and neither touches the 64 MB / 32 MB levels 0 / 1. (A NaN zNear fails both compares and takes the fine path.)
    if (hz.nested && level + kHizCoarseStep < hz.mip_count) {
        const uint32_t cl = level + kHizCoarseStep;  // >= 4: always a (min, max) level
        const int cw = max((int)(hz.width >> cl), 1), ch = max((int)(hz.height >> cl), 1);
        const int cx0 = min(ix0 >> cl, cw - 1), cx1 = min(ix1 >> cl, cw - 1);
        const int cy0 = min(iy0 >> cl, ch - 1), cy1 = min(iy1 >> cl, ch - 1);
        const uint64_t coarse = hz.mip_offset[cl];
        float2 t = hiz_pair(hz, coarse + (uint64_t)cy0 * cw + cx0);
        float cmin = t.x, cmax = t.y;
        if (cx1 != cx0) {
            t = hiz_pair(hz, coarse + (uint64_t)cy0 * cw + cx1);
            cmin = fminf(cmin, t.x);
            cmax = fmaxf(cmax, t.y);
        }
        if (cy1 != cy0) {
            t = hiz_pair(hz, coarse + (uint64_t)cy1 * cw + cx0);
            cmin = fminf(cmin, t.x);
            cmax = fmaxf(cmax, t.y);
            if (cx1 != cx0) {
                t = hiz_pair(hz, coarse + (uint64_t)cy1 * cw + cx1);
                cmin = fminf(cmin, t.x);
                cmax = fmaxf(cmax, t.y);
            }
        }
        if (znear < cmin)
            return true;
        if (znear >= cmax)
            return false;
    }
    const int lw = max((int)(hz.width >> level), 1), lh = max((int)(hz.height >> level), 1);
    const int tx0 = min(ix0 >> level, lw - 1), tx1 = min(ix1 >> level, lw - 1);
    const int ty0 = min(iy0 >> level, lh - 1), ty1 = min(iy1 >> level, lh - 1);
    float zfar = hiz_min_texel(hz, level, lw, tx0, ty0);
    float a = hiz_min_texel(hz, level, lw, tx1, ty0);
    zfar = a < zfar ? a : zfar;
    a = hiz_min_texel(hz, level, lw, tx0, ty1);
    zfar = a < zfar ? a : zfar;
    a = hiz_min_texel(hz, level, lw, tx1, ty1);
    zfar = a < zfar ? a : zfar;
    return znear < zfar;
}

// ------------------------------------------------------------------------------------------------
// K1: cull — one lane per mesh slot
// ------------------------------------------------------------------------------------------------
// Workgroups are handed out round-robin over the 8 XCDs, so with the identity mapping every XCD (own L2, own TLBs)
// touches every 8th 256-entry tile of each stream. With run = R > 0, XCD x takes R consecutive tiles at a time —
// tiles (8q + x)R .. (8q + x + 1)R - 1 for its q-th run — so it streams contiguous stretches (R * 8 KB of the
// 32-byte transform stream), while runs still alternate between the XCDs often enough that the expensive part of
// the pool (the entries inside the frustum are contiguous in the spatial order) stays spread over all of them.
// Measured at 10 M entities, same box A/B: frustum-only cull 110-112 -> 102-104 us (R = 32) -> 99 us (R = 256); the
// Hi-Z variant does not move (R = 32) or loses (R = 256: +4 us; one run per XCD: 134-140 us, the queries pile up on
// two XCDs), the block-bounds variant loses (78 -> 82 -> 89 us), the fused sweep + cull is flat: only the
// frustum-only scan uses it (profiles/r01b_kbench.txt).
// run length for a pool of `tiles` tiles: every XCD gets at least 8 runs (no surplus workgroups to speak of)
inline uint32_t xcd_run_for_tiles(uint32_t tiles)
{
    return tiles >= 8u * 256u * 8u ? 256u : (tiles >= 8u * 32u * 8u ? 32u : 0u);
}

__device__ __forceinline__ uint32_t tile_of_workgroup(uint32_t b, uint32_t run)
{
    if (run == 0)
        return b;
    const uint32_t xcd = b & 7u, k = b >> 3;
    return ((k / run) * 8u + xcd) * run + k % run;
}

// grid size for `tiles` tiles under that mapping: whole groups of 8 runs (surplus workgroups exit at once)
inline uint32_t grid_for_tiles(uint32_t tiles, uint32_t run)
{
    const uint32_t group = run * 8u;
    return run ? (tiles + group - 1) / group * group : tiles;
}

struct CullArgs {
    MeshMirror mesh;
    TransformMirror xf;
    HizDevice hiz;
    ViewParams view;
    ViewBuffers out;
    uint32_t nblocks;
    uint32_t xcd_run;    // tile_of_workgroup(); 0 = workgroup b takes tile b
};

// One mesh entry through the reference's filter chain (mesh.cpp:140-157): candidate / empty-AABB / transform /
// isActive checks, parent-chain model, camera translate, 8 corners. Returns false when the entry is filtered out;
// otherwise `m` holds the camera-relative model (bakedModel) and `c` its corners. Nothing here depends on the
// frustum, so shadow passes that share cameraPosition with the main pass (mesh.cpp:809-843) share this work.
// MAP (MeshMapping) only changes which streams are read and when; the result is the same for any mapping.
// the mesh's model-space AABB (render/mesh.hpp:54) as it sits in the mirror: a = (min.xyz, max.x), b = (max.y, max.z)
template <uint32_t MAP>
__device__ __forceinline__ bool prepare_model(const MeshMirror& mesh, const TransformMirror& xf, const float (&cam)[3],
                                              uint32_t i, Mat34& m, float4& ma, float2& mb)
{
    ma = stream_load(&mesh.a[i]);
    mb = stream_load(&mesh.b[i]);
    uint32_t slot = i;
    bool candidate = true;  // kMapExact: non-candidates carry an empty box and fall out below
    XfRecord r = {};
    if (MAP == kMapGeneral) {
        const uint32_t link = stream_load(&mesh.link[i]);
        slot = link & kSlotMask;
        candidate = (link & kMeshCandidate) && slot != kSlotNone;
        if (candidate)
            r = load_xf(xf, slot);
    } else {
        // the transform loads are issued beside the mesh loads instead of one HBM round trip later
        const bool own = i < xf.count;
        if (own) {
            if (MAP == kMapExact && xf.max_depth == 0) {
                // flat + exactly paired: only the active bit matters (no chain, so modelWithAncestors is moot) and
                // the 64 bits of this wave sit in one word
                r.a = stream_load(&xf.ab[i].a);
                r.b = stream_load(&xf.ab[i].b);
                r.c = stream_load(&xf.c[i]);
                r.flags = (uint32_t)((xf.active_bits[i >> 6] >> (i & 63u)) & 1ull) * kXfActive;
            } else {
                r = stream_xf(xf, i);
            }
        }
        if (MAP == kMapSpeculate) {
            const uint32_t link = stream_load(&mesh.link[i]);
            slot = link & kSlotMask;
            candidate = (link & kMeshCandidate) && slot != kSlotNone;
            if (candidate && !(own && slot == i))
                r = load_xf(xf, slot);  // mis-speculated: this entry maps elsewhere
        } else {
            candidate = own;
        }
    }
    const float mnx = ma.x, mny = ma.y, mnz = ma.z, mxx = ma.w, mxy = mb.x, mxz = mb.y;
    // mesh.cpp:140-142: skip free slots, disabled meshes and all(size <= 0) boxes
    const bool empty = (mxx - mnx <= 0.0f) && (mxy - mny <= 0.0f) && (mxz - mnz <= 0.0f);
    if (!candidate || empty)
        return false;
    if (!(r.flags & kXfActive))  // mesh.cpp:150, transform.hpp:110
        return false;
    const Mat34 world = chain_model(xf, local_model(r), slot, r.flags);
    // math::translate(-cameraPosition, model)  transform.hpp:211,213
    m = translated(world, cam[0], cam[1], cam[2]);
    return true;
}

__device__ __forceinline__ void aabb_corners(const Mat34& m, const float4 a, const float2 b, Corners& c)
{
    aabb_corners(m, a.x, a.y, a.z, a.w, b.x, b.y, c);
}

template <uint32_t MAP>
__device__ __forceinline__ bool prepare_slot(const MeshMirror& mesh, const TransformMirror& xf, const float (&cam)[3],
                                             uint32_t i, Mat34& m, Corners& c)
{
    float4 a;
    float2 b;
    if (!prepare_model<MAP>(mesh, xf, cam, i, m, a, b))
        return false;
    aabb_corners(m, a, b, c);
    return true;
}

// Sphere pre-test for the default predicate: decides most entries without generating a corner, and agrees with the
// exact 8-corner test whenever it decides. Every corner M*(x,y,z) + t lies within
//   r = |c0|_1 ax + |c1|_1 ay + |c2|_1 az   (a = max(|min|, |max|) per axis; L1 column norms bound the L2 norms)
// of the translation t. With d = n.t + w for a unit-normal plane:
//   d < -(r + slack)  =>  every corner's COMPUTED distance is < 0          -> behind this plane, as the exact test says
//   d >  (r + slack)  =>  every corner's computed distance is > 0          -> this plane cannot reject
// where slack covers the fp32 rounding of both evaluations: <= ~20 roundings of relative size 2^-24 on magnitudes
// <= |t|_inf + r + |w| (corner generation, the three fmas of a distance; both use the same matrix bits, so the chain
// depth does not enter), i.e. <= 1.2e-6 * magnitude; the slack is 0.01 + 4e-5 * magnitude, 30x that. Anything else
// — a plane within the band, non-finite values (every comparison false) — is "undecided" and takes the exact test.
// At 10 M entities 0.2-0.5 % of the entries are undecided; 230 -> ~170 VALU instructions per wave with Hi-Z.
enum : uint32_t { kSphereOutside = 0, kSphereInside = 1, kSphereUndecided = 2 };
// r + the magnitude-proportional part of the slack (view independent: shadow passes that share the camera share it)
__device__ __forceinline__ float sphere_reach(const Mat34& m, const float4 a, const float2 b)
{
    // NaN-propagating maxima: a NaN anywhere must reach the comparisons (fmaxf would drop it and decide for the exact test)
    const float ax = max_nan(fabsf(a.x), fabsf(a.w)), ay = max_nan(fabsf(a.y), fabsf(b.x)), az = max_nan(fabsf(a.z), fabsf(b.y));
    const float r = fmaf(fabsf(m.c0x) + fabsf(m.c0y) + fabsf(m.c0z), ax,
                         fmaf(fabsf(m.c1x) + fabsf(m.c1y) + fabsf(m.c1z), ay, (fabsf(m.c2x) + fabsf(m.c2y) + fabsf(m.c2z)) * az));
    const float mag = max_nan(max_nan(fabsf(m.c3x), fabsf(m.c3y)), fabsf(m.c3z)) + r;
    return fmaf(4e-5f, mag, r) + 0.01f;
}
__device__ __forceinline__ uint32_t classify_sphere(const Mat34& m, float reach, const float (&planes)[6][4], uint32_t plane_count)
{
    bool outside = false, decided = true;
#pragma unroll
    for (uint32_t p = 0; p < 6; p++)
        if (p < plane_count) {  // wave-uniform: the coefficients stay in SGPRs
            const float bound = fmaf(4e-5f, fabsf(planes[p][3]), reach);
            const float d = fmaf(planes[p][0], m.c3x, fmaf(planes[p][1], m.c3y, fmaf(planes[p][2], m.c3z, planes[p][3])));
            outside = outside | (d < -bound);  // no short circuit: straight-line code, the masks live in SGPR pairs
            decided = decided & ((d > bound) | (d < -bound));
        }
    return outside ? kSphereOutside : (decided ? kSphereInside : kSphereUndecided);
}
__device__ __forceinline__ uint32_t classify_sphere(const Mat34& m, const float4 a, const float2 b, const float (&planes)[6][4],
                                                    uint32_t plane_count)
{
    return classify_sphere(m, sphere_reach(m, a, b), planes, plane_count);
}

// default getReadyMeshesAsync predicate (render/mesh.hpp:142-146). Fully unrolled with a wave-uniform guard so
// the plane coefficients stay in SGPRs (a runtime-indexed kernarg array would be copied to LDS/scratch).
__device__ __forceinline__ bool behind_frustum(const Corners& c, const float (&planes)[6][4], uint32_t plane_count)
{
    bool behind = false;
#pragma unroll
    for (uint32_t p = 0; p < 6; p++)
        if (p < plane_count)
            behind = behind || all_behind_plane(c, planes[p][0], planes[p][1], planes[p][2], planes[p][3]);
    return behind;
}

template <uint32_t MAP>
__device__ __forceinline__ bool evaluate_slot(const MeshMirror& mesh, const TransformMirror& xf, const ViewParams& view,
                                              uint32_t i, Mat34& m, Corners& c)
{
    return prepare_slot<MAP>(mesh, xf, view.cam, i, m, c) && !behind_frustum(c, view.planes, view.plane_count);
}

}  // namespace gv
