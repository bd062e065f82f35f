// gv_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the visibility pass.
//
//   cull_kernel   replaces the body of prepareUnsortedMeshes / prepareSortedMeshes
//                 (source/system/render/mesh.cpp:137-175 / :213-253): filters, parent-chain model
//                 (include/garden/system/transform.hpp:197-214), 8-corner frustum test
//                 (render/mesh.hpp:142-146), optional Hi-Z occlusion query (build-defined), wave ballot.
//   scan_kernel + emit_kernel   replace drawCount.fetch_add + memcpy into combinedMeshes
//                 (mesh.cpp:177-183) with an order-stable compaction (ballot words + block prefix).
//   cull_multi_kernel   the same for up to 8 views that share cameraPosition (main camera + shadow cascades,
//                 mesh.cpp:795-847) in one pass over the streams.
//   block_bounds_kernel + the BOUNDS variants   opt-in workgroup boxes: conservative block-level frustum rejection.
//   sort_* / radix_*   sortMeshes (mesh.cpp:265-328): stable radix sort of the compact records by distanceSq.
//   sweep_*       TransformComponent::calcModel() for every transform slot (transform.hpp:197-214);
//                 the MFMA form runs the 4x4 chain on v_mfma_f32_4x4x1_16b_f32.
//   sweep_cull_*  sweep and cull of an exactly paired pool in ONE pass (world matrices + cull outputs).
//   hiz_*         HizRenderSystem::downsampleHiz (source/system/render/hiz.cpp:104-167) with the
//                 reduction rule of shaders/hiz.frag:23-63.
//
// All of it is HBM-bound streaming (DESIGN.md has bytes/entity per kernel); loads are 16- or 12-byte
// per lane over SoA streams so each wave-instruction touches 1 KiB / 768 B contiguous.
#include "gv_kernels.hpp"

#include <algorithm>

#include "gv_device_math.hpp"

namespace gv {

// ------------------------------------------------------------------------------------------------
// shared device helpers
// ------------------------------------------------------------------------------------------------
// The mirror streams are read once per frame: nontemporal loads (no L2/MALL allocation priority) measured
// +20 % on this access pattern (tools/kbench.hip: 6.1 -> 7.1 TB/s). Ancestor re-reads use plain loads.
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
    r.a = xf.a[s];
    r.b = xf.b[s];
    r.c = xf.c[s];
    r.flags = xf.flags[s];
    return r;
}
__device__ __forceinline__ XfRecord gather_xf(const TransformMirror& xf, uint32_t s, bool with_flags)
{
    XfRecord r;
    r.a = xf.a[s];
    r.b = xf.b[s];
    r.c = xf.c[s];
    r.flags = with_flags ? xf.flags[s] : 0u;
    return r;
}
__device__ __forceinline__ XfRecord stream_xf(const TransformMirror& xf, uint32_t s)  // the once-per-frame read
{
    XfRecord r;
    r.a = stream_load(&xf.a[s]);
    r.b = stream_load(&xf.b[s]);
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
        return m;
    }
    return hz.mips[hz.mip_offset[level] + (uint64_t)y * lw + x].x;
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
        const float2* coarse = hz.mips + hz.mip_offset[cl];
        float2 t = coarse[(uint64_t)cy0 * cw + cx0];
        float cmin = t.x, cmax = t.y;
        if (cx1 != cx0) {
            t = coarse[(uint64_t)cy0 * cw + cx1];
            cmin = fminf(cmin, t.x);
            cmax = fmaxf(cmax, t.y);
        }
        if (cy1 != cy0) {
            t = coarse[(uint64_t)cy1 * cw + cx0];
            cmin = fminf(cmin, t.x);
            cmax = fmaxf(cmax, t.y);
            if (cx1 != cx0) {
                t = coarse[(uint64_t)cy1 * cw + cx1];
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
struct CullArgs {
    MeshMirror mesh;
    TransformMirror xf;
    HizDevice hiz;
    ViewParams view;
    ViewBuffers out;
    uint32_t nblocks;
    BlockBounds bounds;  // BOUNDS variants only
};

// One mesh entry through the reference's filter chain (mesh.cpp:140-157): candidate / empty-AABB / transform /
// isActive checks, parent-chain model, camera translate, 8 corners. Returns false when the entry is filtered out;
// otherwise `m` holds the camera-relative model (bakedModel) and `c` its corners. Nothing here depends on the
// frustum, so shadow passes that share cameraPosition with the main pass (mesh.cpp:809-843) share this work.
// MAP (MeshMapping) only changes which streams are read and when; the result is the same for any mapping.
template <uint32_t MAP>
__device__ __forceinline__ bool prepare_slot(const MeshMirror& mesh, const TransformMirror& xf, const float (&cam)[3],
                                             uint32_t i, Mat34& m, Corners& c)
{
    const float4 ma = stream_load(&mesh.a[i]);
    const float2 mb = stream_load(&mesh.b[i]);
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
                r.a = stream_load(&xf.a[i]);
                r.b = stream_load(&xf.b[i]);
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
    aabb_corners(m, mnx, mny, mnz, mxx, mxy, mxz, c);
    return true;
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

// K1: one lane per mesh slot: visibility, isVisible byte, one ballot word per wave, per-chunk counts.
// Compaction is two-pass (ballot words -> chunk scan -> emit). Measured alternatives, all within a few % of
// this one in total time or worse (profiles/r01b_compaction_variants.txt): writing 56-byte records from K1
// into per-tile / per-wave staging segments and copying them (sparse partial sectors, +25..50 us in K1);
// LDS-staged fused emission with one global atomic per flush (occupancy, barriers); a decoupled look-back
// scan over 256-slot tiles (inter-workgroup latency and polling traffic dominate such small tiles).
// Conservative workgroup-level frustum rejection. The per-entity test rejects an entity when all 8 of its corners are
// behind one plane (computed distance < 0). The box holds every such corner of the workgroup's candidates as computed
// in world space; the per-frame corners are the same products with c3 - cam in place of c3, so a computed distance
// differs from the box-derived one by rounding only: a few ulps of the coordinate magnitude per chain level
// (<= ~4e-6 * M * (depth + 1)). The margin is an order of magnitude above that, so "box behind by more than the margin"
// implies "every computed corner distance < 0" — the rejected workgroup's entities all fail that plane in the exact
// test too. Boxes with non-finite members are +-inf and never satisfy the comparison (NaN / +inf are not < -margin).
__device__ __forceinline__ bool block_behind_planes(const float4 lo, const float4 hi, const float (&planes)[6][4],
                                                    uint32_t plane_count, const float (&cam)[3], uint32_t max_depth)
{
    const float mag = fmaxf(fabsf(lo.x), fabsf(hi.x)) + fmaxf(fabsf(lo.y), fabsf(hi.y)) + fmaxf(fabsf(lo.z), fabsf(hi.z)) +
                      fabsf(cam[0]) + fabsf(cam[1]) + fabsf(cam[2]);
    const float margin = 0.01f + 4e-5f * (float)(max_depth + 1u) * mag;
    bool behind = false;
#pragma unroll
    for (uint32_t p = 0; p < 6; p++)
        if (p < plane_count) {
            const float nx = planes[p][0], ny = planes[p][1], nz = planes[p][2];
            // the box corner farthest along the normal, camera-relative
            const float x = (nx >= 0.0f ? hi.x : lo.x) - cam[0];
            const float y = (ny >= 0.0f ? hi.y : lo.y) - cam[1];
            const float z = (nz >= 0.0f ? hi.z : lo.z) - cam[2];
            const float d = fmaf(nx, x, fmaf(ny, y, fmaf(nz, z, planes[p][3])));
            behind = behind || (d < -margin);
        }
    return behind;
}
__device__ __forceinline__ bool block_behind_frustum(const float4 lo, const float4 hi, const ViewParams& view, uint32_t max_depth)
{
    return block_behind_planes(lo, hi, view.planes, view.plane_count, view.cam, max_depth);
}

// The per-entity work of one 256-entry workgroup `lb`.
template <bool HIZ, uint32_t MAP>
__device__ __forceinline__ void cull_block(const CullArgs& args, uint32_t lb, uint32_t* wave_count)
{
    const uint32_t i = lb * kCullBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    bool visible = false;
    if (i < args.mesh.count) {
        Mat34 m;
        Corners c;
        visible = evaluate_slot<MAP>(args.mesh, args.xf, args.view, i, m, c);
        // Hi-Z occlusion query on the survivors. Measured (profiles/r01b_hiz_ablation.txt): compacting the
        // survivors across the workgroup through LDS first buys nothing — the stage is bound by the texel
        // gathers (~4.5 M random 64-B sectors per frame), not by divergent VALU work.
        if (HIZ && visible)
            visible = !hiz_occluded(args.hiz, args.view.vp, c);
        if (args.view.write_is_visible)
            args.out.is_visible[i] = visible ? 1 : 0;  // mesh.cpp:144,152,161,166
    }
    const unsigned long long word = __ballot(visible);
    if (lane == 0) {
        args.out.mask[(size_t)lb * (kCullBlock / 64) + wave] = word;
        wave_count[wave] = (uint32_t)__popcll(word);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
#pragma unroll
        for (uint32_t w = 0; w < kCullBlock / 64; w++)
            total += wave_count[w];
        if (total)  // integer adds commute: the sum is deterministic whatever the arrival order
            atomicAdd(&args.out.chunk_count[lb / (kEmitChunk / kCullBlock)], total);
    }
}

// BOUNDS (GV_CONFIG_BLOCK_BOUNDS): the workgroup first tests its box; when the box is behind a plane every entity in it
// is invisible, the outputs say so and the streams stay untouched.
template <bool HIZ, uint32_t MAP, bool BOUNDS>
__global__ __launch_bounds__(kCullBlock) void cull_kernel(const CullArgs args)
{
    __shared__ uint32_t wave_count[kCullBlock / 64];
    const uint32_t lb = blockIdx.x;
    if (BOUNDS) {  // workgroup-uniform
        const uint32_t i = lb * kCullBlock + threadIdx.x;
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        const float4 lo = args.bounds.lo[lb], hi = args.bounds.hi[lb];
        const bool empty = lo.x > hi.x;  // no candidate at all (+inf / -inf)
        const bool skip = empty || block_behind_frustum(lo, hi, args.view, args.xf.max_depth);
        if (threadIdx.x == 0)  // statistics: a plain store per workgroup (a shared counter would serialise ~10^4 atomics)
            args.bounds.examined[lb] = skip ? 0 : 1;
        if (skip) {
            if (args.view.write_is_visible && i < args.mesh.count)
                args.out.is_visible[i] = 0;
            if (lane == 0)
                args.out.mask[(size_t)lb * (kCullBlock / 64) + wave] = 0ull;
            return;
        }
    }
    cull_block<HIZ, MAP>(args, lb, wave_count);
}

hipError_t launch_cull(const MeshMirror& mesh, const TransformMirror& xf, const HizDevice& hiz, const ViewParams& vp,
                       const ViewBuffers& out, hipStream_t stream, const BlockBounds* bounds)
{
    if (mesh.count == 0)
        return hipSuccess;
    CullArgs a;
    a.mesh = mesh;
    a.xf = xf;
    a.hiz = hiz;
    a.view = vp;
    a.out = out;
    a.nblocks = (mesh.count + kCullBlock - 1) / kCullBlock;
    a.bounds = bounds ? *bounds : BlockBounds{};
    const dim3 grid(a.nblocks), block(kCullBlock);
#define GV_LAUNCH_CULL(HIZ, BOUNDS)                                                                                       \
    switch (mesh.mapping) {                                                                                              \
    case kMapExact: hipLaunchKernelGGL((cull_kernel<HIZ, kMapExact, BOUNDS>), grid, block, 0, stream, a); break;         \
    case kMapSpeculate: hipLaunchKernelGGL((cull_kernel<HIZ, kMapSpeculate, BOUNDS>), grid, block, 0, stream, a); break; \
    default: hipLaunchKernelGGL((cull_kernel<HIZ, kMapGeneral, BOUNDS>), grid, block, 0, stream, a); break;              \
    }
    if (vp.use_hiz && bounds) {
        GV_LAUNCH_CULL(true, true)
    } else if (vp.use_hiz) {
        GV_LAUNCH_CULL(true, false)
    } else if (bounds) {
        GV_LAUNCH_CULL(false, true)
    } else {
        GV_LAUNCH_CULL(false, false)
    }
#undef GV_LAUNCH_CULL
    return hipGetLastError();
}

// World-space box of each cull workgroup's candidates (camera at the origin: translate(-0) leaves c3 as it is).
template <uint32_t MAP>
__global__ __launch_bounds__(kCullBlock) void block_bounds_kernel(const MeshMirror mesh, const TransformMirror xf,
                                                                  float4* __restrict__ out_lo, float4* __restrict__ out_hi)
{
    __shared__ float red[kCullBlock / 64][6];
    const uint32_t lb = blockIdx.x;
    const uint32_t i = lb * kCullBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const float inf = __builtin_huge_valf();
    float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    if (i < mesh.count) {
        Mat34 m;
        Corners c;
        const float cam[3] = {0.0f, 0.0f, 0.0f};
        if (prepare_slot<MAP>(mesh, xf, cam, i, m, c)) {
            bool finite = true;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float xs[2] = {c.x[k].x, c.x[k].y}, ys[2] = {c.y[k].x, c.y[k].y}, zs[2] = {c.z[k].x, c.z[k].y};
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    finite = finite && isfinite(xs[h]) && isfinite(ys[h]) && isfinite(zs[h]);
                    lo[0] = fminf(lo[0], xs[h]); hi[0] = fmaxf(hi[0], xs[h]);
                    lo[1] = fminf(lo[1], ys[h]); hi[1] = fmaxf(hi[1], ys[h]);
                    lo[2] = fminf(lo[2], zs[h]); hi[2] = fmaxf(hi[2], zs[h]);
                }
            }
            if (!finite)  // a member the box cannot bound: the workgroup is always examined
                for (int k = 0; k < 3; k++) {
                    lo[k] = -inf;
                    hi[k] = inf;
                }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (uint32_t d = 32; d >= 1; d >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], d, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], d, 64));
        }
    if (lane == 0)
        for (int k = 0; k < 3; k++) {
            red[wave][k] = lo[k];
            red[wave][3 + k] = hi[k];
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < kCullBlock / 64; w++)
            for (int k = 0; k < 3; k++) {
                red[0][k] = fminf(red[0][k], red[w][k]);
                red[0][3 + k] = fmaxf(red[0][3 + k], red[w][3 + k]);
            }
        out_lo[lb] = make_float4(red[0][0], red[0][1], red[0][2], 0.0f);
        out_hi[lb] = make_float4(red[0][3], red[0][4], red[0][5], 0.0f);
    }
}

hipError_t launch_block_bounds(const MeshMirror& mesh, const TransformMirror& xf, float4* lo, float4* hi, hipStream_t stream)
{
    if (mesh.count == 0)
        return hipSuccess;
    const dim3 grid((mesh.count + kCullBlock - 1) / kCullBlock), block(kCullBlock);
    switch (mesh.mapping) {
    case kMapExact: hipLaunchKernelGGL((block_bounds_kernel<kMapExact>), grid, block, 0, stream, mesh, xf, lo, hi); break;
    case kMapSpeculate: hipLaunchKernelGGL((block_bounds_kernel<kMapSpeculate>), grid, block, 0, stream, mesh, xf, lo, hi); break;
    default: hipLaunchKernelGGL((block_bounds_kernel<kMapGeneral>), grid, block, 0, stream, mesh, xf, lo, hi); break;
    }
    return hipGetLastError();
}

// K1, batched over views that share cameraPosition (main camera + shadow cascades, mesh.cpp:795-903): the
// streams are read and the model / corners computed ONCE; each view then costs its plane tests (+ the Hi-Z query
// on view 0 only) and its own outputs. The reference re-runs the whole loop per pass (mesh.cpp:809-843).
struct MultiCullArgs {
    MeshMirror mesh;
    TransformMirror xf;
    HizDevice hiz;
    float cam[3];
    float vp0[16];          // view 0's viewProj (Hi-Z query)
    uint32_t use_hiz0;
    uint32_t nviews;
    MultiViewPlanes planes[kMaxBatchViews];
    ViewBuffers outs[kMaxBatchViews];
    BlockBounds bounds;  // BOUNDS variants only
};

template <bool HIZ, uint32_t MAP, bool BOUNDS>
__global__ __launch_bounds__(kCullBlock) void cull_multi_kernel(const MultiCullArgs args)
{
    const uint32_t lb = blockIdx.x;
    const uint32_t i = lb * kCullBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool in_range = i < args.mesh.count;
    if (BOUNDS) {  // the workgroup is skipped when its box is outside EVERY view of the batch
        const float4 lo = args.bounds.lo[lb], hi = args.bounds.hi[lb];
        bool skip = true;
        if (!(lo.x > hi.x)) {
#pragma unroll
            for (uint32_t v = 0; v < kMaxBatchViews; v++)
                if (v < args.nviews)
                    skip = skip && block_behind_planes(lo, hi, args.planes[v].planes, args.planes[v].plane_count, args.cam,
                                                       args.xf.max_depth);
        }
        if (threadIdx.x == 0)
            args.bounds.examined[lb] = skip ? 0 : 1;
        if (skip) {
#pragma unroll
            for (uint32_t v = 0; v < kMaxBatchViews; v++)
                if (v < args.nviews) {
                    if (args.planes[v].write_is_visible && in_range)
                        args.outs[v].is_visible[i] = 0;
                    if (lane == 0)
                        args.outs[v].mask[(size_t)lb * (kCullBlock / 64) + wave] = 0ull;
                }
            return;
        }
    }
    Mat34 m;
    Corners c;
    const bool candidate = in_range && prepare_slot<MAP>(args.mesh, args.xf, args.cam, i, m, c);
    __shared__ uint32_t wave_count[kMaxBatchViews][kCullBlock / 64];
#pragma unroll
    for (uint32_t v = 0; v < kMaxBatchViews; v++) {
        if (v < args.nviews) {  // uniform
            bool visible = candidate && !behind_frustum(c, args.planes[v].planes, args.planes[v].plane_count);
            if (HIZ && v == 0 && visible)
                visible = !hiz_occluded(args.hiz, args.vp0, c);
            if (args.planes[v].write_is_visible && in_range)
                args.outs[v].is_visible[i] = visible ? 1 : 0;
            const unsigned long long word = __ballot(visible);
            if (lane == 0) {
                args.outs[v].mask[(size_t)lb * (kCullBlock / 64) + wave] = word;
                wave_count[v][wave] = (uint32_t)__popcll(word);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < args.nviews) {
        uint32_t total = 0;
#pragma unroll
        for (uint32_t w = 0; w < kCullBlock / 64; w++)
            total += wave_count[threadIdx.x][w];
        if (total) {
            // outs[] indexed by a lane-varying view: pick the pointer with uniform compares (kernarg stays in SGPRs)
            uint32_t* counts = nullptr;
#pragma unroll
            for (uint32_t v = 0; v < kMaxBatchViews; v++)
                if (threadIdx.x == v)
                    counts = args.outs[v].chunk_count;
            atomicAdd(&counts[lb / (kEmitChunk / kCullBlock)], total);
        }
    }
}

hipError_t launch_cull_multi(const MeshMirror& mesh, const TransformMirror& xf, const HizDevice& hiz,
                             const ViewParams* views, const ViewBuffers* outs, uint32_t nviews, hipStream_t stream,
                             const BlockBounds* bounds)
{
    if (mesh.count == 0)
        return hipSuccess;
    if (nviews == 0 || nviews > kMaxBatchViews)
        return hipErrorInvalidValue;
    MultiCullArgs a;
    a.mesh = mesh;
    a.xf = xf;
    a.hiz = hiz;
    for (int k = 0; k < 3; k++)
        a.cam[k] = views[0].cam[k];
    for (int k = 0; k < 16; k++)
        a.vp0[k] = views[0].vp[k];
    a.use_hiz0 = views[0].use_hiz;
    a.nviews = nviews;
    for (uint32_t v = 0; v < kMaxBatchViews; v++) {
        const ViewParams& src = views[v < nviews ? v : 0];
        for (int p = 0; p < 6; p++)
            for (int k = 0; k < 4; k++)
                a.planes[v].planes[p][k] = src.planes[p][k];
        a.planes[v].plane_count = src.plane_count;
        a.planes[v].write_is_visible = src.write_is_visible;
        a.outs[v] = outs[v < nviews ? v : 0];
    }
    const dim3 grid((mesh.count + kCullBlock - 1) / kCullBlock), block(kCullBlock);
    a.bounds = bounds ? *bounds : BlockBounds{};
#define GV_LAUNCH_MULTI(HIZ, BOUNDS)                                                                                            \
    switch (mesh.mapping) {                                                                                                    \
    case kMapExact: hipLaunchKernelGGL((cull_multi_kernel<HIZ, kMapExact, BOUNDS>), grid, block, 0, stream, a); break;         \
    case kMapSpeculate: hipLaunchKernelGGL((cull_multi_kernel<HIZ, kMapSpeculate, BOUNDS>), grid, block, 0, stream, a); break; \
    default: hipLaunchKernelGGL((cull_multi_kernel<HIZ, kMapGeneral, BOUNDS>), grid, block, 0, stream, a); break;              \
    }
    if (a.use_hiz0 && bounds) {
        GV_LAUNCH_MULTI(true, true)
    } else if (a.use_hiz0) {
        GV_LAUNCH_MULTI(true, false)
    } else if (bounds) {
        GV_LAUNCH_MULTI(false, true)
    } else {
        GV_LAUNCH_MULTI(false, false)
    }
#undef GV_LAUNCH_MULTI
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// K2: exclusive scan of the per-block visible counts (one workgroup; <= ~400k blocks at 10^8 slots)
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kScanBlock = 1024;

// n = slots / 4096 chunk totals (2 442 at 10^7 slots): 1024 per pass, coalesced, carry across passes.
// Reads each count once and writes 0 back so the next frame's cull workgroups can add into it again.
__global__ __launch_bounds__(kScanBlock) void scan_kernel(uint32_t* __restrict__ counts,
                                                          uint32_t* __restrict__ offsets,
                                                          uint32_t* __restrict__ total, uint32_t n)
{
    __shared__ uint32_t wave_sum[kScanBlock / 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += kScanBlock) {
        const uint32_t idx = base + threadIdx.x;
        uint32_t v = 0;
        if (idx < n) {
            v = counts[idx];
            counts[idx] = 0;
        }
        uint32_t incl = v;  // wave-level inclusive scan
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d)
                incl += up;
        }
        if (lane == 63)
            wave_sum[wave] = incl;
        __syncthreads();
        uint32_t wave_prefix = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < kScanBlock / 64; w++) {
            wave_prefix += w < wave ? wave_sum[w] : 0u;
            all += wave_sum[w];
        }
        if (idx < n)
            offsets[idx] = carry + wave_prefix + incl - v;
        carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0)
        *total = carry;
}

hipError_t launch_scan(const ViewBuffers& out, uint32_t chunk_count, hipStream_t stream)
{
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(kScanBlock), 0, stream, out.chunk_count, out.chunk_offset,
                       out.draw_count, chunk_count);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// K3: emit — order-stable compaction of the records of visible slots (mesh.cpp:169-173)
// ------------------------------------------------------------------------------------------------
struct EmitArgs {
    MeshMirror mesh;
    TransformMirror xf;
    ViewParams view;
    ViewBuffers out;
    uint32_t nchunks;
    uint32_t clear_chunks;  // SELF: entries of chunk_count_next to clear (a larger pool may have used it last)
};

// The record of visible mirror entry i at output position `rank` (mesh.cpp:169-173). Visible entries passed every
// filter in K1: only the transform entry and its model are needed here.
__device__ __forceinline__ void write_record(const EmitArgs& args, uint32_t i, size_t rank)
{
    // every gather here is a sparse 64-byte fetch for a few useful bytes: the flag byte is only read when the pool has
    // chains at all
    const bool chains = args.xf.max_depth != 0;  // uniform
    uint32_t slot = i;
    XfRecord rec = {};
    if (args.mesh.mapping == kMapGeneral) {  // uniform
        slot = args.mesh.link[i] & kSlotMask;
        rec = gather_xf(args.xf, slot, chains);
    } else {
        rec = gather_xf(args.xf, i, chains);  // same speculation as K1: entry i beside (or instead of) the link word
        if (args.mesh.mapping == kMapSpeculate) {
            slot = args.mesh.link[i] & kSlotMask;
            if (slot != i)
                rec = gather_xf(args.xf, slot, chains);
        }
    }
    const Mat34 world = chain_model(args.xf, local_model(rec), slot, rec.flags);
    const Mat34 m = translated(world, args.view.cam[0], args.view.cam[1], args.view.cam[2]);
    args.out.visible_idx[rank] = args.mesh.orig ? args.mesh.orig[i] : i;  // pool slot: componentOffset = slot * componentSize  mesh.cpp:170
    float4* bm = reinterpret_cast<float4*>(args.out.baked_model + rank * 12);
    bm[0] = make_float4(m.c0x, m.c0y, m.c0z, m.c1x);
    bm[1] = make_float4(m.c1y, m.c1z, m.c2x, m.c2y);
    bm[2] = make_float4(m.c2z, m.c3x, m.c3y, m.c3z);
    const float tx = m.c3x + args.view.cam_offset[0];
    const float ty = m.c3y + args.view.cam_offset[1];
    const float tz = m.c3z + args.view.cam_offset[2];
    args.out.distance_sq[rank] = args.view.distance_2d ? m.c3z + 1.0f : fmaf(tz, tz, fmaf(ty, ty, tx * tx));
}

// position of the k-th (0-based) set bit of `word`
__device__ __forceinline__ uint32_t select_bit(unsigned long long word, uint32_t k)
{
    uint32_t pos = 0;
#pragma unroll
    for (uint32_t width = 32; width >= 1; width >>= 1) {
        const uint32_t c = (uint32_t)__popcll(word & ((1ull << width) - 1ull));
        if (k >= c) {
            k -= c;
            pos += width;
            word >>= width;
        }
    }
    return pos;
}

constexpr uint32_t kEmitParts = 4;  // workgroups per 4096-slot chunk: 1024 slots = 16 ballot words each

// Four workgroups per 4096-slot chunk, each owning 16 of its 64 ballot words. Every workgroup prefix-sums the
// chunk's 64 words (512 B, L2), then lane r takes the r-th visible slot of its quarter (binary search over the
// word prefix + select of the k-th set bit), so the model recompute and the 56-byte record store run on dense
// waves and only the visible fraction costs instructions. Output rank = chunk base + r: ascending slot order,
// whatever order the workgroups run in.
// SELF: no scan launch in front — while wave 0 prefixes the ballot words, waves 1-3 sum the chunk totals below this
// chunk (a few KB of L2 reads) to get its base; workgroup 0 also writes the grand total and clears the OTHER totals
// buffer for the next frame's cull (the two buffers alternate, so nobody is still reading the one being cleared).
template <bool SELF>
__global__ __launch_bounds__(256) void emit_kernel(const EmitArgs args)
{
    __shared__ unsigned long long words[64];
    __shared__ uint32_t prefix[65];
    __shared__ uint32_t below[4];
    const uint32_t chunk = blockIdx.x / kEmitParts, part = blockIdx.x % kEmitParts;
    const uint32_t first_word = chunk * 64;
    const uint32_t total_words = ((args.mesh.count + kCullBlock - 1) / kCullBlock) * (kCullBlock / 64);
    if (threadIdx.x < 64) {
        const uint32_t w = first_word + threadIdx.x;
        const unsigned long long word = w < total_words ? args.out.mask[w] : 0ull;
        uint32_t incl = (uint32_t)__popcll(word);
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (threadIdx.x >= d)
                incl += up;
        }
        words[threadIdx.x] = word;
        prefix[threadIdx.x + 1] = incl;
        if (threadIdx.x == 0)
            prefix[0] = 0;
    } else if (SELF) {
        const uint32_t upto = blockIdx.x == 0 ? args.nchunks : chunk;  // workgroup 0: the grand total
        uint32_t sum = 0;
        for (uint32_t c = threadIdx.x - 64; c < upto; c += 192)
            sum += args.out.chunk_count[c];
#pragma unroll
        for (uint32_t d = 32; d >= 1; d >>= 1)
            sum += __shfl_xor(sum, d, 64);
        if ((threadIdx.x & 63u) == 0)
            below[threadIdx.x >> 6] = sum;
    }
    __syncthreads();
    uint32_t base;
    if (SELF) {
        base = below[1] + below[2] + below[3];
        if (blockIdx.x == 0) {
            if (threadIdx.x == 0)
                *args.out.draw_count = base;
            for (uint32_t c = threadIdx.x; c < args.clear_chunks; c += 256)
                args.out.chunk_count_next[c] = 0;
            base = 0;  // chunk 0 starts the list
        }
    } else {
        base = args.out.chunk_offset[chunk];
    }
    const uint32_t wlo = part * (64 / kEmitParts), whi = wlo + 64 / kEmitParts;
    const uint32_t total = prefix[whi];
    for (uint32_t r = prefix[wlo] + threadIdx.x; r < total; r += 256) {
        uint32_t lo = wlo, hi = whi;  // word w in [wlo, whi) with prefix[w] <= r < prefix[w + 1]
#pragma unroll
        for (int step = 0; step < 4; step++) {
            const uint32_t mid = (lo + hi) >> 1;
            if (prefix[mid] <= r)
                lo = mid;
            else
                hi = mid;
        }
        const uint32_t pos = select_bit(words[lo], r - prefix[lo]);
        const uint32_t i = (first_word + lo) * 64 + pos;
        write_record(args, i, (size_t)base + r);
    }
}

hipError_t launch_emit(const MeshMirror& mesh, const TransformMirror& xf, const ViewParams& vp, const ViewBuffers& out,
                       hipStream_t stream, bool self_prefix, uint32_t clear_chunks)
{
    if (mesh.count == 0)
        return hipSuccess;
    EmitArgs a;
    a.mesh = mesh;
    a.xf = xf;
    a.view = vp;
    a.out = out;
    a.nchunks = (mesh.count + kEmitChunk - 1) / kEmitChunk;
    a.clear_chunks = clear_chunks;
    if (self_prefix)
        hipLaunchKernelGGL(emit_kernel<true>, dim3(a.nchunks * kEmitParts), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(emit_kernel<false>, dim3(a.nchunks * kEmitParts), dim3(256), 0, stream, a);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void pack_active_kernel(const uint8_t* __restrict__ flags, uint32_t count,
                                                          unsigned long long* __restrict__ bits)
{
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    const bool active = e < count && (flags[e] & kXfActive);
    const unsigned long long word = __ballot(active);
    if ((threadIdx.x & 63u) == 0 && (e & ~63u) < count)
        bits[e >> 6] = word;
}

hipError_t launch_pack_active(const uint8_t* flags, uint32_t count, unsigned long long* bits, hipStream_t stream)
{
    if (count == 0)
        return hipSuccess;
    hipLaunchKernelGGL(pack_active_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, flags, count, bits);
    return hipGetLastError();
}

// Dirty TransformComponents shipped as raw AoS bytes (slots [first, first + count) of the caller's pool, copied as they
// lie): the AoS -> SoA gather that the host otherwise does runs here, at HBM speed. Parent links are not touched
// (this path serves GV_DIRTY_TRANSFORM; link changes go through the host, which also validates depth and cycles).
__global__ __launch_bounds__(256) void aos_transforms_kernel(const uint8_t* __restrict__ raw, AosTransformLayout L,
                                                             uint32_t first, uint32_t count,
                                                             const uint32_t* __restrict__ xinv, float4* __restrict__ a,
                                                             float4* __restrict__ b, float2* __restrict__ c,
                                                             uint8_t* __restrict__ flags)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count)
        return;
    const uint8_t* t = raw + (size_t)k * L.stride;
    float pos[3], scl[3], rot[4];
    uint32_t entity;
    memcpy(pos, t + L.position, 12);
    memcpy(scl, t + L.scale, 12);
    memcpy(rot, t + L.rotation, 16);
    memcpy(&entity, t + L.entity, 4);
    uint8_t f = 0;
    if (entity)
        f |= kXfLive;
    if (t[L.self_active] && t[L.ancestors_active])
        f |= kXfActive;
    if (t[L.model_with_ancestors])
        f |= kXfWithAncestors;
    const uint32_t s = first + k;
    const uint32_t j = xinv ? xinv[s] : s;
    a[j] = make_float4(pos[0], pos[1], pos[2], scl[0]);
    b[j] = make_float4(rot[0], rot[1], rot[2], rot[3]);
    c[j] = make_float2(scl[1], scl[2]);
    flags[j] = f;
}

hipError_t launch_aos_transforms(const uint8_t* raw, const AosTransformLayout& layout, uint32_t first, uint32_t count,
                                 const uint32_t* xinv, float4* a, float4* b, float2* c, uint8_t* flags, hipStream_t stream)
{
    if (count == 0)
        return hipSuccess;
    hipLaunchKernelGGL(aos_transforms_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, raw, layout, first, count, xinv,
                       a, b, c, flags);
    return hipGetLastError();
}

template <typename T>
__global__ __launch_bounds__(256) void scatter_kernel(const uint32_t* __restrict__ idx, uint32_t count,
                                                      const T* __restrict__ src, T* __restrict__ dst)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x)
        dst[idx[k]] = src[k];
}

hipError_t launch_scatter(const uint32_t* idx, uint32_t count, const void* src, void* dst, uint32_t elem_bytes,
                          hipStream_t stream)
{
    if (count == 0)
        return hipSuccess;
    const dim3 grid(min((count + 255u) / 256u, 4096u)), block(256);
    switch (elem_bytes) {
    case 1: hipLaunchKernelGGL(scatter_kernel<uint8_t>, grid, block, 0, stream, idx, count, (const uint8_t*)src, (uint8_t*)dst); break;
    case 4: hipLaunchKernelGGL(scatter_kernel<uint32_t>, grid, block, 0, stream, idx, count, (const uint32_t*)src, (uint32_t*)dst); break;
    case 8: hipLaunchKernelGGL(scatter_kernel<float2>, grid, block, 0, stream, idx, count, (const float2*)src, (float2*)dst); break;
    case 16: hipLaunchKernelGGL(scatter_kernel<float4>, grid, block, 0, stream, idx, count, (const float4*)src, (float4*)dst); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void gather_world_kernel(const float4* __restrict__ world, const uint32_t* __restrict__ xinv,
                                                           uint32_t first, uint32_t count, float4* __restrict__ out)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < count * 3; k += gridDim.x * blockDim.x) {
        const uint32_t s = k / 3, part = k - s * 3;
        out[k] = world[(size_t)xinv[first + s] * 3 + part];
    }
}

hipError_t launch_gather_world(const float4* world, const uint32_t* xinv, uint32_t first, uint32_t count, float4* out,
                               hipStream_t stream)
{
    if (count == 0)
        return hipSuccess;
    hipLaunchKernelGGL(gather_world_kernel, dim3(min((count * 3 + 255u) / 256u, 4096u)), dim3(256), 0, stream, world, xinv,
                       first, count, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void copy_idx_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ count,
                                                       uint32_t* __restrict__ dst, uint32_t capacity, uint32_t base)
{
    const uint32_t n = min(*count, capacity);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dst[i] = src[i] + base;
}

hipError_t launch_copy_idx(const uint32_t* src, const uint32_t* count, uint32_t* dst, uint32_t capacity, uint32_t base,
                           hipStream_t stream)
{
    hipLaunchKernelGGL(copy_idx_kernel, dim3(2048), dim3(256), 0, stream, src, count, dst, capacity, base);
    return hipGetLastError();
}

// exchange shard: [draw_count, idx + base ...]; the header is the true count even when it exceeds `capacity`
__global__ __launch_bounds__(256) void copy_shard_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ count,
                                                         uint32_t* __restrict__ dst, uint32_t capacity, uint32_t base)
{
    const uint32_t total = *count, n = min(total, capacity);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        dst[0] = total;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dst[1 + i] = src[i] + base;
}

hipError_t launch_copy_shard(const uint32_t* src, const uint32_t* count, uint32_t* dst, uint32_t capacity, uint32_t base,
                             hipStream_t stream)
{
    const uint32_t blocks = std::max(1u, std::min(2048u, (capacity + 255u) / 256u));
    hipLaunchKernelGGL(copy_shard_kernel, dim3(blocks), dim3(256), 0, stream, src, count, dst, capacity, base);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// sortMeshes (mesh.cpp:265-328): order the compact records by distanceSq — ascending for unsorted buffers
// (front to back, operator< at render/mesh.hpp:196), descending for the sorted / translucent ones (:204).
// Stable LSD radix sort on a 32-bit order-preserving key, 8 bits per pass; ties keep ascending slot order
// (std::sort in the reference is unstable, so any tie order is within its contract).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kSortTile = 4096;  // keys per workgroup per pass (16 sub-tiles of 256)

__global__ __launch_bounds__(256) void sort_keys_kernel(const float* __restrict__ dist, const uint32_t* __restrict__ count,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                        uint32_t descending)
{
    const uint32_t n = *count;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const uint32_t u = __float_as_uint(dist[j]);
        uint32_t k = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);  // ascending float order == ascending key order
        if (descending)
            k = ~k;
        keys[j] = k;
        vals[j] = j;
    }
}

// Launch geometry is sized for the pool capacity (known on the host); the record count lives on the device, so
// every kernel derives the live tile count from it and surplus workgroups exit at once.
// per-workgroup digit histogram, bin-major with a fixed stride: hist[bin * stride + tile]
__global__ __launch_bounds__(256) void radix_hist_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ count,
                                                         uint32_t* __restrict__ hist, uint32_t shift, uint32_t stride)
{
    const uint32_t n = *count;
    const uint32_t lo = blockIdx.x * kSortTile;
    if (lo >= n)
        return;
    __shared__ uint32_t bins[256];
    bins[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t hi = min(lo + kSortTile, n);
    for (uint32_t j = lo + threadIdx.x; j < hi; j += 256)
        atomicAdd(&bins[(keys[j] >> shift) & 255u], 1u);
    __syncthreads();
    hist[threadIdx.x * stride + blockIdx.x] = bins[threadIdx.x];
}

// one workgroup per digit: exclusive scan of that digit's per-tile counts in place + the digit's total
__global__ __launch_bounds__(256) void radix_bin_scan_kernel(uint32_t* __restrict__ hist, uint32_t* __restrict__ bin_total,
                                                             const uint32_t* __restrict__ count, uint32_t stride)
{
    __shared__ uint32_t wave_sum[4];
    const uint32_t tiles = (*count + kSortTile - 1) / kSortTile;
    uint32_t* row = hist + (size_t)blockIdx.x * stride;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < tiles; base += 256) {
        const uint32_t idx = base + threadIdx.x;
        const uint32_t v = idx < tiles ? row[idx] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d)
                incl += up;
        }
        if (lane == 63)
            wave_sum[wave] = incl;
        __syncthreads();
        uint32_t wave_prefix = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; w++) {
            wave_prefix += w < wave ? wave_sum[w] : 0u;
            all += wave_sum[w];
        }
        if (idx < tiles)
            row[idx] = carry + wave_prefix + incl - v;
        carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0)
        bin_total[blockIdx.x] = carry;
}

// stable scatter: sub-tiles of 256 keys in order; rank inside a wave by digit matching (8 ballots), across waves
// and sub-tiles through LDS counters. Digit d of this tile starts at (sum of lower digits' totals) + hist[d][tile].
__global__ __launch_bounds__(256) void radix_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                            const uint32_t* __restrict__ vals_in,
                                                            uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                            const uint32_t* __restrict__ count,
                                                            const uint32_t* __restrict__ hist,
                                                            const uint32_t* __restrict__ bin_total, uint32_t shift,
                                                            uint32_t stride)
{
    const uint32_t n = *count;
    const uint32_t lo = blockIdx.x * kSortTile;
    if (lo >= n)
        return;
    __shared__ uint32_t base[256];        // next output position of each digit for this workgroup
    __shared__ uint32_t wcount[4][256];   // per-wave digit counts of the current sub-tile
    __shared__ uint32_t wave_sum[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    {  // exclusive scan of the 256 digit totals, one digit per lane
        const uint32_t v = bin_total[threadIdx.x];
        uint32_t incl = v;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d)
                incl += up;
        }
        if (lane == 63)
            wave_sum[wave] = incl;
        __syncthreads();
        uint32_t wave_prefix = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; w++)
            wave_prefix += w < wave ? wave_sum[w] : 0u;
        base[threadIdx.x] = wave_prefix + incl - v + hist[threadIdx.x * stride + blockIdx.x];
    }
#pragma unroll
    for (uint32_t w = 0; w < 4; w++)
        wcount[w][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t hi = min(lo + kSortTile, n);
    for (uint32_t t = lo; t < hi; t += 256) {  // uniform trip count
        const uint32_t j = t + threadIdx.x;
        const bool valid = j < hi;
        const uint32_t key = valid ? keys_in[j] : 0u, val = valid ? vals_in[j] : 0u;
        const uint32_t d = (key >> shift) & 255u;
        unsigned long long peer = __ballot(valid);
#pragma unroll
        for (uint32_t b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peer &= bit ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peer & ((1ull << lane) - 1ull));
        const bool leader = valid && rank == 0;
        if (leader)
            wcount[wave][d] = (uint32_t)__popcll(peer);
        __syncthreads();
        if (valid) {
            uint32_t pos = base[d] + rank;
#pragma unroll
            for (uint32_t w = 0; w < 4; w++)
                pos += w < wave ? wcount[w][d] : 0u;
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __syncthreads();
        if (leader) {
            atomicAdd(&base[d], (uint32_t)__popcll(peer));
            wcount[wave][d] = 0;
        }
        __syncthreads();
    }
}

// permute the 56-byte records by the sorted positions
__global__ __launch_bounds__(256) void sort_gather_kernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ count,
                                                          const uint32_t* __restrict__ idx_in, const float* __restrict__ model_in,
                                                          const float* __restrict__ dist_in, uint32_t* __restrict__ idx_out,
                                                          float* __restrict__ model_out, float* __restrict__ dist_out)
{
    const uint32_t n = *count;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const uint32_t src = order[j];
        idx_out[j] = idx_in[src];
        dist_out[j] = dist_in[src];
        const float4* sm = reinterpret_cast<const float4*>(model_in + (size_t)src * 12);
        float4* dm = reinterpret_cast<float4*>(model_out + (size_t)j * 12);
        const float4 m0 = sm[0], m1 = sm[1], m2 = sm[2];
        dm[0] = m0;
        dm[1] = m1;
        dm[2] = m2;
    }
}

hipError_t launch_sort(const SortBuffers& b, uint32_t capacity, bool descending, hipStream_t stream)
{
    if (capacity == 0)
        return hipSuccess;
    const uint32_t stride = (capacity + kSortTile - 1) / kSortTile;  // tiles at full capacity = hist row stride
    const uint32_t wide = min((capacity + 255u) / 256u, 4096u);
    hipLaunchKernelGGL(sort_keys_kernel, dim3(wide), dim3(256), 0, stream, b.dist_in, b.count, b.keys[0], b.vals[0],
                       descending ? 1u : 0u);
    for (uint32_t pass = 0; pass < 4; pass++) {
        const uint32_t src = pass & 1u, dst = src ^ 1u;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(stride), dim3(256), 0, stream, b.keys[src], b.count, b.hist, pass * 8, stride);
        hipLaunchKernelGGL(radix_bin_scan_kernel, dim3(256), dim3(256), 0, stream, b.hist, b.bin_total, b.count, stride);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(stride), dim3(256), 0, stream, b.keys[src], b.vals[src], b.keys[dst],
                           b.vals[dst], b.count, b.hist, b.bin_total, pass * 8, stride);
    }
    // 4 passes: the sorted order ends in vals[0]
    hipLaunchKernelGGL(sort_gather_kernel, dim3(wide), dim3(256), 0, stream, b.vals[0], b.count, b.idx_in, b.model_in, b.dist_in,
                       b.idx_out, b.model_out, b.dist_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// world-matrix sweep (camera = 0), VALU form: one lane per transform slot
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sweep_valu_kernel(const TransformMirror xf, float4* __restrict__ world)
{
    const uint32_t lb = blockIdx.x;
    const uint32_t s = lb * 256 + threadIdx.x;
    if (s >= xf.count)
        return;
    const XfRecord r = stream_xf(xf, s);
    float4 w0 = make_float4(0, 0, 0, 0), w1 = w0, w2 = w0;
    if (r.flags & kXfLive) {
        const Mat34 m = chain_model(xf, local_model(r), s, r.flags);
        w0 = make_float4(m.c0x, m.c0y, m.c0z, m.c1x);
        w1 = make_float4(m.c1y, m.c1z, m.c2x, m.c2y);
        w2 = make_float4(m.c2z, m.c3x, m.c3y, m.c3z);
    }
    stream_store(&world[(size_t)s * 3 + 0], w0);
    stream_store(&world[(size_t)s * 3 + 1], w1);
    stream_store(&world[(size_t)s * 3 + 2], w2);
}

hipError_t launch_sweep_valu(const TransformMirror& xf, float4* world, hipStream_t stream)
{
    if (xf.count == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sweep_valu_kernel, dim3((xf.count + 255) / 256), dim3(256), 0, stream, xf, world);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// world-matrix sweep, MFMA form.
// v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 blocks per wave; block = lane >> 2; the A operand of lane
// (block, i) is A[i][k], the B operand of lane (block, j) is B[k][j], D register r of lane (block, j) is D[r][j].
// Four issues k = 0..3 into one accumulator give parentModel * model with the per-element order
// fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, +0)))) — the canonical chain, bit-identical to the VALU form.
//
// Layout: the memory side is one lane per transform slot (coalesced 16/12-byte streams, one calcModel per
// slot, 64 slots per wave); the matrix side is 4 lanes per slot. Local models cross from one to the other
// through a per-wave LDS tile (13-word row pitch: conflict-free). A wave's 64 slots are multiplied in 4
// rounds of 16; lane (e, j) keeps column j of the running product of slot 16r+e in 4 registers per round,
// which is also its B operand for the next ancestor — no movement between chain steps.
// ------------------------------------------------------------------------------------------------
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr uint32_t kPitch = 13;

__device__ __forceinline__ void lds_put_model(float* row, const Mat34& m)
{
    row[0] = m.c0x; row[1] = m.c0y; row[2] = m.c0z;
    row[3] = m.c1x; row[4] = m.c1y; row[5] = m.c1z;
    row[6] = m.c2x; row[7] = m.c2y; row[8] = m.c2z;
    row[9] = m.c3x; row[10] = m.c3y; row[11] = m.c3z;
}

// Each wave owns its LDS tile, so the matrix/memory-side hand-over only needs wave-level ordering: LDS operations of
// one wave execute in issue order; the fences keep the compiler from moving them across the hand-over.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256) void sweep_mfma_kernel(const TransformMirror xf, float* __restrict__ world)
{
    __shared__ float tile[4][64 * kPitch];  // one tile per wave
    __shared__ uint32_t has_parent[4][64];
    const uint32_t lb = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t q = lane & 3u, e = lane >> 2;  // matrix side: column/row q of slot 16r + e
    const uint32_t s = lb * 256 + threadIdx.x;    // memory side: this lane's slot
    float* my_tile = tile[wave];
    uint32_t flags = 0;
    Mat34 m = {};
    const bool in_range = s < xf.count;
    if (in_range) {
        const XfRecord r = stream_xf(xf, s);
        flags = r.flags;
        m = local_model(r);
    }
    const bool live = in_range && (flags & kXfLive);
    lds_put_model(my_tile + lane * kPitch, m);
    wave_lds_sync();
    float x[4][4];  // [round][row]: column q of the product of slot 16r + e (row 3 = bottom-row element)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float* row = my_tile + (16 * r + e) * kPitch + 3 * q;
        x[r][0] = row[0];
        x[r][1] = row[1];
        x[r][2] = row[2];
        x[r][3] = q == 3 ? 1.0f : 0.0f;
    }
    uint32_t p = (live && xf.max_depth != 0 && (flags & kXfWithAncestors)) ? xf.parent[s] : kSlotNone;
    for (uint32_t d = 0; d < xf.max_depth; d++) {
        const bool has = p != kSlotNone;
        if (!__any(has))
            break;  // wave-uniform exit (MFMA ignores EXEC: every lane of the wave takes every step)
        uint32_t next = kSlotNone;
        Mat34 pm = {};
        if (has) {
            pm = local_model(load_xf(xf, p));
            next = xf.parent[p];
        }
        lds_put_model(my_tile + lane * kPitch, pm);
        has_parent[wave][lane] = has ? 1u : 0u;
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 4; r++) {
            // row q of the parent's local model of slot 16r + e: A[q][k] = column k, row q
            const float* prow = my_tile + (16 * r + e) * kPitch;
            const float a0 = q < 3 ? prow[q] : 0.0f;
            const float a1 = q < 3 ? prow[3 + q] : 0.0f;
            const float a2 = q < 3 ? prow[6 + q] : 0.0f;
            const float a3 = q < 3 ? prow[9 + q] : 1.0f;
            const bool step = has_parent[wave][16 * r + e] != 0;
            f32x4_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, x[r][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, x[r][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, x[r][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a3, x[r][3], acc, 0, 0, 0);
            // slots whose chain has ended keep their product untouched (bit-exact, incl. -0)
            x[r][0] = step ? acc[0] : x[r][0];
            x[r][1] = step ? acc[1] : x[r][1];
            x[r][2] = step ? acc[2] : x[r][2];
            // acc[3] (the bottom-row element) is not taken: models are affine and x[r][3] stays the constant 0 / 1,
            // as in every other implementation (it only differs from acc[3] when the operands are non-finite)
        }
        p = next;
    }
    // liveness of slot 16r + e on the matrix side
    has_parent[wave][lane] = live ? 1u : 0u;
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t slot = lb * 256 + wave * 64 + 16 * r + e;
        if (slot < xf.count) {
            const bool ok = has_parent[wave][16 * r + e] != 0;
            // float4x3 order: column q's xyz at 12 floats per slot -> 768 contiguous bytes per round
            float* dst = world + (size_t)slot * 12 + q * 3;
            stream_store(dst + 0, ok ? x[r][0] : 0.0f);
            stream_store(dst + 1, ok ? x[r][1] : 0.0f);
            stream_store(dst + 2, ok ? x[r][2] : 0.0f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA sweep + cull in one pass (cfg4: "hierarchy recomputed each frame + cull"). For a mesh pool that is exactly
// paired with the transform pool the world matrix of slot s IS the model of mesh entry s before the camera translate
// (transform.hpp:211-213), so the sweep's product is handed back to the memory side through the wave's LDS tile,
// stored (48 B, nontemporal) and culled from registers: the TRS streams are read once per frame instead of twice and
// the cull's own chain walk disappears. Same arithmetic as sweep_mfma_kernel followed by cull_kernel, bit for bit.
// ------------------------------------------------------------------------------------------------
struct SweepCullArgs {
    CullArgs cull;
    float4* world;
};

template <bool HIZ>
__global__ __launch_bounds__(256) void sweep_cull_mfma_kernel(const SweepCullArgs args)
{
    __shared__ float tile[4][64 * kPitch];  // one tile per wave
    __shared__ uint32_t has_parent[4][64];
    __shared__ uint32_t wave_count[4];
    const TransformMirror& xf = args.cull.xf;
    const MeshMirror& mesh = args.cull.mesh;
    const uint32_t lb = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t q = lane & 3u, e = lane >> 2;  // matrix side: column/row q of slot 16r + e
    const uint32_t s = lb * 256 + threadIdx.x;    // memory side: this lane's slot = its mesh entry
    float* my_tile = tile[wave];
    uint32_t flags = 0;
    Mat34 m = {};
    const bool in_range = s < xf.count, has_mesh = s < mesh.count;
    float4 ma = make_float4(0, 0, 0, 0);
    float2 mb = make_float2(0, 0);
    if (has_mesh) {  // issued beside the transform streams
        ma = stream_load(&mesh.a[s]);
        mb = stream_load(&mesh.b[s]);
    }
    if (in_range) {
        const XfRecord r = stream_xf(xf, s);
        flags = r.flags;
        m = local_model(r);
    }
    const bool live = in_range && (flags & kXfLive);
    lds_put_model(my_tile + lane * kPitch, m);
    wave_lds_sync();
    float x[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float* row = my_tile + (16 * r + e) * kPitch + 3 * q;
        x[r][0] = row[0];
        x[r][1] = row[1];
        x[r][2] = row[2];
        x[r][3] = q == 3 ? 1.0f : 0.0f;
    }
    uint32_t p = (live && xf.max_depth != 0 && (flags & kXfWithAncestors)) ? xf.parent[s] : kSlotNone;
    for (uint32_t d = 0; d < xf.max_depth; d++) {
        const bool has = p != kSlotNone;
        if (!__any(has))
            break;
        uint32_t next = kSlotNone;
        Mat34 pm = {};
        if (has) {
            pm = local_model(load_xf(xf, p));
            next = xf.parent[p];
        }
        lds_put_model(my_tile + lane * kPitch, pm);
        has_parent[wave][lane] = has ? 1u : 0u;
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float* prow = my_tile + (16 * r + e) * kPitch;
            const float a0 = q < 3 ? prow[q] : 0.0f;
            const float a1 = q < 3 ? prow[3 + q] : 0.0f;
            const float a2 = q < 3 ? prow[6 + q] : 0.0f;
            const float a3 = q < 3 ? prow[9 + q] : 1.0f;
            const bool step = has_parent[wave][16 * r + e] != 0;
            f32x4_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, x[r][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, x[r][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, x[r][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a3, x[r][3], acc, 0, 0, 0);
            x[r][0] = step ? acc[0] : x[r][0];
            x[r][1] = step ? acc[1] : x[r][1];
            x[r][2] = step ? acc[2] : x[r][2];
        }
        p = next;
        wave_lds_sync();  // the tile is rewritten by the next step / the hand-back below
    }
    // hand the products back: lane (e, q) holds column q of slot 16r + e
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float* row = my_tile + (16 * r + e) * kPitch + 3 * q;
        row[0] = x[r][0];
        row[1] = x[r][1];
        row[2] = x[r][2];
    }
    wave_lds_sync();
    Mat34 world = {};
    {
        const float* row = my_tile + lane * kPitch;
        world.c0x = row[0]; world.c0y = row[1]; world.c0z = row[2];
        world.c1x = row[3]; world.c1y = row[4]; world.c1z = row[5];
        world.c2x = row[6]; world.c2y = row[7]; world.c2z = row[8];
        world.c3x = row[9]; world.c3y = row[10]; world.c3z = row[11];
    }
    if (in_range) {
        float4 w0 = make_float4(0, 0, 0, 0), w1 = w0, w2 = w0;
        if (live) {
            w0 = make_float4(world.c0x, world.c0y, world.c0z, world.c1x);
            w1 = make_float4(world.c1y, world.c1z, world.c2x, world.c2y);
            w2 = make_float4(world.c2z, world.c3x, world.c3y, world.c3z);
        }
        stream_store(&args.world[(size_t)s * 3 + 0], w0);
        stream_store(&args.world[(size_t)s * 3 + 1], w1);
        stream_store(&args.world[(size_t)s * 3 + 2], w2);
    }
    // ---- cull_kernel's tail on the model in registers (mesh.cpp:140-175) ----
    bool visible = false;
    if (has_mesh) {
        const float mnx = ma.x, mny = ma.y, mnz = ma.z, mxx = ma.w, mxy = mb.x, mxz = mb.y;
        const bool empty = (mxx - mnx <= 0.0f) && (mxy - mny <= 0.0f) && (mxz - mnz <= 0.0f);
        if (in_range && !empty && (flags & kXfActive)) {
            const Mat34 model = translated(world, args.cull.view.cam[0], args.cull.view.cam[1], args.cull.view.cam[2]);
            Corners c;
            aabb_corners(model, mnx, mny, mnz, mxx, mxy, mxz, c);
            visible = !behind_frustum(c, args.cull.view.planes, args.cull.view.plane_count);
            if (HIZ && visible)
                visible = !hiz_occluded(args.cull.hiz, args.cull.view.vp, c);
        }
        if (args.cull.view.write_is_visible)
            args.cull.out.is_visible[s] = visible ? 1 : 0;
    }
    const unsigned long long word = __ballot(visible);
    const bool mesh_block = lb < args.cull.nblocks;  // workgroups past the mesh range only sweep
    if (lane == 0) {
        if (mesh_block)
            args.cull.out.mask[(size_t)lb * 4 + wave] = word;
        wave_count[wave] = (uint32_t)__popcll(word);
    }
    __syncthreads();
    if (threadIdx.x == 0 && mesh_block) {
        const uint32_t total = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        if (total)
            atomicAdd(&args.cull.out.chunk_count[lb / (kEmitChunk / kCullBlock)], total);
    }
}

// The same pass with the v_fma_f32 chain (one lane per slot end to end, no LDS hand-over).
template <bool HIZ>
__global__ __launch_bounds__(256) void sweep_cull_valu_kernel(const SweepCullArgs args)
{
    __shared__ uint32_t wave_count[4];
    const TransformMirror& xf = args.cull.xf;
    const MeshMirror& mesh = args.cull.mesh;
    const uint32_t lb = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t s = lb * 256 + threadIdx.x;
    const bool in_range = s < xf.count, has_mesh = s < mesh.count;
    float4 ma = make_float4(0, 0, 0, 0);
    float2 mb = make_float2(0, 0);
    if (has_mesh) {
        ma = stream_load(&mesh.a[s]);
        mb = stream_load(&mesh.b[s]);
    }
    uint32_t flags = 0;
    Mat34 world = {};
    if (in_range) {
        const XfRecord r = stream_xf(xf, s);
        flags = r.flags;
        float4 w0 = make_float4(0, 0, 0, 0), w1 = w0, w2 = w0;
        if (flags & kXfLive) {
            world = chain_model(xf, local_model(r), s, flags);
            w0 = make_float4(world.c0x, world.c0y, world.c0z, world.c1x);
            w1 = make_float4(world.c1y, world.c1z, world.c2x, world.c2y);
            w2 = make_float4(world.c2z, world.c3x, world.c3y, world.c3z);
        }
        stream_store(&args.world[(size_t)s * 3 + 0], w0);
        stream_store(&args.world[(size_t)s * 3 + 1], w1);
        stream_store(&args.world[(size_t)s * 3 + 2], w2);
    }
    bool visible = false;
    if (has_mesh) {
        const float mnx = ma.x, mny = ma.y, mnz = ma.z, mxx = ma.w, mxy = mb.x, mxz = mb.y;
        const bool empty = (mxx - mnx <= 0.0f) && (mxy - mny <= 0.0f) && (mxz - mnz <= 0.0f);
        if (in_range && !empty && (flags & kXfActive)) {
            const Mat34 model = translated(world, args.cull.view.cam[0], args.cull.view.cam[1], args.cull.view.cam[2]);
            Corners c;
            aabb_corners(model, mnx, mny, mnz, mxx, mxy, mxz, c);
            visible = !behind_frustum(c, args.cull.view.planes, args.cull.view.plane_count);
            if (HIZ && visible)
                visible = !hiz_occluded(args.cull.hiz, args.cull.view.vp, c);
        }
        if (args.cull.view.write_is_visible)
            args.cull.out.is_visible[s] = visible ? 1 : 0;
    }
    const unsigned long long word = __ballot(visible);
    const bool mesh_block = lb < args.cull.nblocks;
    if (lane == 0) {
        if (mesh_block)
            args.cull.out.mask[(size_t)lb * 4 + wave] = word;
        wave_count[wave] = (uint32_t)__popcll(word);
    }
    __syncthreads();
    if (threadIdx.x == 0 && mesh_block) {
        const uint32_t total = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        if (total)
            atomicAdd(&args.cull.out.chunk_count[lb / (kEmitChunk / kCullBlock)], total);
    }
}

hipError_t launch_sweep_cull(const MeshMirror& mesh, const TransformMirror& xf, const HizDevice& hiz,
                             const ViewParams& vp, const ViewBuffers& out, float4* world, bool mfma, hipStream_t stream)
{
    if (xf.count == 0)
        return hipSuccess;
    SweepCullArgs a;
    a.cull.mesh = mesh;
    a.cull.xf = xf;
    a.cull.hiz = hiz;
    a.cull.view = vp;
    a.cull.out = out;
    a.cull.nblocks = (mesh.count + kCullBlock - 1) / kCullBlock;
    a.world = world;
    const uint32_t blocks = (std::max(xf.count, mesh.count) + 255) / 256;
    const bool hz = vp.use_hiz && hiz.mip_count;
    if (mfma && hz)
        hipLaunchKernelGGL(sweep_cull_mfma_kernel<true>, dim3(blocks), dim3(256), 0, stream, a);
    else if (mfma)
        hipLaunchKernelGGL(sweep_cull_mfma_kernel<false>, dim3(blocks), dim3(256), 0, stream, a);
    else if (hz)
        hipLaunchKernelGGL(sweep_cull_valu_kernel<true>, dim3(blocks), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(sweep_cull_valu_kernel<false>, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_sweep_mfma(const TransformMirror& xf, float4* world, hipStream_t stream)
{
    if (xf.count == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sweep_mfma_kernel, dim3((xf.count + 255) / 256), dim3(256), 0, stream, xf,
                       reinterpret_cast<float*>(world));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Hi-Z pyramid
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 hiz_src(const float* d, const float2* p, uint32_t sw, uint32_t x, uint32_t y)
{
    if (d) {  // HIZ_VARIANT_FIRST: (d, d)  hiz.frag:57-60
        const float v = d[(size_t)y * sw + x];
        return make_float2(v, v);
    }
    return p[(size_t)y * sw + x];
}
__device__ __forceinline__ void hiz_acc(float2& mm, float2 t)
{
    mm.x = t.x < mm.x ? t.x : mm.x;  // MIN_DEPTH  depth.gsl:30-31
    mm.y = t.y > mm.y ? t.y : mm.y;  // MAX_DEPTH  depth.gsl:32-33
}

// One destination texel per lane; any size (hiz.frag:27-56 with the odd-size branches).
__global__ __launch_bounds__(256) void hiz_level_kernel(const float* __restrict__ src_depth,
                                                        const float2* __restrict__ src_pairs,
                                                        float2* __restrict__ dst, uint32_t sw, uint32_t sh, uint32_t dw,
                                                        uint32_t dh, uint32_t rule)
{
    const uint32_t px = blockIdx.x * 64 + (threadIdx.x & 63u);
    const uint32_t py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= dw || py >= dh)
        return;
    const bool odd_x = (sw & 1u) != 0, odd_y = (sh & 1u) != 0;
    const uint32_t x0 = 2 * px, y0 = 2 * py;
    const uint32_t x1 = min(x0 + 1, sw - 1), y1 = min(y0 + 1, sh - 1);
    const uint32_t x2 = min(x0 + 2, sw - 1), y2 = min(y0 + 2, sh - 1);
    float2 mm = hiz_src(src_depth, src_pairs, sw, x0, y0);
    hiz_acc(mm, hiz_src(src_depth, src_pairs, sw, x1, y0));
    hiz_acc(mm, hiz_src(src_depth, src_pairs, sw, x0, y1));
    hiz_acc(mm, hiz_src(src_depth, src_pairs, sw, x1, y1));
    if (odd_x) {  // hiz.frag:36-41
        hiz_acc(mm, hiz_src(src_depth, src_pairs, sw, x2, y1));
        hiz_acc(mm, hiz_src(src_depth, src_pairs, sw, x2, y0));
        if (odd_y)  // hiz.frag:43-47
            hiz_acc(mm, hiz_src(src_depth, src_pairs, sw, x2, y2));
    }
    if (odd_y) {  // hiz.frag:49-55 reads gather components .y/.z = (2p.x+1, 2p.y+2), (2p.x+1, 2p.y+1)
        hiz_acc(mm, hiz_src(src_depth, src_pairs, sw, x1, y2));
        if (rule == 1u)  // GV_HIZ_RULE_CONSERVATIVE: the whole extra row
            hiz_acc(mm, hiz_src(src_depth, src_pairs, sw, x0, y2));
    }
    dst[(size_t)py * dw + px] = mm;
}

hipError_t launch_hiz_level(const float* src_depth, const float2* src_pairs, float2* dst, uint32_t sw, uint32_t sh,
                            uint32_t dw, uint32_t dh, uint32_t rule, hipStream_t stream)
{
    hipLaunchKernelGGL(hiz_level_kernel, dim3((dw + 63) / 64, (dh + 3) / 4), dim3(256), 0, stream, src_depth, src_pairs,
                       dst, sw, sh, dw, dh, rule);
    return hipGetLastError();
}

// Fused: one workgroup reduces a 64x64 source tile to 32^2, 16^2, 8^2, 4^2, 2^2 and 1 texel — six
// levels in one pass, the source read once, intermediate levels staged in LDS instead of re-read
// from HBM (the reference re-reads every mip in its own render pass, hiz.cpp:155-164).
template <bool PAIRS>
__global__ __launch_bounds__(256) void hiz_fused_kernel(const float* __restrict__ src_depth,
                                                        const float2* __restrict__ src_pairs, const HizFusedDst dst,
                                                        uint32_t sw, uint32_t sh)
{
    __shared__ float2 lds16[16][17];
    __shared__ float2 lds8[8][9];
    __shared__ float2 lds4[4][5];
    __shared__ float2 lds2[2][3];
    const uint32_t tx = threadIdx.x & 15u, ty = threadIdx.x >> 4;
    const uint32_t ox = blockIdx.x * 64, oy = blockIdx.y * 64;
    const uint32_t px = ox + 4 * tx, py = oy + 4 * ty;
    float mn[4][4], mx[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (PAIRS) {
            const float4* row = reinterpret_cast<const float4*>(src_pairs + (size_t)(py + r) * sw + px);
            const float4 lo = row[0], hi = row[1];
            mn[r][0] = lo.x; mx[r][0] = lo.y; mn[r][1] = lo.z; mx[r][1] = lo.w;
            mn[r][2] = hi.x; mx[r][2] = hi.y; mn[r][3] = hi.z; mx[r][3] = hi.w;
        } else {
            const float4 v = stream_load(reinterpret_cast<const float4*>(src_depth + (size_t)(py + r) * sw + px));
            mn[r][0] = mx[r][0] = v.x; mn[r][1] = mx[r][1] = v.y;
            mn[r][2] = mx[r][2] = v.z; mn[r][3] = mx[r][3] = v.w;
        }
    }
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
            q[a][b] = mm;
        }
    const uint32_t w1 = sw >> 1;
    if (dst.level[0]) {  // null: the level stays virtual (queries reduce the source themselves)
#pragma unroll
        for (int a = 0; a < 2; a++) {
            float4* o = reinterpret_cast<float4*>(dst.level[0] + (size_t)(oy / 2 + 2 * ty + a) * w1 + ox / 2 + 2 * tx);
            *o = make_float4(q[a][0].x, q[a][0].y, q[a][1].x, q[a][1].y);
        }
    }
    // level +2: one texel per lane
    float2 m2 = q[0][0];
    hiz_acc(m2, q[0][1]);
    hiz_acc(m2, q[1][0]);
    hiz_acc(m2, q[1][1]);
    dst.level[1][(size_t)(oy / 4 + ty) * (sw >> 2) + ox / 4 + tx] = m2;
    lds16[ty][tx] = m2;
    __syncthreads();
    if (threadIdx.x < 64) {  // level +3: 8x8
        const uint32_t x = threadIdx.x & 7u, y = threadIdx.x >> 3;
        float2 mm = lds16[2 * y][2 * x];
        hiz_acc(mm, lds16[2 * y][2 * x + 1]);
        hiz_acc(mm, lds16[2 * y + 1][2 * x]);
        hiz_acc(mm, lds16[2 * y + 1][2 * x + 1]);
        dst.level[2][(size_t)(oy / 8 + y) * (sw >> 3) + ox / 8 + x] = mm;
        lds8[y][x] = mm;
    }
    __syncthreads();
    if (threadIdx.x < 16) {  // level +4: 4x4
        const uint32_t x = threadIdx.x & 3u, y = threadIdx.x >> 2;
        float2 mm = lds8[2 * y][2 * x];
        hiz_acc(mm, lds8[2 * y][2 * x + 1]);
        hiz_acc(mm, lds8[2 * y + 1][2 * x]);
        hiz_acc(mm, lds8[2 * y + 1][2 * x + 1]);
        dst.level[3][(size_t)(oy / 16 + y) * (sw >> 4) + ox / 16 + x] = mm;
        lds4[y][x] = mm;
    }
    __syncthreads();
    if (threadIdx.x < 4) {  // level +5: 2x2
        const uint32_t x = threadIdx.x & 1u, y = threadIdx.x >> 1;
        float2 mm = lds4[2 * y][2 * x];
        hiz_acc(mm, lds4[2 * y][2 * x + 1]);
        hiz_acc(mm, lds4[2 * y + 1][2 * x]);
        hiz_acc(mm, lds4[2 * y + 1][2 * x + 1]);
        dst.level[4][(size_t)(oy / 32 + y) * (sw >> 5) + ox / 32 + x] = mm;
        lds2[y][x] = mm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // level +6: 1 texel
        float2 mm = lds2[0][0];
        hiz_acc(mm, lds2[0][1]);
        hiz_acc(mm, lds2[1][0]);
        hiz_acc(mm, lds2[1][1]);
        dst.level[5][(size_t)(oy / 64) * (sw >> 6) + ox / 64] = mm;
    }
    (void)sh;
}

hipError_t launch_hiz_fused(const float* src_depth, const float2* src_pairs, const HizFusedDst& dst, uint32_t sw,
                            uint32_t sh, hipStream_t stream)
{
    const dim3 grid(sw / 64, sh / 64);
    if (src_depth)
        hipLaunchKernelGGL(hiz_fused_kernel<false>, grid, dim3(256), 0, stream, src_depth, src_pairs, dst, sw, sh);
    else
        hipLaunchKernelGGL(hiz_fused_kernel<true>, grid, dim3(256), 0, stream, src_depth, src_pairs, dst, sw, sh);
    return hipGetLastError();
}

}  // namespace gv
