// gv_cull.hip — hand-written gfx950 (CDNA4, wave64) kernels of the visibility pass: cull, compaction and the mirror
// maintenance kernels. (Sort: gv_sort.hip; sweeps: gv_sweep.hip; pyramid: gv_hiz.hip; shared helpers: gv_device.hpp.)
//
//   cull_kernel   replaces the body of prepareUnsortedMeshes / prepareSortedMeshes
//                 (source/system/render/mesh.cpp:137-175 / :213-253): filters, parent-chain model
//                 (include/garden/system/transform.hpp:197-214), 8-corner frustum test
//                 (render/mesh.hpp:142-146), optional Hi-Z occlusion query (build-defined), wave ballot.
//   scan_kernel + emit_kernel   replace drawCount.fetch_add + memcpy into combinedMeshes
//                 (mesh.cpp:177-183) with an order-stable compaction (ballot words + block prefix).
//   cull_multi_kernel   the same for up to 8 views that share cameraPosition (main camera + shadow cascades,
//                 mesh.cpp:795-847) in one pass over the streams.
//   block_bounds_kernel, block_classify / _window / cull_list_kernel   workgroup boxes: conservative block-level rejection.
//   sort_* / radix_*   sortMeshes (mesh.cpp:265-328): stable radix sort of the compact records by distanceSq.
//   sweep_*       TransformComponent::calcModel() for every transform slot (transform.hpp:197-214);
//                 the MFMA form runs the 4x4 chain on v_mfma_f32_4x4x1_16b_f32.
//   sweep_cull_*  sweep and cull of an exactly paired pool in ONE pass (world matrices + cull outputs).
//   hiz_*         HizRenderSystem::downsampleHiz (source/system/render/hiz.cpp:104-167) with the
//                 reduction rule of shaders/hiz.frag:23-63.
//
// All of it is HBM-bound streaming (DESIGN.md has bytes/entity per kernel); loads are 16- or 12-byte
// per lane over SoA streams so each wave-instruction touches 1 KiB / 768 B contiguous.
#include "gv_device.hpp"

namespace gv {

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
            behind = behind || (d < -fmaf(4e-5f, fabsf(planes[p][3]), margin));  // the plane offset rounds too
        }
    return behind;
}
__device__ __forceinline__ bool block_behind_frustum(const float4 lo, const float4 hi, const ViewParams& view, uint32_t max_depth)
{
    return block_behind_planes(lo, hi, view.planes, view.plane_count, view.cam, max_depth);
}

// the kernel's first argument (a CullArgs at the start of the kernarg segment), read again through a pointer the compiler cannot
// see through
typedef const CullArgs __attribute__((address_space(4))) * ConstCullArgs;
__device__ __forceinline__ void reload_cull_args(CullArgs& out)
{
    ConstCullArgs kernarg = (ConstCullArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kernarg));
    __builtin_memcpy(&out, kernarg, sizeof(out));
}

// The per-entity work of one 256-entry workgroup `lb`. RELOAD (cull_list_kernel, which calls this in a loop): the kernel arguments
// are read afresh at each stage — inlined into a loop, the ~70 SGPRs of planes, matrices and pointers would otherwise be read once
// in front of it and stay live across it (106 SGPRs, the overflow spilled into VGPRs: six waves per SIMD where cull_kernel runs
// eight); per stage each field lives only where it is used.
template <bool HIZ, uint32_t MAP, bool RELOAD = false>
__device__ __forceinline__ void cull_block(const CullArgs& args0, uint32_t lb, uint32_t* wave_count)
{
    CullArgs stage1;
    if (RELOAD)
        reload_cull_args(stage1);
    const CullArgs& args = RELOAD ? stage1 : args0;
    uint32_t tid = threadIdx.x;
    if (RELOAD)  // (... and so is everything that only depends on the lane: hoisted out of the loop it would sit in VGPRs across it)
        asm volatile("" : "+v"(tid));
    const uint32_t i = lb * kCullBlock + tid;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    bool visible = false;
    if (i < args.mesh.count) {
        Mat34 m;
        float4 box_a;
        float2 box_b;
        Corners c;
        uint32_t where = kSphereOutside;
        if (prepare_model<MAP>(args.mesh, args.xf, args.view.cam, i, m, box_a, box_b))
            where = classify_sphere(m, box_a, box_b, args.view.planes, args.view.plane_count);
        visible = where == kSphereInside;
        if (where == kSphereUndecided) {  // near a plane (or non-finite): the exact 8-corner test; rare, skipped wave-wide otherwise
            aabb_corners(m, box_a, box_b, c);
            visible = !behind_frustum(c, args.view.planes, args.view.plane_count);
        } else if (HIZ && visible) {
            aabb_corners(m, box_a, box_b, c);
        }
        // Hi-Z occlusion query on the survivors. Measured (profiles/r01b_hiz_ablation.txt): compacting the
        // survivors across the workgroup through LDS first buys nothing — the stage is bound by the texel
        // gathers (~4.5 M random 64-B sectors per frame), not by divergent VALU work.
        if (HIZ && visible) {
            CullArgs stage2;
            if (RELOAD)
                reload_cull_args(stage2);
            const CullArgs& a2 = RELOAD ? stage2 : args0;
            visible = !hiz_occluded(a2.hiz, a2.view.vp, c);
        }
    }
    CullArgs stage3;
    if (RELOAD)
        reload_cull_args(stage3);
    const CullArgs& out_args = RELOAD ? stage3 : args0;
    if (i < out_args.mesh.count && out_args.view.write_is_visible)
        out_args.out.is_visible[i] = visible ? 1 : 0;  // mesh.cpp:144,152,161,166
    __shared__ unsigned long long wave_word[kCullBlock / 64];
    const unsigned long long word = __ballot(visible);
    if (lane == 0) {
        wave_word[wave] = word;
        wave_count[wave] = (uint32_t)__popcll(word);
    }
    __syncthreads();
    // the tile's four ballot words leave as ONE 32-byte store (small stores are what a bandwidth-bound read kernel pays for:
    // round-3 probe tools/read_probe.hip, in the history), the tile's count as one atomic
    if (tid < kCullBlock / 64)
        out_args.out.mask[(size_t)lb * (kCullBlock / 64) + tid] = wave_word[tid];
    if (tid == 0) {
        uint32_t total = 0;
#pragma unroll
        for (uint32_t w = 0; w < kCullBlock / 64; w++)
            total += wave_count[w];
        if (total)  // integer adds commute: the sum is deterministic whatever the arrival order
            atomicAdd(&out_args.out.chunk_count[lb / (kEmitChunk / kCullBlock)], total);
    }
}

template <bool HIZ, uint32_t MAP>
__global__ __launch_bounds__(kCullBlock) void cull_kernel(const CullArgs args)
{
    __shared__ uint32_t wave_count[kCullBlock / 64];
    const uint32_t lb = tile_of_workgroup(blockIdx.x, args.xcd_run);
    if (lb >= args.nblocks)
        return;
    cull_block<HIZ, MAP>(args, lb, wave_count);
}

// ------------------------------------------------------------------------------------------------
// Block bounds (GV_CONFIG_BLOCK_BOUNDS) as two or three launches: classify, then cull the kept workgroups only.
// (Round 2 tested the box inside the cull kernel: a latency chain in front of every workgroup — box load -> 8 projections ->
// texel loads -> reduction — and 30 k skipped workgroups were still 30 k dispatches: with every frustum-surviving workgroup
// found occluded that kernel still took 60 us; profiles/withdrawn.md.) block_classify_kernel runs the
// frustum test with ONE LANE per 256-entry block (block_classify_kernel: 5 us at 39 k blocks), lists the survivors (one atomic per
// wave of 64 blocks) and, for a Hi-Z view, the texel window of each; block_window_kernel then reads every listed window with one
// WAVE per block (a lane walking its own window measured 66 us, a wave per block for the whole test 25 us: the launch-to-decision
// chain was paid 39 k times); cull_list_kernel runs the per-entity path of the listed blocks that were not proven occluded.
// Skipped blocks get their outputs (zero ballot words; zero isVisible bytes when the cull owns them) from whichever launch
// decides them. Same tests, same outputs.
// ------------------------------------------------------------------------------------------------
// Block-level Hi-Z: a block is skipped when EVERY candidate of the workgroup would be found occluded by its own query (hiz_occluded),
// like one behind a frustum plane. block_window works out what to compare (one lane per block), block_window_kernel reads the texels.
//
// An entity e is occluded iff zNear_e < zFar_e, zNear_e = max z/w over its 8 corners, zFar_e = min over the <= 2x2 texels
// that cover its pixel rect R_e at its level L_e. For the block:
//   * the box [lo, hi] holds every corner of every candidate (world space); inflated by the rounding margin of
//     block_behind_planes it also holds them as the per-frame arithmetic computes them (camera-relative products, fma order), and
//     the margin's effect on every projected quantity (>= 1e-7 relative: margin >= 0.01, w <= 1e5) dominates the fp32 rounding
//     of both evaluations. x/w, y/w, z/w are monotone along any segment with w > 0, so their extrema over the box are at its
//     corners: zNear_b = max z/w >= zNear_e, and the block's pixel rect R_b (one more pixel each way) contains every R_e;
//     a box that reaches w <= 0 is never tested (its entities may take the "cannot bound: visible" exit).
//   * lo.w = the largest sphere reach r of the block's candidates (block_bounds_kernel): two corners of one entity differ by
//     <= 2 r in L1, so its rect spans at most n_max pixels (below) and its level is at most L_max = floor(log2 n_max) + 1.
//   * at any level L_b >= L_max of a NESTED pyramid every texel of level L_e <= L_b that touches R_e lies inside a level-L_b texel
//     that touches R_b, and a texel's min bounds everything under it: zFar_b = min over ALL level-L_b texels touching R_b <= zFar_e.
//   So zNear_b < zFar_b  =>  zNear_e <= zNear_b < zFar_b <= zFar_e for every candidate: all occluded. L_b is also raised until R_b
//   spans <= 16 x 16 texels: four texels per lane of one wave, one reduction. A NaN texel (an entity's own compare would fail on it) or any
//   non-finite intermediate declines the shortcut.
// What that test needs from the block's box for one view, worked out by ONE LANE (block_classify_kernel runs a lane per
// block): the texel window (level, first texel, extent <= 16 x 16) and the box's nearest depth. level 0xFF: the block cannot
// be tested (pyramid not nested, box reaches w <= 0, non-finite, window too large) and is kept.
struct BlockWindow {
    uint32_t block;   // the 256-entry block
    uint32_t level;   // | (tx1 - tx0) << 8 | (ty1 - ty0) << 12
    uint32_t origin;  // tx0 | ty0 << 16
    float znear;      // already widened by its rounding allowance
};
__device__ __forceinline__ BlockWindow block_window(const HizDevice& hz, const ViewParams& view, const float4 lo, const float4 hi,
                                                    uint32_t max_depth, uint32_t lb)
{
    BlockWindow out{lb, 0xFFu, 0u, 0.0f};
    if (!hz.nested)
        return out;
    const float (&vp)[16] = view.vp;
    const float mag = fmaxf(fabsf(lo.x), fabsf(hi.x)) + fmaxf(fabsf(lo.y), fabsf(hi.y)) + fmaxf(fabsf(lo.z), fabsf(hi.z)) +
                      fabsf(view.cam[0]) + fabsf(view.cam[1]) + fabsf(view.cam[2]);
    const float margin = 0.01f + 4e-5f * (float)(max_depth + 1u) * mag;
    const float bx[2] = {lo.x - view.cam[0] - margin, hi.x - view.cam[0] + margin};
    const float by[2] = {lo.y - view.cam[1] - margin, hi.y - view.cam[1] + margin};
    const float bz[2] = {lo.z - view.cam[2] - margin, hi.z - view.cam[2] + margin};
    const float inf = __builtin_huge_valf();
    float nx0 = inf, nx1 = -inf, ny0 = inf, ny1 = -inf, znear = -inf, wmin = inf;
    bool bounded = true;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float x = bx[k & 1], y = by[(k >> 1) & 1], z = bz[k >> 2];
        const float clx = fmaf(vp[0], x, fmaf(vp[4], y, fmaf(vp[8], z, vp[12])));
        const float cly = fmaf(vp[1], x, fmaf(vp[5], y, fmaf(vp[9], z, vp[13])));
        const float clz = fmaf(vp[2], x, fmaf(vp[6], y, fmaf(vp[10], z, vp[14])));
        const float clw = fmaf(vp[3], x, fmaf(vp[7], y, fmaf(vp[11], z, vp[15])));
        bounded = bounded && (clw > 0.0f);
        const float rcp = 1.0f / clw;
        const float ndx = clx * rcp, ndy = cly * rcp, ndz = clz * rcp;
        nx0 = fminf(nx0, ndx); nx1 = fmaxf(nx1, ndx);
        ny0 = fminf(ny0, ndy); ny1 = fmaxf(ny1, ndy);
        znear = fmaxf(znear, ndz);
        wmin = fminf(wmin, clw);
    }
    if (!bounded)
        return out;
    const float r = lo.w;
    const float r0 = fmaxf(fmaxf(fabsf(vp[0]), fabsf(vp[4])), fabsf(vp[8])), r1 = fmaxf(fmaxf(fabsf(vp[1]), fabsf(vp[5])), fabsf(vp[9]));
    const float r3 = fmaxf(fmaxf(fabsf(vp[3]), fabsf(vp[7])), fabsf(vp[11]));
    const float span = 2.0f * r / wmin;
    const float ext_x = 0.5f * (float)hz.width * span * fmaf(fmaxf(fabsf(nx0), fabsf(nx1)), r3, r0) + 3.0f;
    const float ext_y = 0.5f * (float)hz.height * span * fmaf(fmaxf(fabsf(ny0), fabsf(ny1)), r3, r1) + 3.0f;
    const float n_max = fmaxf(ext_x, ext_y);
    if (!(n_max < 32768.0f))
        return out;
    const uint32_t l_max = (31u - (uint32_t)__clz((int)n_max)) + 1u;
    const int W = (int)hz.width, H = (int)hz.height;
    const float umin = clamp01(fmaf(nx0, 0.5f, 0.5f)), umax = clamp01(fmaf(nx1, 0.5f, 0.5f));
    const float vmin = clamp01(fmaf(ny0, 0.5f, 0.5f)), vmax = clamp01(fmaf(ny1, 0.5f, 0.5f));
    const int ix0 = max((int)(umin * (float)W) - 1, 0), ix1 = min((int)(umax * (float)W) + 1, W - 1);
    const int iy0 = max((int)(vmin * (float)H) - 1, 0), iy1 = min((int)(vmax * (float)H) + 1, H - 1);
    uint32_t level = min(l_max, hz.mip_count - 1u);
    while (level + 1u < hz.mip_count && (((ix1 >> level) - (ix0 >> level)) > 15 || ((iy1 >> level) - (iy0 >> level)) > 15))
        level++;
    const int lw = max(W >> level, 1), lh = max(H >> level, 1);
    const int tx0 = min(ix0 >> level, lw - 1), tx1 = min(ix1 >> level, lw - 1);
    const int ty0 = min(iy0 >> level, lh - 1), ty1 = min(iy1 >> level, lh - 1);
    if (tx1 - tx0 > 15 || ty1 - ty0 > 15)
        return out;
    out.level = level | ((uint32_t)(tx1 - tx0) << 8) | ((uint32_t)(ty1 - ty0) << 12);
    out.origin = (uint32_t)tx0 | ((uint32_t)ty0 << 16);
    out.znear = fmaf(fabsf(znear), 2e-6f, znear);
    return out;
}

struct ClassifyArgs {
    BlockBounds bounds;
    HizDevice hiz;
    ViewParams view;
    unsigned long long* mask;   // the view's ballot words: zeroed for the skipped blocks
    uint8_t* is_visible;        // the view's bytes: zeroed for the skipped blocks when view.write_is_visible
    uint32_t* kept_count;       // this launch's counter (zero on entry)
    BlockWindow* kept;          // the frustum-surviving blocks, ascending within a wave's 64 blocks, waves in order of arrival
    uint8_t* kept_flag;         // HIZ: per list entry, 1 = not proven occluded (written by block_window_kernel)
    uint32_t nblocks, count, max_depth;
};

__device__ __forceinline__ void write_skipped_block(const ClassifyArgs& a, uint32_t lb)
{
    uint4* words = reinterpret_cast<uint4*>(a.mask + (size_t)lb * (kCullBlock / 64));  // 4 ballot words = 32 bytes
    words[0] = make_uint4(0, 0, 0, 0);
    words[1] = make_uint4(0, 0, 0, 0);
    if (a.view.write_is_visible) {  // (only when no emit follows the cull: count-only main views)
        const uint32_t first = lb * kCullBlock, end = min(first + kCullBlock, a.count);
        for (uint32_t s = first; s < end; s++)
            a.is_visible[s] = 0;
    }
}

// launch 1: one LANE per 256-entry block, one wave per workgroup. Frustum test; the survivors go to the list (HIZ: with their
// texel windows).
template <bool HIZ>
__global__ __launch_bounds__(64) void block_classify_kernel(const ClassifyArgs a)
{
    const uint32_t lb = blockIdx.x * 64u + threadIdx.x;
    bool kept = false;
    BlockWindow win{lb, 0xFFu, 0u, 0.0f};
    if (lb < a.nblocks) {
        const float4 lo = a.bounds.lo[lb], hi = a.bounds.hi[lb];
        kept = !(lo.x > hi.x || block_behind_frustum(lo, hi, a.view, a.max_depth));
        if (HIZ && kept)
            win = block_window(a.hiz, a.view, lo, hi, a.max_depth, lb);
        if (!HIZ || !kept)  // (HIZ survivors: decided by block_window_kernel)
            a.bounds.examined[lb] = kept ? 1 : 0;
        if (!kept)
            write_skipped_block(a, lb);
    }
    const unsigned long long keep = __ballot(kept);
    if (keep == 0ull)
        return;
    uint32_t base = 0;
    if (threadIdx.x == 0)
        base = atomicAdd(a.kept_count, (uint32_t)__popcll(keep));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (kept)
        a.kept[base + (uint32_t)__popcll(keep & ((1ull << threadIdx.x) - 1ull))] = win;
}

// launch 2 (Hi-Z views): one WAVE per listed block reads its window, four texels per lane, and decides.
__global__ __launch_bounds__(256) void block_window_kernel(const ClassifyArgs a)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves = gridDim.x * 4u;
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
    BlockWindow next = a.kept[min(first, a.nblocks - 1u)];  // (asked for together with the count: the list is sized for every block of the pool)
    const uint32_t listed = *a.kept_count;
    const float inf = __builtin_huge_valf();
    for (uint32_t j = first; j < listed; j += waves) {
        const BlockWindow w = next;  // wave-uniform
        if (j + waves < listed)
            next = a.kept[j + waves];
        bool occluded = false;
        const uint32_t level = w.level & 0xFFu;
        if (level != 0xFFu) {
            const uint32_t dx = (w.level >> 8) & 15u, dy = (w.level >> 12) & 15u, tx0 = w.origin & 0xFFFFu, ty0 = w.origin >> 16;
            const uint32_t lw = max(a.hiz.width >> level, 1u);
            const uint32_t lx = lane & 15u;
            float m = inf;
            bool nan = false;
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) {  // all four loads in flight together
                const uint32_t ly = (lane >> 4) + 4u * q;
                float t = inf;
                if (lx <= dx && ly <= dy)
                    t = hiz_min_texel(a.hiz, level, lw, tx0 + lx, ty0 + ly);
                nan = nan || t != t;
                m = fminf(m, t);  // (a NaN is dropped here and reported through `nan`)
            }
#pragma unroll
            for (uint32_t d = 32; d >= 1; d >>= 1)
                m = fminf(m, __shfl_xor(m, d, 64));
            occluded = __ballot(nan) == 0ull && w.znear < m;
        }
        if (lane == 0) {
            a.kept_flag[j] = occluded ? 0 : 1;
            a.bounds.examined[w.block] = occluded ? 0 : 1;
        }
        if (occluded) {
            if (lane < kCullBlock / 64)
                a.mask[(size_t)w.block * (kCullBlock / 64) + lane] = 0ull;
            if (a.view.write_is_visible) {
                const uint32_t slot = w.block * kCullBlock + 4u * lane;
                if (slot + 3u < a.count)
                    *reinterpret_cast<uint32_t*>(a.is_visible + slot) = 0u;
                else
                    for (uint32_t s = 0; slot + s < a.count && s < 4u; s++)
                        a.is_visible[slot + s] = 0;
            }
        }
    }
}

struct CullListArgs {
    const uint32_t* kept_count;
    const BlockWindow* kept;
    const uint8_t* kept_flag;  // NULL: every listed block is culled (no Hi-Z block test)
    uint32_t* next_count;      // the counter of the NEXT classify launch (the two alternate): cleared here
};

// last launch: the per-entity path of the listed blocks
template <bool HIZ, uint32_t MAP>
__global__ __launch_bounds__(kCullBlock, 8) void cull_list_kernel(const CullArgs args, const CullListArgs la)
{
    __shared__ uint32_t wave_count[kCullBlock / 64];
    // the workgroup's first list entry is asked for together with the count that says whether it exists (the list and the flags are
    // sized for every block of the pool): one round trip in front of the block's own loads instead of three
    uint32_t flag = la.kept_flag ? la.kept_flag[blockIdx.x] : 1u;
    uint32_t block = la.kept[blockIdx.x].block;
    const uint32_t listed = *la.kept_count;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *la.next_count = 0;
    for (uint32_t j = blockIdx.x; j < listed;) {  // workgroup-uniform trip count
        if (__builtin_amdgcn_readfirstlane((int)flag)) {  // (loaded values: told to the compiler to be wave-uniform, so that the block's
            cull_block<HIZ, MAP, true>(args, (uint32_t)__builtin_amdgcn_readfirstlane((int)block), wave_count);  // addresses are scalar as in cull_kernel)
            __syncthreads();  // the LDS words are reused by the next block
        }
        j += gridDim.x;
        if (j < listed) {
            flag = la.kept_flag ? la.kept_flag[j] : 1u;
            block = la.kept[j].block;
        }
    }
}

hipError_t launch_cull_listed(const MeshMirror& mesh, const TransformMirror& xf, const HizDevice& hiz, const ViewParams& vp,
                              const ViewBuffers& out, const BlockBounds& bounds, uint32_t* kept_count, uint32_t* next_count,
                              void* kept_list, uint8_t* kept_flag, hipStream_t stream)
{
    if (mesh.count == 0)
        return hipSuccess;
    const uint32_t nblocks = (mesh.count + kCullBlock - 1) / kCullBlock;
    const bool window_test = vp.use_hiz && hiz.nested;
    ClassifyArgs c{};
    c.bounds = bounds;
    c.hiz = hiz;
    c.view = vp;
    c.mask = out.mask;
    c.is_visible = out.is_visible;
    c.kept_count = kept_count;
    c.kept = static_cast<BlockWindow*>(kept_list);
    c.kept_flag = kept_flag;
    c.nblocks = nblocks;
    c.count = mesh.count;
    c.max_depth = xf.max_depth;
    // a quarter of the workgroups a flat launch has (behind a frustum test a quarter of the blocks is kept: one each; a view that
    // keeps everything walks four per workgroup) — a launch of 39 k workgroups that mostly return at once costs 9 us of dispatch
    const uint32_t list_grid = (nblocks + 3u) / 4u;
    if (window_test) {
        hipLaunchKernelGGL(block_classify_kernel<true>, dim3((nblocks + 63) / 64), dim3(64), 0, stream, c);
        hipLaunchKernelGGL(block_window_kernel, dim3((list_grid + 3u) / 4u), dim3(256), 0, stream, c);
    } else {
        hipLaunchKernelGGL(block_classify_kernel<false>, dim3((nblocks + 63) / 64), dim3(64), 0, stream, c);
    }
    CullArgs a{};
    a.mesh = mesh;
    a.xf = xf;
    a.hiz = hiz;
    a.view = vp;
    a.out = out;
    a.nblocks = nblocks;
    const CullListArgs la{kept_count, static_cast<const BlockWindow*>(kept_list), window_test ? kept_flag : nullptr, next_count};
    const dim3 grid(list_grid), block(kCullBlock);
#define GV_LAUNCH_LIST(HIZ)                                                                                              \
    switch (mesh.mapping) {                                                                                             \
    case kMapExact: hipLaunchKernelGGL((cull_list_kernel<HIZ, kMapExact>), grid, block, 0, stream, a, la); break;         \
    case kMapSpeculate: hipLaunchKernelGGL((cull_list_kernel<HIZ, kMapSpeculate>), grid, block, 0, stream, a, la); break; \
    default: hipLaunchKernelGGL((cull_list_kernel<HIZ, kMapGeneral>), grid, block, 0, stream, a, la); break;              \
    }
    if (vp.use_hiz) {
        GV_LAUNCH_LIST(true)
    } else {
        GV_LAUNCH_LIST(false)
    }
#undef GV_LAUNCH_LIST
    return hipGetLastError();
}
size_t cull_list_entry_bytes() { return sizeof(BlockWindow); }

// ------------------------------------------------------------------------------------------------
// K1 + K3 in one launch: cull, order-stable compaction and record emission of one view.
// A workgroup culls its 256-entry tile, learns how many records the tiles before it hold by DECOUPLED LOOK-BACK (a
// 64-bit status word per tile: epoch | flag | count; wave 0 inspects 64 predecessors per step) and writes its records —
// the models are still in registers, so the emit kernel's second gather of the TRS streams and its second walk up the
// parent chain disappear with its launch — through LDS as whole contiguous rows. Same records in the same order as
// cull_kernel + emit_kernel (ascending mirror entry). Tiles are taken in order of arrival (a ticket per workgroup), so
// a workgroup only ever waits for tiles whose workgroups are already running. The status words are stamped with the
// launch's epoch and never cleared. Used where it measures faster than the two launches (gv_context.cpp picks).
// ------------------------------------------------------------------------------------------------
struct FusedEmit {
    unsigned long long* status;  // one word per tile
    uint32_t* ticket;            // running ticket counter (never reset: ticket_base is its value before this launch)
    uint32_t ticket_base;
    uint32_t epoch;              // != 0, different from the previous launches that used these words
};
constexpr unsigned long long kTileAggregate = 1ull << 30, kTilePrefix = 2ull << 30, kTileFlagMask = 3ull << 30, kTileCountMask = (1ull << 30) - 1ull;

template <bool HIZ, uint32_t MAP>
__global__ __launch_bounds__(kCullBlock) void cull_emit_kernel(const CullArgs args, const FusedEmit fe)
{
    __shared__ uint32_t wave_count[kCullBlock / 64];
    __shared__ uint32_t tile_s, base_s;
    __shared__ float4 stage[kCullBlock * 3];
    if (threadIdx.x == 0)
        tile_s = atomicAdd(fe.ticket, 1u) - fe.ticket_base;
    __syncthreads();
    const uint32_t lb = tile_s;  // < nblocks: the grid has exactly nblocks workgroups
    const uint32_t i = lb * kCullBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    bool visible = false;
    Mat34 m = {};
    if (i < args.mesh.count) {
        float4 box_a;
        float2 box_b;
        Corners c;
        uint32_t where = kSphereOutside;
        if (prepare_model<MAP>(args.mesh, args.xf, args.view.cam, i, m, box_a, box_b))
            where = classify_sphere(m, box_a, box_b, args.view.planes, args.view.plane_count);
        visible = where == kSphereInside;
        if (where == kSphereUndecided) {
            aabb_corners(m, box_a, box_b, c);
            visible = !behind_frustum(c, args.view.planes, args.view.plane_count);
        } else if (HIZ && visible) {
            aabb_corners(m, box_a, box_b, c);
        }
        if (HIZ && visible)
            visible = !hiz_occluded(args.hiz, args.view.vp, c);
        if (args.view.write_is_visible)
            args.out.is_visible[i] = visible ? 1 : 0;  // mesh.cpp:144,152,161,166
    }
    const unsigned long long word = __ballot(visible);
    if (lane == 0)
        wave_count[wave] = (uint32_t)__popcll(word);
    __syncthreads();
    uint32_t total = 0, wave_prefix = 0;
#pragma unroll
    for (uint32_t w = 0; w < kCullBlock / 64; w++) {
        wave_prefix += w < wave ? wave_count[w] : 0u;
        total += wave_count[w];
    }
    if (wave == 0) {  // look-back: 64 predecessors per step
        const unsigned long long stamp = (unsigned long long)fe.epoch << 32;
        uint32_t before = 0;
        if (lb == 0) {
            if (lane == 0)
                __hip_atomic_store(&fe.status[0], stamp | kTilePrefix | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0)
                __hip_atomic_store(&fe.status[lb], stamp | kTileAggregate | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int32_t p = (int32_t)lb - 1;  // nearest predecessor not consumed yet
            for (;;) {
                const int32_t q = p - (int32_t)lane;
                unsigned long long v = stamp | kTilePrefix;  // in front of tile 0: nothing
                if (q >= 0)
                    v = __hip_atomic_load(&fe.status[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long flag = (uint32_t)(v >> 32) == fe.epoch ? (v & kTileFlagMask) : 0ull;  // other epochs: not published yet
                const unsigned long long empties = __ballot(flag == 0ull), prefixes = __ballot(flag == kTilePrefix);
                const uint32_t first_empty = empties ? (uint32_t)__builtin_ctzll(empties) : 64u;
                const uint32_t first_prefix = prefixes ? (uint32_t)__builtin_ctzll(prefixes) : 64u;
                const uint32_t upto = min(first_empty, first_prefix + 1u);  // lanes [0, upto) can be consumed
                uint32_t part = lane < upto ? (uint32_t)(v & kTileCountMask) : 0u;
#pragma unroll
                for (uint32_t d = 32; d >= 1; d >>= 1)
                    part += __shfl_xor(part, d, 64);
                before += part;
                if (first_prefix < first_empty)
                    break;
                p -= (int32_t)upto;
                if (upto == 0)
                    __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0)
                __hip_atomic_store(&fe.status[lb], stamp | kTilePrefix | (before + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            base_s = before;
            if (lb + 1 == args.nblocks)
                *args.out.draw_count = before + total;
        }
    }
    __syncthreads();
    const uint32_t base = base_s;
    if (visible) {  // the record of mesh.cpp:169-173, at its final place
        const uint32_t local = wave_prefix + (uint32_t)__popcll(word & ((1ull << lane) - 1ull));
        const size_t rank = (size_t)base + local;
        args.out.visible_idx[rank] = args.mesh.orig ? args.mesh.orig[i] : i;
        const float tx = m.c3x + args.view.cam_offset[0], ty = m.c3y + args.view.cam_offset[1], tz = m.c3z + args.view.cam_offset[2];
        args.out.distance_sq[rank] = args.view.distance_2d ? m.c3z + 1.0f : fmaf(tz, tz, fmaf(ty, ty, tx * tx));
        float4* row = stage + local * 3;
        row[0] = make_float4(m.c0x, m.c0y, m.c0z, m.c1x);
        row[1] = make_float4(m.c1y, m.c1z, m.c2x, m.c2y);
        row[2] = make_float4(m.c2z, m.c3x, m.c3y, m.c3z);
    }
    __syncthreads();
    float4* dst = reinterpret_cast<float4*>(args.out.baked_model) + (size_t)base * 3;
    for (uint32_t q = threadIdx.x; q < total * 3u; q += kCullBlock)
        dst[q] = stage[q];
}

hipError_t launch_cull_emit(const MeshMirror& mesh, const TransformMirror& xf, const HizDevice& hiz, const ViewParams& vp,
                            const ViewBuffers& out, unsigned long long* status, uint32_t* ticket, uint32_t ticket_base, uint32_t epoch,
                            hipStream_t stream)
{
    if (mesh.count == 0)
        return hipSuccess;
    CullArgs a{};
    a.mesh = mesh;
    a.xf = xf;
    a.hiz = hiz;
    a.view = vp;
    a.out = out;
    a.nblocks = (mesh.count + kCullBlock - 1) / kCullBlock;
    const FusedEmit fe{status, ticket, ticket_base, epoch};
    const dim3 grid(a.nblocks), block(kCullBlock);
#define GV_LAUNCH_FUSED(HIZ)                                                                                          \
    switch (mesh.mapping) {                                                                                          \
    case kMapExact: hipLaunchKernelGGL((cull_emit_kernel<HIZ, kMapExact>), grid, block, 0, stream, a, fe); break;         \
    case kMapSpeculate: hipLaunchKernelGGL((cull_emit_kernel<HIZ, kMapSpeculate>), grid, block, 0, stream, a, fe); break; \
    default: hipLaunchKernelGGL((cull_emit_kernel<HIZ, kMapGeneral>), grid, block, 0, stream, a, fe); break;              \
    }
    if (vp.use_hiz) {
        GV_LAUNCH_FUSED(true)
    } else {
        GV_LAUNCH_FUSED(false)
    }
#undef GV_LAUNCH_FUSED
    return hipGetLastError();
}

hipError_t launch_cull(const MeshMirror& mesh, const TransformMirror& xf, const HizDevice& hiz, const ViewParams& vp,
                       const ViewBuffers& out, hipStream_t stream)
{
    if (mesh.count == 0)
        return hipSuccess;
    CullArgs a{};
    a.mesh = mesh;
    a.xf = xf;
    a.hiz = hiz;
    a.view = vp;
    a.out = out;
    a.nblocks = (mesh.count + kCullBlock - 1) / kCullBlock;
    // measured: the frustum-only scan gains 6 % from per-XCD runs, the Hi-Z variant does not
    a.xcd_run = vp.use_hiz ? 0 : xcd_run_for_tiles(a.nblocks);
    const dim3 grid(grid_for_tiles(a.nblocks, a.xcd_run)), block(kCullBlock);
#define GV_LAUNCH_CULL(HIZ)                                                                                       \
    switch (mesh.mapping) {                                                                                      \
    case kMapExact: hipLaunchKernelGGL((cull_kernel<HIZ, kMapExact>), grid, block, 0, stream, a); break;         \
    case kMapSpeculate: hipLaunchKernelGGL((cull_kernel<HIZ, kMapSpeculate>), grid, block, 0, stream, a); break; \
    default: hipLaunchKernelGGL((cull_kernel<HIZ, kMapGeneral>), grid, block, 0, stream, a); break;              \
    }
    if (vp.use_hiz) {
        GV_LAUNCH_CULL(true)
    } else {
        GV_LAUNCH_CULL(false)
    }
#undef GV_LAUNCH_CULL
    return hipGetLastError();
}

// World-space box of each cull workgroup's candidates (camera at the origin: translate(-0) leaves c3 as it is).
template <uint32_t MAP>
__device__ __forceinline__ void block_bounds_of(const MeshMirror& mesh, const TransformMirror& xf, const uint32_t lb, float (*red)[7],
                                                float4* __restrict__ out_lo, float4* __restrict__ out_hi)
{
    const uint32_t i = lb * kCullBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const float inf = __builtin_huge_valf();
    float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    float reach = 0.0f;  // the largest sphere reach of the workgroup's candidates (block_window: bounds an entity's pixel extent)
    if (i < mesh.count) {
        Mat34 m;
        Corners c;
        float4 box_a;
        float2 box_b;
        const float cam[3] = {0.0f, 0.0f, 0.0f};
        if (prepare_model<MAP>(mesh, xf, cam, i, m, box_a, box_b)) {
            aabb_corners(m, box_a, box_b, c);
            reach = sphere_reach(m, box_a, box_b);
            bool finite = reach == reach;
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
            if (!finite) {  // a member the box cannot bound: the workgroup is always examined
                for (int k = 0; k < 3; k++) {
                    lo[k] = -inf;
                    hi[k] = inf;
                }
                reach = inf;
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
#pragma unroll
    for (uint32_t d = 32; d >= 1; d >>= 1)
        reach = fmaxf(reach, __shfl_xor(reach, d, 64));
    if (lane == 0) {
        for (int k = 0; k < 3; k++) {
            red[wave][k] = lo[k];
            red[wave][3 + k] = hi[k];
        }
        red[wave][6] = reach;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < kCullBlock / 64; w++) {
            for (int k = 0; k < 3; k++) {
                red[0][k] = fminf(red[0][k], red[w][k]);
                red[0][3 + k] = fmaxf(red[0][3 + k], red[w][3 + k]);
            }
            red[0][6] = fmaxf(red[0][6], red[w][6]);
        }
        out_lo[lb] = make_float4(red[0][0], red[0][1], red[0][2], red[0][6]);
        out_hi[lb] = make_float4(red[0][3], red[0][4], red[0][5], 0.0f);
    }
}

template <uint32_t MAP>
__global__ __launch_bounds__(kCullBlock) void block_bounds_kernel(const MeshMirror mesh, const TransformMirror xf,
                                                                  float4* __restrict__ out_lo, float4* __restrict__ out_hi)
{
    __shared__ float red[kCullBlock / 64][7];
    block_bounds_of<MAP>(mesh, xf, blockIdx.x, red, out_lo, out_hi);
}

// one emit seed per entry of a flat, exactly paired pool (gv_kernels.hpp EmitSeed)
__device__ __forceinline__ void emit_seed_of(const MeshMirror& mesh, const TransformMirror& xf, EmitSeed* __restrict__ seeds, uint32_t i)
{
    if (i >= mesh.count || i >= xf.count)
        return;
    float4* out = reinterpret_cast<float4*>(seeds + i);
    const float2 c = xf.c[i];
    out[0] = xf.ab[i].a;
    out[1] = xf.ab[i].b;
    out[2] = make_float4(c.x, c.y, __uint_as_float(mesh.orig ? mesh.orig[i] : i), 0.0f);
    out[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// What a pool AT REST has derived from its mirror — block bounds, emit seeds — kept current while a few entries change per frame
// (round 3; flat, exactly paired pools): every sync flags the 256-entry blocks that hold a re-mirrored entry
// (mark_dirty_blocks_kernel); the next cull re-derives just those. A workgroup reads the flags of 16 consecutive blocks with one
// load and walks the flagged ones: a tenth of the workgroups of one per block (a launch whose workgroups mostly leave at once is
// still paid per workgroup: 9 us for 39 k).
constexpr uint32_t kPatchSpan = 16;
template <uint32_t MAP>
__global__ __launch_bounds__(kCullBlock) void block_patch_kernel(const MeshMirror mesh, const TransformMirror xf, float4* __restrict__ out_lo,
                                                                 float4* __restrict__ out_hi, EmitSeed* __restrict__ seeds /* or NULL */,
                                                                 uint8_t* __restrict__ flags /* padded to a multiple of kPatchSpan */,
                                                                 uint32_t nblocks)
{
    __shared__ float red[kCullBlock / 64][7];
    const uint4 word = *reinterpret_cast<const uint4*>(flags + (size_t)blockIdx.x * kPatchSpan);  // workgroup-uniform
    if ((word.x | word.y | word.z | word.w) == 0u)
        return;
    const uint32_t w[4] = {word.x, word.y, word.z, word.w};
#pragma unroll 1
    for (uint32_t k = 0; k < kPatchSpan; k++) {
        const uint32_t lb = blockIdx.x * kPatchSpan + k;
        if (!((w[k >> 2] >> (8u * (k & 3u))) & 0xFFu) || lb >= nblocks)
            continue;
        block_bounds_of<MAP>(mesh, xf, lb, red, out_lo, out_hi);
        if (seeds)
            emit_seed_of(mesh, xf, seeds, lb * kCullBlock + threadIdx.x);
        __syncthreads();  // `red` is reused by the next block
    }
    if (threadIdx.x == 0)  // (every lane has read the flags into `word` before anyone gets here: the load precedes the first barrier)
        *reinterpret_cast<uint4*>(flags + (size_t)blockIdx.x * kPatchSpan) = make_uint4(0, 0, 0, 0);
}

hipError_t launch_block_patch(const MeshMirror& mesh, const TransformMirror& xf, float4* lo, float4* hi, EmitSeed* seeds, uint8_t* flags,
                              hipStream_t stream)
{
    if (mesh.count == 0)
        return hipSuccess;
    const uint32_t nblocks = (mesh.count + kCullBlock - 1) / kCullBlock;
    const dim3 grid((nblocks + kPatchSpan - 1) / kPatchSpan), block(kCullBlock);
    switch (mesh.mapping) {
    case kMapExact: hipLaunchKernelGGL((block_patch_kernel<kMapExact>), grid, block, 0, stream, mesh, xf, lo, hi, seeds, flags, nblocks); break;
    case kMapSpeculate: hipLaunchKernelGGL((block_patch_kernel<kMapSpeculate>), grid, block, 0, stream, mesh, xf, lo, hi, seeds, flags, nblocks); break;
    default: hipLaunchKernelGGL((block_patch_kernel<kMapGeneral>), grid, block, 0, stream, mesh, xf, lo, hi, seeds, flags, nblocks); break;
    }
    return hipGetLastError();
}

// flags[entry >> 8] = 1 for every entry re-mirrored by a sync: thread t is the t-th dirty slot of the sync's ranges (start[k] = slots
// in the ranges before range k, first[k] = its first slot); inv: slot -> mirror entry, a table of `slots` elements (NULL: the
// mirror is in slot order). The ranges may be TRANSFORM slots (transform-side syncs flag the exactly paired mesh pools through the
// transform pool's own table): a slot is bounded by the table it indexes, the ENTRY by the flagged pool's occupancy — a paired
// pool with fewer meshes than transforms maps transform slots beyond its occupancy to entries inside it.
__global__ __launch_bounds__(256) void mark_dirty_blocks_kernel(const uint32_t* __restrict__ start, const uint32_t* __restrict__ first, uint32_t nranges,
                                                                const uint32_t* __restrict__ inv, uint32_t slots, uint32_t entries,
                                                                uint8_t* __restrict__ flags)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= start[nranges])
        return;
    uint32_t lo = 0, hi = nranges;  // the range k with start[k] <= t < start[k + 1]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (start[mid] <= t)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t slot = first[lo] + (t - start[lo]);
    if (slot >= slots)
        return;
    const uint32_t entry = inv ? inv[slot] : slot;
    if (entry < entries)
        flags[entry / kCullBlock] = 1;
}

hipError_t launch_mark_dirty_blocks(const uint32_t* start, const uint32_t* first, uint32_t nranges, uint32_t total, const uint32_t* inv,
                                    uint32_t slots, uint32_t entries, uint8_t* flags, hipStream_t stream)
{
    if (total == 0 || nranges == 0)
        return hipSuccess;
    hipLaunchKernelGGL(mark_dirty_blocks_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, start, first, nranges, inv, slots, entries, flags);
    return hipGetLastError();
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

// One view's part of the batch arguments, read where the view is processed: the view loops below are NOT unrolled and fetch
// planes[v] / outs[v] with scalar loads at a runtime offset from `base` (the argument block in the constant address space: the
// kernarg segment, or the job's table entry). Unrolled over eight views the ~45 dwords per view were all live at once — 106 SGPRs
// and 200-290 more spilled into VGPR lanes, a v_readlane per use.
typedef const MultiCullArgs __attribute__((address_space(4))) * ConstMultiArgs;
__device__ __forceinline__ MultiViewPlanes view_planes(ConstMultiArgs base, uint32_t v)
{
    MultiViewPlanes out;
    __builtin_memcpy(&out, &base->planes[v], sizeof(out));
    return out;
}
__device__ __forceinline__ ViewBuffers view_outputs(ConstMultiArgs base, uint32_t v)
{
    ViewBuffers out;
    __builtin_memcpy(&out, &base->outs[v], sizeof(out));
    return out;
}

template <bool HIZ, uint32_t MAP, bool BOUNDS>
__device__ __forceinline__ void cull_multi_block(const MultiCullArgs& args, ConstMultiArgs base, const uint32_t lb,
                                                 uint32_t (&wave_count)[kMaxBatchViews][kCullBlock / 64])
{
    asm volatile("" : "+s"(base));  // (opaque: nothing read through it is moved out of the loops)
    const uint32_t i = lb * kCullBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool in_range = i < args.mesh.count;
    const uint32_t nviews = min(args.nviews, kMaxBatchViews);
    if (BOUNDS) {  // the workgroup is skipped when its box is outside EVERY view of the batch
        const float4 lo = args.bounds.lo[lb], hi = args.bounds.hi[lb];
        bool skip = true;
        if (!(lo.x > hi.x)) {
#pragma unroll 1
            for (uint32_t v = 0; v < nviews && skip; v++) {
                const MultiViewPlanes pl = view_planes(base, v);
                skip = block_behind_planes(lo, hi, pl.planes, pl.plane_count, args.cam, args.xf.max_depth);
            }
        }
        if (threadIdx.x == 0)
            args.bounds.examined[lb] = skip ? 0 : 1;
        if (skip) {
#pragma unroll 1
            for (uint32_t v = 0; v < nviews; v++) {
                const ViewBuffers out = view_outputs(base, v);
                if (base->planes[v].write_is_visible && in_range)
                    out.is_visible[i] = 0;
                if (lane == 0)
                    out.mask[(size_t)lb * (kCullBlock / 64) + wave] = 0ull;
            }
            return;
        }
    }
    Mat34 m;
    float4 box_a;
    float2 box_b;
    Corners c;
    const bool candidate = in_range && prepare_model<MAP>(args.mesh, args.xf, args.cam, i, m, box_a, box_b);
    const float reach = candidate ? sphere_reach(m, box_a, box_b) : 0.0f;  // one sphere for every view of the batch
    bool have_corners = false;  // generated once, by the first view that needs them
#pragma unroll 1
    for (uint32_t v = 0; v < nviews; v++) {
        const MultiViewPlanes pl = view_planes(base, v);
        const uint32_t where = candidate ? classify_sphere(m, reach, pl.planes, pl.plane_count) : kSphereOutside;
        bool visible = where == kSphereInside;
        if (where == kSphereUndecided || (HIZ && v == 0 && visible)) {
            if (!have_corners) {
                aabb_corners(m, box_a, box_b, c);
                have_corners = true;
            }
            if (where == kSphereUndecided)
                visible = !behind_frustum(c, pl.planes, pl.plane_count);
        }
        if (HIZ && v == 0 && visible)
            visible = !hiz_occluded(args.hiz, args.vp0, c);
        const ViewBuffers out = view_outputs(base, v);
        if (pl.write_is_visible && in_range)
            out.is_visible[i] = visible ? 1 : 0;
        const unsigned long long word = __ballot(visible);
        if (lane == 0) {
            out.mask[(size_t)lb * (kCullBlock / 64) + wave] = word;
            wave_count[v][wave] = (uint32_t)__popcll(word);
        }
    }
    __syncthreads();
    if (threadIdx.x < nviews) {
        uint32_t total = 0;
#pragma unroll
        for (uint32_t w = 0; w < kCullBlock / 64; w++)
            total += wave_count[threadIdx.x][w];
        if (total) {
            uint32_t* counts = base->outs[threadIdx.x].chunk_count;  // (a lane-varying view: an ordinary load from the argument block)
            atomicAdd(&counts[lb / (kEmitChunk / kCullBlock)], total);
        }
    }
}

template <bool HIZ, uint32_t MAP, bool BOUNDS>
__global__ __launch_bounds__(kCullBlock) void cull_multi_kernel(const MultiCullArgs args)
{
    __shared__ uint32_t wave_count[kMaxBatchViews][kCullBlock / 64];
    cull_multi_block<HIZ, MAP, BOUNDS>(args, (ConstMultiArgs)__builtin_amdgcn_kernarg_segment_ptr(), blockIdx.x, wave_count);
}

// ------------------------------------------------------------------------------------------------
// Table-driven forms for a TICK of engine-sized pools (gv_cull_batch_begin): the culls of ALL mesh systems of a frame —
// each with its main camera and shadow passes — in ONE launch (blockIdx.y = job), their emits in ONE launch
// (blockIdx.y = (job, view)). The job descriptors live in a device table; the kernels read it through the constant
// address space (uniform index -> scalar loads, exactly what a by-value kernel argument compiles to).
// ------------------------------------------------------------------------------------------------
typedef ConstMultiArgs ConstCullTable;

__global__ __launch_bounds__(kCullBlock) void cull_table_kernel(const MultiCullArgs* __restrict__ table)
{
    __shared__ uint32_t wave_count[kMaxBatchViews][kCullBlock / 64];
    MultiCullArgs args;  // (a plain struct copy cannot read from an address-space-qualified source: memcpy can)
    __builtin_memcpy(&args, (ConstCullTable)table + blockIdx.y, sizeof(args));
    if (blockIdx.x * kCullBlock >= args.mesh.count)
        return;  // the grid is as wide as the largest pool of the tick
#define GV_TABLE_CULL(HIZ)                                                                              \
    switch (args.mesh.mapping) {                                                                        \
    case kMapExact: cull_multi_block<HIZ, kMapExact, false>(args, (ConstCullTable)table + blockIdx.y, blockIdx.x, wave_count); break;       \
    case kMapSpeculate: cull_multi_block<HIZ, kMapSpeculate, false>(args, (ConstCullTable)table + blockIdx.y, blockIdx.x, wave_count); break; \
    default: cull_multi_block<HIZ, kMapGeneral, false>(args, (ConstCullTable)table + blockIdx.y, blockIdx.x, wave_count); break;            \
    }
    if (args.use_hiz0) {
        GV_TABLE_CULL(true)
    } else {
        GV_TABLE_CULL(false)
    }
#undef GV_TABLE_CULL
}

hipError_t launch_cull_table(const void* device_table, uint32_t jobs, uint32_t max_slots, hipStream_t stream)
{
    if (jobs == 0 || max_slots == 0)
        return hipSuccess;
    hipLaunchKernelGGL(cull_table_kernel, dim3((max_slots + kCullBlock - 1) / kCullBlock, jobs), dim3(kCullBlock), 0, stream,
                       static_cast<const MultiCullArgs*>(device_table));
    return hipGetLastError();
}

static void fill_multi_args(MultiCullArgs& a, const MeshMirror& mesh, const TransformMirror& xf, const HizDevice& hiz,
                            const ViewParams* views, const ViewBuffers* outs, uint32_t nviews)
{
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
    a.bounds = BlockBounds{};
}

size_t cull_table_entry_bytes() { return sizeof(MultiCullArgs); }
void fill_cull_table_entry(void* entry, const MeshMirror& mesh, const TransformMirror& xf, const HizDevice& hiz,
                           const ViewParams* views, const ViewBuffers* outs, uint32_t nviews)
{
    fill_multi_args(*static_cast<MultiCullArgs*>(entry), mesh, xf, hiz, views, outs, nviews);
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
    fill_multi_args(a, mesh, xf, hiz, views, outs, nviews);
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
    const float4* world;    // world matrices of every transform entry (3 float4 each) when a sweep of the CURRENT mirror
                            // has just written them (gv_sweep / the fused sweep + cull), else NULL
    const EmitSeed* seeds;  // one 64-byte record per mirror entry of a flat, exactly paired pool at rest (gv_kernels.hpp), else NULL
};

// The camera-relative model (bakedModel) of visible mirror entry i (mesh.cpp:169-173). Visible entries passed every
// filter in K1: only the transform entry and its model are needed here. When the world matrices of this mirror are
// resident (args.world) the record takes world[slot] — the very product chain_model would rebuild, already in HBM as
// 48 contiguous bytes — instead of re-walking the parent chain (4 x (32 + 8 + 4) bytes of dependent gathers at depth 3).
__device__ __forceinline__ Mat34 record_model(const EmitArgs& args, uint32_t i, bool use_seed)
{
    // every gather here is a sparse 64-byte fetch for a few useful bytes: the flag byte is only read when the pool has
    // chains at all
    const bool chains = args.xf.max_depth != 0;  // uniform
    Mat34 world;
    if (use_seed) {  // workgroup-uniform: one sector holds all of it (flat + exactly paired: no chain, slot == i)
        const EmitSeed* s = args.seeds + i;
        const float4 a = s->a, b = s->b;
        const float2 c = s->c;
        world = calc_model(a.x, a.y, a.z, b.x, b.y, b.z, b.w, a.w, c.x, c.y);
    } else if (args.world) {  // uniform
        uint32_t slot = i;
        if (args.mesh.mapping != kMapExact)
            slot = args.mesh.link[i] & kSlotMask;
        const float4* w = args.world + (size_t)slot * 3;
        const float4 w0 = w[0], w1 = w[1], w2 = w[2];
        world.c0x = w0.x; world.c0y = w0.y; world.c0z = w0.z;
        world.c1x = w0.w; world.c1y = w1.x; world.c1z = w1.y;
        world.c2x = w1.z; world.c2y = w1.w; world.c2z = w2.x;
        world.c3x = w2.y; world.c3y = w2.z; world.c3z = w2.w;
    } else {
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
        world = chain_model(args.xf, local_model(rec), slot, rec.flags);
    }
    return translated(world, args.view.cam[0], args.view.cam[1], args.view.cam[2]);
}

__device__ __forceinline__ float record_distance(const EmitArgs& args, const Mat34& m)
{
    const float tx = m.c3x + args.view.cam_offset[0];
    const float ty = m.c3y + args.view.cam_offset[1];
    const float tz = m.c3z + args.view.cam_offset[2];
    return args.view.distance_2d ? m.c3z + 1.0f : fmaf(tz, tz, fmaf(ty, ty, tx * tx));
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

constexpr uint32_t kSelfPrefixLoads = (kSelfPrefixMaxChunks / 4 + 191) / 192;  // uint4 loads per lane of waves 1-3

// Four workgroups per 4096-slot chunk, each owning 16 of its 64 ballot words. Every workgroup prefix-sums the
// chunk's 64 words (512 B, L2), then lane r takes the r-th visible slot of its quarter (binary search over the
// word prefix + select of the k-th set bit), so the model recompute and the 56-byte record store run on dense
// waves and only the visible fraction costs instructions. Output rank = chunk base + r: ascending slot order,
// whatever order the workgroups run in.
// SELF: no scan launch in front — while wave 0 prefixes the ballot words, waves 1-3 sum the chunk totals below this
// chunk (a few KB of L2 reads) to get its base; workgroup 0 also writes the grand total and clears the OTHER totals
// buffer for the next frame's cull (the two buffers alternate, so nobody is still reading the one being cleared).
#ifndef GV_EMIT_PIPELINED  // (A/B builds: tools/ab_lib.sh)
#define GV_EMIT_PIPELINED 1
#endif
template <bool SELF>
__device__ __forceinline__ void emit_block(const EmitArgs& args, const uint32_t block)
{
    __shared__ unsigned long long words[64];
    __shared__ uint32_t prefix[65];
    __shared__ uint32_t below[4];
    __shared__ float4 stage[256 * 3];  // the round's 256 models, record-major
    const uint32_t chunk = block / kEmitParts, part = block % kEmitParts;
    const uint32_t first_word = chunk * 64;
    const uint32_t total_words = ((args.mesh.count + kCullBlock - 1) / kCullBlock) * (kCullBlock / 64);
    // An EMPTY chunk (behind an occlusion pass most are: cfg3 keeps 3 % of the entities, in a fifth of the chunks) has no record
    // to emit; its isVisible bytes are zeros, which they already are unless the quarter's flag says otherwise. Such a workgroup
    // leaves after one load instead of walking the whole chain of dependent loads (ballot words -> prefix -> barrier -> ...):
    // the launch was bound by 9 768 workgroups taking turns at that chain, not by bytes. Workgroup 0 keeps its global duties.
    if (SELF && args.out.vis_flags && block != 0 && args.out.chunk_count[chunk] == 0) {  // workgroup-uniform
        const bool dirty = args.view.write_is_visible && args.out.vis_flags[block];
        __syncthreads();  // every wave has read the flag before thread 0 may clear it (a late wave would otherwise see "clean" and skip its stores)
        if (!dirty)
            return;
        const uint32_t slot = chunk * kEmitChunk + part * (kEmitChunk / kEmitParts) + 4u * threadIdx.x;
        if (slot + 3u < args.mesh.count) {
            *reinterpret_cast<uint32_t*>(args.out.is_visible + slot) = 0u;
        } else {
            for (uint32_t k = 0; slot + k < args.mesh.count; k++)
                args.out.is_visible[slot + k] = 0;
        }
        // (a quarter that reaches past the pool's end is never marked clean: the bytes behind the end may be a larger pool's)
        if (threadIdx.x == 0 && chunk * kEmitChunk + (part + 1u) * (kEmitChunk / kEmitParts) <= args.mesh.count)
            args.out.vis_flags[block] = 0;
        return;
    }
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
        const uint32_t upto = block == 0 ? args.nchunks : chunk;  // workgroup 0: the grand total
        // all loads of the sum are issued before the first is consumed: one L2 round trip for up to 4096 totals, not
        // one per 192 of them (the serial form took ~13 dependent round trips for the last chunks of a 10 M pool)
        const uint4* __restrict__ totals4 = reinterpret_cast<const uint4*>(args.out.chunk_count);
        const uint32_t t = threadIdx.x - 64, groups = upto >> 2;
        uint4 v[kSelfPrefixLoads];
#pragma unroll
        for (uint32_t k = 0; k < kSelfPrefixLoads; k++) {
            const uint32_t q = t + 192 * k;
            v[k] = q < groups ? totals4[q] : make_uint4(0, 0, 0, 0);
        }
        uint32_t sum = t < (upto & 3u) ? args.out.chunk_count[(groups << 2) + t] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < kSelfPrefixLoads; k++)
            sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
#pragma unroll
        for (uint32_t d = 32; d >= 1; d >>= 1)
            sum += __shfl_xor(sum, d, 64);
        if ((threadIdx.x & 63u) == 0)
            below[threadIdx.x >> 6] = sum;
    }
    __syncthreads();
    if (args.view.write_is_visible) {
        // isVisible of this workgroup's quarter of the chunk (mesh.cpp:144,152,161,166), expanded from the ballot words:
        // lane t owns slots 4t .. 4t + 3 -> one 4-byte store, 1 KB of whole sectors per workgroup
        const uint32_t local = (block % kEmitParts) * (kEmitChunk / kEmitParts) + 4u * threadIdx.x;  // slot inside the chunk
        const uint32_t slot = chunk * kEmitChunk + local;
        if (slot < args.mesh.count) {
            const uint32_t nibble = (uint32_t)(words[local >> 6] >> (local & 63u)) & 15u;
            const uint32_t bytes = (nibble & 1u) | ((nibble & 2u) << 7) | ((nibble & 4u) << 14) | ((nibble & 8u) << 21);
            if (slot + 3u < args.mesh.count) {
                *reinterpret_cast<uint32_t*>(args.out.is_visible + slot) = bytes;
            } else {
                for (uint32_t k = 0; slot + k < args.mesh.count; k++)
                    args.out.is_visible[slot + k] = (uint8_t)((bytes >> (8u * k)) & 1u);
            }
        }
        if (SELF && args.out.vis_flags && threadIdx.x == 0)  // does this quarter now hold a non-zero byte (or bytes this pool does not reach)?
            args.out.vis_flags[block] = prefix[(part + 1) * (64 / kEmitParts)] != prefix[part * (64 / kEmitParts)] ||
                                                chunk * kEmitChunk + (part + 1u) * (kEmitChunk / kEmitParts) > args.mesh.count
                                            ? 1 : 0;
    }
    uint32_t base;
    if (SELF) {
        base = below[1] + below[2] + below[3];
        if (block == 0) {
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
    // Seeds pay where the visible entries are SPARSE (behind an occlusion pass: one sector instead of three per record); in a
    // dense chunk neighbouring records share the sectors of the streams and the 64-byte seeds move more, measured: at 21 %
    // visible the emit took 59.9 us with seeds against 45.8 us without, at 12 % (cfg3's in-frustum chunks) 18.4 against 21.4.
    const bool use_seed = args.seeds != nullptr && prefix[64] <= kEmitChunk * 5 / 32;
    // the record r of this workgroup's quarter -> its mirror entry (binary search over the quarter's words + k-th set bit)
    auto entry_of = [&](uint32_t r) {
        uint32_t lo = wlo, hi = whi;  // word w in [wlo, whi) with prefix[w] <= r < prefix[w + 1]
#pragma unroll
        for (uint32_t span = 64 / kEmitParts; span > 1; span >>= 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (prefix[mid] <= r)
                lo = mid;
            else
                hi = mid;
        }
        return (first_word + lo) * 64 + select_bit(words[lo], r - prefix[lo]);
    };
    // DENSE chunks of a flat, exactly paired pool (no seeds, no world matrices, no chains: the record is the local model of entry
    // i): the gathers of round k + 1 are issued before round k's records go through LDS and out, so that a round waits for its
    // stores and the next round's loads together instead of one after the other (a quarter of a full chunk is four rounds). Same-box A/B: emit 53.5 -> 51.2 us at 2.6 M records, 45.3 -> 43.8 us at 2.06 M (a barrier that
    // orders LDS only, so that the gathers fly across it, measured the same).
    const bool pipelined = GV_EMIT_PIPELINED && !use_seed && !args.world && args.xf.max_depth == 0 &&
                           args.mesh.mapping == kMapExact;  // uniform
    if (pipelined) {
        uint32_t r0 = prefix[wlo];
        uint32_t i_cur = 0xFFFFFFFFu, orig_cur = 0;
        float4 a_cur = {}, b_cur = {};
        float2 c_cur = {};
        if (r0 + threadIdx.x < total) {
            i_cur = entry_of(r0 + threadIdx.x);
            a_cur = args.xf.ab[i_cur].a;
            b_cur = args.xf.ab[i_cur].b;
            c_cur = args.xf.c[i_cur];
            orig_cur = args.mesh.orig ? args.mesh.orig[i_cur] : i_cur;
        }
        for (; r0 < total; r0 += 256) {
            uint32_t i_nxt = 0xFFFFFFFFu, orig_nxt = 0;
            float4 a_nxt = {}, b_nxt = {};
            float2 c_nxt = {};
            if (r0 + 256u + threadIdx.x < total) {
                i_nxt = entry_of(r0 + 256u + threadIdx.x);
                a_nxt = args.xf.ab[i_nxt].a;
                b_nxt = args.xf.ab[i_nxt].b;
                c_nxt = args.xf.c[i_nxt];
                orig_nxt = args.mesh.orig ? args.mesh.orig[i_nxt] : i_nxt;
            }
            if (i_cur != 0xFFFFFFFFu) {
                XfRecord rec;
                rec.a = a_cur, rec.b = b_cur, rec.c = c_cur, rec.flags = 0u;
                const Mat34 m = translated(local_model(rec), args.view.cam[0], args.view.cam[1], args.view.cam[2]);
                const size_t rank = (size_t)base + r0 + threadIdx.x;
                args.out.visible_idx[rank] = orig_cur;
                args.out.distance_sq[rank] = record_distance(args, m);
                float4* row = stage + threadIdx.x * 3;
                row[0] = make_float4(m.c0x, m.c0y, m.c0z, m.c1x);
                row[1] = make_float4(m.c1y, m.c1z, m.c2x, m.c2y);
                row[2] = make_float4(m.c2z, m.c3x, m.c3y, m.c3z);
            }
            __syncthreads();
            const uint32_t quads = min(256u, total - r0) * 3u;
            float4* dst = reinterpret_cast<float4*>(args.out.baked_model) + ((size_t)base + r0) * 3;
            for (uint32_t q = threadIdx.x; q < quads; q += 256)
                dst[q] = stage[q];
            __syncthreads();  // the stage is rewritten by the next round
            i_cur = i_nxt, orig_cur = orig_nxt, a_cur = a_nxt, b_cur = b_nxt, c_cur = c_nxt;
        }
        return;
    }
    // 256 consecutive output records per round (uniform trip count). Each lane builds one record; the 48-byte models go
    // through LDS so that they leave as whole rows — lane k stores float4 k, k + 256, k + 512 of the round's contiguous
    // 12 KB — instead of three 16-byte pieces per lane at a 48-byte stride.
    for (uint32_t r0 = prefix[wlo]; r0 < total; r0 += 256) {
        const uint32_t r = r0 + threadIdx.x;
        if (r < total) {
            uint32_t lo = wlo, hi = whi;  // word w in [wlo, whi) with prefix[w] <= r < prefix[w + 1]
#pragma unroll
            for (uint32_t span = 64 / kEmitParts; span > 1; span >>= 1) {  // log2(words per workgroup) halvings
                const uint32_t mid = (lo + hi) >> 1;
                if (prefix[mid] <= r)
                    lo = mid;
                else
                    hi = mid;
            }
            const uint32_t pos = select_bit(words[lo], r - prefix[lo]);
            const uint32_t i = (first_word + lo) * 64 + pos;
            const Mat34 m = record_model(args, i, use_seed);
            const size_t rank = (size_t)base + r;
            // pool slot: componentOffset = slot * componentSize  mesh.cpp:170
            args.out.visible_idx[rank] = use_seed ? args.seeds[i].orig : (args.mesh.orig ? args.mesh.orig[i] : i);
            args.out.distance_sq[rank] = record_distance(args, m);
            float4* row = stage + threadIdx.x * 3;
            row[0] = make_float4(m.c0x, m.c0y, m.c0z, m.c1x);
            row[1] = make_float4(m.c1y, m.c1z, m.c2x, m.c2y);
            row[2] = make_float4(m.c2z, m.c3x, m.c3y, m.c3z);
        }
        __syncthreads();
        const uint32_t quads = min(256u, total - r0) * 3u;
        float4* dst = reinterpret_cast<float4*>(args.out.baked_model) + ((size_t)base + r0) * 3;
        for (uint32_t q = threadIdx.x; q < quads; q += 256)
            dst[q] = stage[q];
        __syncthreads();  // the stage is rewritten by the next round
    }
}

template <bool SELF>
__global__ __launch_bounds__(256) void emit_kernel(const EmitArgs args)
{
    emit_block<SELF>(args, blockIdx.x);
}

// The views of one batched cull (main camera + shadow passes over a small pool, where every launch counts) emitted by
// ONE launch: blockIdx.y picks the view, the pool-side arguments are shared.
struct EmitBatchArgs {
    MeshMirror mesh;
    TransformMirror xf;
    const float4* world;
    uint32_t nchunks;
    uint32_t clear_chunks[kMaxBatchViews];
    ViewParams view[kMaxBatchViews];
    ViewBuffers out[kMaxBatchViews];
};
__global__ __launch_bounds__(256) void emit_batch_kernel(const EmitBatchArgs batch)
{
    EmitArgs args;
    args.mesh = batch.mesh;
    args.xf = batch.xf;
    args.view = batch.view[blockIdx.y];
    args.out = batch.out[blockIdx.y];
    args.nchunks = batch.nchunks;
    args.clear_chunks = batch.clear_chunks[blockIdx.y];
    args.world = batch.world;
    args.seeds = nullptr;
    emit_block<true>(args, blockIdx.x);
}

// one EmitArgs per (job, view) of a tick; blockIdx.y picks the entry
typedef const EmitArgs __attribute__((address_space(4))) * ConstEmitTable;
__global__ __launch_bounds__(256) void emit_table_kernel(const EmitArgs* __restrict__ table)
{
    EmitArgs args;
    __builtin_memcpy(&args, (ConstEmitTable)table + blockIdx.y, sizeof(args));
    if (blockIdx.x >= args.nchunks * kEmitParts)
        return;  // the grid is as wide as the largest pool of the tick
    emit_block<true>(args, blockIdx.x);
}

size_t emit_table_entry_bytes() { return sizeof(EmitArgs); }
void fill_emit_table_entry(void* entry, const MeshMirror& mesh, const TransformMirror& xf, const ViewParams& vp,
                           const ViewBuffers& out, uint32_t clear_chunks, const float4* world)
{
    EmitArgs& a = *static_cast<EmitArgs*>(entry);
    a.mesh = mesh;
    a.xf = xf;
    a.view = vp;
    a.out = out;
    a.nchunks = (mesh.count + kEmitChunk - 1) / kEmitChunk;
    a.clear_chunks = clear_chunks;
    a.world = world;
    a.seeds = nullptr;
}

hipError_t launch_emit_table(const void* device_table, uint32_t entries, uint32_t max_slots, hipStream_t stream)
{
    if (entries == 0 || max_slots == 0)
        return hipSuccess;
    const uint32_t chunks = (max_slots + kEmitChunk - 1) / kEmitChunk;
    hipLaunchKernelGGL(emit_table_kernel, dim3(chunks * kEmitParts, entries), dim3(256), 0, stream, static_cast<const EmitArgs*>(device_table));
    return hipGetLastError();
}

hipError_t launch_emit_batch(const MeshMirror& mesh, const TransformMirror& xf, const ViewParams* views, const ViewBuffers* outs,
                             const uint32_t* clear_chunks, uint32_t nviews, hipStream_t stream, const float4* world)
{
    if (mesh.count == 0 || nviews == 0)
        return hipSuccess;
    EmitBatchArgs a;
    a.mesh = mesh;
    a.xf = xf;
    a.world = world;
    a.nchunks = (mesh.count + kEmitChunk - 1) / kEmitChunk;
    for (uint32_t v = 0; v < kMaxBatchViews; v++) {
        const uint32_t k = v < nviews ? v : 0;
        a.view[v] = views[k];
        a.out[v] = outs[k];
        a.clear_chunks[v] = clear_chunks[k];
    }
    hipLaunchKernelGGL(emit_batch_kernel, dim3(a.nchunks * kEmitParts, nviews), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_emit(const MeshMirror& mesh, const TransformMirror& xf, const ViewParams& vp, const ViewBuffers& out,
                       hipStream_t stream, bool self_prefix, uint32_t clear_chunks, const float4* world, const EmitSeed* seeds)
{
    if (mesh.count == 0)
        return hipSuccess;
    EmitArgs a;
    a.world = world;
    a.seeds = seeds;
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

__global__ __launch_bounds__(256) void emit_seeds_kernel(const MeshMirror mesh, const TransformMirror xf, EmitSeed* __restrict__ seeds)
{
    emit_seed_of(mesh, xf, seeds, blockIdx.x * 256 + threadIdx.x);
}

hipError_t launch_emit_seeds(const MeshMirror& mesh, const TransformMirror& xf, EmitSeed* seeds, hipStream_t stream)
{
    if (mesh.count == 0)
        return hipSuccess;
    hipLaunchKernelGGL(emit_seeds_kernel, dim3((mesh.count + 255) / 256), dim3(256), 0, stream, mesh, xf, seeds);
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
                                                             const uint32_t* __restrict__ xinv, XfAB* __restrict__ ab,
                                                             float2* __restrict__ c, uint8_t* __restrict__ flags,
                                                             uint8_t* __restrict__ dirty)
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
    ab[j].a = make_float4(pos[0], pos[1], pos[2], scl[0]);
    ab[j].b = make_float4(rot[0], rot[1], rot[2], rot[3]);
    c[j] = make_float2(scl[1], scl[2]);
    flags[j] = f;
    if (dirty)
        dirty[j] = 1;
}

hipError_t launch_aos_transforms(const uint8_t* raw, const AosTransformLayout& layout, uint32_t first, uint32_t count,
                                 const uint32_t* xinv, XfAB* ab, float2* c, uint8_t* flags, uint8_t* dirty, hipStream_t stream)
{
    if (count == 0)
        return hipSuccess;
    hipLaunchKernelGGL(aos_transforms_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, raw, layout, first, count, xinv,
                       ab, c, flags, dirty);
    return hipGetLastError();
}

// Dirty MeshRenderComponents shipped as raw AoS bytes (slots [first, first + count) of the caller's pool): the gather the host
// otherwise does (gather_meshes, gv_mirror.cpp) — Manager::tryGet<TransformComponent>(entity) through the entity -> slot table
// (mesh.cpp:149), the candidate rule (mesh.cpp:142), the empty box of a non-candidate — at HBM speed. *demoted is set when a
// candidate no longer pairs with its own mirror index (a pool mapped kMapExact must then be demoted by the host).
__global__ __launch_bounds__(256) void aos_meshes_kernel(const uint8_t* __restrict__ raw, AosMeshLayout L, uint32_t first, uint32_t count,
                                                         const uint32_t* __restrict__ inv, const uint32_t* __restrict__ e2t,
                                                         uint32_t entity_capacity, uint32_t xf_occupancy, const uint32_t* __restrict__ xinv,
                                                         float4* __restrict__ a, float2* __restrict__ b, uint32_t* __restrict__ link,
                                                         uint32_t* __restrict__ demoted)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count)
        return;
    const uint8_t* m = raw + (size_t)k * L.stride;
    float mn[3], mx[3];
    uint32_t entity;
    memcpy(mn, m + L.aabb_min, 12);
    memcpy(mx, m + L.aabb_max, 12);
    memcpy(&entity, m + L.entity, 4);
    uint32_t slot = kSlotNone;
    if (entity != 0 && entity < entity_capacity) {
        const uint32_t s = e2t[entity];
        if (s != 0xFFFFFFFFu && s < xf_occupancy)
            slot = xinv ? xinv[s] : s;
    }
    const bool candidate = entity != 0 && m[L.is_enabled] != 0 && slot != kSlotNone;
    const uint32_t i = first + k;
    const uint32_t j = inv ? inv[i] : i;
    a[j] = candidate ? make_float4(mn[0], mn[1], mn[2], mx[0]) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    b[j] = candidate ? make_float2(mx[1], mx[2]) : make_float2(0.0f, 0.0f);
    link[j] = slot | (candidate ? kMeshCandidate : 0u);
    if (candidate && slot != j)
        atomicOr(demoted, 1u);
}

hipError_t launch_aos_meshes(const uint8_t* raw, const AosMeshLayout& layout, uint32_t first, uint32_t count, const uint32_t* inv,
                             const uint32_t* e2t, uint32_t entity_capacity, uint32_t xf_occupancy, const uint32_t* xinv, float4* a, float2* b,
                             uint32_t* link, uint32_t* demoted, hipStream_t stream)
{
    if (count == 0)
        return hipSuccess;
    hipLaunchKernelGGL(aos_meshes_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, raw, layout, first, count, inv, e2t, entity_capacity,
                       xf_occupancy, xinv, a, b, link, demoted);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void mark_bytes_kernel(const uint32_t* __restrict__ idx, uint32_t count, uint8_t* __restrict__ dst)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x)
        dst[idx[k]] = 1;
}

hipError_t launch_mark_bytes(const uint32_t* idx, uint32_t count, uint8_t* dst, hipStream_t stream)
{
    if (count == 0)
        return hipSuccess;
    hipLaunchKernelGGL(mark_bytes_kernel, dim3(min((count + 255u) / 256u, 4096u)), dim3(256), 0, stream, idx, count, dst);
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
    case 32: hipLaunchKernelGGL(scatter_kernel<XfAB>, grid, block, 0, stream, idx, count, (const XfAB*)src, (XfAB*)dst); break;
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

// `map` (gv_pool_set_index_map; may be NULL): pool slot -> the caller's global id (a spatial tile's slots are not a
// contiguous range of the world's), applied before `base` is added.
__global__ __launch_bounds__(256) void copy_idx_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ count,
                                                       uint32_t* __restrict__ dst, uint32_t capacity, uint32_t base,
                                                       const uint32_t* __restrict__ map)
{
    const uint32_t n = min(*count, capacity);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dst[i] = (map ? map[src[i]] : src[i]) + base;
}

hipError_t launch_copy_idx(const uint32_t* src, const uint32_t* count, uint32_t* dst, uint32_t capacity, uint32_t base,
                           const uint32_t* map, hipStream_t stream)
{
    hipLaunchKernelGGL(copy_idx_kernel, dim3(2048), dim3(256), 0, stream, src, count, dst, capacity, base, map);
    return hipGetLastError();
}

// exchange shard: [draw_count, idx + base ...]; the header is the true count even when it exceeds `capacity`
__global__ __launch_bounds__(256) void copy_shard_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ count,
                                                         uint32_t* __restrict__ dst, uint32_t capacity, uint32_t base,
                                                         const uint32_t* __restrict__ map)
{
    const uint32_t total = *count, n = min(total, capacity);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        dst[0] = total;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dst[1 + i] = (map ? map[src[i]] : src[i]) + base;
}

hipError_t launch_copy_shard(const uint32_t* src, const uint32_t* count, uint32_t* dst, uint32_t capacity, uint32_t base,
                             const uint32_t* map, hipStream_t stream)
{
    const uint32_t blocks = std::max(1u, std::min(2048u, (capacity + 255u) / 256u));
    hipLaunchKernelGGL(copy_shard_kernel, dim3(blocks), dim3(256), 0, stream, src, count, dst, capacity, base, map);
    return hipGetLastError();
}

// Small pools: every result of every view of the last cull goes to the caller-visible pinned host buffers in ONE
// launch (blockIdx.y = view) — the count, the records [0, count) and (main pass) the isVisible bytes — so gv_results_fetch is one launch and one
// stream synchronisation instead of a count read-back, four copies and a second synchronisation. 16-byte stores:
// consecutive lanes fill whole PCIe write bursts.
// Records [first, first + 64) of a view in the caller's struct layout (GvRecordLayout): built in LDS by the first 64 lanes,
// written out by the whole workgroup as contiguous 16-byte pieces (whole PCIe bursts when dst is host memory).
__device__ __forceinline__ void pack_records_chunk(uint4* stage, const uint32_t* __restrict__ idx, const float* __restrict__ model,
                                                   const float* __restrict__ dist, const RecordLayout& L, uint32_t first, uint32_t n,
                                                   uint8_t* __restrict__ dst)
{
    const uint32_t live = min(64u, n - first), quads = L.stride >> 4;
    for (uint32_t q = threadIdx.x; q < live * quads; q += 256)
        stage[q] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (threadIdx.x < live) {
        const uint32_t j = first + threadIdx.x;
        uint8_t* rec = reinterpret_cast<uint8_t*>(stage) + threadIdx.x * L.stride;
        const unsigned long long offset = (unsigned long long)record_slot(L, idx[j]) * L.component_stride;  // componentOffset  mesh.cpp:170
        uint32_t* w = reinterpret_cast<uint32_t*>(rec + L.component_offset);
        w[0] = (uint32_t)offset;
        w[1] = (uint32_t)(offset >> 32);
        float* bm = reinterpret_cast<float*>(rec + L.baked_model);
#pragma unroll
        for (int k = 0; k < 12; k++)
            bm[k] = model[(size_t)j * 12 + k];
        *reinterpret_cast<float*>(rec + L.distance_sq) = dist[j];
        if (L.buffer_index != 0xFFFFFFFFu)
            *reinterpret_cast<uint32_t*>(rec + L.buffer_index) = L.buffer_index_value;
    }
    __syncthreads();
    uint4* out = reinterpret_cast<uint4*>(dst + (size_t)first * L.stride);
    for (uint32_t q = threadIdx.x; q < live * quads; q += 256)
        out[q] = stage[q];
    __syncthreads();  // the stage is reused by the next chunk
}

__global__ __launch_bounds__(256) void pack_records_kernel(const uint32_t* __restrict__ count, const uint32_t* __restrict__ idx,
                                                           const float* __restrict__ model, const float* __restrict__ dist,
                                                           const RecordLayout L, uint32_t capacity, uint8_t* __restrict__ dst)
{
    __shared__ uint4 stage[64 * kMaxRecordStride / 16];
    const uint32_t n = min(*count, capacity);
    for (uint32_t first = blockIdx.x * 64; first < n; first += gridDim.x * 64)  // workgroup-uniform
        pack_records_chunk(stage, idx, model, dist, L, first, n, dst);
}

hipError_t launch_pack_records(const uint32_t* count, const uint32_t* idx, const float* model, const float* dist, const RecordLayout& layout,
                               uint32_t capacity, uint8_t* dst, hipStream_t stream)
{
    if (capacity == 0)
        return hipSuccess;
    const uint32_t blocks = std::max(1u, std::min(2048u, (capacity + 63u) / 64u));
    hipLaunchKernelGGL(pack_records_kernel, dim3(blocks), dim3(256), 0, stream, count, idx, model, dist, layout, capacity, dst);
    return hipGetLastError();
}

// One view's results into its pinned host buffers; `block` of `nblocks` workgroups share the copies.
__device__ __forceinline__ void publish_block(const PublishArgs& a, const uint32_t block, const uint32_t nblocks)
{
    const uint32_t n = *a.count;
    const uint32_t tid = block * 256 + threadIdx.x, threads = nblocks * 256;
    if (tid == 0)
        *a.host_count = n;
    if (a.host_records) {  // the caller's own record structs instead of the three arrays
        __shared__ uint4 stage[64 * kMaxRecordStride / 16];
        for (uint32_t first = block * 64; first < n; first += nblocks * 64)  // workgroup-uniform
            pack_records_chunk(stage, a.idx, a.model, a.dist, a.layout, first, n, a.host_records);
    } else if (a.host_idx) {
        for (uint32_t j = tid; j < n; j += threads) {
            a.host_idx[j] = a.idx[j];
            a.host_dist[j] = a.dist[j];
        }
        const float4* __restrict__ src = reinterpret_cast<const float4*>(a.model);
        float4* __restrict__ dst = reinterpret_cast<float4*>(a.host_model);
        for (uint32_t q = tid; q < 3 * n; q += threads)
            dst[q] = src[q];
    }
    if (a.host_is_visible && a.orig) {
        // spatially ordered mirror: workgroup 0 puts the bytes back into pool-slot order in LDS (the pool is at most
        // kPublishMaxSlots bytes) and writes them out as whole words — no scattered single-byte stores over PCIe
        if (block != 0)
            return;
        __shared__ uint8_t slots[(kPublishLdsSlots + 3u) & ~3u];
        for (uint32_t j = threadIdx.x; j < a.occupancy; j += 256)
            slots[a.orig[j]] = a.is_visible[j];
        __syncthreads();
        const uint32_t words = a.occupancy >> 2;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(slots);
        uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(a.host_is_visible);
        for (uint32_t w = threadIdx.x; w < words; w += 256)
            dst[w] = src[w];
        for (uint32_t j = (words << 2) + threadIdx.x; j < a.occupancy; j += 256)
            a.host_is_visible[j] = slots[j];
    } else if (a.host_is_visible) {
        const uint32_t words = a.occupancy >> 2;
        const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(a.is_visible);
        uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(a.host_is_visible);
        for (uint32_t w = tid; w < words; w += threads)
            dst[w] = src[w];
        for (uint32_t j = (words << 2) + tid; j < a.occupancy; j += threads)
            a.host_is_visible[j] = a.is_visible[j];
    }
}

__global__ __launch_bounds__(256) void publish_kernel(const PublishBatch batch)
{
    publish_block(batch.view[blockIdx.y], blockIdx.x, gridDim.x);
}
// isVisible bytes of a spatially ordered mirror back into pool-slot order (dst[orig[j]] = src[j]) on the device, where a
// random byte scatter is cheap; the host then only streams them into the components
__global__ __launch_bounds__(256) void unpermute_bytes_kernel(const uint8_t* __restrict__ src, const uint32_t* __restrict__ orig,
                                                              uint32_t count, uint8_t* __restrict__ dst)
{
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < count; j += gridDim.x * 256)
        dst[orig[j]] = src[j];
}
hipError_t launch_unpermute_bytes(const uint8_t* src, const uint32_t* orig, uint32_t count, uint8_t* dst, hipStream_t stream)
{
    const uint32_t blocks = std::max(1u, std::min(4096u, (count + 255u) / 256u));
    hipLaunchKernelGGL(unpermute_bytes_kernel, dim3(blocks), dim3(256), 0, stream, src, orig, count, dst);
    return hipGetLastError();
}

hipError_t launch_publish(const PublishBatch& batch, uint32_t views, uint32_t occupancy, hipStream_t stream)
{
    const uint32_t blocks = std::max(1u, std::min(256u, (occupancy + 255u) / 256u));
    hipLaunchKernelGGL(publish_kernel, dim3(blocks, views), dim3(256), 0, stream, batch);
    return hipGetLastError();
}

}  // namespace gv
