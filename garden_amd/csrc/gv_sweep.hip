// gv_sweep.hip — world-matrix sweeps (TransformComponent::calcModel for every slot, transform.hpp:197-214): VALU and
// MFMA forms, alone and fused with the cull of an exactly paired pool.
#include "gv_device.hpp"

namespace gv {

// ------------------------------------------------------------------------------------------------
// world-matrix sweep (camera = 0), VALU form: one lane per transform slot
// ------------------------------------------------------------------------------------------------
// The 256 world matrices of a tile leave as whole rows: every lane parks its 48 bytes in LDS and lane k then stores float4 k,
// k + 256 and k + 512 of the tile's contiguous 12 KB — three 16-byte stores per lane at a 48-byte stride are un-coalesced per
// instruction (the emit kernel gained 24 % from the same change, profiles/r02b_emit_ab.txt). `rows`: live slots of the tile
// (256, less in the last one); every thread of the workgroup must call it (it synchronises).
__device__ __forceinline__ void store_world_tile(float4* stage /* LDS [768] */, float4* __restrict__ world, uint32_t lb, uint32_t rows,
                                                 float4 w0, float4 w1, float4 w2)
{
    float4* mine = stage + threadIdx.x * 3;
    mine[0] = w0;
    mine[1] = w1;
    mine[2] = w2;
    __syncthreads();
    float4* __restrict__ dst = world + (size_t)lb * 768;
    const uint32_t quads = rows * 3;
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) {
        const uint32_t q = threadIdx.x + 256 * k;
        if (q < quads)
            stream_store(dst + q, stage[q]);
    }
}

__global__ __launch_bounds__(256) void sweep_valu_kernel(const TransformMirror xf, float4* __restrict__ world)
{
    __shared__ float4 stage[768];
    const uint32_t lb = blockIdx.x;
    const uint32_t s = lb * 256 + threadIdx.x;
    float4 w0 = make_float4(0, 0, 0, 0), w1 = w0, w2 = w0;
    if (s < xf.count) {
        const XfRecord r = stream_xf(xf, s);
        if (r.flags & kXfLive) {
            const Mat34 m = chain_model(xf, local_model(r), s, r.flags);
            w0 = make_float4(m.c0x, m.c0y, m.c0z, m.c1x);
            w1 = make_float4(m.c1y, m.c1z, m.c2x, m.c2y);
            w2 = make_float4(m.c2z, m.c3x, m.c3y, m.c3z);
        }
    }
    store_world_tile(stage, world, lb, min(256u, xf.count - lb * 256u), w0, w1, w2);
}

// ------------------------------------------------------------------------------------------------
// Subtree-scoped sweep (the reference recomputes lazily per calcModel call, transform.hpp:197-214, and
// marks nothing): `dirty[e]` is 1 for every mirror entry whose TRS / flags / parent link were re-mirrored since the
// world-matrix cache was last brought up to date. An entry's world matrix is stale iff its chain — itself or any
// ancestor — contains a dirty entry, so every lane walks its chain over the 1-byte flags and the 4-byte links (the
// ancestors' lines are shared by their whole subtree: cache hits) and only stale entries pay the TRS gathers, the
// products and the 48-byte store: 5-10 B per clean entry instead of 92. Same bits as the full sweeps (same chain_model).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sweep_subtree_kernel(const TransformMirror xf, const uint8_t* __restrict__ dirty,
                                                            float4* __restrict__ world)
{
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= xf.count)
        return;
    bool stale = dirty[s] != 0;
    if (xf.max_depth != 0 && !stale) {
        // (the walk ignores modelWithAncestors: an entry that does not use its chain is merely recomputed to the same value)
        uint32_t p = xf.parent[s];
        for (uint32_t d = 0; d < xf.max_depth && p != kSlotNone; d++) {
            if (dirty[p]) {
                stale = true;
                break;
            }
            p = xf.parent[p];
        }
    }
    if (!stale)
        return;
    const XfRecord r = load_xf(xf, s);
    float4 w0 = make_float4(0, 0, 0, 0), w1 = w0, w2 = w0;
    if (r.flags & kXfLive) {
        const Mat34 m = chain_model(xf, local_model(r), s, r.flags);
        w0 = make_float4(m.c0x, m.c0y, m.c0z, m.c1x);
        w1 = make_float4(m.c1y, m.c1z, m.c2x, m.c2y);
        w2 = make_float4(m.c2z, m.c3x, m.c3y, m.c3z);
    }
    world[(size_t)s * 3 + 0] = w0;
    world[(size_t)s * 3 + 1] = w1;
    world[(size_t)s * 3 + 2] = w2;
}

hipError_t launch_sweep_subtree(const TransformMirror& xf, const uint8_t* dirty, float4* world, hipStream_t stream)
{
    if (xf.count == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sweep_subtree_kernel, dim3((xf.count + 255) / 256), dim3(256), 0, stream, xf, dirty, world);
    return hipGetLastError();
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
// Per-wave LDS tile: 64 slots x 20 floats (16 used; the pitch makes the memory side's 16-byte stores conflict-free).
// Everything crosses it as whole float4s — three or four 16-byte LDS instructions where round 1 issued twelve 4-byte
// ones (the hand-over was ~150 LDS instructions per lane per slot at depth 3 and the reason the MFMA form lost to the
// VALU form when fused with the cull; now ~50):
//   self models go in COLUMN-major, padded with the bottom-row element: quad j = (c_j.x, c_j.y, c_j.z, j == 3) — lane
//   (e, q) of the matrix side reads quad q = its B operand / running product column in one load;
//   parent models go in ROW-major: quad i = (c0[i], c1[i], c2[i], c3[i]), quad 3 = the constant bottom row (0,0,0,1) —
//   lane (e, q) reads quad q = its four A operands in one load, no per-lane selects; "this slot has a parent" lives
//   in a compact per-wave word array beside the tile (conflict-free 4-byte accesses).
constexpr uint32_t kPitch4 = 5;            // float4s per slot

__device__ __forceinline__ void lds_put_columns(float4* slot, const Mat34& m)
{
    slot[0] = make_float4(m.c0x, m.c0y, m.c0z, 0.0f);
    slot[1] = make_float4(m.c1x, m.c1y, m.c1z, 0.0f);
    slot[2] = make_float4(m.c2x, m.c2y, m.c2z, 0.0f);
    slot[3] = make_float4(m.c3x, m.c3y, m.c3z, 1.0f);
}
__device__ __forceinline__ void lds_put_rows(float4* slot, const Mat34& m)
{
    slot[0] = make_float4(m.c0x, m.c1x, m.c2x, m.c3x);
    slot[1] = make_float4(m.c0y, m.c1y, m.c2y, m.c3y);
    slot[2] = make_float4(m.c0z, m.c1z, m.c2z, m.c3z);
    slot[3] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
}

// Each wave owns its LDS tile, so the matrix/memory-side hand-over only needs wave-level ordering: LDS operations of
// one wave execute in issue order; the fences keep the compiler from moving them across the hand-over.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// store_world_tile for ONE wave (no workgroup barrier, 3 KB of LDS that may be a region the wave no longer needs): lane k
// stores float4 k, k + 64 and k + 128 of the wave's 64 contiguous matrices. `rows`: live slots among the wave's 64.
__device__ __forceinline__ void store_world_wave(float4* wave_stage /* LDS [192] */, float4* __restrict__ world, uint32_t first_slot, uint32_t rows,
                                                 float4 w0, float4 w1, float4 w2)
{
    const uint32_t lane = threadIdx.x & 63u;
    wave_lds_sync();  // earlier reads of the region
    float4* mine = wave_stage + lane * 3;
    mine[0] = w0;
    mine[1] = w1;
    mine[2] = w2;
    wave_lds_sync();
    float4* __restrict__ dst = world + (size_t)first_slot * 3;
    const uint32_t quads = rows * 3;
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) {
        const uint32_t q = lane + 64 * k;
        if (q < quads)
            stream_store(dst + q, wave_stage[q]);
    }
}

// One ancestor step on the matrix side: x[r] = column q of the running product of slot 16r + e (w = bottom-row element),
// the tile holds the parents' local models row-major. Four issues k = 0..3 into a zero accumulator = the canonical
// fma chain. Slots whose chain has ended keep their product untouched (bit-exact, incl. -0). acc[3] (the bottom-row
// element) is not taken: models are affine and x[r].w stays the constant 0 / 1, as in every other implementation (it only
// differs from acc[3] when the operands are non-finite).
__device__ __forceinline__ void mfma_chain_step(const float4* tile4, const uint32_t* has_parent, uint32_t e, uint32_t q, float4 (&x)[4])
{
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float4 row = tile4[(16 * r + e) * kPitch4 + q];
        const bool step = has_parent[16 * r + e] != 0;
        f32x4_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(row.x, x[r].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(row.y, x[r].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(row.z, x[r].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(row.w, x[r].w, acc, 0, 0, 0);
        x[r].x = step ? acc[0] : x[r].x;
        x[r].y = step ? acc[1] : x[r].y;
        x[r].z = step ? acc[2] : x[r].z;
    }
}

__global__ __launch_bounds__(256) void sweep_mfma_kernel(const TransformMirror xf, float* __restrict__ world)
{
    __shared__ float4 tile[4][64 * kPitch4];  // one tile per wave
    __shared__ uint32_t has_parent[4][64];
    const uint32_t lb = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t q = lane & 3u, e = lane >> 2;  // matrix side: column/row q of slot 16r + e
    const uint32_t s = lb * 256 + threadIdx.x;    // memory side: this lane's slot
    float4* my_tile = tile[wave];
    uint32_t flags = 0;
    Mat34 m = {};
    const bool in_range = s < xf.count;
    if (in_range) {
        const XfRecord r = stream_xf(xf, s);
        flags = r.flags;
        m = local_model(r);
    }
    const bool live = in_range && (flags & kXfLive);
    lds_put_columns(my_tile + lane * kPitch4, m);
    wave_lds_sync();
    float4 x[4];  // [round]: column q of the product of slot 16r + e (w = bottom-row element)
#pragma unroll
    for (int r = 0; r < 4; r++)
        x[r] = my_tile[(16 * r + e) * kPitch4 + q];
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
        wave_lds_sync();  // the previous step's reads of the tile
        lds_put_rows(my_tile + lane * kPitch4, pm);
        has_parent[wave][lane] = has ? 1u : 0u;
        wave_lds_sync();
        mfma_chain_step(my_tile, has_parent[wave], e, q, x);
        p = next;
    }
    // liveness of slot 16r + e on the matrix side
    wave_lds_sync();
    has_parent[wave][lane] = live ? 1u : 0u;
    wave_lds_sync();
    // float4x3 order: column q's xyz at 12 floats per slot. The wave's 64 matrices are assembled in its (now free) tile and
    // leave as whole rows — lane k stores float4 k, k + 64, k + 128 of 3 KB — instead of three 4-byte stores per lane and round
    float* stage = reinterpret_cast<float*>(my_tile);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const bool ok = has_parent[wave][16 * r + e] != 0;
        float* dst = stage + (16 * r + e) * 12 + q * 3;
        dst[0] = ok ? x[r].x : 0.0f;
        dst[1] = ok ? x[r].y : 0.0f;
        dst[2] = ok ? x[r].z : 0.0f;
    }
    wave_lds_sync();
    const uint32_t wave_first = lb * 256 + wave * 64;
    if (wave_first < xf.count) {
        const uint32_t quads = min(64u, xf.count - wave_first) * 3;
        float4* __restrict__ out = reinterpret_cast<float4*>(world) + (size_t)wave_first * 3;
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) {
            const uint32_t i4 = lane + 64 * k;
            if (i4 < quads)
                stream_store(out + i4, my_tile[i4]);
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
    __shared__ float4 tile[4][64 * kPitch4];  // one tile per wave
    __shared__ uint32_t has_parent[4][64];
    __shared__ uint32_t wave_count[4];
    const TransformMirror& xf = args.cull.xf;
    const MeshMirror& mesh = args.cull.mesh;
    const uint32_t lb = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t q = lane & 3u, e = lane >> 2;  // matrix side: column/row q of slot 16r + e
    const uint32_t s = lb * 256 + threadIdx.x;    // memory side: this lane's slot = its mesh entry
    float4* my_tile = tile[wave];
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
    uint32_t p = (live && xf.max_depth != 0 && (flags & kXfWithAncestors)) ? xf.parent[s] : kSlotNone;
    Mat34 world = m;
    if (__any(p != kSlotNone)) {  // wave-uniform: a wave of roots / flat entries keeps its local models as they are
        lds_put_columns(my_tile + lane * kPitch4, m);
        wave_lds_sync();
        float4 x[4];
#pragma unroll
        for (int r = 0; r < 4; r++)
            x[r] = my_tile[(16 * r + e) * kPitch4 + q];
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
            wave_lds_sync();  // the previous step's reads of the tile
            lds_put_rows(my_tile + lane * kPitch4, pm);
            has_parent[wave][lane] = has ? 1u : 0u;
            wave_lds_sync();
            mfma_chain_step(my_tile, has_parent[wave], e, q, x);
            p = next;
        }
        // hand the products back: lane (e, q) holds column q of slot 16r + e
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 4; r++)
            my_tile[(16 * r + e) * kPitch4 + q] = x[r];
        wave_lds_sync();
        const float4* slot = my_tile + lane * kPitch4;
        const float4 k0 = slot[0], k1 = slot[1], k2 = slot[2], k3 = slot[3];
        world.c0x = k0.x; world.c0y = k0.y; world.c0z = k0.z;
        world.c1x = k1.x; world.c1y = k1.y; world.c1z = k1.z;
        world.c2x = k2.x; world.c2y = k2.y; world.c2z = k2.z;
        world.c3x = k3.x; world.c3y = k3.y; world.c3z = k3.z;
    }
    {
        float4 w0 = make_float4(0, 0, 0, 0), w1 = w0, w2 = w0;
        if (live) {
            w0 = make_float4(world.c0x, world.c0y, world.c0z, world.c1x);
            w1 = make_float4(world.c1y, world.c1z, world.c2x, world.c2y);
            w2 = make_float4(world.c2z, world.c3x, world.c3y, world.c3z);
        }
        const uint32_t wave_first = lb * 256u + wave * 64u;  // (the wave's tile is free again: its products sit in registers)
        store_world_wave(my_tile, args.world, wave_first, wave_first < xf.count ? min(64u, xf.count - wave_first) : 0u, w0, w1, w2);
    }
    // ---- cull_kernel's tail on the model in registers (mesh.cpp:140-175) ----
    bool visible = false;
    if (has_mesh) {
        const float mnx = ma.x, mny = ma.y, mnz = ma.z, mxx = ma.w, mxy = mb.x, mxz = mb.y;
        const bool empty = (mxx - mnx <= 0.0f) && (mxy - mny <= 0.0f) && (mxz - mnz <= 0.0f);
        if (in_range && !empty && (flags & kXfActive)) {
            const Mat34 model = translated(world, args.cull.view.cam[0], args.cull.view.cam[1], args.cull.view.cam[2]);
            Corners c;
            // sphere pre-test first (gv_device.hpp): corners and the exact test only for entries near a plane
            const uint32_t where = classify_sphere(model, ma, mb, args.cull.view.planes, args.cull.view.plane_count);
            visible = where == kSphereInside;
            if (where == kSphereUndecided) {
                aabb_corners(model, mnx, mny, mnz, mxx, mxy, mxz, c);
                visible = !behind_frustum(c, args.cull.view.planes, args.cull.view.plane_count);
            } else if (HIZ && visible) {
                aabb_corners(model, mnx, mny, mnz, mxx, mxy, mxz, c);
            }
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
    __shared__ float4 world_stage[768];
    uint32_t flags = 0;
    Mat34 world = {};
    {
        float4 w0 = make_float4(0, 0, 0, 0), w1 = w0, w2 = w0;
        if (in_range) {
            const XfRecord r = stream_xf(xf, s);
            flags = r.flags;
            if (flags & kXfLive) {
                world = chain_model(xf, local_model(r), s, flags);
                w0 = make_float4(world.c0x, world.c0y, world.c0z, world.c1x);
                w1 = make_float4(world.c1y, world.c1z, world.c2x, world.c2y);
                w2 = make_float4(world.c2z, world.c3x, world.c3y, world.c3z);
            }
        }
        const uint32_t wave_first = lb * 256u + wave * 64u;
        store_world_wave(world_stage + wave * 192u, args.world, wave_first, wave_first < xf.count ? min(64u, xf.count - wave_first) : 0u, w0, w1, w2);
    }
    bool visible = false;
    if (has_mesh) {
        const float mnx = ma.x, mny = ma.y, mnz = ma.z, mxx = ma.w, mxy = mb.x, mxz = mb.y;
        const bool empty = (mxx - mnx <= 0.0f) && (mxy - mny <= 0.0f) && (mxz - mnz <= 0.0f);
        if (in_range && !empty && (flags & kXfActive)) {
            const Mat34 model = translated(world, args.cull.view.cam[0], args.cull.view.cam[1], args.cull.view.cam[2]);
            Corners c;
            // sphere pre-test first (gv_device.hpp): corners and the exact test only for entries near a plane
            const uint32_t where = classify_sphere(model, ma, mb, args.cull.view.planes, args.cull.view.plane_count);
            visible = where == kSphereInside;
            if (where == kSphereUndecided) {
                aabb_corners(model, mnx, mny, mnz, mxx, mxy, mxz, c);
                visible = !behind_frustum(c, args.cull.view.planes, args.cull.view.plane_count);
            } else if (HIZ && visible) {
                aabb_corners(model, mnx, mny, mnz, mxx, mxy, mxz, c);
            }
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
    SweepCullArgs a{};
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

}  // namespace gv
