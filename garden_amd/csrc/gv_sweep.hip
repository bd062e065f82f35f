// gv_sweep.hip — world-matrix sweeps (TransformComponent::calcModel for every slot, transform.hpp:197-214): VALU and
// MFMA forms, alone and fused with the cull of an exactly paired pool.
#include "gv_device.hpp"

namespace gv {

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
