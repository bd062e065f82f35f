// gv_context.cpp — host side of libgarden_vis.so: the C-ABI of include/garden_vis.h — context, binds, per-frame
// dispatch, sweeps, Hi-Z, statistics. (The device mirror behind the binds lives in gv_mirror.cpp, the reader side — waits,
// fetches, records, sorts, device accessors — in gv_results.cpp, the multi-GPU exchange in gv_exchange.cpp.)
//
// Replaces (reference paths): the scratch sizing / dispatch / wait of MeshRenderSystem::prepareMeshes
// (source/system/render/mesh.cpp:331-553), the per-entity Manager::tryGet lookup (mesh.cpp:149) —
// resolved once into the mirror — and HizRenderSystem::downsampleHiz's per-mip pass loop
// (source/system/render/hiz.cpp:148-164).
//
// There is NO CPU fallback: without a gfx950 device gv_create fails with GV_E_NODEVICE.
#include "gv_ctx.hpp"
#include "gv_hiz_kernels.hpp"

#include <sys/mman.h>
#include <unistd.h>

using namespace gv;

namespace gv {

thread_local std::string g_create_error;


// Frustum(viewProj) (mesh.cpp:815,867,869,900,902) — host side, once per view: Gribb-Hartmann rows of
// the column-major matrix for a [0,1] clip depth, normalised; degenerate planes (|n|^2 < 1e-12, the
// z >= 0 plane of the infinite reversed-Z projection) are dropped. DESIGN.md §"Canonical arithmetic".
void build_view_params(const GvView& v, ViewParams* out)
{
    float row[4][4];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++)
            row[r][c] = v.view_proj[c * 4 + r];
    float p[6][4];
    for (int c = 0; c < 4; c++) {
        p[0][c] = row[3][c] + row[0][c];
        p[1][c] = row[3][c] - row[0][c];
        p[2][c] = row[3][c] + row[1][c];
        p[3][c] = row[3][c] - row[1][c];
        p[4][c] = row[2][c];
        p[5][c] = row[3][c] - row[2][c];
    }
    memset(out, 0, sizeof(*out));
    for (int i = 0; i < 6; i++) {
        const float len2 = std::fmaf(p[i][2], p[i][2], std::fmaf(p[i][1], p[i][1], p[i][0] * p[i][0]));
        if (!(len2 >= 1e-12f))
            continue;
        const float inv = 1.0f / std::sqrt(len2);
        float* q = out->planes[out->plane_count++];
        for (int c = 0; c < 4; c++)
            q[c] = p[i][c] * inv;
    }
    for (int c = 0; c < 3; c++) {
        out->cam[c] = v.camera_position[c];
        out->cam_offset[c] = v.camera_offset[c];
    }
    memcpy(out->vp, v.view_proj, sizeof(out->vp));
    out->write_is_visible = v.shadow_pass < 0 ? 1u : 0u;  // isNotShadowPass  mesh.cpp:121
    out->use_hiz = v.use_hiz ? 1u : 0u;
    out->distance_2d = v.distance_2d ? 1u : 0u;
}

int reserve_view(GvCtx* ctx, ViewState& vs, uint32_t occupancy, bool emit)
{
    const size_t n = std::max<uint32_t>(occupancy, 1);
    const size_t blocks = (n + kCullBlock - 1) / kCullBlock;
    const size_t chunks = (n + kEmitChunk - 1) / kEmitChunk;
    GV_HIP(ctx, vs.mask.reserve(blocks * (kCullBlock / 64)));
    if (chunks > vs.chunk_count.cap) {
        GV_HIP(ctx, vs.chunk_count.reserve(chunks));
        GV_HIP(ctx, vs.chunk_count2.reserve(chunks));
        // the cull workgroups add into the totals; scan (or the self-prefixing emit) re-zeroes them; fresh ones start at zero
        GV_HIP(ctx, hipMemsetAsync(vs.chunk_count.ptr, 0, vs.chunk_count.cap * sizeof(uint32_t), ctx->stream));
        GV_HIP(ctx, hipMemsetAsync(vs.chunk_count2.ptr, 0, vs.chunk_count2.cap * sizeof(uint32_t), ctx->stream));
        vs.count_parity = 0;
        vs.stale_chunks[0] = vs.stale_chunks[1] = 0;
    }
    GV_HIP(ctx, vs.chunk_offset.reserve(chunks));
    GV_HIP(ctx, vs.draw_count.reserve(4));
    if (n > vs.is_visible.cap || chunks * kEmitParts > vs.vis_flags.cap) {
        GV_HIP(ctx, vs.is_visible.reserve(n));
        GV_HIP(ctx, vs.vis_flags.reserve(chunks * kEmitParts));
        vs.vis_flags_current = false;  // new bytes: unknown contents
    }
    GV_HIP(ctx, vs.h_draw_count.reserve(4));
    if (emit) {
        GV_HIP(ctx, vs.visible_idx.reserve(n));
        GV_HIP(ctx, vs.baked_model.reserve(n * 12));
        GV_HIP(ctx, vs.distance_sq.reserve(n));
    }
    return GV_OK;
}

ViewBuffers view_buffers(ViewState& vs)
{
    ViewBuffers b;
    b.mask = vs.mask.ptr;
    b.chunk_count = vs.count_parity ? vs.chunk_count2.ptr : vs.chunk_count.ptr;
    b.chunk_count_next = vs.count_parity ? vs.chunk_count.ptr : vs.chunk_count2.ptr;
    b.chunk_offset = vs.chunk_offset.ptr;
    b.draw_count = vs.draw_count.ptr;
    b.is_visible = vs.is_visible.ptr;
    b.vis_flags = vs.vis_flags.ptr;
    b.visible_idx = vs.visible_idx.ptr;
    b.baked_model = vs.baked_model.ptr;
    b.distance_sq = vs.distance_sq.ptr;
    return b;
}

// In front of a self-prefixing emit of this view: its quarter-chunk flags must describe the isVisible bytes. They do when the
// previous writer of those bytes was such an emit too; otherwise (a cull that stored the bytes itself, the one-launch cull + emit,
// the scan-path emit, new buffers) every quarter is marked "may hold a non-zero" and the emit rewrites it once.
int emit_flags_ready(GvCtx* ctx, ViewState& vs)
{
    if (!vs.vis_flags_current && vs.vis_flags.ptr)
        GV_HIP(ctx, hipMemsetAsync(vs.vis_flags.ptr, 1, vs.vis_flags.cap, ctx->stream));
    vs.vis_flags_current = true;
    return GV_OK;
}

// The world-matrix cache has just been brought up to date (any sweep): the re-mirror flags start over.
int world_current(GvCtx* ctx)
{
    if (ctx->xdirty_set && ctx->d_xdirty.ptr)
        GV_HIP(ctx, hipMemsetAsync(ctx->d_xdirty.ptr, 0, std::min<size_t>(ctx->d_xdirty.cap, ctx->xf.occupancy), ctx->stream));
    ctx->xdirty_set = false;
    ctx->world_partial = false;
    ctx->world_valid = true;
    return GV_OK;
}

// level k >= 1 of the pyramid: float2 texels, or packed binary16 pairs (4 bytes) under GV_CONFIG_HIZ_RG16F
static float2* mip_ptr(GvCtx* ctx, uint32_t k)
{
    if (ctx->config.flags & GV_CONFIG_HIZ_RG16F)
        return reinterpret_cast<float2*>(reinterpret_cast<uint32_t*>(ctx->d_mips.ptr) + ctx->mip_off[k]);
    return ctx->d_mips.ptr + ctx->mip_off[k];
}

int hiz_reduce(GvCtx* ctx)
{
    const bool rg16f = (ctx->config.flags & GV_CONFIG_HIZ_RG16F) != 0;
    ctx->hiz_level1_stored = false;
    ZoneScope zone("HiZ Downsample");
    KernelTimer timer(ctx, GV_K_HIZ);
    uint32_t k = 1;
    while (k < ctx->hiz_mips) {
        const uint32_t sw = ctx->mip_w[k - 1], sh = ctx->mip_h[k - 1];
        const float* src_d = k == 1 ? ctx->depth_ptr : nullptr;
        const float2* src_p = k == 1 ? nullptr : mip_ptr(ctx, k - 1);
        if (sw % 64 == 0 && sh % 64 == 0 && k + 5 < ctx->hiz_mips) {
            HizFusedDst dst;
            for (int l = 0; l < 6; l++)
                dst.level[l] = mip_ptr(ctx, k + l);
            if (k == 1 && ctx->hiz_level1_virtual)
                dst.level[0] = nullptr;  // not written: 3/4 of the pyramid's bytes (gv_hiz_read_level materialises it on demand)
            GV_HIP(ctx, launch_hiz_fused(src_d, src_p, dst, sw, sh, rg16f, ctx->stream));
            k += 6;
        } else if ((uint64_t)ctx->mip_w[k] * ctx->mip_h[k] <= kHizTailTexels) {
            // the rest of the pyramid is small: one workgroup, one launch (frame sizes are rarely divisible by 64)
            static_assert(GV_MAX_MIPS <= 16, "HizTailArgs holds 16 levels");
            HizTailArgs tail{};
            tail.depth = ctx->depth_ptr;
            tail.mips = ctx->d_mips.ptr;
            for (uint32_t m = 0; m < ctx->hiz_mips; m++) {
                tail.offset[m] = ctx->mip_off[m];
                tail.w[m] = ctx->mip_w[m];
                tail.h[m] = ctx->mip_h[m];
            }
            tail.first = k;
            tail.count = ctx->hiz_mips - k;
            tail.rule = ctx->config.hiz_rule;
            GV_HIP(ctx, launch_hiz_tail(tail, rg16f, ctx->stream));
            k = ctx->hiz_mips;
        } else if (k + 3 < ctx->hiz_mips && sw >= 2 && sh >= 2 &&
                   ((ctx->mip_w[k] + 63) / 64) * ((ctx->mip_h[k] + 63) / 64) >= 96 &&
                   ((ctx->mip_w[k] + 63) / 64) * ((ctx->mip_h[k] + 63) / 64) <= 200 &&
                   (uint64_t)ctx->mip_w[k + 2] * ctx->mip_h[k + 2] > kHizTailTexels && (uint64_t)ctx->mip_w[k + 3] * ctx->mip_h[k + 3] > kHizTailTexels / 2) {
            // Four levels per launch where that is what makes the one-workgroup tail start small: three levels would leave it a
            // first level of more than 4096 texels to read from memory (1920 x 1080: 8040), and the frame is large enough for
            // 64 x 64 workgroups to fill the GPU but not so large that the heavier workgroups cost more than the tail saves
            // (measured: 1920 x 1080 22.8 -> 17.1 us, 1600 x 900 17.7 -> 16.3; 2560 x 1440 21.0 -> 22.4 and 3840 x 2160 27.2 -> 32.6 keep three levels)
            HizFused4Args f{};
            f.depth = src_d;
            f.src_pairs = src_p;
            for (uint32_t l = 0; l < 5; l++) {
                f.w[l] = ctx->mip_w[k - 1 + l];
                f.h[l] = ctx->mip_h[k - 1 + l];
            }
            for (uint32_t l = 0; l < 4; l++)
                f.dst[l] = mip_ptr(ctx, k + l);
            f.rule = ctx->config.hiz_rule;
            GV_HIP(ctx, launch_hiz_fused4(f, rg16f, ctx->stream));
            k += 4;
        } else if (k + 2 < ctx->hiz_mips && sw >= 2 && sh >= 2) {
            // any size: three levels per launch, a rim of the two intermediate levels recomputed per workgroup (gv_hiz.hip)
            HizFused3Args f{};
            f.depth = src_d;
            f.src_pairs = src_p;
            for (uint32_t l = 0; l < 4; l++) {
                f.w[l] = ctx->mip_w[k - 1 + l];
                f.h[l] = ctx->mip_h[k - 1 + l];
            }
            for (uint32_t l = 0; l < 3; l++)
                f.dst[l] = mip_ptr(ctx, k + l);
            f.rule = ctx->config.hiz_rule;
            GV_HIP(ctx, launch_hiz_fused3(f, rg16f, ctx->stream));
            k += 3;
        } else {
            GV_HIP(ctx, launch_hiz_level(src_d, src_p, mip_ptr(ctx, k), sw, sh, ctx->mip_w[k], ctx->mip_h[k], ctx->config.hiz_rule, rg16f,
                                         ctx->stream));
            k += 1;
        }
    }
    return GV_OK;
}

HizDevice hiz_device(const GvCtx* ctx)
{
    HizDevice hz{};
    if (ctx->hiz_valid) {
        hz.depth = ctx->depth_ptr;
        hz.mips = ctx->d_mips.ptr;
        hz.mip_offset = ctx->d_mip_offset.ptr;
        hz.width = ctx->hiz_w;
        hz.height = ctx->hiz_h;
        hz.mip_count = ctx->hiz_mips;
        hz.nested = ctx->hiz_nested ? 1u : 0u;
        hz.rg16f = (ctx->config.flags & GV_CONFIG_HIZ_RG16F) ? 1u : 0u;
        hz.level1_virtual = ctx->hiz_level1_virtual ? 1u : 0u;
    }
    return hz;
}

// The launches of one gv_cull (views already reserved and marked valid by gv_cull): the sweep riding on it, block
// bounds, the cull itself in the form that fits (fused sweep + cull, cull + emit in one launch, batched views, ...),
// compaction and emission.
int cull_launch(GvCtx* ctx, uint32_t pool_id, const ViewParams* vps, uint32_t view_count, bool batched)
{
    PoolState& p = ctx->pools[pool_id];
    const MeshMirror mesh{p.d_a.ptr, p.d_b.ptr, p.d_link.ptr, p.occupancy, p.mapping, p.perm.empty() ? nullptr : p.d_orig.ptr};
    const TransformMirror xf = xf_mirror(ctx);
    const HizDevice hz = hiz_device(ctx);
    const uint32_t chunks = (p.occupancy + kEmitChunk - 1) / kEmitChunk;
    ViewBuffers vbs[GV_MAX_VIEWS];
    ViewParams cvps[GV_MAX_VIEWS];  // as the cull kernels see the views: isVisible bytes are left to the emit that follows
    for (uint32_t v = 0; v < view_count; v++) {
        vbs[v] = view_buffers(ctx->views[pool_id][v]);
        cvps[v] = vps[v];
        if (ctx->views[pool_id][v].emitted)
            cvps[v].write_is_visible = 0;
        if (cvps[v].write_is_visible)
            ctx->views[pool_id][v].vis_flags_current = false;  // the cull stores the bytes itself
    }
    int rc = GV_OK;
    // GV_SWEEP_WITH_CULL: an exactly paired pool takes the fused MFMA sweep + cull for its first view; anything else
    // gets the same results from the plain MFMA sweep followed by the ordinary cull
    const bool sweep_requested = ctx->sweep_with_cull;
    ctx->sweep_with_cull = false;
    const bool fused = sweep_requested && !batched && p.occupancy != 0 && mesh.mapping == kMapExact && mesh.count <= xf.count;
    if (sweep_requested) {
        GV_HIP(ctx, ctx->d_world.reserve((size_t)std::max(ctx->xf.occupancy, 1u) * 3));
        if (!fused) {
            KernelTimer t(ctx, GV_K_SWEEP);
            if (ctx->sweep_with_cull_mfma)
                GV_HIP(ctx, launch_sweep_mfma(xf, ctx->d_world.ptr, ctx->stream));
            else
                GV_HIP(ctx, launch_sweep_valu(xf, ctx->d_world.ptr, ctx->stream));
        }
        if ((rc = world_current(ctx)) != GV_OK)  // (the fused form below writes every slot of the same buffer)
            return rc;
    }
    // What is derived from a pool AT REST — block bounds, emit seeds — is (re)built when the pool's mirror is clean, or has just
    // changed after a quiet frame; a pool that changes frame after frame (dynamic scene) goes without
    const bool changed = p.seen_epoch != p.epoch || p.seen_xf_epoch != ctx->xf_epoch;
    // ... or keeps changing only a little (kMinSmallStreak syncs in a row that re-mirrored a few entries each): then it gets them
    // once more and they are patched from there on (below)
    constexpr uint32_t kMinSmallStreak = 4;
    // (only pools whose boxes can then be patched — entry i <-> transform entry i, no chains —: any other would be rebuilt every frame)
    const bool patchable = mesh.mapping == kMapExact && xf.max_depth == 0 && mesh.count <= xf.count;
    const bool may_rebuild = !(changed && p.changed_prev) || (patchable && changed && p.small_streak >= kMinSmallStreak);
    p.changed_prev = changed;
    p.seen_epoch = p.epoch;
    p.seen_xf_epoch = ctx->xf_epoch;
    BlockBounds bounds;
    bool use_bounds = false;
    const bool bounds_wanted = (ctx->config.flags & GV_CONFIG_BLOCK_BOUNDS) ||
                               (!(ctx->config.flags & GV_CONFIG_LINEAR_SCAN) && p.occupancy > kAutoBoundsMinSlots);
    if (bounds_wanted && p.occupancy != 0 && !fused) {
        bool current = p.bounds_epoch == p.epoch && p.bounds_xf_epoch == ctx->xf_epoch;
        if (!current && p.patch_valid && patchable && p.d_blk_lo.ptr && p.d_blk_dirty.ptr) {
            // every change since the boxes were current is on record (sync_mirror flagged the blocks): re-derive those blocks — and
            // their entries' emit seeds when the seeds were in step with the boxes — instead of culling without boxes until the
            // pool comes to rest (a scene in which a few entities move every frame never does)
            const bool seeds_in_step = p.d_seed.ptr && p.d_seed.cap >= p.occupancy && p.seed_epoch == p.bounds_epoch && p.seed_xf_epoch == p.bounds_xf_epoch;
            KernelTimer t(ctx, GV_K_SWEEP);
            GV_HIP(ctx, launch_block_patch(mesh, xf, p.d_blk_lo.ptr, p.d_blk_hi.ptr, seeds_in_step ? p.d_seed.ptr : nullptr, p.d_blk_dirty.ptr, ctx->stream));
            if (seeds_in_step) {
                p.seed_epoch = p.epoch;
                p.seed_xf_epoch = ctx->xf_epoch;
            }
            p.bounds_epoch = p.epoch;
            p.bounds_xf_epoch = ctx->xf_epoch;
            current = true;
        }
        if (!current && may_rebuild) {
            const size_t nb = (p.occupancy + kCullBlock - 1) / kCullBlock;
            GV_HIP(ctx, p.d_blk_lo.reserve(nb));
            GV_HIP(ctx, p.d_blk_hi.reserve(nb));
            GV_HIP(ctx, p.d_blk_dirty.reserve((nb + 15) & ~(size_t)15));
            GV_HIP(ctx, hipMemsetAsync(p.d_blk_dirty.ptr, 0, p.d_blk_dirty.cap, ctx->stream));
            KernelTimer t(ctx, GV_K_SWEEP);  // accounted with the other per-change passes
            GV_HIP(ctx, launch_block_bounds(mesh, xf, p.d_blk_lo.ptr, p.d_blk_hi.ptr, ctx->stream));
            p.bounds_epoch = p.epoch;
            p.bounds_xf_epoch = ctx->xf_epoch;
            p.patch_valid = patchable;  // from here on every sync records what it re-mirrors (gv_mirror.cpp mark_dirty_blocks)
            current = true;
        }
        if (current) {
            const size_t nb = (p.occupancy + kCullBlock - 1) / kCullBlock;
            GV_HIP(ctx, ctx->d_examined.reserve(nb));
            const size_t list_words = 4 + nb * (cull_list_entry_bytes() / sizeof(uint32_t));
            GV_HIP(ctx, p.d_kept_flag.reserve(nb));
            if (list_words > p.d_kept.cap) {
                GV_HIP(ctx, p.d_kept.reserve(list_words));
                GV_HIP(ctx, hipMemsetAsync(p.d_kept.ptr, 0, 2 * sizeof(uint32_t), ctx->stream));
                p.kept_parity = 0;
            }
            use_bounds = true;
        }
    }
    // emit seeds: a flat, exactly paired pool of some size whose views want records (gv_kernels.hpp)
    const EmitSeed* seeds = nullptr;
    bool wants_records = false;
    for (uint32_t v = 0; v < view_count; v++)
        wants_records = wants_records || ctx->views[pool_id][v].emitted;
    if (wants_records && !batched && p.occupancy >= kEmitSeedMinSlots && mesh.mapping == kMapExact && xf.max_depth == 0 &&
        mesh.count <= xf.count) {
        bool current = p.seed_epoch == p.epoch && p.seed_xf_epoch == ctx->xf_epoch && p.d_seed.cap >= p.occupancy;
        if (!current && may_rebuild) {
            GV_HIP(ctx, p.d_seed.reserve(p.occupancy));
            KernelTimer t(ctx, GV_K_SWEEP);
            GV_HIP(ctx, launch_emit_seeds(mesh, xf, p.d_seed.ptr, ctx->stream));
            p.seed_epoch = p.epoch;
            p.seed_xf_epoch = ctx->xf_epoch;
            current = true;
        }
        if (current)
            seeds = p.d_seed.ptr;
    }
    // records take the resident world matrices when a sweep of the current mirror has written them (this call's fused
    // or leading sweep, or an earlier gv_sweep with no transform change since): same bits as the chain walk
    const float4* emit_world = (ctx->world_valid && !ctx->world_partial && ctx->max_depth != 0) ? ctx->d_world.ptr : nullptr;
    if (p.occupancy != 0) {
        if (use_bounds) {
            bounds.lo = p.d_blk_lo.ptr;
            bounds.hi = p.d_blk_hi.ptr;
            bounds.examined = ctx->d_examined.ptr;
            ctx->bounds_blocks_total = (p.occupancy + kCullBlock - 1) / kCullBlock;
        }
        if (batched) {
            KernelTimer t(ctx, GV_K_CULL);
            GV_HIP(ctx, launch_cull_multi(mesh, xf, hz, cvps, vbs, view_count, ctx->stream, use_bounds ? &bounds : nullptr));
        }
        constexpr uint32_t self_max = kSelfPrefixMaxChunks;
        // a batched cull whose views all want records: ONE self-prefixing emit launch for all of them
        bool emit_batched = batched && chunks <= self_max;
        for (uint32_t v = 0; v < view_count && emit_batched; v++)
            emit_batched = ctx->views[pool_id][v].emitted;
        if (emit_batched) {
            uint32_t clear[GV_MAX_VIEWS];
            for (uint32_t v = 0; v < view_count; v++) {
                ViewState& vs = ctx->views[pool_id][v];
                const uint32_t cur = vs.count_parity, other = cur ^ 1u;
                clear[v] = std::max(chunks, vs.stale_chunks[other]);
                vs.stale_chunks[other] = 0;
                vs.stale_chunks[cur] = chunks;
                vs.count_parity = other;
                if ((rc = emit_flags_ready(ctx, vs)) != GV_OK)
                    return rc;
            }
            KernelTimer t(ctx, GV_K_EMIT);
            GV_HIP(ctx, launch_emit_batch(mesh, xf, vps, vbs, clear, view_count, ctx->stream, emit_world));
        }
        // one view, records wanted, pool small enough for the look-back form to win: cull + emit in ONE launch
        if (!batched && !fused && !use_bounds && view_count == 1 && ctx->views[pool_id][0].emitted && p.occupancy <= kFusedEmitMaxSlots) {
            ViewState& vs = ctx->views[pool_id][0];
            const size_t nb = (p.occupancy + kCullBlock - 1) / kCullBlock;
            if (nb > vs.tile_status.cap || !vs.tile_ticket.ptr) {
                GV_HIP(ctx, vs.tile_status.reserve(nb));
                GV_HIP(ctx, vs.tile_ticket.reserve(1));
                GV_HIP(ctx, hipMemsetAsync(vs.tile_status.ptr, 0, vs.tile_status.cap * sizeof(unsigned long long), ctx->stream));
                GV_HIP(ctx, hipMemsetAsync(vs.tile_ticket.ptr, 0, sizeof(uint32_t), ctx->stream));
                vs.tile_ticket_base = 0;
                vs.tile_epoch = 0;
            }
            vs.ballots_current = false;
            vs.vis_flags_current = false;  // cull_emit_kernel stores the bytes
            vs.tile_epoch = vs.tile_epoch == UINT32_MAX ? 1u : vs.tile_epoch + 1u;
            {
                KernelTimer t(ctx, GV_K_CULL);
                GV_HIP(ctx, launch_cull_emit(mesh, xf, hz, vps[0], vbs[0], vs.tile_status.ptr, vs.tile_ticket.ptr, vs.tile_ticket_base,
                                             vs.tile_epoch, ctx->stream));
            }
            vs.tile_ticket_base += (uint32_t)nb;
            for (uint32_t v = view_count; v < GV_MAX_VIEWS; v++)
                ctx->views[pool_id][v].valid = false;
            return GV_OK;
        }
        for (uint32_t v = 0; v < view_count && !emit_batched; v++) {
            if (!batched) {
                KernelTimer t(ctx, GV_K_CULL);
                // bounds: classify the workgroups first and cull the kept ones from a list
                if (fused && v == 0)
                    GV_HIP(ctx, launch_sweep_cull(mesh, xf, hz, cvps[v], vbs[v], ctx->d_world.ptr, ctx->sweep_with_cull_mfma, ctx->stream));
                else if (use_bounds) {
                    uint32_t* counters = p.d_kept.ptr;
                    GV_HIP(ctx, launch_cull_listed(mesh, xf, hz, cvps[v], vbs[v], bounds, counters + p.kept_parity, counters + (p.kept_parity ^ 1u),
                                                   counters + 4, p.d_kept_flag.ptr, ctx->stream));
                    p.kept_parity ^= 1u;
                } else {
                    GV_HIP(ctx, launch_cull(mesh, xf, hz, cvps[v], vbs[v], ctx->stream));
                }
            }
            if (ctx->views[pool_id][v].emitted && chunks <= self_max) {
                // no scan launch: emit derives the chunk bases itself and leaves THIS totals buffer as it is; the
                // next cull of this view adds into the other one, which this emit has cleared
                ViewState& vs = ctx->views[pool_id][v];
                const uint32_t cur = vs.count_parity, other = cur ^ 1u;
                if ((rc = emit_flags_ready(ctx, vs)) != GV_OK)
                    return rc;
                {
                    KernelTimer t(ctx, GV_K_EMIT);
                    GV_HIP(ctx, launch_emit(mesh, xf, vps[v], vbs[v], ctx->stream, true, std::max(chunks, vs.stale_chunks[other]), emit_world, seeds));
                }
                vs.stale_chunks[other] = 0;
                vs.stale_chunks[cur] = chunks;
                vs.count_parity = other;
                continue;
            }
            {
                KernelTimer t(ctx, GV_K_SCAN);
                GV_HIP(ctx, launch_scan(vbs[v], chunks, ctx->stream));
            }
            if (ctx->views[pool_id][v].emitted) {
                ctx->views[pool_id][v].vis_flags_current = false;  // (the scan-path emit writes every byte and keeps no flags)
                KernelTimer t(ctx, GV_K_EMIT);
                GV_HIP(ctx, launch_emit(mesh, xf, vps[v], vbs[v], ctx->stream, false, 0, emit_world, seeds));
            }
        }
    }
    for (uint32_t v = view_count; v < GV_MAX_VIEWS; v++)
        ctx->views[pool_id][v].valid = false;
    return GV_OK;  // (ctx->last_pool is gv_cull's to set: a recorded cull launched later — by a reader of ANOTHER pool's results — must not
                   // turn the view-indexed calls towards its own pool; round 4, found by tools/schedule_soak.py)
}

// The culls recorded since gv_cull_batch_begin: ONE cull launch for all of them (blockIdx.y = job) and ONE emit launch
// for all their views. Descriptors are built from the pools' CURRENT mirrors and shipped as one small table.
int flush_culls(GvCtx* ctx)
{
    ctx->cull_batching = false;  // the batch ends with its first read
    if (ctx->cull_jobs.empty())
        return GV_OK;
    std::vector<Context::CullJob> jobs;
    jobs.swap(ctx->cull_jobs);
    GV_HIP(ctx, hipSetDevice(ctx->device));
    if (jobs.size() == 1) {  // nothing to batch: the ordinary launches (cull + emit in one for a single view) are shorter
        bool batched = jobs[0].view_count > 1;
        return cull_launch(ctx, jobs[0].pool_id, jobs[0].vps, jobs[0].view_count, batched);
    }
    const size_t cull_bytes = cull_table_entry_bytes(), emit_bytes = emit_table_entry_bytes();
    uint32_t emit_entries = 0, max_slots = 0;
    for (const auto& j : jobs)
        emit_entries += j.view_count;
    const size_t cull_off = 0, emit_off = (jobs.size() * cull_bytes + 255) & ~(size_t)255;
    const size_t total = emit_off + (size_t)emit_entries * emit_bytes;
    const uint32_t turn = ctx->tick_turn;
    ctx->tick_turn ^= 1u;
    if (!ctx->tick_done[turn])
        GV_HIP(ctx, hipEventCreateWithFlags(&ctx->tick_done[turn], hipEventDisableTiming));
    GV_HIP(ctx, hipEventSynchronize(ctx->tick_done[turn]));  // (a never-recorded event is complete)
    GV_HIP(ctx, ctx->h_tick[turn].reserve(total));
    if (total > ctx->d_tick.cap) {
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // kernels of the previous tick may still read the old table
        GV_HIP(ctx, ctx->d_tick.reserve(total + total / 2));
    }
    uint8_t* host = ctx->h_tick[turn].ptr;
    const TransformMirror xf = xf_mirror(ctx);
    const HizDevice hz = hiz_device(ctx);
    uint32_t e = 0;
    for (size_t k = 0; k < jobs.size(); k++) {
        const auto& j = jobs[k];
        PoolState& p = ctx->pools[j.pool_id];
        const MeshMirror mesh{p.d_a.ptr, p.d_b.ptr, p.d_link.ptr, p.occupancy, p.mapping, p.perm.empty() ? nullptr : p.d_orig.ptr};
        const uint32_t chunks = (p.occupancy + kEmitChunk - 1) / kEmitChunk;
        ViewBuffers vbs[GV_MAX_VIEWS];
        for (uint32_t v = 0; v < j.view_count; v++)
            vbs[v] = view_buffers(ctx->views[j.pool_id][v]);
        ViewParams cvps[GV_MAX_VIEWS];  // (every view of a recorded job emits records: the emit expands isVisible)
        for (uint32_t v = 0; v < j.view_count; v++) {
            cvps[v] = j.vps[v];
            cvps[v].write_is_visible = 0;
        }
        fill_cull_table_entry(host + cull_off + k * cull_bytes, mesh, xf, hz, cvps, vbs, j.view_count);
        for (uint32_t v = 0; v < j.view_count; v++) {
            ViewState& vs = ctx->views[j.pool_id][v];
            const uint32_t cur = vs.count_parity, other = cur ^ 1u;
            if (int rc = emit_flags_ready(ctx, vs))
                return rc;
            fill_emit_table_entry(host + emit_off + (size_t)(e++) * emit_bytes, mesh, xf, j.vps[v], vbs[v],
                                  std::max(chunks, vs.stale_chunks[other]), nullptr);
            vs.stale_chunks[other] = 0;
            vs.stale_chunks[cur] = chunks;
            vs.count_parity = other;
        }
        max_slots = std::max(max_slots, p.occupancy);
    }
    GV_HIP(ctx, hipMemcpyAsync(ctx->d_tick.ptr, host, total, hipMemcpyHostToDevice, ctx->stream));
    GV_HIP(ctx, hipEventRecord(ctx->tick_done[turn], ctx->stream));
    {
        ZoneScope zone("Meshes Prepare");
        {
            KernelTimer t(ctx, GV_K_CULL);
            GV_HIP(ctx, launch_cull_table(ctx->d_tick.ptr + cull_off, (uint32_t)jobs.size(), max_slots, ctx->stream));
        }
        KernelTimer t(ctx, GV_K_EMIT);
        GV_HIP(ctx, launch_emit_table(ctx->d_tick.ptr + emit_off, emit_entries, max_slots, ctx->stream));
    }
    return GV_OK;
}

// A recorded cull (gv_cull_batch_begin) is launched with the pools' mirrors as they are at the FLUSH: a bind or a dirty mark
// of a pool it reads, in between, would let it run on another occupancy than its view buffers were sized for. Such a call
// therefore launches what has been recorded so far first; recording then goes on. pool: the mesh pool about to change, or
// GV_MAX_POOLS for the transform pool (every recorded cull reads it). A frame that binds and culls its mesh systems one after
// the other (the shim does) keeps its one batch: a system's bind does not touch the pools recorded before it.
int flush_recorded_culls(GvCtx* ctx, uint32_t pool)
{
    bool reads_it = false;
    for (const auto& j : ctx->cull_jobs)
        reads_it = reads_it || pool == GV_MAX_POOLS || j.pool_id == pool;
    if (!reads_it)
        return GV_OK;
    const bool batching = ctx->cull_batching;
    const int rc = flush_culls(ctx);
    ctx->cull_batching = batching;
    return rc;
}

}  // namespace gv

// ================================================================================================
// C-ABI
// ================================================================================================
extern "C" {

uint32_t gv_abi_version(void) { return GV_ABI_VERSION; }

const char* gv_last_error(const GvCtx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int gv_create(const GvConfig* config, GvCtx** out_ctx)
{
    if (!out_ctx) {
        g_create_error = "gv_create: out_ctx is NULL";
        return GV_E_ARG;
    }
    *out_ctx = nullptr;
    GvConfig cfg{};
    cfg.struct_size = sizeof(GvConfig);
    if (config) {
        if (config->struct_size != sizeof(GvConfig)) {
            g_create_error = "gv_create: GvConfig.struct_size mismatch (ABI)";
            return GV_E_ARG;
        }
        cfg = *config;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_error = std::string("gv_create: no HIP device (") + hipGetErrorString(e) +
                         "); libgarden_vis has no CPU fallback";
        return GV_E_NODEVICE;
    }
    if (cfg.device < 0 || cfg.device >= count) {
        g_create_error = "gv_create: device ordinal out of range";
        return GV_E_ARG;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, cfg.device);
    if (e != hipSuccess) {
        g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        return GV_E_HIP;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("gv_create: device is ") + prop.gcnArchName +
                         ", kernels are built for gfx950 only (no fallback)";
        return GV_E_NODEVICE;
    }
    GvCtx* ctx = new (std::nothrow) GvCtx();
    if (!ctx) {
        g_create_error = "gv_create: out of host memory";
        return GV_E_OOM;
    }
    ctx->config = cfg;
    ctx->device = cfg.device;
    ctx->profile_mask = !(cfg.flags & GV_CONFIG_PROFILE_EVENTS) ? 0u : (cfg.flags & GV_CONFIG_PROFILE_CULL_ONLY) ? 1u << GV_K_CULL : (1u << GV_K_COUNT) - 1u;
    if ((e = hipSetDevice(cfg.device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_error = std::string("gv_create: ") + hipGetErrorString(e);
        delete ctx;
        return GV_E_HIP;
    }
    *out_ctx = ctx;
    return GV_OK;
}

void gv_destroy(GvCtx* ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    // the exchange stream first, with its bounded wait: the context's stream may be waiting for a frame's rows behind a collective
    // that a peer has left (a timeout aborts the communicator, which lets both streams run out)
    (void)exchange_drain(ctx);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    drain_events(ctx);
    exchange_release(ctx);
    for (auto& ev : ctx->free_events) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    ctx->d_xab.release(); ctx->d_xc.release(); ctx->d_xflags.release(); ctx->d_xactive.release(); ctx->d_xparent.release();
    ctx->h_xab.release(); ctx->h_xc.release(); ctx->h_xflags.release(); ctx->h_xparent.release();
    for (auto& p : ctx->pools) {
        for (auto& target : p.record_target)
            release_record_target(target);
        p.d_a.release(); p.d_b.release(); p.d_link.release(); p.h_a.release(); p.h_b.release(); p.h_link.release(); p.d_orig.release(); p.d_inv.release(); p.d_index_map.release(); p.d_blk_lo.release(); p.d_blk_hi.release(); p.d_seed.release(); p.d_kept.release(); p.d_kept_flag.release(); p.d_blk_dirty.release();
    }
    for (auto& per_pool : ctx->views)
      for (auto& v : per_pool) {
        v.mask.release(); v.chunk_count.release(); v.chunk_count2.release(); v.chunk_offset.release(); v.draw_count.release();
        v.is_visible.release(); v.vis_flags.release(); v.visible_idx.release(); v.baked_model.release(); v.distance_sq.release();
        v.alt_idx.release(); v.alt_model.release(); v.alt_dist.release(); v.sort_hist.release(); v.sort_ranks.release(); 
        for (int k = 0; k < 2; k++) { v.sort_keys[k].release(); v.sort_vals[k].release(); v.sort_slots[k].release(); }
        v.h_visible_idx.release(); v.h_draw_count.release(); v.h_baked_model.release();
        v.h_distance_sq.release(); v.h_is_visible.release(); v.is_visible_slots.release();
        v.h_records.release(); v.d_records.release();
        v.tile_status.release(); v.tile_ticket.release();
    }
    ctx->d_world.release(); ctx->d_xdirty.release(); ctx->d_raw.release(); ctx->d_examined.release();
    ctx->d_e2t.release(); ctx->d_flag.release(); ctx->h_flag.release(); ctx->h_ranges.release(); ctx->d_ranges.release();
    for (int k = 0; k < 2; k++) {
        ctx->h_raw[k].release();
        if (ctx->raw_done[k])
            (void)hipEventDestroy(ctx->raw_done[k]);
    }
    if (ctx->upload_done) {
        (void)hipEventDestroy(ctx->upload_done);
        ctx->upload_done = nullptr;
    }
    ctx->d_xinv.release(); ctx->sc_xf.release(); ctx->dsc_xf.release(); ctx->sc_mesh.release(); ctx->dsc_mesh.release(); ctx->dsc_a.release(); ctx->dsc_c.release();
    ctx->d_depth.release(); ctx->d_mips.release(); ctx->d_mip_offset.release(); ctx->d_tick.release(); ctx->h_done.release();
    for (int k = 0; k < 2; k++) {
        ctx->h_tick[k].release();
        if (ctx->tick_done[k])
            (void)hipEventDestroy(ctx->tick_done[k]);
    }
    if (ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int gv_transform_bind(GvCtx* ctx, const void* base, size_t stride, uint32_t occupancy,
                      const GvTransformLayout* layout, const uint32_t* entity_to_transform,
                      uint32_t entity_capacity)
{
    if (!ctx)
        return GV_E_ARG;
    if (!layout || (occupancy && !base) || (entity_capacity && !entity_to_transform))
        return ctx->fail(GV_E_ARG, "gv_transform_bind: NULL argument");
    if (occupancy >= kSlotNone)
        return ctx->fail(GV_E_ARG, "gv_transform_bind: occupancy %u exceeds the 28-bit slot range", occupancy);
    const uint32_t need = std::max({layout->position, layout->scale, layout->rotation}) + 16u;
    if (occupancy && stride < need)
        return ctx->fail(GV_E_ARG, "gv_transform_bind: stride %zu smaller than layout (%u)", stride, need);
    GvTransformColumns c{};
    const uint8_t* b = static_cast<const uint8_t*>(base);
    auto col = [&](uint32_t offset) { return GvColumn{b ? b + offset : nullptr, (uint32_t)stride}; };
    c.entity = col(layout->entity);
    c.parent = col(layout->parent);
    c.position = col(layout->position);
    c.scale = col(layout->scale);
    c.rotation = col(layout->rotation);
    c.self_active = col(layout->self_active);
    c.ancestors_active = col(layout->ancestors_active);
    c.model_with_ancestors = col(layout->model_with_ancestors);
    return gv_transform_bind_columns(ctx, &c, occupancy, entity_to_transform, entity_capacity);
}

int gv_transform_bind_columns(GvCtx* ctx, const GvTransformColumns* columns, uint32_t occupancy,
                              const uint32_t* entity_to_transform, uint32_t entity_capacity)
{
    if (!ctx)
        return GV_E_ARG;
    if (!columns || (entity_capacity && !entity_to_transform))
        return ctx->fail(GV_E_ARG, "gv_transform_bind_columns: NULL argument");
    if (occupancy >= kSlotNone)
        return ctx->fail(GV_E_ARG, "gv_transform_bind_columns: occupancy %u exceeds the 28-bit slot range", occupancy);
    if (int rc = flush_recorded_culls(ctx, GV_MAX_POOLS))
        return rc;
    const GvColumn* all[8] = {&columns->entity, &columns->parent, &columns->position, &columns->scale, &columns->rotation,
                              &columns->self_active, &columns->ancestors_active, &columns->model_with_ancestors};
    const uint32_t width[8] = {4, 4, 12, 12, 16, 1, 1, 1};
    for (int k = 0; k < 8; k++)
        if (occupancy && (!all[k]->data || all[k]->stride < width[k]))
            return ctx->fail(GV_E_ARG, "gv_transform_bind_columns: column %d is NULL or its stride is below %u bytes", k, width[k]);
    auto col = [](const GvColumn& g) { return Column{static_cast<const uint8_t*>(g.data), g.stride}; };
    // first bind or a pool that shrank: rebuild. A pool that GREW keeps its mirror; gv_sync appends the new slots.
    const bool moved = !ctx->xf.bound || occupancy < ctx->xf.occupancy;
    ctx->xf.entity = col(columns->entity);
    ctx->xf.parent = col(columns->parent);
    ctx->xf.position = col(columns->position);
    ctx->xf.scale = col(columns->scale);
    ctx->xf.rotation = col(columns->rotation);
    ctx->xf.self_active = col(columns->self_active);
    ctx->xf.ancestors_active = col(columns->ancestors_active);
    ctx->xf.model_with_ancestors = col(columns->model_with_ancestors);
    ctx->xf.occupancy = occupancy;
    ctx->xf.entity_to_transform = entity_to_transform;
    ctx->xf.entity_capacity = entity_capacity;
    ctx->xf.bound = true;
    if (moved)
        ctx->xf_need_full = true;  // first bind or pool resized: rebuild the mirror
    return GV_OK;
}

int gv_pool_bind(GvCtx* ctx, uint32_t pool_id, void* base, size_t stride, uint32_t occupancy,
                 const GvMeshLayout* layout)
{
    if (!ctx)
        return GV_E_ARG;
    if (pool_id >= GV_MAX_POOLS || !layout || (occupancy && !base))
        return ctx->fail(GV_E_ARG, "gv_pool_bind: bad argument (pool_id %u)", pool_id);
    if (occupancy >= kSlotNone)
        return ctx->fail(GV_E_ARG, "gv_pool_bind: occupancy %u exceeds the 28-bit slot range", occupancy);
    const uint32_t need = std::max(layout->aabb_min, layout->aabb_max) + 12u;
    if (occupancy && stride < need)
        return ctx->fail(GV_E_ARG, "gv_pool_bind: stride %zu smaller than layout (%u)", stride, need);
    GvMeshColumns c{};
    uint8_t* b = static_cast<uint8_t*>(base);
    auto col = [&](uint32_t offset) { return GvColumn{b ? b + offset : nullptr, (uint32_t)stride}; };
    c.entity = col(layout->entity);
    c.is_enabled = col(layout->is_enabled);
    c.aabb_min = col(layout->aabb_min);
    c.aabb_max = col(layout->aabb_max);
    c.is_visible = b ? b + layout->is_visible : nullptr;
    c.is_visible_stride = (uint32_t)stride;
    return gv_pool_bind_columns(ctx, pool_id, &c, occupancy);
}

int gv_pool_bind_columns(GvCtx* ctx, uint32_t pool_id, const GvMeshColumns* columns, uint32_t occupancy)
{
    if (!ctx)
        return GV_E_ARG;
    if (pool_id >= GV_MAX_POOLS || !columns)
        return ctx->fail(GV_E_ARG, "gv_pool_bind_columns: bad argument (pool_id %u)", pool_id);
    if (occupancy >= kSlotNone)
        return ctx->fail(GV_E_ARG, "gv_pool_bind_columns: occupancy %u exceeds the 28-bit slot range", occupancy);
    if (int rc = flush_recorded_culls(ctx, pool_id))
        return rc;
    const GvColumn* all[4] = {&columns->entity, &columns->is_enabled, &columns->aabb_min, &columns->aabb_max};
    const uint32_t width[4] = {4, 1, 12, 12};
    for (int k = 0; k < 4; k++)
        if (occupancy && (!all[k]->data || all[k]->stride < width[k]))
            return ctx->fail(GV_E_ARG, "gv_pool_bind_columns: column %d is NULL or its stride is below %u bytes", k, width[k]);
    if (columns->is_visible && columns->is_visible_stride == 0)
        return ctx->fail(GV_E_ARG, "gv_pool_bind_columns: is_visible_stride is 0");
    auto col = [](const GvColumn& g) { return Column{static_cast<const uint8_t*>(g.data), g.stride}; };
    PoolState& p = ctx->pools[pool_id];
    const bool moved = !p.bound || occupancy < p.occupancy;
    p.entity = col(columns->entity);
    p.is_enabled = col(columns->is_enabled);
    p.aabb_min = col(columns->aabb_min);
    p.aabb_max = col(columns->aabb_max);
    p.is_visible = static_cast<uint8_t*>(columns->is_visible);
    p.is_visible_stride = columns->is_visible_stride;
    p.occupancy = occupancy;
    p.bound = true;
    if (moved)
        p.need_full = true;
    return GV_OK;
}

int gv_pool_bind_ready(GvCtx* ctx, uint32_t pool_id, const void* data, uint32_t stride, uint32_t width)
{
    if (!ctx)
        return GV_E_ARG;
    if (pool_id >= GV_MAX_POOLS || (data && ((width != 1 && width != 4) || stride < width)))
        return ctx->fail(GV_E_ARG, "gv_pool_bind_ready: bad argument (pool %u, width %u, stride %u)", pool_id, width, stride);
    if (int rc = flush_recorded_culls(ctx, pool_id))
        return rc;
    PoolState& p = ctx->pools[pool_id];
    const bool had = p.ready.ptr != nullptr;
    p.ready = Column{static_cast<const uint8_t*>(data), data ? stride : 0};
    p.ready_width = data ? width : 0;
    if (had != (data != nullptr) && p.bound)
        p.dirty.add(0, p.occupancy);  // the candidate bits of the whole pool may change
    return GV_OK;
}

int gv_mark_dirty(GvCtx* ctx, uint32_t kind, uint32_t first, uint32_t count)
{
    if (!ctx)
        return GV_E_ARG;
    if (int rc = flush_recorded_culls(ctx, kind == GV_DIRTY_MESH ? std::min(first >> 28, GV_MAX_POOLS - 1u) : GV_MAX_POOLS))
        return rc;
    switch (kind) {
    case GV_DIRTY_TRANSFORM:
        ctx->xf_dirty.add(first, count);
        return GV_OK;
    case GV_DIRTY_HIERARCHY:
        if (count == 0) {
            ctx->xf_need_full = true;  // everything: entities came or went; the mirror is re-ordered too
        } else {
            ctx->xf_dirty.add(first, count);  // re-parented slots: links re-gathered in place, order kept
            ctx->xf_links_dirty = true;
        }
        return GV_OK;
    case GV_DIRTY_MESH: {
        const uint32_t pool = first >> 28, lo = first & kSlotMask;
        if (pool >= GV_MAX_POOLS)
            return ctx->fail(GV_E_ARG, "gv_mark_dirty: pool id %u", pool);
        ctx->pools[pool].dirty.add(lo, count);
        return GV_OK;
    }
    default:
        return ctx->fail(GV_E_ARG, "gv_mark_dirty: unknown kind %u", kind);
    }
}

int gv_hierarchy_rebuild(GvCtx* ctx)
{
    if (!ctx)
        return GV_E_ARG;
    if (int rc = flush_recorded_culls(ctx, GV_MAX_POOLS))
        return rc;
    ctx->xf_need_full = true;
    return sync_mirror(ctx);
}

int gv_sync(GvCtx* ctx)
{
    if (!ctx)
        return GV_E_ARG;
    if (int rc = flush_recorded_culls(ctx, GV_MAX_POOLS))
        return rc;
    return sync_mirror(ctx);
}

int gv_cull(GvCtx* ctx, uint32_t pool_id, const GvView* views, uint32_t view_count)
{
    if (!ctx)
        return GV_E_ARG;
    if (pool_id >= GV_MAX_POOLS || !views || view_count == 0 || view_count > GV_MAX_VIEWS)
        return ctx->fail(GV_E_ARG, "gv_cull: bad argument (pool %u, %u views)", pool_id, view_count);
    ZoneScope zone("Meshes Prepare");
    PoolState& p = ctx->pools[pool_id];
    if (!p.bound || !ctx->xf.bound)
        return ctx->fail(GV_E_STATE, "gv_cull: pools not bound");
    int rc = sync_mirror(ctx);
    if (rc != GV_OK)
        return rc;
    GV_HIP(ctx, hipSetDevice(ctx->device));
    for (uint32_t v = 0; v < view_count; v++) {
        if (views[v].use_hiz && !ctx->hiz_valid)
            return ctx->fail(GV_E_STATE, "gv_cull: view %u asks for Hi-Z but gv_hiz_build has not run", v);
    }
    if (ctx->cull_batching)
        for (const auto& j : ctx->cull_jobs)
            if (j.pool_id == pool_id) {
                // A second cull of a pool inside one batch: run the first one now — BEFORE this cull's view states are set up
                // (round 4, found by the random schedules of tests/schedules.py: the launch of the first cull ends by invalidating
                // the views beyond ITS view count, which used to switch off the views this cull had just made valid)
                if ((rc = flush_culls(ctx)) != GV_OK)
                    return rc;
                ctx->cull_batching = true;
                break;
            }
    // Views that share cameraPosition (the main camera and its shadow cascades: mesh.cpp:809-843 passes the same
    // cameraPosition to every prepareMeshes) are culled in ONE pass over the streams; Hi-Z only on view 0.
    bool batched = view_count > 1 && view_count <= kMaxBatchViews && p.occupancy > 0;
    for (uint32_t v = 1; v < view_count && batched; v++)
        batched = memcmp(views[v].camera_position, views[0].camera_position, 12) == 0 && !views[v].use_hiz;
    ViewParams vps[GV_MAX_VIEWS];
    for (uint32_t v = 0; v < view_count; v++) {
        ViewState& vs = ctx->views[pool_id][v];
        const bool emit = views[v].emit_records != 0;
        rc = reserve_view(ctx, vs, p.occupancy, emit);
        if (rc != GV_OK)
            return rc;
        vs.pool_id = pool_id;
        vs.occupancy = p.occupancy;
        vs.main_pass = views[v].shadow_pass < 0;
        vs.emitted = emit;
        vs.valid = true;
        vs.published = false, vs.records_fetched = false;
        vs.sort_pending = 0;  // a sort of the previous results that nobody asked for any more
        vs.ballots_current = true;  // every cull launch but the one-launch cull + emit of a small pool writes them
        build_view_params(views[v], &vps[v]);
        if (p.occupancy == 0)
            GV_HIP(ctx, hipMemsetAsync(vs.draw_count.ptr, 0, 4, ctx->stream));
    }
    // gv_cull_batch_begin: an engine-sized pool whose views all want records is only RECORDED here; the first read
    // launches every recorded cull together (flush_culls)
    if (ctx->cull_batching) {
        bool eligible = p.occupancy != 0 && p.occupancy <= kSmallSortMaxSlots && !ctx->sweep_with_cull &&
                        !(ctx->config.flags & GV_CONFIG_BLOCK_BOUNDS) && (view_count == 1 || batched) &&
                        ctx->cull_jobs.size() < GV_MAX_POOLS;
        for (uint32_t v = 0; v < view_count && eligible; v++)
            eligible = ctx->views[pool_id][v].emitted;
        if (eligible) {
            Context::CullJob job;
            job.pool_id = pool_id;
            job.view_count = view_count;
            for (uint32_t v = 0; v < view_count; v++)
                job.vps[v] = vps[v];
            ctx->cull_jobs.push_back(job);
            for (uint32_t v = view_count; v < GV_MAX_VIEWS; v++)
                ctx->views[pool_id][v].valid = false;
            ctx->last_pool = pool_id;
            return GV_OK;
        }
    }
    ctx->last_pool = pool_id;
    return cull_launch(ctx, pool_id, vps, view_count, batched);
}

int gv_cull_batch_begin(GvCtx* ctx)
{
    if (!ctx)
        return GV_E_ARG;
    ctx->cull_batching = true;
    return GV_OK;
}

int gv_cull_batch_end(GvCtx* ctx)
{
    if (!ctx)
        return GV_E_ARG;
    return flush_culls(ctx);
}

int gv_sweep(GvCtx* ctx, uint32_t mode)
{
    if (!ctx)
        return GV_E_ARG;
    if (mode == GV_SWEEP_WITH_CULL || mode == GV_SWEEP_WITH_CULL_VALU) {  // nothing to launch now: the next gv_cull carries the sweep
        // "the next gv_cull" is a cull ISSUED from here on: culls recorded earlier in a batch are launched first, so that none of
        // them takes the request when it is launched later (round 4, tools/schedule_soak.py: the sweep then ran over the mirror of
        // the earlier moment and the cull it was meant for went without)
        if (int rc = flush_recorded_culls(ctx, GV_MAX_POOLS))
            return rc;
        ctx->sweep_with_cull = true;
        ctx->sweep_with_cull_mfma = mode == GV_SWEEP_WITH_CULL;
        return GV_OK;
    }
    if (mode != GV_SWEEP_VALU && mode != GV_SWEEP_MFMA && mode != GV_SWEEP_INCREMENTAL)
        return ctx->fail(GV_E_ARG, "gv_sweep: unknown mode %u", mode);
    int rc = sync_mirror(ctx);
    if (rc != GV_OK)
        return rc;
    GV_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t n = ctx->xf.occupancy;
    if (mode == GV_SWEEP_INCREMENTAL && ctx->world_valid && !ctx->world_partial)
        return GV_OK;  // the cache is current: nothing to launch
    const bool subtree = mode == GV_SWEEP_INCREMENTAL && ctx->world_valid && ctx->world_partial;
    if (!subtree)
        GV_HIP(ctx, ctx->d_world.reserve((size_t)std::max(n, 1u) * 3));
    {
        KernelTimer t(ctx, GV_K_SWEEP);
        if (subtree)
            GV_HIP(ctx, launch_sweep_subtree(xf_mirror(ctx), ctx->d_xdirty.ptr, ctx->d_world.ptr, ctx->stream));
        else if (mode == GV_SWEEP_MFMA)
            GV_HIP(ctx, launch_sweep_mfma(xf_mirror(ctx), ctx->d_world.ptr, ctx->stream));
        else
            GV_HIP(ctx, launch_sweep_valu(xf_mirror(ctx), ctx->d_world.ptr, ctx->stream));
    }
    return world_current(ctx);
}

int gv_get_world(GvCtx* ctx, uint32_t first, uint32_t count, float* out12)
{
    if (!ctx)
        return GV_E_ARG;
    if (!ctx->world_valid || ctx->world_partial)
        return ctx->fail(GV_E_STATE, "gv_get_world: gv_sweep has not run since the last transform change");
    if (!out12 || (uint64_t)first + count > ctx->xf.occupancy)
        return ctx->fail(GV_E_ARG, "gv_get_world: range [%u, +%u) outside the pool", first, count);
    GV_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->xinv.empty()) {
        GV_HIP(ctx, hipMemcpyAsync(out12, ctx->d_world.ptr + (size_t)first * 3, (size_t)count * 48, hipMemcpyDeviceToHost, ctx->stream));
    } else if (count) {  // the cache is in mirror order: gather the requested pool slots on the device first
        GV_HIP(ctx, ctx->dsc_a.reserve((size_t)count * 3));
        GV_HIP(ctx, launch_gather_world(ctx->d_world.ptr, ctx->d_xinv.ptr, first, count, ctx->dsc_a.ptr, ctx->stream));
        GV_HIP(ctx, hipMemcpyAsync(out12, ctx->dsc_a.ptr, (size_t)count * 48, hipMemcpyDeviceToHost, ctx->stream));
    }
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drain_events(ctx);
    return GV_OK;
}

int gv_hiz_build(GvCtx* ctx, const float* depth, uint32_t width, uint32_t height, uint32_t mem_kind)
{
    if (!ctx)
        return GV_E_ARG;
    if (!depth || width == 0 || height == 0 || width > 32768 || height > 32768)
        return ctx->fail(GV_E_ARG, "gv_hiz_build: bad depth image %ux%u", width, height);
    if (int rc = flush_culls(ctx))  // recorded culls query the pyramid as it was when they were recorded
        return rc;
    GV_HIP(ctx, hipSetDevice(ctx->device));
    // calcMipCount(frameSize) hiz.cpp:27; sizes max(size / 2, 1) hiz.cpp:55
    uint32_t mips = 0;
    for (uint32_t m = std::max(width, height); m; m >>= 1)
        mips++;
    if (mips > GV_MAX_MIPS)
        return ctx->fail(GV_E_ARG, "gv_hiz_build: %u mips exceed GV_MAX_MIPS", mips);
    uint64_t off = 0;
    uint32_t cw = width, ch = height;
    for (uint32_t k = 0; k < mips; k++) {
        ctx->mip_w[k] = cw;
        ctx->mip_h[k] = ch;
        ctx->mip_off[k] = off;
        if (k >= 1)
            off += (uint64_t)cw * ch;
        cw = std::max(cw / 2, 1u);
        ch = std::max(ch / 2, 1u);
    }
    ctx->hiz_w = width;
    ctx->hiz_h = height;
    ctx->hiz_mips = mips;
    // The reference rule (hiz.frag:49-55) skips one texel of the extra row on odd heights, so a level is only
    // guaranteed to bound everything below it when no source level is odd, or under the conservative rule.
    ctx->hiz_nested = ctx->config.hiz_rule == GV_HIZ_RULE_CONSERVATIVE;
    if (!ctx->hiz_nested) {
        ctx->hiz_nested = true;
        for (uint32_t k = 0; k + 1 < mips; k++)
            if ((ctx->mip_w[k] > 1 && (ctx->mip_w[k] & 1u)) || (ctx->mip_h[k] > 1 && (ctx->mip_h[k] & 1u)))
                ctx->hiz_nested = false;
    }
    // level 1 stays virtual when the first six levels come from the fused kernel (sizes divisible by 64: plain 2x2 rule)
    ctx->hiz_level1_virtual = width % 64 == 0 && height % 64 == 0 && mips > 6;
    // (RG16F texels are half the size: the same buffer type, half the elements)
    GV_HIP(ctx, ctx->d_mips.reserve(std::max<uint64_t>((ctx->config.flags & GV_CONFIG_HIZ_RG16F) ? (off + 1) / 2 : off, 1)));
    GV_HIP(ctx, ctx->d_mip_offset.reserve(GV_MAX_MIPS));
    GV_HIP(ctx, hipMemcpyAsync(ctx->d_mip_offset.ptr, ctx->mip_off, sizeof(uint64_t) * GV_MAX_MIPS, hipMemcpyHostToDevice, ctx->stream));
    if (mem_kind == GV_MEM_DEVICE) {
        ctx->depth_ptr = depth;
    } else {
        GV_HIP(ctx, ctx->d_depth.reserve((size_t)width * height));
        GV_HIP(ctx, hipMemcpyAsync(ctx->d_depth.ptr, depth, (size_t)width * height * 4, hipMemcpyHostToDevice, ctx->stream));
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // caller's (pageable) buffer may go away
        ctx->depth_ptr = ctx->d_depth.ptr;
    }
    ctx->hiz_valid = true;
    return hiz_reduce(ctx);
}

int gv_hiz_rebuild(GvCtx* ctx)
{
    if (!ctx)
        return GV_E_ARG;
    if (!ctx->hiz_valid)
        return ctx->fail(GV_E_STATE, "gv_hiz_rebuild: no depth image resident");
    if (int rc = flush_culls(ctx))
        return rc;
    GV_HIP(ctx, hipSetDevice(ctx->device));
    return hiz_reduce(ctx);
}

int gv_hiz_mip_count(GvCtx* ctx, uint32_t* mip_count)
{
    if (!ctx)
        return GV_E_ARG;
    if (!ctx->hiz_valid || !mip_count)
        return ctx->fail(GV_E_STATE, "gv_hiz_mip_count: no pyramid");
    *mip_count = ctx->hiz_mips;
    return GV_OK;
}

int gv_hiz_read_level(GvCtx* ctx, uint32_t level, float* out_pairs, uint32_t* w, uint32_t* h)
{
    if (!ctx)
        return GV_E_ARG;
    if (!ctx->hiz_valid)
        return ctx->fail(GV_E_STATE, "gv_hiz_read_level: no pyramid");
    if (level == 0 || level >= ctx->hiz_mips || !out_pairs)
        return ctx->fail(GV_E_ARG, "gv_hiz_read_level: level %u (valid 1..%u)", level, ctx->hiz_mips - 1);
    GV_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)ctx->mip_w[level] * ctx->mip_h[level];
    if (level == 1 && ctx->hiz_level1_virtual && !ctx->hiz_level1_stored) {  // on demand: one generic level pass
        GV_HIP(ctx, launch_hiz_level(ctx->depth_ptr, nullptr, mip_ptr(ctx, 1), ctx->mip_w[0], ctx->mip_h[0], ctx->mip_w[1], ctx->mip_h[1],
                                     ctx->config.hiz_rule, (ctx->config.flags & GV_CONFIG_HIZ_RG16F) != 0, ctx->stream));
        ctx->hiz_level1_stored = true;
    }
    if (ctx->config.flags & GV_CONFIG_HIZ_RG16F) {
        // the texels arrive packed in the upper half of the caller's buffer and are widened in place, front to back
        // (binary16 -> binary32 is exact): out_pairs holds the values the RG16F image holds
        uint32_t* packed = reinterpret_cast<uint32_t*>(out_pairs) + n;
        GV_HIP(ctx, hipMemcpyAsync(packed, mip_ptr(ctx, level), n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < n; i++) {
            const uint32_t t = packed[i];  // read before out_pairs[2i + 1] can reach it (2i + 1 <= n + i)
            out_pairs[2 * i] = (float)__builtin_bit_cast(_Float16, (unsigned short)(t & 0xFFFFu));
            out_pairs[2 * i + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(t >> 16));
        }
    } else {
        GV_HIP(ctx, hipMemcpyAsync(out_pairs, ctx->d_mips.ptr + ctx->mip_off[level], n * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    drain_events(ctx);
    if (w)
        *w = ctx->mip_w[level];
    if (h)
        *h = ctx->mip_h[level];
    return GV_OK;
}

int gv_stats(GvCtx* ctx, GvStats* out)
{
    if (!ctx || !out)
        return GV_E_ARG;
    if (!ctx->pending.empty()) {
        GV_HIP(ctx, hipSetDevice(ctx->device));
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        drain_events(ctx);
    }
    ctx->stats.bounds_blocks_total = ctx->bounds_blocks_total;
    ctx->stats.bounds_blocks_examined = 0;
    if (ctx->d_examined.ptr && ctx->bounds_blocks_total) {
        std::vector<uint8_t> flags(ctx->bounds_blocks_total);
        GV_HIP(ctx, hipSetDevice(ctx->device));
        GV_HIP(ctx, hipMemcpyAsync(flags.data(), ctx->d_examined.ptr, flags.size(), hipMemcpyDeviceToHost, ctx->stream));
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (uint8_t f : flags)
            ctx->stats.bounds_blocks_examined += f;
    }
    ctx->stats.max_depth = ctx->max_depth;
    ctx->stats.transform_count = ctx->xf.occupancy;
    for (uint32_t i = 0; i < GV_MAX_POOLS; i++)
        ctx->stats.mesh_count[i] = ctx->pools[i].bound ? ctx->pools[i].occupancy : 0;
    *out = ctx->stats;
    return GV_OK;
}

int gv_stats_reset(GvCtx* ctx)
{
    if (!ctx)
        return GV_E_ARG;
    if (!ctx->pending.empty()) {
        GV_HIP(ctx, hipSetDevice(ctx->device));
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        drain_events(ctx);
    }
    memset(ctx->stats.launches, 0, sizeof(ctx->stats.launches));
    memset(ctx->stats.device_ms, 0, sizeof(ctx->stats.device_ms));
    ctx->stats.upload_bytes = 0;
    ctx->stats.exchanges = ctx->stats.exchange_tail_rounds = 0;
    ctx->bounds_blocks_total = 0;
    memset(ctx->profile_seen, 0, sizeof(ctx->profile_seen));
    memset(ctx->profile_timed, 0, sizeof(ctx->profile_timed));
    return GV_OK;
}

int gv_profile_sampling(GvCtx* ctx, uint32_t every)
{
    if (!ctx)
        return GV_E_ARG;
    if (every == 0)
        return ctx->fail(GV_E_ARG, "gv_profile_sampling: every must be at least 1");
    ctx->profile_every = every;
    memset(ctx->profile_seen, 0, sizeof(ctx->profile_seen));
    return GV_OK;
}

int gv_profile_kernels(GvCtx* ctx, uint32_t kernel_mask)
{
    if (!ctx)
        return GV_E_ARG;
    if (kernel_mask >> GV_K_COUNT)
        return ctx->fail(GV_E_ARG, "gv_profile_kernels: mask 0x%x names kernels beyond GV_K_COUNT", kernel_mask);
    ctx->profile_mask = kernel_mask;
    memset(ctx->profile_seen, 0, sizeof(ctx->profile_seen));
    return GV_OK;
}

int gv_profile_samples(GvCtx* ctx, uint64_t samples[GV_K_COUNT])
{
    if (!ctx || !samples)
        return GV_E_ARG;
    memcpy(samples, ctx->profile_timed, sizeof(ctx->profile_timed));
    return GV_OK;
}

int gv_debug_stream_peak(GvCtx* ctx, uint32_t pool_id, uint32_t launches, double* gb_per_s)
{
    if (!ctx)
        return GV_E_ARG;
    if (pool_id >= GV_MAX_POOLS || !gb_per_s || launches == 0 || launches > 1000)
        return ctx->fail(GV_E_ARG, "gv_debug_stream_peak: bad argument");
    PoolState& p = ctx->pools[pool_id];
    if (!p.bound || !ctx->xf.bound)
        return ctx->fail(GV_E_STATE, "gv_debug_stream_peak: pools not bound");
    if (int rc = sync_mirror(ctx))
        return rc;
    GV_HIP(ctx, hipSetDevice(ctx->device));
    const MeshMirror mesh{p.d_a.ptr, p.d_b.ptr, p.d_link.ptr, p.occupancy, p.mapping, nullptr};
    const TransformMirror xf = xf_mirror(ctx);
    const uint32_t n = std::min(mesh.count, xf.count);
    *gb_per_s = 0.0;
    if (n == 0)
        return GV_OK;
    GV_HIP(ctx, ctx->dsc_c.reserve(1));
    float* sink = reinterpret_cast<float*>(ctx->dsc_c.ptr);
    hipEvent_t e0, e1;
    GV_HIP(ctx, hipEventCreate(&e0));
    GV_HIP(ctx, hipEventCreate(&e1));
    std::vector<float> ms;
    hipError_t err = hipSuccess;
    for (uint32_t k = 0; k < launches + 3 && err == hipSuccess; k++) {  // three untimed warm-ups
        (void)hipEventRecord(e0, ctx->stream);
        err = launch_stream_probe(mesh, xf, sink, ctx->stream);
        (void)hipEventRecord(e1, ctx->stream);
        if (err == hipSuccess)
            err = hipEventSynchronize(e1);
        float t = 0.0f;
        if (err == hipSuccess)
            err = hipEventElapsedTime(&t, e0, e1);
        if (k >= 3)
            ms.push_back(t);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (err != hipSuccess)
        return ctx->hip_fail(err, "gv_debug_stream_peak");
    std::sort(ms.begin(), ms.end());
    const double median = ms[ms.size() / 2];
    *gb_per_s = median > 0.0 ? 65.0 * n / (median * 1e-3) / 1e9 : 0.0;
    return GV_OK;
}

void* gv_stream(GvCtx* ctx) { return ctx ? static_cast<void*>(ctx->stream) : nullptr; }

void gv_host_parallel_tasks(uint32_t count, void (*fn)(void* user, uint32_t task), void* user)
{
    if (!fn || !count)
        return;
    static const uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
    const uint32_t parts = std::min(count, std::min(hw, 16u));
    std::atomic<uint32_t> next{0};
    run_parts(parts, [&](uint32_t) {  // every part takes tasks until none is left
        for (uint32_t task = next.fetch_add(1, std::memory_order_relaxed); task < count; task = next.fetch_add(1, std::memory_order_relaxed))
            fn(user, task);
    });
}

void gv_host_parallel_ranges(uint32_t first, uint32_t count, void (*fn)(void* user, uint32_t lo, uint32_t hi), void* user)
{
    if (fn && count)
        parallel_ranges(first, count, [&](uint32_t lo, uint32_t hi) { fn(user, lo, hi); });
}

}  // extern "C"
