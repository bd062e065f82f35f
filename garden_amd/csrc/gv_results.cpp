// gv_results.cpp — the reader side of the C-ABI of include/garden_vis.h: what delivers a view's results — waits, counts, fetches
// (the publish launch of engine-sized pools, record structs, record targets, instance bases), device-side accessors and shard
// copies, the device sort — and the flushes every reader starts with (recorded culls, deferred small-pool sorts). Split from
// gv_context.cpp in round 4 (binds, per-frame dispatch, Hi-Z and sweeps stay there); shared pieces are declared in gv_ctx.hpp.
//
// Replaces (reference paths): the wait + sort at the end of MeshRenderSystem::prepareMeshes (source/system/render/mesh.cpp:548-553,
// sortMeshes :265-328) and the hand-over of combinedMeshes / drawCount / instanceCount to the render passes (:556-770).
#include "gv_ctx.hpp"
#include "gv_hiz_kernels.hpp"

#include <sys/mman.h>
#include <unistd.h>

using namespace gv;

namespace gv {

// gv_sort of a pool too large for the one-launch batch: the record count on the device picks rank or radix sort (launch_sort)
// Buffers and launch arguments of a large sort of `vs`: inputs = the view's current records, outputs = its alternate set.
static int sort_buffers_of(GvCtx* ctx, ViewState& vs, SortBuffers& b)
{
    const size_t n = vs.occupancy;  // upper bound of draw_count, known without a readback
    const size_t nblocks = sort_tile_count((uint32_t)n);
    GV_HIP(ctx, vs.alt_idx.reserve(n));
    GV_HIP(ctx, vs.alt_model.reserve(n * 12));
    GV_HIP(ctx, vs.alt_dist.reserve(n));
    for (int k = 0; k < 2; k++) {
        GV_HIP(ctx, vs.sort_keys[k].reserve(n));
        GV_HIP(ctx, vs.sort_vals[k].reserve(n));
        GV_HIP(ctx, vs.sort_slots[k].reserve(n));
    }
    // sort_hist: [2 sets of counters (global + per-group digit histograms)] + the tiles' digit counts
    const size_t set_words = sort_set_words((uint32_t)n);
    const size_t want = 2 * set_words + nblocks * 256;
    if (want > vs.sort_hist.cap) {
        GV_HIP(ctx, vs.sort_hist.reserve(want));
        vs.sort_set_words = 0;
    }
    if (vs.sort_set_words != set_words) {  // (a pool that changed size moves the sets: both start at zero again)
        GV_HIP(ctx, hipMemsetAsync(vs.sort_hist.ptr, 0, 2 * set_words * sizeof(uint32_t), ctx->stream));
        vs.sort_set_words = set_words;
        vs.sort_parity = 0;
    }
    GV_HIP(ctx, vs.sort_ranks.reserve(n));
    b = SortBuffers{};
    b.count = vs.draw_count.ptr;
    b.idx_in = vs.visible_idx.ptr;
    b.model_in = vs.baked_model.ptr;
    b.dist_in = vs.distance_sq.ptr;
    b.idx_out = vs.alt_idx.ptr;
    b.model_out = vs.alt_model.ptr;
    b.dist_out = vs.alt_dist.ptr;
    b.ranks = vs.sort_ranks.ptr;
    for (int k = 0; k < 2; k++) {
        b.keys[k] = vs.sort_keys[k].ptr;
        b.vals[k] = vs.sort_vals[k].ptr;
        b.slots[k] = vs.sort_slots[k].ptr;
        b.counters[k] = vs.sort_hist.ptr + k * set_words;
    }
    b.tile_hist = vs.sort_hist.ptr + 2 * set_words;
    b.parity = vs.sort_parity;
    return GV_OK;
}

// gv_sort of a pool too large for the one-launch batch: the record count on the device picks rank or radix sort (launch_sort).
// sort_large_prepare: the view's buffers and what to enqueue for it; sort_large_done: the sorted records are the view's records.
static int sort_large_prepare(GvCtx* ctx, ViewState& vs, bool descending, SortBatchEntry& e)
{
    const size_t n = vs.occupancy;
    if (int rc = sort_buffers_of(ctx, vs, e.b))
        return rc;
    // the previous frame's count says what to enqueue for a mid-sized pool: a short list gets the rank sort alone
    e.mode = vs.count_hint == 0xFFFFFFFFu ? kSortBoth
             : vs.count_hint <= kRankOnlyHintRecords ? kSortRankOnly
             : vs.count_hint > 2 * kRankSortMaxRecords ? kSortRadixOnly : kSortBoth;
    e.capacity = (uint32_t)n;
    e.descending = descending ? 1u : 0u;
    if (!sort_is_rank_only((uint32_t)n, e.mode))
        vs.sort_parity ^= 1u;  // the radix passes leave the other set of counters zeroed for the next sort
    return GV_OK;
}
static void sort_large_done(ViewState& vs)
{
    vs.published = false, vs.records_fetched = false;
    // the sorted records now live in the alternate set: swap it in
    std::swap(vs.visible_idx, vs.alt_idx);
    std::swap(vs.baked_model, vs.alt_model);
    std::swap(vs.distance_sq, vs.alt_dist);
}
static int sort_large(GvCtx* ctx, ViewState& vs, bool descending)
{
    GV_HIP(ctx, hipSetDevice(ctx->device));
    SortBatchEntry e{};
    if (int rc = sort_large_prepare(ctx, vs, descending, e))
        return rc;
    {
        KernelTimer t(ctx, GV_K_SORT);
        GV_HIP(ctx, launch_sort(e.b, e.capacity, descending, ctx->stream, e.mode));
    }
    sort_large_done(vs);
    return GV_OK;
}

// gv_sort on a small pool only records the request; the first call that needs the records (fetch, device accessors,
// gv_wait) sorts every pending view of EVERY pool in ONE launch — five mesh systems with a main camera and three shadow
// passes each cost one launch, not twenty.
// Everything queued on the context's stream has finished — where hipStreamSynchronize would do, for the end of an engine-sized
// tick: a one-lane kernel behind the queue writes a sequence number into pinned memory and the host polls it
// (launch_done_flag). Falls back to the synchronisation when profiling events wait to be read or when the word does not arrive
// within 2 ms (a long queue: let the runtime sleep).
int wait_for_stream(GvCtx* ctx)
{
    if (!ctx->pending.empty()) {
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return GV_OK;
    }
    if (!ctx->h_done.ptr) {
        GV_HIP(ctx, ctx->h_done.reserve(16));
        memset(ctx->h_done.ptr, 0, 16 * sizeof(uint32_t));
    }
    const uint32_t seq = ++ctx->done_seq;
    GV_HIP(ctx, launch_done_flag(ctx->h_done.ptr, seq, ctx->stream));
    volatile uint32_t* word = ctx->h_done.ptr;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; *word != seq; spins++) {
        if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
            GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return GV_OK;
}

// What delivers the results of (pool pid, view v) to the host: the count, the records [0, count) (three arrays, or the
// pool's record structs — in the library's pinned buffer or the caller's own page-locked array) and, for a main pass, the
// isVisible bytes in pool-slot order. Buffers are reserved here; used by the publish launch of gv_pool_results_fetch and by the
// small-pool sort, which publishes what it has just sorted.
static int delivered_layout(GvCtx* ctx, const PoolState& pool, uint32_t pool_id, uint32_t occupancy, RecordLayout& L);
int publish_args_of(GvCtx* ctx, uint32_t pid, uint32_t v, PublishArgs& a)
{
    PoolState& wp = ctx->pools[pid];
    ViewState& w = ctx->views[pid][v];
    const bool wperm = !wp.perm.empty() && wp.perm.size() == w.occupancy;
    a = PublishArgs{};
    a.count = w.draw_count.ptr;
    a.idx = w.visible_idx.ptr;
    a.model = w.baked_model.ptr;
    a.dist = w.distance_sq.ptr;
    a.is_visible = w.is_visible.ptr;
    GV_HIP(ctx, w.h_draw_count.reserve(4));
    a.host_count = w.h_draw_count.ptr;
    w.records_staged = false;
    if (w.emitted && wp.record_layout.stride) {
        const PoolState::RecordTarget& target = wp.record_target[v];
        const size_t need = (size_t)w.occupancy * wp.record_layout.stride;
        if (target.host && target.bytes < need)
            return ctx->fail(GV_E_ARG, "gv_results_fetch: the record target of pool %u view %u holds %zu bytes, occupancy * stride = %zu", pid, v,
                             target.bytes, need);
        GV_HIP(ctx, w.h_records.reserve(need));
        a.host_records = w.h_records.ptr;
        w.records_at = target.host ? target.host : w.h_records.ptr;
        w.records_staged = target.host != nullptr;  // (never page-locked: filled from h_records after the synchronisation)
        if (int rc = delivered_layout(ctx, wp, pid, w.occupancy, a.layout))
            return rc;
        w.records_fetched = true;
    } else if (w.emitted) {
        GV_HIP(ctx, w.h_visible_idx.reserve(w.occupancy));
        GV_HIP(ctx, w.h_baked_model.reserve((size_t)w.occupancy * 12));
        GV_HIP(ctx, w.h_distance_sq.reserve(w.occupancy));
        a.host_idx = w.h_visible_idx.ptr;
        a.host_model = w.h_baked_model.ptr;
        a.host_dist = w.h_distance_sq.ptr;
    }
    a.orig = wperm ? wp.d_orig.ptr : nullptr;
    if (w.main_pass) {
        GV_HIP(ctx, w.h_is_visible.reserve(w.occupancy));
        a.host_is_visible = w.h_is_visible.ptr;
        if (wperm && w.occupancy > kPublishLdsSlots) {  // too large for the in-LDS un-permutation
            GV_HIP(ctx, w.is_visible_slots.reserve(w.occupancy));
            GV_HIP(ctx, launch_unpermute_bytes(w.is_visible.ptr, wp.d_orig.ptr, w.occupancy, w.is_visible_slots.ptr, ctx->stream));
            a.is_visible = w.is_visible_slots.ptr;
            a.orig = nullptr;
        }
    }
    a.occupancy = w.occupancy;
    return GV_OK;
}

int flush_sorts(GvCtx* ctx)
{
    if (int rc = flush_culls(ctx))  // the records about to be sorted / read may still be waiting to be culled
        return rc;
    // Mid-sized pools (beyond the one-launch batch, up to kMidSortMaxSlots slots): ALL the pending lists of the frame by ONE set of
    // launches — the rank-sort launch and the eight radix launches with blockIdx.y = list (launch_sort_batch). A frame of seven mesh
    // systems x four passes at 10^5 slots each was 150-250 short dependent launches, ~2 ms of a 4 ms tick.
    for (;;) {
        SortBatchEntry batch[kMaxSortBatch];
        ViewState* taken[kMaxSortBatch];
        uint32_t n = 0;
        int failed = GV_OK;  // a view whose buffers could not be had keeps its request; the views in front of it are sorted all the same
        for (uint32_t pool = 0; pool < GV_MAX_POOLS && n < kMaxSortBatch && failed == GV_OK; pool++)
            for (uint32_t v = 0; v < GV_MAX_VIEWS && n < kMaxSortBatch; v++) {
                ViewState& vs = ctx->views[pool][v];
                if (!vs.valid || !vs.sort_pending || vs.occupancy <= kBatchSortMaxSlots)
                    continue;
                if ((failed = sort_large_prepare(ctx, vs, vs.sort_pending == 2, batch[n])) != GV_OK)
                    break;
                vs.sort_pending = 0;
                taken[n++] = &vs;
            }
        if (n) {
            GV_HIP(ctx, hipSetDevice(ctx->device));
            {
                ZoneScope zone("Meshes Sort");
                KernelTimer t(ctx, GV_K_SORT);
                GV_HIP(ctx, launch_sort_batch(batch, n, ctx->stream));
            }
            for (uint32_t k = 0; k < n; k++)
                sort_large_done(*taken[k]);
        }
        if (failed != GV_OK)
            return failed;
        if (n == 0)
            break;
    }
    for (;;) {
        uint32_t widest = 0, views = 0;
        ViewState* taken[kMaxPublishViews];
        SortBatch batch{};
        // The sort publishes what it has sorted (count, records at their sorted places, isVisible: one kernel boundary and the
        // publish kernel's own dependent loads less per tick) — unless other small views wait for a publish launch anyway
        // (a tick with unsorted OIT buffers): then that launch takes these views along and the sort stays lean.
        bool fuse_publish = true;
        for (uint32_t pool = 0; pool < GV_MAX_POOLS && fuse_publish; pool++)
            for (uint32_t v = 0; v < GV_MAX_VIEWS; v++) {
                const ViewState& w = ctx->views[pool][v];
                if (w.valid && !w.published && !w.sort_pending && w.occupancy != 0 && w.occupancy <= kPublishMaxSlots)
                    fuse_publish = false;
                // ... and only records that leave as whole structs: a record written at its sorted place is one aligned 64- /
                // 80-byte piece; the three arrays would leave as scattered 4- and 48-byte pieces, which measured slower
                // (10 k entities: 50.8 vs 46.3 us per tick) than the publish kernel's contiguous rows
                if (w.valid && w.sort_pending && w.occupancy <= kBatchSortMaxSlots && !ctx->pools[pool].record_layout.stride)
                    fuse_publish = false;
            }
        for (uint32_t pool = 0; pool < GV_MAX_POOLS && views < kMaxPublishViews; pool++)
            for (uint32_t v = 0; v < GV_MAX_VIEWS && views < kMaxPublishViews; v++) {
                ViewState& vs = ctx->views[pool][v];
                if (!vs.valid || !vs.sort_pending)
                    continue;
                const size_t n = vs.occupancy;
                GV_HIP(ctx, vs.alt_idx.reserve(n));
                GV_HIP(ctx, vs.alt_model.reserve(n * 12));
                GV_HIP(ctx, vs.alt_dist.reserve(n));
                SmallSortEntry& b = batch.view[views];
                b.count = vs.draw_count.ptr;
                b.idx_in = vs.visible_idx.ptr;
                b.model_in = vs.baked_model.ptr;
                b.dist_in = vs.distance_sq.ptr;
                b.idx_out = vs.alt_idx.ptr;
                b.model_out = vs.alt_model.ptr;
                b.dist_out = vs.alt_dist.ptr;
                b.capacity = vs.occupancy;
                b.descending = vs.sort_pending == 2 ? 1u : 0u;
                b.fused_publish = fuse_publish ? 1u : 0u;
                if (fuse_publish)
                    if (int rc = publish_args_of(ctx, pool, v, b.publish))
                        return rc;
                widest = std::max(widest, vs.occupancy);
                taken[views++] = &vs;
            }
        if (views == 0)
            return GV_OK;
        GV_HIP(ctx, hipSetDevice(ctx->device));
        {
            ZoneScope zone("Meshes Sort");
            KernelTimer t(ctx, GV_K_SORT);
            GV_HIP(ctx, launch_sort_small_batch(batch, views, widest, ctx->stream));
        }
        for (uint32_t k = 0; k < views; k++) {  // the sorted records now live in the alternate set: swap it in
            ViewState& vs = *taken[k];
            std::swap(vs.visible_idx, vs.alt_idx);
            std::swap(vs.baked_model, vs.alt_model);
            std::swap(vs.distance_sq, vs.alt_dist);
            vs.sort_pending = 0;
            if (batch.view[k].fused_publish) {
                vs.published = true;  // ... once the stream has been synchronised
                ctx->publish_sync_pending = true;
            } else {
                vs.published = false, vs.records_fetched = false;
            }
        }
    }
}

// The pool's record layout as the kernels take it: with GV_RESULTS_MAP_RECORDS the records carry the caller's WORLD slots.
static int delivered_layout(GvCtx* ctx, const PoolState& pool, uint32_t pool_id, uint32_t occupancy, RecordLayout& L)
{
    L = pool.record_layout;
    L.slot_map = nullptr;
    if (pool.result_flags & GV_RESULTS_MAP_RECORDS) {
        if (pool.index_map_count < occupancy)
            return ctx->fail(GV_E_STATE, "gv_results_fetch: pool %u delivers records in world slots (gv_pool_set_result_mapping), but its index map covers %u of "
                             "%u slots", pool_id, pool.index_map_count, occupancy);
        L.slot_map = pool.d_index_map.ptr;
    }
    return GV_OK;
}

ViewState* view_of(GvCtx* ctx, uint32_t pool_id, uint32_t view_index)
{
    if (pool_id >= GV_MAX_POOLS || view_index >= GV_MAX_VIEWS || !ctx->views[pool_id][view_index].valid)
        return nullptr;
    return &ctx->views[pool_id][view_index];
}


// Is every page of [p, p + bytes) still mapped? (msync fails with ENOMEM otherwise.) A freed std::vector of this size — the
// engine's combinedMeshes that was let go, or reallocated, while it was still the record target — is an unmapped range.
static bool range_mapped(const void* p, size_t bytes)
{
    const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
    const uintptr_t lo = (uintptr_t)p & ~(page - 1), hi = ((uintptr_t)p + bytes + page - 1) & ~(page - 1);
    return msync(reinterpret_cast<void*>(lo), hi - lo, MS_ASYNC) == 0;
}

// false: the caller let go of the memory while it was still the record target (include/garden_vis.h: the range must stay
// allocated until it is replaced, removed or the context destroyed); gv_pool_set_record_target reports that
bool release_record_target(PoolState::RecordTarget& target)
{
    bool intact = true;
    if (target.host)
        intact = range_mapped(target.host, target.bytes);
    target = PoolState::RecordTarget{};
    return intact;
}

}  // namespace gv

// ================================================================================================
// C-ABI
// ================================================================================================
extern "C" {

int gv_wait(GvCtx* ctx)
{
    if (!ctx)
        return GV_E_ARG;
    if (int rc = flush_sorts(ctx))
        return rc;
    GV_HIP(ctx, hipSetDevice(ctx->device));
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drain_events(ctx);
    return GV_OK;
}

int gv_result_count(GvCtx* ctx, uint32_t view_index, uint32_t* draw_count)
{
    return ctx ? gv_pool_result_count(ctx, ctx->last_pool, view_index, draw_count) : GV_E_ARG;
}

int gv_pool_result_count(GvCtx* ctx, uint32_t pool_id, uint32_t view_index, uint32_t* draw_count)
{
    if (!ctx)
        return GV_E_ARG;
    if (!draw_count || !view_of(ctx, pool_id, view_index))
        return ctx->fail(GV_E_ARG, "gv_result_count: pool %u view %u has no results", pool_id, view_index);
    if (int rc = flush_culls(ctx))
        return rc;
    ViewState& vs = *view_of(ctx, pool_id, view_index);
    GV_HIP(ctx, hipSetDevice(ctx->device));
    GV_HIP(ctx, hipMemcpyAsync(vs.h_draw_count.ptr, vs.draw_count.ptr, 4, hipMemcpyDeviceToHost, ctx->stream));
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drain_events(ctx);
    *draw_count = vs.h_draw_count.ptr[0];
    return GV_OK;
}

int gv_results_fetch(GvCtx* ctx, uint32_t view_index, int write_back, GvResult* out)
{
    return ctx ? gv_pool_results_fetch(ctx, ctx->last_pool, view_index, write_back, out) : GV_E_ARG;
}

// a byte per component at the component's stride: every store is a cache line of its own
#ifndef GV_WRITE_BACK_FLOOR  // (A/B builds)
#define GV_WRITE_BACK_FLOOR 49152
#endif
constexpr uint32_t kWriteBackFloor = GV_WRITE_BACK_FLOOR;

int gv_pool_results_fetch(GvCtx* ctx, uint32_t pool_id, uint32_t view_index, int write_back, GvResult* out)
{
    if (!ctx)
        return GV_E_ARG;
    if (!out)
        return ctx->fail(GV_E_ARG, "gv_results_fetch: out is NULL");
    if (!view_of(ctx, pool_id, view_index))
        return ctx->fail(GV_E_ARG, "gv_results_fetch: pool %u view %u has no results", pool_id, view_index);
    if (int rc = flush_sorts(ctx))
        return rc;
    ViewState& vs = *view_of(ctx, pool_id, view_index);
    PoolState& pool = ctx->pools[vs.pool_id];
    const bool permuted = !pool.perm.empty() && pool.perm.size() == vs.occupancy;
    const bool small = vs.occupancy != 0 && vs.occupancy <= kPublishMaxSlots;
    auto reserve_records = [&]() -> int {
        GV_HIP(ctx, vs.h_visible_idx.reserve(vs.occupancy));
        GV_HIP(ctx, vs.h_baked_model.reserve((size_t)vs.occupancy * 12));
        GV_HIP(ctx, vs.h_distance_sq.reserve(vs.occupancy));
        return GV_OK;
    };
    const bool want_vis = vs.main_pass && vs.occupancy;
    if (want_vis)
        GV_HIP(ctx, vs.h_is_visible.reserve(vs.occupancy));
    uint32_t count = 0;
    if (small) {
        // engine-sized pools are launch- and round-trip-bound: one kernel writes count, records and isVisible of EVERY
        // view of this cull (the main camera and its shadow passes are fetched one after the other, mesh.cpp:809-843)
        // straight into the pinned host buffers, one synchronisation ends the frame; the sibling views' fetches find
        // their results already there
        if (!vs.published || ctx->publish_sync_pending) {
            GV_HIP(ctx, hipSetDevice(ctx->device));
            static_assert(kMaxPublishViews >= GV_MAX_VIEWS, "PublishBatch holds at least one pool's views");
            // ... of EVERY pool culled since the last fetch: a frame that culls all its mesh systems first and reads
            // afterwards (gv_pool_results_fetch) ends with this one launch and one synchronisation. Views whose small-pool
            // sort has published them already (flush_sorts) only wait for that synchronisation.
            PublishBatch batch{};
            uint32_t views = 0, widest = 0;
            ViewState* sent[kMaxPublishViews];
            for (uint32_t q = 0; q < GV_MAX_POOLS && views < kMaxPublishViews; q++) {
                const uint32_t pid = (pool_id + q) % GV_MAX_POOLS;  // the pool asked for first: it always fits
                for (uint32_t v = 0; v < GV_MAX_VIEWS && views < kMaxPublishViews; v++) {
                    ViewState& w = ctx->views[pid][v];
                    if (!w.valid || w.published || w.occupancy == 0 || w.occupancy > kPublishMaxSlots)
                        continue;
                    if (int rc = publish_args_of(ctx, pid, v, batch.view[views]))
                        return rc;
                    widest = std::max(widest, w.occupancy);
                    sent[views++] = &w;
                }
            }
            if (views)
                GV_HIP(ctx, launch_publish(batch, views, widest, ctx->stream));
            if (int rc = wait_for_stream(ctx))
                return rc;
            drain_events(ctx);
            ctx->publish_sync_pending = false;
            for (uint32_t k = 0; k < views; k++)
                sent[k]->published = true;
            // record targets are the caller's pageable arrays: the records cross the host once more, from the pinned buffer the publish
            // kernel wrote. All views of the frame in ONE pass — in 256 KB pieces over the worker threads from 1 MB up (seven mesh
            // systems' lists at 10^6 entities: 8 MB, ~1 ms of one thread)
            struct Piece {
                uint8_t* to;
                const uint8_t* from;
                size_t bytes;
            };
            std::vector<Piece> pieces;
            size_t staged_bytes = 0;
            constexpr size_t kPiece = (size_t)256 << 10;
            for (auto& per_pool : ctx->views)
                for (ViewState& w : per_pool)
                    if (w.valid && w.published && w.records_staged) {
                        const size_t bytes = (size_t)w.h_draw_count.ptr[0] * ctx->pools[w.pool_id].record_layout.stride;
                        if (bytes && !range_mapped(w.records_at, bytes))
                            return ctx->fail(GV_E_STATE, "gv_results_fetch: the record target of pool %u is not mapped any more (freed while it was "
                                                         "still the target?)", w.pool_id);
                        for (size_t at = 0; at < bytes; at += kPiece)
                            pieces.push_back(Piece{w.records_at + at, w.h_records.ptr + at, std::min(kPiece, bytes - at)});
                        staged_bytes += bytes;
                        w.records_staged = false;
                    }
            const uint32_t parts = staged_bytes >= ((size_t)1 << 20) ? std::min<uint32_t>((uint32_t)pieces.size(), worker_parts((size_t)1 << 30)) : 1u;
            run_parts(parts, [&](uint32_t t) {
                for (size_t k = t; k < pieces.size(); k += parts)
                    memcpy(pieces[k].to, pieces[k].from, pieces[k].bytes);
            });
        }
        count = vs.h_draw_count.ptr[0];
    } else {
        int rc = gv_pool_result_count(ctx, pool_id, view_index, &count);
        if (rc != GV_OK)
            return rc;
        if (vs.emitted && pool.record_layout.stride)
            vs.records_fetched = true;
        const PoolState::RecordTarget& target = pool.record_target[view_index];
        uint8_t* staged_for = nullptr;  // a target that could not be page-locked: filled from h_records after the copies
        if (vs.emitted && pool.record_layout.stride) {
            const size_t need = (size_t)vs.occupancy * pool.record_layout.stride;
            if (target.host && target.bytes < need)
                return ctx->fail(GV_E_ARG, "gv_results_fetch: the record target of pool %u view %u holds %zu bytes, occupancy * stride = %zu",
                                 pool_id, view_index, target.bytes, need);
            GV_HIP(ctx, vs.h_records.reserve(need));
            vs.records_at = target.host ? target.host : vs.h_records.ptr;
        }
        if (vs.emitted && count && pool.record_layout.stride) {  // packed on the device, one copy
            const size_t bytes = (size_t)count * pool.record_layout.stride;
            GV_HIP(ctx, vs.d_records.reserve((size_t)vs.occupancy * pool.record_layout.stride));
            RecordLayout delivered;
            if ((rc = delivered_layout(ctx, pool, pool_id, vs.occupancy, delivered)) != GV_OK)
                return rc;
            GV_HIP(ctx, launch_pack_records(vs.draw_count.ptr, vs.visible_idx.ptr, vs.baked_model.ptr, vs.distance_sq.ptr, delivered,
                                            count, vs.d_records.ptr, ctx->stream));
            staged_for = target.host;
            GV_HIP(ctx, hipMemcpyAsync(vs.h_records.ptr, vs.d_records.ptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
        } else if (vs.emitted && count) {
            if ((rc = reserve_records()) != GV_OK)
                return rc;
            GV_HIP(ctx, hipMemcpyAsync(vs.h_visible_idx.ptr, vs.visible_idx.ptr, (size_t)count * 4, hipMemcpyDeviceToHost, ctx->stream));
            GV_HIP(ctx, hipMemcpyAsync(vs.h_baked_model.ptr, vs.baked_model.ptr, (size_t)count * 48, hipMemcpyDeviceToHost, ctx->stream));
            GV_HIP(ctx, hipMemcpyAsync(vs.h_distance_sq.ptr, vs.distance_sq.ptr, (size_t)count * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        if (want_vis) {
            const uint8_t* src = vs.is_visible.ptr;
            if (permuted) {  // back into pool-slot order on the device: the random half of the write-back
                GV_HIP(ctx, vs.is_visible_slots.reserve(vs.occupancy));
                GV_HIP(ctx, launch_unpermute_bytes(vs.is_visible.ptr, pool.d_orig.ptr, vs.occupancy, vs.is_visible_slots.ptr, ctx->stream));
                src = vs.is_visible_slots.ptr;
            }
            GV_HIP(ctx, hipMemcpyAsync(vs.h_is_visible.ptr, src, vs.occupancy, hipMemcpyDeviceToHost, ctx->stream));
        }
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (staged_for) {
            if (!range_mapped(staged_for, (size_t)count * pool.record_layout.stride))
                return ctx->fail(GV_E_STATE, "gv_results_fetch: the record target of pool %u is not mapped any more (freed while it was still "
                                             "the target?)", pool_id);
            // (the caller's array is pageable: the records cross the host once more — on the worker threads from 128 Ki records up,
            // 14 MB at 10^6 entities was ~1 ms of one thread)
            const size_t stride = pool.record_layout.stride;
            uint8_t* const from = vs.h_records.ptr;
            parallel_ranges(0, count, [&](uint32_t a, uint32_t b) { memcpy(staged_for + (size_t)a * stride, from + (size_t)a * stride, (size_t)(b - a) * stride); });
        }
    }
    vs.count_hint = count;  // (what the next frame's sort of this view expects)
    memset(out, 0, sizeof(*out));
    out->draw_count = count;
    out->instance_count = count;  // default getReadyMeshesAsync returns 0/1 (render/mesh.hpp:142-146)
    const bool as_records = vs.emitted && vs.records_fetched;
    if (vs.emitted && count && !as_records) {
        out->visible_idx = vs.h_visible_idx.ptr;
        out->baked_model = vs.h_baked_model.ptr;
        out->distance_sq = vs.h_distance_sq.ptr;
    }
    if (pool.ready.ptr && pool.result_flags)
        return ctx->fail(GV_E_STATE, "gv_results_fetch: pool %u has a ready column AND a result mapping (records / isVisible in world slots): not available "
                         "together", pool_id);
    if (pool.ready.ptr && pool.bound && pool.occupancy == vs.occupancy && count) {
        // instanceCount += readyCount (mesh.cpp:174): the drawn meshes' own counts, summed over the fetched list (or, for
        // a count-only main pass, over the isVisible bytes); a count-only shadow view keeps draw_count
        std::atomic<uint64_t> total{0};
        if (as_records) {  // the slot is componentOffset / component size
            const RecordLayout L = pool.record_layout;
            const uint8_t* field = vs.records_at + L.component_offset;
            parallel_ranges(0, count, [&](uint32_t a, uint32_t b) {
                uint64_t sum = 0;
                for (uint32_t k = a; k < b; k++) {
                    uint64_t offset;
                    memcpy(&offset, field + (size_t)k * L.stride, 8);
                    sum += pool.ready_count((uint32_t)(offset / L.component_stride));
                }
                total += sum;
            });
            out->instance_count = (uint32_t)total.load();
        } else if (vs.emitted) {
            const uint32_t* idx = vs.h_visible_idx.ptr;
            parallel_ranges(0, count, [&](uint32_t a, uint32_t b) {
                uint64_t sum = 0;
                for (uint32_t k = a; k < b; k++)
                    sum += pool.ready_count(idx[k]);
                total += sum;
            });
            out->instance_count = (uint32_t)total.load();
        } else if (want_vis) {
            const uint8_t* vis = vs.h_is_visible.ptr;
            parallel_ranges(0, vs.occupancy, [&](uint32_t a, uint32_t b) {
                uint64_t sum = 0;
                for (uint32_t i = a; i < b; i++)
                    if (vis[i])
                        sum += pool.ready_count(i);
                total += sum;
            });
            out->instance_count = (uint32_t)total.load();
        }
    }
    if (vs.main_pass && vs.occupancy) {
        // the bytes arrive in pool-slot order; write_back streams them into the components themselves:
        // meshRenderView->isVisible = ...  mesh.cpp:144,152,161,166
        uint8_t* component_vis = nullptr;
        size_t component_stride = 0;
        if (write_back) {
            if (!pool.bound || pool.occupancy != vs.occupancy)
                return ctx->fail(GV_E_STATE, "gv_results_fetch: pool %u rebound since gv_cull", vs.pool_id);
            component_vis = pool.is_visible;
            component_stride = pool.is_visible_stride;
        }
        uint8_t* out_vis = vs.h_is_visible.ptr;
        if (write_back && (pool.result_flags & GV_RESULTS_MAP_VISIBLE)) {
            // straight into the caller's WORLD pool, through the share's slot -> world slot table (holes and slots past the world's
            // occupancy are skipped); ranks own disjoint world slots
            if (pool.h_index_map.size() < vs.occupancy || !pool.visible_base)
                return ctx->fail(GV_E_STATE, "gv_results_fetch: pool %u writes isVisible through its index map (gv_pool_set_result_mapping), which covers %zu "
                                 "of %u slots", pool_id, pool.h_index_map.size(), vs.occupancy);
            const uint32_t* map = pool.h_index_map.data();
            uint8_t* base = pool.visible_base;
            const size_t stride = pool.visible_stride;
            const uint32_t limit = pool.visible_count;
            parallel_ranges(0, vs.occupancy, [&](uint32_t a, uint32_t b) {
                for (uint32_t i = a; i < b; i++)
                    if (map[i] < limit)  // (GV_NONE: a hole)
                        base[(size_t)map[i] * stride] = out_vis[i];
            }, kWriteBackFloor);
        } else if (component_vis)
            parallel_ranges(0, vs.occupancy, [&](uint32_t a, uint32_t b) {
                for (uint32_t i = a; i < b; i++)
                    component_vis[(size_t)i * component_stride] = out_vis[i];
            }, kWriteBackFloor);
        out->is_visible = out_vis;
    }
    return GV_OK;
}

int gv_pool_set_record_layout(GvCtx* ctx, uint32_t pool_id, const GvRecordLayout* layout)
{
    if (!ctx)
        return GV_E_ARG;
    if (pool_id >= GV_MAX_POOLS)
        return ctx->fail(GV_E_ARG, "gv_pool_set_record_layout: pool %u out of range", pool_id);
    RecordLayout L{};  // read when results are delivered, not when they are computed: queued culls are left alone
    if (layout) {
        const uint32_t stride = layout->stride;
        auto inside = [&](uint32_t offset, uint32_t bytes) { return offset % 4 == 0 && offset <= stride && bytes <= stride - offset; };
        struct Span { uint32_t at, bytes; } spans[4] = {{layout->component_offset, 8}, {layout->baked_model, 48}, {layout->distance_sq, 4},
                                                       {layout->buffer_index, 4}};
        const uint32_t fields = layout->buffer_index == GV_NONE ? 3 : 4;
        bool ok = stride != 0 && stride % 16 == 0 && stride <= kMaxRecordStride && layout->component_stride != 0;
        for (uint32_t i = 0; ok && i < fields; i++) {
            ok = inside(spans[i].at, spans[i].bytes);
            for (uint32_t j = 0; ok && j < i; j++)
                ok = spans[i].at + spans[i].bytes <= spans[j].at || spans[j].at + spans[j].bytes <= spans[i].at;
        }
        if (!ok)
            return ctx->fail(GV_E_ARG, "gv_pool_set_record_layout: stride %u (a multiple of 16, at most %u) with fields at %u/%u/%u/%u: "
                             "fields must be 4-byte aligned, inside the record and disjoint", stride, kMaxRecordStride,
                             layout->component_offset, layout->baked_model, layout->distance_sq, layout->buffer_index);
        L = RecordLayout{stride, layout->component_offset, layout->baked_model, layout->distance_sq, layout->buffer_index,
                         layout->component_stride, layout->buffer_index_value, 0u, nullptr};
    }
    if (memcmp(&ctx->pools[pool_id].record_layout, &L, sizeof(L)) == 0)
        return GV_OK;  // set every frame by callers that re-bind every frame
    ctx->pools[pool_id].record_layout = L;
    for (uint32_t v = 0; v < GV_MAX_VIEWS; v++)  // the next fetch delivers this pool's results again, in the new form
        ctx->views[pool_id][v].published = false, ctx->views[pool_id][v].records_fetched = false;
    return GV_OK;
}

int gv_pool_set_record_target(GvCtx* ctx, uint32_t pool_id, uint32_t view_index, void* records, size_t bytes)
{
    if (!ctx)
        return GV_E_ARG;
    if (pool_id >= GV_MAX_POOLS || view_index >= GV_MAX_VIEWS)
        return ctx->fail(GV_E_ARG, "gv_pool_set_record_target: pool %u view %u out of range", pool_id, view_index);
    if (records && ((uintptr_t)records % 16 != 0 || bytes == 0))
        return ctx->fail(GV_E_ARG, "gv_pool_set_record_target: records must be 16-byte aligned and bytes non-zero");
    PoolState::RecordTarget& target = ctx->pools[pool_id].record_target[view_index];
    if (target.host == records && (target.bytes == bytes || !records))
        return GV_OK;  // set every frame by callers that re-bind every frame
    const bool intact = release_record_target(target);
    ViewState& vs = ctx->views[pool_id][view_index];
    vs.published = false, vs.records_fetched = false;  // the next fetch delivers this view's records again, to the new place
    // An error return means NO state change, and here the state does change — the new target goes in — so a previous
    // range that was found unmapped when it was let go is a diagnostic, not a failure: GvStats::record_targets_lost counts it and
    // gv_last_error holds the text (callers such as the shim's check() abort the tick on a non-zero return).
    if (!intact) {
        ctx->stats.record_targets_lost++;
        (void)ctx->fail(GV_OK, "gv_pool_set_record_target: the previous record target of pool %u view %u was no longer mapped when it was let "
                               "go: the range must stay allocated until it is replaced or removed (the new target is in place)", pool_id, view_index);
    }
    if (!records)
        return GV_OK;
    target.host = static_cast<uint8_t*>(records);
    target.bytes = bytes;
    // The caller's array is NOT page-locked: the records arrive in the library's own pinned buffer and the fetch copies them
    // into the array (one memcpy of draw_count records: 3.4 us at 10 k entities, ~50 us for the 1.4 MB of a 100 k-entity pool).
    // Round 2 let the device write the array in place (hipHostRegister once per address): 4 us less per tick at 10 k entities —
    // and, measured in round 3, the GPU test tier then ABORTED inside the runtime in 4 of 16 runs (profiles/r03_record_target_soak.txt),
    // in an unrelated later copy into pageable memory that reused the addresses of an array that had been registered and
    // un-registered. Application memory is therefore never registered (profiles/withdrawn.md).
    return GV_OK;
}

int gv_pool_results_records(GvCtx* ctx, uint32_t pool_id, uint32_t view_index, const void** records, uint32_t* count)
{
    if (!ctx)
        return GV_E_ARG;
    if (!records || !count || !view_of(ctx, pool_id, view_index))
        return ctx->fail(GV_E_ARG, "gv_pool_results_records: pool %u view %u has no results", pool_id, view_index);
    ViewState& vs = *view_of(ctx, pool_id, view_index);
    if (!ctx->pools[pool_id].record_layout.stride)
        return ctx->fail(GV_E_STATE, "gv_pool_results_records: pool %u has no record layout", pool_id);
    if (!vs.emitted)
        return ctx->fail(GV_E_STATE, "gv_pool_results_records: pool %u view %u was culled count-only (GV_CULL_NO_RECORDS)", pool_id, view_index);
    if (!vs.records_fetched || ctx->publish_sync_pending) {  // not fetched yet (or published by a sort that nobody has waited for)
        GvResult unused;
        if (int rc = gv_pool_results_fetch(ctx, pool_id, view_index, 0, &unused))
            return rc;
    }
    *count = vs.h_draw_count.ptr[0];
    *records = *count && vs.records_fetched ? vs.records_at : nullptr;
    return GV_OK;
}

int gv_pool_results_instance_bases(GvCtx* ctx, uint32_t pool_id, uint32_t view_index, const uint32_t** bases, uint32_t* count)
{
    if (!ctx)
        return GV_E_ARG;
    if (!bases || !count || !view_of(ctx, pool_id, view_index))
        return ctx->fail(GV_E_ARG, "gv_pool_results_instance_bases: pool %u view %u has no results", pool_id, view_index);
    if (!view_of(ctx, pool_id, view_index)->emitted)
        return ctx->fail(GV_E_STATE, "gv_pool_results_instance_bases: pool %u view %u was culled count-only (GV_CULL_NO_RECORDS)", pool_id,
                         view_index);
    GvResult r;
    if (int rc = gv_pool_results_fetch(ctx, pool_id, view_index, 0, &r))  // (published results are only looked up)
        return rc;
    ViewState& vs = *view_of(ctx, pool_id, view_index);
    PoolState& pool = ctx->pools[pool_id];
    const uint32_t n = r.draw_count;
    vs.instance_bases.resize((size_t)n + 1);
    uint32_t* out = vs.instance_bases.data();
    const bool counted = pool.ready.ptr && pool.bound && pool.occupancy == vs.occupancy;
    const bool as_records = vs.records_fetched;
    const RecordLayout L = pool.record_layout;
    auto slot_of = [&](uint32_t k) -> uint32_t {
        if (!as_records)
            return vs.h_visible_idx.ptr[k];
        uint64_t offset;
        memcpy(&offset, vs.records_at + (size_t)k * L.stride + L.component_offset, 8);
        return (uint32_t)(offset / L.component_stride);
    };
    // exclusive prefix in two passes over fixed chunks: chunk sums in parallel, their prefix serially, the fill in parallel
    constexpr uint32_t kChunk = 1u << 16;
    const uint32_t chunks = (n + kChunk - 1) / kChunk;
    std::vector<uint64_t> chunk_base((size_t)chunks + 1, 0);
    if (counted)
        parallel_ranges(0, chunks, [&](uint32_t ca, uint32_t cb) {
            for (uint32_t c = ca; c < cb; c++) {
                uint64_t sum = 0;
                for (uint32_t k = c * kChunk, e = std::min(n, (c + 1) * kChunk); k < e; k++)
                    sum += pool.ready_count(slot_of(k));
                chunk_base[c + 1] = sum;
            }
        });
    else
        for (uint32_t c = 0; c < chunks; c++)
            chunk_base[c + 1] = std::min(n, (c + 1) * kChunk) - c * kChunk;
    for (uint32_t c = 0; c < chunks; c++)
        chunk_base[c + 1] += chunk_base[c];
    parallel_ranges(0, chunks, [&](uint32_t ca, uint32_t cb) {
        for (uint32_t c = ca; c < cb; c++) {
            uint32_t at = (uint32_t)chunk_base[c];
            for (uint32_t k = c * kChunk, e = std::min(n, (c + 1) * kChunk); k < e; k++) {
                out[k] = at;
                at += counted ? pool.ready_count(slot_of(k)) : 1u;
            }
        }
    });
    out[n] = (uint32_t)chunk_base[chunks];
    *bases = out;
    *count = n;
    return GV_OK;
}

int gv_results_device(GvCtx* ctx, uint32_t view_index, GvDeviceResult* out)
{
    return ctx ? gv_pool_results_device(ctx, ctx->last_pool, view_index, out) : GV_E_ARG;
}

int gv_pool_results_device(GvCtx* ctx, uint32_t pool_id, uint32_t view_index, GvDeviceResult* out)
{
    if (!ctx)
        return GV_E_ARG;
    if (!out || !view_of(ctx, pool_id, view_index))
        return ctx->fail(GV_E_ARG, "gv_results_device: pool %u view %u has no results", pool_id, view_index);
    if (int rc = flush_sorts(ctx))
        return rc;
    ViewState& vs = *view_of(ctx, pool_id, view_index);
    out->visible_idx = vs.emitted ? vs.visible_idx.ptr : nullptr;
    out->baked_model = vs.emitted ? vs.baked_model.ptr : nullptr;
    out->distance_sq = vs.emitted ? vs.distance_sq.ptr : nullptr;
    out->is_visible = vs.main_pass ? vs.is_visible.ptr : nullptr;
    out->draw_count = vs.draw_count.ptr;
    return GV_OK;
}

int gv_results_copy_idx_device(GvCtx* ctx, uint32_t view_index, void* dst_device, uint32_t capacity,
                               uint32_t index_base)
{
    if (!ctx)
        return GV_E_ARG;
    if (!dst_device || !view_of(ctx, ctx->last_pool, view_index) || !view_of(ctx, ctx->last_pool, view_index)->emitted)
        return ctx->fail(GV_E_ARG, "gv_results_copy_idx_device: view %u has no emitted records", view_index);
    if (int rc = flush_sorts(ctx))
        return rc;
    ViewState& vs = *view_of(ctx, ctx->last_pool, view_index);
    GV_HIP(ctx, hipSetDevice(ctx->device));
    const PoolState& pool = ctx->pools[vs.pool_id];
    GV_HIP(ctx, launch_copy_idx(vs.visible_idx.ptr, vs.draw_count.ptr, static_cast<uint32_t*>(dst_device), capacity,
                                index_base, pool.index_map_count >= vs.occupancy ? pool.d_index_map.ptr : nullptr, ctx->stream));
    return GV_OK;
}

int gv_results_copy_shard_device(GvCtx* ctx, uint32_t view_index, void* dst_device, uint32_t capacity,
                                 uint32_t index_base)
{
    if (!ctx)
        return GV_E_ARG;
    return gv::copy_shard_of_pool(ctx, ctx->last_pool, view_index, dst_device, capacity, index_base);
}

int gv_results_copy_mask_device(GvCtx* ctx, uint32_t view_index, void* dst_device, uint32_t word_count)
{
    if (!ctx)
        return GV_E_ARG;
    if (!dst_device || !view_of(ctx, ctx->last_pool, view_index))
        return ctx->fail(GV_E_ARG, "gv_results_copy_mask_device: view %u has no results", view_index);
    if (int rc = flush_culls(ctx))
        return rc;
    ViewState& vs = *view_of(ctx, ctx->last_pool, view_index);
    if (word_count < (vs.occupancy + 31u) / 32u)
        return ctx->fail(GV_E_ARG, "gv_results_copy_mask_device: %u words for a pool of %u slots", word_count, vs.occupancy);
    if (!vs.ballots_current && !vs.main_pass)
        return ctx->fail(GV_E_STATE, "gv_results_copy_mask_device: view %u has neither ballot words nor isVisible bytes", view_index);
    GV_HIP(ctx, hipSetDevice(ctx->device));
    GV_HIP(ctx, launch_mask_shard(vs.ballots_current ? vs.mask.ptr : nullptr, vs.is_visible.ptr, vs.draw_count.ptr, vs.occupancy,
                                  static_cast<uint32_t*>(dst_device), word_count, ctx->stream));
    return GV_OK;
}

int gv_pool_mirror_slots(GvCtx* ctx, uint32_t pool_id, uint32_t* entry_to_slot, uint32_t capacity)
{
    if (!ctx)
        return GV_E_ARG;
    if (pool_id >= GV_MAX_POOLS || !entry_to_slot)
        return ctx->fail(GV_E_ARG, "gv_pool_mirror_slots: bad argument (pool %u)", pool_id);
    PoolState& p = ctx->pools[pool_id];
    if (!p.bound)
        return ctx->fail(GV_E_STATE, "gv_pool_mirror_slots: pool %u is not bound", pool_id);
    if (int rc = sync_mirror(ctx))
        return rc;
    const uint32_t n = std::min(p.occupancy, capacity);
    const bool permuted = !p.perm.empty() && p.perm.size() == p.occupancy;
    for (uint32_t e = 0; e < n; e++)
        entry_to_slot[e] = permuted ? p.perm[e] : e;
    return GV_OK;
}

int gv_pool_mirror_epoch(GvCtx* ctx, uint32_t pool_id, uint64_t* epoch)
{
    if (!ctx)
        return GV_E_ARG;
    if (pool_id >= GV_MAX_POOLS || !epoch)
        return ctx->fail(GV_E_ARG, "gv_pool_mirror_epoch: bad argument (pool %u)", pool_id);
    if (!ctx->pools[pool_id].bound)
        return ctx->fail(GV_E_STATE, "gv_pool_mirror_epoch: pool %u is not bound", pool_id);
    if (int rc = sync_mirror(ctx))
        return rc;
    *epoch = ctx->pools[pool_id].order_epoch;
    return GV_OK;
}

int gv_pool_set_index_map(GvCtx* ctx, uint32_t pool_id, const uint32_t* global_ids, uint32_t count)
{
    if (!ctx)
        return GV_E_ARG;
    if (pool_id >= GV_MAX_POOLS || (count && !global_ids))
        return ctx->fail(GV_E_ARG, "gv_pool_set_index_map: bad argument (pool %u)", pool_id);
    PoolState& p = ctx->pools[pool_id];
    GV_HIP(ctx, hipSetDevice(ctx->device));
    if (count) {
        GV_HIP(ctx, p.d_index_map.reserve(count));
        GV_HIP(ctx, hipMemcpyAsync(p.d_index_map.ptr, global_ids, (size_t)count * 4, hipMemcpyHostToDevice, ctx->stream));
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller's (pageable) table may go away
    }
    p.index_map_count = count;
    p.h_index_map.assign(global_ids, global_ids + count);
    for (uint32_t v = 0; v < GV_MAX_VIEWS; v++)  // records that carry mapped slots are delivered again
        if (p.result_flags & GV_RESULTS_MAP_RECORDS)
            ctx->views[pool_id][v].published = false, ctx->views[pool_id][v].records_fetched = false;
    return GV_OK;
}

int gv_pool_update_index_map(GvCtx* ctx, uint32_t pool_id, uint32_t first, const uint32_t* global_ids, uint32_t count)
{
    if (!ctx)
        return GV_E_ARG;
    if (pool_id >= GV_MAX_POOLS || (count && !global_ids) || (uint64_t)first + count > 0xFFFFFFFFull)
        return ctx->fail(GV_E_ARG, "gv_pool_update_index_map: bad argument (pool %u, first %u, count %u)", pool_id, first, count);
    if (count == 0)
        return GV_OK;
    PoolState& p = ctx->pools[pool_id];
    if (first > p.index_map_count)
        return ctx->fail(GV_E_ARG, "gv_pool_update_index_map: first %u leaves a gap behind the table's %u entries", first, p.index_map_count);
    GV_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t end = first + count;
    if (end > p.index_map_count) {
        GV_HIP(ctx, p.d_index_map.grow(end, p.index_map_count, ctx->stream));
        p.index_map_count = end;
        p.h_index_map.resize(end, GV_NONE);
    }
    std::copy(global_ids, global_ids + count, p.h_index_map.begin() + first);
    // (out of the library's own host copy, which stays put; stream order keeps queued readers of the old entries in front)
    GV_HIP(ctx, hipMemcpyAsync(p.d_index_map.ptr + first, p.h_index_map.data() + first, (size_t)count * 4, hipMemcpyHostToDevice, ctx->stream));
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (pageable source: the runtime may still be reading it)
    return GV_OK;
}

int gv_pool_set_result_mapping(GvCtx* ctx, uint32_t pool_id, uint32_t flags, void* visible_base, size_t visible_stride, uint32_t visible_count)
{
    if (!ctx)
        return GV_E_ARG;
    if (pool_id >= GV_MAX_POOLS || (flags & ~(GV_RESULTS_MAP_RECORDS | GV_RESULTS_MAP_VISIBLE)) ||
        ((flags & GV_RESULTS_MAP_VISIBLE) && (!visible_base || !visible_stride)))
        return ctx->fail(GV_E_ARG, "gv_pool_set_result_mapping: pool %u, flags 0x%x, visible base %p stride %zu", pool_id, flags, visible_base, visible_stride);
    PoolState& p = ctx->pools[pool_id];
    if ((p.result_flags ^ flags) & GV_RESULTS_MAP_RECORDS)
        for (uint32_t v = 0; v < GV_MAX_VIEWS; v++)  // the next fetch delivers this pool's records again, in the other numbering
            ctx->views[pool_id][v].published = false, ctx->views[pool_id][v].records_fetched = false;
    p.result_flags = flags;
    p.visible_base = (flags & GV_RESULTS_MAP_VISIBLE) ? static_cast<uint8_t*>(visible_base) : nullptr;
    p.visible_stride = (flags & GV_RESULTS_MAP_VISIBLE) ? visible_stride : 0;
    p.visible_count = (flags & GV_RESULTS_MAP_VISIBLE) ? visible_count : 0;
    return GV_OK;
}

int gv_sort(GvCtx* ctx, uint32_t view_index, int descending)
{
    return ctx ? gv_pool_sort(ctx, ctx->last_pool, view_index, descending) : GV_E_ARG;
}

int gv_pool_sort(GvCtx* ctx, uint32_t pool_id, uint32_t view_index, int descending)
{
    if (!ctx)
        return GV_E_ARG;
    if (!view_of(ctx, pool_id, view_index) || !view_of(ctx, pool_id, view_index)->emitted)
        return ctx->fail(GV_E_ARG, "gv_sort: pool %u view %u has no emitted records", pool_id, view_index);
    ZoneScope zone("Meshes Sort");
    ViewState& vs = *view_of(ctx, pool_id, view_index);
    if (vs.occupancy == 0)
        return GV_OK;
    // launched with the other views' sorts when the records are asked for: small pools share one launch, mid-sized ones one set of
    // launches (flush_sorts)
    if (vs.occupancy <= kMidSortMaxSlots) {
        vs.sort_pending = descending ? 2 : 1;
        vs.published = false, vs.records_fetched = false;
        return GV_OK;
    }
    return sort_large(ctx, vs, descending != 0);
}

}  // extern "C"

namespace gv {

// (pool, view)'s shard [draw_count, visible_idx + index_base ...] into dst_device: gv_results_copy_shard_device for the pool of the most
// recent cull, the exchange for the pool it is asked for (several mesh systems culled in one batch, each exchanged afterwards)
int copy_shard_of_pool(GvCtx* ctx, uint32_t pool_id, uint32_t view_index, void* dst_device, uint32_t capacity, uint32_t index_base)
{
    if (!dst_device || !view_of(ctx, pool_id, view_index) || !view_of(ctx, pool_id, view_index)->emitted)
        return ctx->fail(GV_E_ARG, "shard copy: pool %u view %u has no emitted records", pool_id, view_index);
    if (int rc = flush_sorts(ctx))
        return rc;
    ViewState& vs = *view_of(ctx, pool_id, view_index);
    GV_HIP(ctx, hipSetDevice(ctx->device));
    const PoolState& pool = ctx->pools[vs.pool_id];
    GV_HIP(ctx, launch_copy_shard(vs.visible_idx.ptr, vs.draw_count.ptr, static_cast<uint32_t*>(dst_device), capacity,
                                  index_base, pool.index_map_count >= vs.occupancy ? pool.d_index_map.ptr : nullptr, ctx->stream));
    return GV_OK;
}

}  // namespace gv
