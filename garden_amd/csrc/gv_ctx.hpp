// gv_ctx.hpp — the context behind the opaque GvCtx of include/garden_vis.h and the small host-side utilities shared
// by gv_mirror.cpp (AoS/columns -> device mirror) and gv_context.cpp (the C-ABI, per-frame dispatch, results).
#pragma once
#include <hip/hip_runtime.h>
#include <rocprofiler-sdk-roctx/roctx.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/garden_vis.h"
#include "gv_kernels.hpp"
#include "gv_workers.hpp"
#include "gv_dirty_ranges.hpp"


namespace gv {

template <typename T>
struct DeviceBuf {  // grow-only device allocation (scratch vectors grow, never shrink: mesh.cpp:377-395)
    T* ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= cap)
            return hipSuccess;
        if (ptr)
            (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T));
        if (e == hipSuccess) {
            cap = n;
            // GV_DEBUG_POISON: fresh device memory is filled with a pattern instead of whatever (often zeros) the allocator hands
            // out, so that a kernel that reads what nobody wrote fails every time, not once in a long session
            static const bool poison = getenv("GV_DEBUG_POISON") != nullptr;
            if (poison) {  // (the fill must have landed before any stream touches the buffer: hipMemset alone may still be in flight)
                e = hipMemset(ptr, 0xCD, n * sizeof(T));
                if (e == hipSuccess)
                    e = hipDeviceSynchronize();
            }
        }
        return e;
    }
    // like reserve, but the first `keep` elements survive (pool growth: the mirror is appended to, not rebuilt);
    // capacity grows by half so that steady appends do not reallocate every frame
    hipError_t grow(size_t n, size_t keep, hipStream_t stream)
    {
        if (n <= cap)
            return hipSuccess;
        const size_t want = std::max(n, cap + cap / 2);
        T* fresh = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&fresh), want * sizeof(T));
        if (e != hipSuccess)
            return e;
        if (ptr && keep) {
            e = hipMemcpyAsync(fresh, ptr, std::min(keep, cap) * sizeof(T), hipMemcpyDeviceToDevice, stream);
            if (e == hipSuccess)
                e = hipStreamSynchronize(stream);
            if (e != hipSuccess) {
                (void)hipFree(fresh);
                return e;
            }
        }
        if (ptr)
            (void)hipFree(ptr);
        ptr = fresh;
        cap = want;
        return hipSuccess;
    }
    void release()
    {
        if (ptr)
            (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

template <typename T>
struct PinnedBuf {
    T* ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= cap)
            return hipSuccess;
        if (ptr)
            (void)hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess)
            cap = n;
        return e;
    }
    hipError_t grow(size_t n, size_t keep)  // the caller has drained every async copy that reads this buffer
    {
        if (n <= cap)
            return hipSuccess;
        const size_t want = std::max(n, cap + cap / 2);
        T* fresh = nullptr;
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&fresh), want * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess)
            return e;
        if (ptr && keep)
            memcpy(fresh, ptr, std::min(keep, cap) * sizeof(T));
        if (ptr)
            (void)hipHostFree(ptr);
        ptr = fresh;
        cap = want;
        return hipSuccess;
    }
    void release()
    {
        if (ptr)
            (void)hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

struct DirtyRange {
    uint32_t lo = UINT32_MAX, hi = 0;
    bool any() const { return lo < hi; }
    void add(uint32_t first, uint32_t count)
    {
        if (count == 0)
            return;
        lo = std::min(lo, first);
        // saturating: first + count must not wrap to a tiny `hi` (the mark would be dropped and the mirror go stale);
        // users clamp `hi` to the pool's occupancy
        hi = std::max(hi, (uint32_t)std::min<uint64_t>((uint64_t)first + count, UINT32_MAX));
    }
    void clear()
    {
        lo = UINT32_MAX;
        hi = 0;
    }
};

// One field of a bound pool: element i lives at ptr + i * stride. An AoS pool binds every field with the component
// stride and its offset folded into ptr; column (SoA) storage binds each field with its own array and element size.
struct Column {
    const uint8_t* ptr = nullptr;
    size_t stride = 0;
    const uint8_t* at(size_t i) const { return ptr + i * stride; }
    uint32_t u32(size_t i) const
    {
        uint32_t v;
        memcpy(&v, ptr + i * stride, 4);
        return v;
    }
    const float* f32(size_t i) const { return reinterpret_cast<const float*>(ptr + i * stride); }
    uint8_t u8(size_t i) const { return ptr[i * stride]; }
};

struct TransformBinding {
    Column entity, parent, position, scale, rotation, self_active, ancestors_active, model_with_ancestors;
    uint32_t occupancy = 0;
    const uint32_t* entity_to_transform = nullptr;
    uint32_t entity_capacity = 0;
    bool bound = false;
};

struct PoolState {
    Column entity, is_enabled, aabb_min, aabb_max;
    RecordLayout record_layout{};  // gv_pool_set_record_layout (stride 0: none)
    struct RecordTarget {        // gv_pool_set_record_target: the caller's own array for a view's records
        uint8_t* host = nullptr;
        size_t bytes = 0;
    };
    RecordTarget record_target[GV_MAX_VIEWS];
    Column ready;              // gv_pool_bind_ready: per-slot ready count (ptr NULL: none, every slot counts 1)
    uint32_t ready_width = 0;  // 1 or 4 bytes
    uint32_t ready_count(size_t i) const { return !ready.ptr ? 1u : (ready_width == 4 ? ready.u32(i) : (uint32_t)ready.u8(i)); }
    uint8_t* is_visible = nullptr;  // write-back target (NULL: none), element i at is_visible + i * is_visible_stride
    size_t is_visible_stride = 0;
    uint32_t occupancy = 0;
    bool bound = false;
    bool need_full = false;
    uint32_t mapping = kMapGeneral;  // MeshMapping, chosen at full gather (kMapExact may only be demoted afterwards)
    DirtyRanges dirty;               // itemised (scattered enable / ready / AABB edits re-mirror what they touched)
    // spatial mirror order (empty = slot order): perm[j] = pool slot held by mirror entry j, inv = its inverse
    std::vector<uint32_t> perm, inv;
    uint64_t order_epoch = 0;    // bumped whenever the entry -> slot table changes (full build, appended slots, re-order): gv_pool_mirror_epoch
    DeviceBuf<uint32_t> d_orig;  // perm on the device: emit reports original pool slots
    DeviceBuf<uint32_t> d_inv;   // inv on the device: the device-side gather of dirty mesh ranges (upload_meshes_device)
    DirtyRange staging_stale;    // slots whose host staging entries lag behind the device (written by that path)
    DeviceBuf<uint32_t> d_index_map;  // gv_pool_set_index_map: pool slot -> the caller's global id (exchange shards)
    uint32_t index_map_count = 0;     // 0: none
    std::vector<uint32_t> h_index_map;  // the same table on the host (the mapped write-back of isVisible walks it)
    // gv_pool_set_result_mapping: results in the caller's WORLD numbering
    uint32_t result_flags = 0;          // GV_RESULTS_MAP_*
    uint8_t* visible_base = nullptr;    // GV_RESULTS_MAP_VISIBLE: byte of slot i -> visible_base + h_index_map[i] * visible_stride
    size_t visible_stride = 0;
    uint32_t visible_count = 0;
    // GV_CONFIG_BLOCK_BOUNDS: per-workgroup world boxes, valid for (bounds_xf_epoch, bounds_epoch)
    DeviceBuf<float4> d_blk_lo, d_blk_hi;
    DeviceBuf<uint8_t> d_blk_dirty;  // one byte per block: holds an entry re-mirrored since the boxes (and seeds) were last current
    uint32_t small_streak = 0;       // syncs in a row that re-mirrored only a few entries of this pool (a pool that keeps changing a
                                     // little gets its boxes rebuilt once, then patched; one that keeps changing a lot goes without)
    bool patch_valid = false;        // every change of the mirror since then is recorded in d_blk_dirty (flat, exactly paired pools):
                                     // the next cull re-derives the flagged blocks instead of going without boxes
    DeviceBuf<EmitSeed> d_seed;    // emit seeds (gv_kernels.hpp), valid for (seed_xf_epoch, seed_epoch)
    uint64_t seed_epoch = 0, seed_xf_epoch = 0;
    DeviceBuf<uint32_t> d_kept;    // [2 alternating counters, 2 words of padding | list entries] of launch_cull_listed
    DeviceBuf<uint8_t> d_kept_flag;  // per list entry (Hi-Z views)
    uint32_t kept_parity = 0;      // which counter the next classify launch adds into
    uint32_t mirrored = 0, appended = 0;  // entries the mirror holds / of those, appended (unsorted) since the last full build
    uint64_t epoch = 1, bounds_epoch = 0, bounds_xf_epoch = 0;  // epoch: bumped whenever this pool's mirror changes
    uint64_t seen_epoch = 0, seen_xf_epoch = 0;                 // state at this pool's previous gv_cull
    bool changed_prev = false;                                  // ... and whether it had changed then too (dynamic pool)
    // device mirror + pinned staging
    DeviceBuf<float4> d_a;
    DeviceBuf<float2> d_b;
    DeviceBuf<uint32_t> d_link;
    PinnedBuf<float4> h_a;
    PinnedBuf<float2> h_b;
    PinnedBuf<uint32_t> h_link;
};

struct ViewState {
    DeviceBuf<unsigned long long> mask;
    DeviceBuf<uint32_t> chunk_count, chunk_count2, chunk_offset, draw_count;
    uint32_t count_parity = 0;  // which totals buffer the next cull adds into (see launch_emit self_prefix)
    uint32_t stale_chunks[2] = {0, 0};  // entries of each totals buffer that may be non-zero right now

    DeviceBuf<uint8_t> is_visible;        // mirror order, written by the cull
    DeviceBuf<uint8_t> vis_flags;         // ViewBuffers::vis_flags: which quarter-chunks of is_visible may hold a non-zero
    bool vis_flags_current = false;       // false: something other than the self-prefixing emit wrote is_visible since (or the
                                          // buffers are new): the flags are set to "may be non-zero" before the next emit
    DeviceBuf<uint8_t> is_visible_slots;  // pool-slot order, filled by gv_results_fetch of a large, spatially ordered pool
    DeviceBuf<uint32_t> visible_idx;
    DeviceBuf<float> baked_model, distance_sq;
    // gv_sort: alternate record set + radix-sort scratch (allocated on first use)
    DeviceBuf<uint32_t> alt_idx, sort_keys[2], sort_vals[2], sort_slots[2], sort_hist;
    DeviceBuf<float> alt_model, alt_dist;
    PinnedBuf<uint32_t> h_visible_idx, h_draw_count;
    PinnedBuf<float> h_baked_model, h_distance_sq;
    PinnedBuf<uint8_t> h_is_visible;
    PinnedBuf<uint8_t> h_records;    // results in the pool's record layout (gv_pool_results_records)
    DeviceBuf<uint8_t> d_records;    // ... packed on the device first for pools too large to publish directly
    bool records_fetched = false;    // records_at holds this cull's records
    uint8_t* records_at = nullptr;   // h_records, or the caller's array (gv_pool_set_record_target)
    uint32_t count_hint = 0xFFFFFFFFu;  // draw count of this view's previous fetch (unknown: none)
    bool records_staged = false;     // the caller's array could not be page-locked: h_records is copied into it after the synchronisation
    bool ballots_current = false;    // `mask` holds this cull's ballot words (not after the one-launch cull + emit of a small pool)
    std::vector<uint32_t> instance_bases;  // gv_pool_results_instance_bases (built on request)
    uint32_t pool_id = 0, occupancy = 0;
    bool main_pass = false, emitted = false, valid = false;
    // fused cull + emit (launch_cull_emit): look-back words, ticket counter and their running epoch / base
    DeviceBuf<unsigned long long> tile_status;
    DeviceBuf<uint32_t> tile_ticket;
    uint32_t tile_ticket_base = 0, tile_epoch = 0;
    uint32_t sort_parity = 0;  // which of the two counter sets in sort_hist the next large sort uses (gv_sort.hip)
    size_t sort_set_words = 0; // size of one set as laid out in sort_hist (0: not initialised)
    DeviceBuf<uint16_t> sort_ranks;
    uint8_t sort_pending = 0;  // small pool: gv_sort asked for (1 ascending, 2 descending), not launched yet (flush_sorts)
    bool published = false;  // small pool: the host buffers already hold this view's results (gv_results_fetch of a sibling view)
};

struct PendingEvent {
    hipEvent_t start, stop;
    int kernel;
};


// Everything a context owns. The C-ABI's opaque `GvCtx` (a global-scope name) derives from it below.
struct Context {
    GvConfig config{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;

    // ---- transform pool: binding, change tracking, device mirror + pinned staging ----
    TransformBinding xf;
    bool xf_need_full = false;     // (re)build the whole mirror at the next sync
    bool xf_links_dirty = false;   // a ranged GV_DIRTY_HIERARCHY: parent links changed -> re-validate depth / cycles
    DirtyRanges xf_dirty;
    uint64_t xf_epoch = 1;         // bumped whenever the transform mirror changes
    uint32_t xf_mirrored = 0, xf_appended = 0;  // as PoolState::mirrored / appended, for the transform pool
    uint32_t max_depth = 0;        // longest parent chain in the mirror
    DeviceBuf<XfAB> d_xab;         // {pos|scale.x, quat} per entry
    DeviceBuf<float2> d_xc;
    DeviceBuf<uint8_t> d_xflags;
    DeviceBuf<unsigned long long> d_xactive;  // bit-plane of kXfActive, derived on the device
    DeviceBuf<uint32_t> d_xparent;
    PinnedBuf<XfAB> h_xab;
    PinnedBuf<float2> h_xc;
    PinnedBuf<uint8_t> h_xflags;
    PinnedBuf<uint32_t> h_xparent;
    // spatial mirror order of the transform pool (empty = slot order)
    std::vector<uint32_t> xperm, xinv;
    DeviceBuf<uint32_t> d_xinv;    // slot -> mirror entry (gv_get_world, device-side gather)
    // device-side gather of dirty AoS ranges
    DeviceBuf<uint8_t> d_raw;      // raw component bytes of the dirty slot range
    PinnedBuf<uint8_t> h_raw[2];   // the library's own pinned chunks the span travels through (double-buffered)
    hipEvent_t raw_done[2] = {nullptr, nullptr};  // chunk buffer k may be rewritten once its last copy has run
    DirtyRange staging_stale;      // slots whose host staging entries lag behind the device (written by that path)
    hipEvent_t upload_done = nullptr;  // behind the last copies that read the pinned staging / packet buffers: the next sync waits for THIS before
    bool upload_pending = false;       // it rewrites them, not for the whole stream (the previous frame's cull may still be running)
    PinnedBuf<uint32_t> h_ranges;  // a sync's dirty ranges on their way to mark_dirty_blocks_kernel
    DeviceBuf<uint32_t> d_ranges;
    DeviceBuf<uint32_t> d_e2t;     // entity -> transform slot table on the device, refreshed by every device-side mesh gather
    DeviceBuf<uint32_t> d_flag;    // one word: "a candidate no longer pairs with its own index" (aos_meshes_kernel)
    PinnedBuf<uint32_t> h_flag;
    // scratch of the scattered (dirty-range) host upload path
    PinnedBuf<XfPacket> sc_xf;       // one packet of {entry, record} per sync and side (gv_sort_kernels.hpp)
    DeviceBuf<XfPacket> dsc_xf;
    PinnedBuf<MeshPacket> sc_mesh;
    DeviceBuf<MeshPacket> dsc_mesh;
    DeviceBuf<float4> dsc_a;         // device scratch of gv_get_world (the gathered matrices of a permuted mirror) ...
    DeviceBuf<float2> dsc_c;         // ... and of gv_debug_stream_peak (its sink)

    PoolState pools[GV_MAX_POOLS];
    // results are kept per (pool, view): every mesh system's cull can be issued before the first result is read
    // (gv_pool_results_fetch ...); the view-indexed entry points address the pool of the most recent gv_cull
    ViewState views[GV_MAX_POOLS][GV_MAX_VIEWS];
    uint32_t last_pool = 0;

    // ---- a tick of engine-sized pools (gv_cull_batch_begin): culls recorded, launched together at the first read ----
    struct CullJob {
        uint32_t pool_id, view_count;
        ViewParams vps[GV_MAX_VIEWS];
    };
    bool cull_batching = false;
    std::vector<CullJob> cull_jobs;
    PinnedBuf<uint8_t> h_tick[2];  // descriptor tables on their way to d_tick (double-buffered: the host may run ahead)
    hipEvent_t tick_done[2] = {nullptr, nullptr};
    uint32_t tick_turn = 0;
    DeviceBuf<uint8_t> d_tick;

    // ---- world-matrix cache (gv_sweep) ----
    DeviceBuf<float4> d_world;
    bool world_valid = false;          // d_world holds the world matrices of the current mirror, except ...
    bool world_partial = false;        // ... for the chains through entries flagged in d_xdirty (subtree-scoped sweep)
    bool xdirty_set = false;           // some flag in d_xdirty may be 1
    DeviceBuf<uint8_t> d_xdirty;       // 1 byte per transform mirror entry: re-mirrored since the cache was brought up to date
    bool sweep_with_cull = false;      // GV_SWEEP_WITH_CULL[_VALU] requested: the next gv_cull also writes the world matrices
    bool sweep_with_cull_mfma = true;  // ... with the MFMA or the VALU chain

    // ---- block bounds (GV_CONFIG_BLOCK_BOUNDS) statistics ----
    DeviceBuf<uint8_t> d_examined;     // of the LAST bounded cull: 1 byte per workgroup
    uint64_t bounds_blocks_total = 0;

    // ---- Hi-Z ----
    DeviceBuf<float> d_depth;
    const float* depth_ptr = nullptr;  // d_depth.ptr or caller's device memory
    DeviceBuf<float2> d_mips;
    DeviceBuf<uint64_t> d_mip_offset;
    uint32_t hiz_w = 0, hiz_h = 0, hiz_mips = 0;
    uint32_t mip_w[GV_MAX_MIPS]{}, mip_h[GV_MAX_MIPS]{};
    uint64_t mip_off[GV_MAX_MIPS]{};
    bool hiz_valid = false;
    bool hiz_nested = false;           // every level bounds all the texels it covers (see HizDevice::nested)
    PinnedBuf<uint32_t> h_done;  // the word the done-flag kernel writes (wait_for_stream)
    uint32_t done_seq = 0;
    bool publish_sync_pending = false;  // a small-pool sort has published its views; nobody has synchronised the stream since
    bool hiz_level1_virtual = false;   // decided in gv_hiz_build: sizes whose first six levels take the fused kernel
    bool hiz_level1_stored = false;    // ... and whether gv_hiz_read_level has materialised it since the last build

    // ---- multi-GPU exchange (gv_exchange.cpp) ----
    void* exchange_comm = nullptr;     // ncclComm_t
    int exchange_rank = 0, exchange_world = 1;
    uint32_t exchange_mode = GV_EXCHANGE_ALLGATHER;  // GvExchangeMode
    DeviceBuf<uint32_t> d_shard;       // [count, indices...] of this rank
    // gv_exchange_visible: library-owned rows, sized from the previous frame's headers, completed when a list outgrew its room
    struct ExchangeSlot {
        DeviceBuf<uint32_t> rows;      // [world][row_words]
        DeviceBuf<uint32_t> shard;     // this rank's WHOLE [count, indices ...] of the slot's frame: the first exchange reads its leading
                                       // 1 + room[me] words, a completing exchange the tail behind them (both on exchange_stream)
        hipEvent_t produced = nullptr; // on ctx->stream behind the shard copy: exchange_stream waits for it
        hipEvent_t done = nullptr;     // on exchange_stream behind collective + headers (+ tails): GvExchangeFrame::ready_event
        // GV_EXCHANGE_PEER (gv_exchange_init_peers): `sent` on this rank's exchange stream behind its stores into everybody's rows;
        // on rank 0's exchange stream — the hub — `all_produced` behind every rank's `produced` (nobody still reads the rows about to
        // be overwritten) and `all_sent` behind every rank's `sent` (every row has arrived everywhere)
        hipEvent_t sent = nullptr, all_produced = nullptr, all_sent = nullptr;
        PinnedBuf<uint32_t> hdr;       // [1] sequence word + [world][hdr_words] leading words of every row, written by exchange_headers_kernel
        uint32_t hdr_words = 1;        // 1 (the count header) + the lists of a gv_exchange_views frame
        uint32_t items = 0;            // gv_exchange_views: lists per rank in this slot's frame (0: a single-list frame)
        std::vector<uint32_t> item_counts;  // settled: [world][items] per-list counts (GvExchangeFrame::item_counts)
        PinnedBuf<ShardItem> h_items;  // the frame's list descriptors on their way to d_items
        DeviceBuf<ShardItem> d_items;
        std::vector<ShardItem> items_uploaded, items_wanted;  // what d_items holds (empty: unknown) / this frame's descriptors
        uint32_t row_words = 0;
        uint32_t room[GV_EXCHANGE_MAX_RANKS] = {};        // list entries rank r's row was predicted to need in this slot's frame
        uint32_t travelled[GV_EXCHANGE_MAX_RANKS] = {};   // words of row r on the links in the first exchange
        uint32_t counts[GV_EXCHANGE_MAX_RANKS] = {};      // settled: the frame's own headers
        uint32_t tail_words[GV_EXCHANGE_MAX_RANKS] = {};  // settled: what the completing exchange carried
        uint64_t cut = 0;              // settled: ranks whose list outgrew its room
        uint64_t frame = 0;
        uint32_t mode = 0;
        bool in_flight = false;        // sent; its headers have not been read yet
        bool settled = false;          // headers read, rooms updated, tails delivered: the rows are complete behind `done`
    } exchange_slots[2];
    hipStream_t exchange_stream = nullptr;              // EVERY collective of the communicator runs here (one communicator, one stream): the next frame's cull (ctx->stream) does not wait for the links
    hipEvent_t exchange_in = nullptr, exchange_out = nullptr;  // hand-over events of the caller-owned forms (gv_exchange_shards / _masks)
    uint64_t exchange_frame = 0;                        // the next frame's number
    uint32_t exchange_room[GV_EXCHANGE_MAX_RANKS] = {};  // room the next frame gives each rank
    uint32_t exchange_timeout_ms = 30000;               // bound of every host wait of the exchange (gv_exchange_set_timeout)
    bool exchange_broken = false;                       // a wait ran out: the communicator is aborted, not destroyed
    bool exchange_by_group = false;                     // made by gv_exchange_init_all / _init_peers: driven through the *_all forms only
    std::vector<GvCtx*> exchange_peers;                 // gv_exchange_init_peers: the group's contexts by rank (empty: a communicator, or nothing)

    // ---- profiling ----
    std::vector<PendingEvent> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
    GvStats stats{};
    uint32_t profile_every = 1;                      // gv_profile_sampling
    uint32_t profile_mask = 0;                       // bit k: launches of GvKernelId k are bracketed (gv_create from the config flags; gv_profile_kernels)
    uint64_t profile_seen[GV_K_COUNT] = {}, profile_timed[GV_K_COUNT] = {};

    int fail(int code, const char* fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        error = buf;
        return code;
    }
    int hip_fail(hipError_t e, const char* what)
    {
        return fail(e == hipErrorOutOfMemory ? GV_E_OOM : GV_E_HIP, "%s: %s", what, hipGetErrorString(e));
    }
};

}  // namespace gv

struct GvCtx final : gv::Context {};


#define GV_HIP(ctx, call)                                   \
    do {                                                    \
        hipError_t e__ = (call);                            \
        if (e__ != hipSuccess)                              \
            return (ctx)->hip_fail(e__, #call);             \
    } while (0)

namespace gv {

// roctx ranges named after the reference's profiler zones (SET_CPU_ZONE_SCOPED / SET_GPU_DEBUG_LABEL:
// "Meshes Prepare" source/system/render/mesh.cpp:334, "Meshes Sort" :267, "HiZ Downsample" hiz.cpp:146), so a
// rocprofv3 --marker-trace timeline reads like the engine's Tracy capture.
struct ZoneScope {
    explicit ZoneScope(const char* name) { roctxRangePushA(name); }
    ~ZoneScope() { roctxRangePop(); }
};

// ---- profiling events ----
struct KernelTimer {
    GvCtx* ctx;
    int kernel;
    hipEvent_t start = nullptr, stop = nullptr;
    bool on = false;
    KernelTimer(GvCtx* c, int k) : ctx(c), kernel(k)
    {
        ctx->stats.launches[k]++;
        if (!((ctx->profile_mask >> k) & 1u))
            return;
        if (ctx->profile_seen[k]++ % ctx->profile_every != 0)  // gv_profile_sampling
            return;
        if (!ctx->free_events.empty()) {
            start = ctx->free_events.back().first;
            stop = ctx->free_events.back().second;
            ctx->free_events.pop_back();
        } else if (hipEventCreate(&start) != hipSuccess || hipEventCreate(&stop) != hipSuccess) {
            return;
        }
        on = hipEventRecord(start, ctx->stream) == hipSuccess;
        if (on)
            ctx->profile_timed[k]++;
    }
    ~KernelTimer()
    {
        if (!on)
            return;
        (void)hipEventRecord(stop, ctx->stream);
        ctx->pending.push_back({start, stop, kernel});
    }
};

// host worker threads for the gathers (AoS component pools -> SoA staging) and the isVisible write-back: gv_workers.hpp

void drain_events(GvCtx* ctx);

// gv_exchange.cpp
void exchange_release(GvCtx* ctx);            // destroys the communicator, if any
int exchange_drain(GvCtx* ctx);    // bounded wait for what is queued on the exchange stream; a timeout aborts the communicator (GV_E_TIMEOUT)

// gv_context.cpp (the cull side)
int flush_culls(GvCtx* ctx);                             // launches the culls recorded since gv_cull_batch_begin; ends the batch
int flush_recorded_culls(GvCtx* ctx, uint32_t pool);    // ... those that read `pool` (GV_MAX_POOLS: any), recording goes on
ViewBuffers view_buffers(ViewState& vs);
// gv_results.cpp (the reader side)
int flush_sorts(GvCtx* ctx);                             // flush_culls + the sorts of small pools that were asked for and not launched yet
int wait_for_stream(GvCtx* ctx);                         // until the stream has drained (a polled word; hipStreamSynchronize as fallback)
ViewState* view_of(GvCtx* ctx, uint32_t pool_id, uint32_t view_index);  // NULL: no valid results
bool release_record_target(PoolState::RecordTarget& target);            // false: the range was found unmapped
int copy_shard_of_pool(GvCtx* ctx, uint32_t pool_id, uint32_t view_index, void* dst_device, uint32_t capacity, uint32_t index_base);

// gv_mirror.cpp
int sync_mirror(GvCtx* ctx);                 // brings the device mirror up to date with the bound pools + dirty ranges
TransformMirror xf_mirror(const GvCtx* ctx);  // the transform mirror as the kernels see it

}  // namespace gv
