// gv_exchange.cpp — the multi-GPU exchange step of SURVEY.md §8e in the C-ABI, for hosts without torch.distributed
// (a C++ engine, one process per GPU): every rank's compact visible list goes out as a fixed-capacity shard
// [draw_count, global indices ...] and all ranks gather the shards with ONE equal-size ncclAllGather enqueued on the
// library's exchange stream (every operation of the communicator goes on that one stream, ordered against the context's stream
// by events) — no host synchronisation; counts are read from the shard headers. Same wire format as
// garden_amd/multi.py::VisibleListExchange (the torch.distributed variant in bench.py).
// The node's xGMI fabric is fully connected point to point, and a ring all-gather serialises world-1 hops over it, so
// two direct patterns sit beside the all-gather for an A/B on real hardware (gv_exchange_set_mode, or the environment
// variable GV_EXCHANGE_MODE = allgather | p2p | broadcast read at gv_exchange_init): one ncclGroup of send/recv pairs
// with every peer (each shard crosses exactly one link), or one ncclBroadcast per root. Same bytes in the same place.
//
// RCCL is bound at run time (dlopen): a process that already carries an RCCL — PyTorch bundles one — keeps using that
// copy, and libgarden_vis.so has no link-time dependency on it. GV_RCCL_LIBRARY names another library with the same
// entry points (a site's own RCCL build; the tests' shared-memory transport, tests/cpp/rccl_stub, which lets N ranks share
// one GPU — RCCL itself refuses two ranks on one device).
//
// gv_exchange_visible is the per-frame form the engine calls: the library owns the rows and sizes them from the headers
// of earlier frames, which reach the host through pinned memory (exchange_headers_kernel) — see include/garden_vis.h.
#include <dlfcn.h>

#include "gv_ctx.hpp"

namespace {

struct NcclId {
    char bytes[GV_EXCHANGE_ID_BYTES];
};
using ncclComm_t = void*;

struct Rccl {
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, NcclId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        if (const char* named = getenv("GV_RCCL_LIBRARY")) {
            h = dlopen(named, RTLD_NOW | RTLD_LOCAL);
            if (!h) {
                const char* why = dlerror();  // (once: the call clears the message)
                r.why = std::string("GV_RCCL_LIBRARY=") + named + " not loadable: " + (why ? why : "?");
                return;
            }
        }
        for (const char* name : {"librccl.so", "librccl.so.1"}) {  // an already loaded copy first
            if (h)
                break;
            h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        }
        if (!h)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (h)
                    break;
            }
        if (!h) {
            const char* why = dlerror();  // (once: the call clears the message — asking twice handed std::string a NULL)
            r.why = std::string("librccl not loadable: ") + (why ? why : "?");
            return;
        }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(h, "ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(h, "ncclRecv"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(h, "ncclBroadcast"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString && r.Send && r.Recv &&
               r.Broadcast && r.GroupStart && r.GroupEnd;
        if (!r.ok)
            r.why = "librccl lacks an expected symbol";
    });
    return r;
}

constexpr int kNcclUint32 = 3;  // ncclUint32 (rccl.h: ncclDataType_t)

}  // namespace

namespace gv {

void exchange_release(GvCtx* ctx)
{
    if (ctx->exchange_comm) {
        Rccl& r = rccl();
        if (r.ok)
            (void)r.CommDestroy(ctx->exchange_comm);
        ctx->exchange_comm = nullptr;
    }
    ctx->d_shard.release();
    if (ctx->exchange_stream)
        (void)hipStreamSynchronize(ctx->exchange_stream);
    for (auto& slot : ctx->exchange_slots) {
        slot.rows.release();
        slot.shard.release();
        slot.hdr.release();
        if (slot.produced)
            (void)hipEventDestroy(slot.produced);
        if (slot.done)
            (void)hipEventDestroy(slot.done);
        slot.produced = slot.done = nullptr;
        slot.in_flight = false;
        slot.row_words = 0;
    }
    if (ctx->exchange_stream)
        (void)hipStreamDestroy(ctx->exchange_stream);
    ctx->exchange_stream = nullptr;
    if (ctx->exchange_in)
        (void)hipEventDestroy(ctx->exchange_in);
    if (ctx->exchange_out)
        (void)hipEventDestroy(ctx->exchange_out);
    ctx->exchange_in = ctx->exchange_out = nullptr;
    ctx->d_xcounts.release();
    ctx->h_xcounts.release();
    ctx->exchange_frame = 0;
    ctx->exchange_need_exact = true;
    ctx->exchange_counts_frame = UINT64_MAX;
    ctx->exchange_cut = 0;
}

}  // namespace gv

extern "C" {

int gv_exchange_unique_id(void* out_id)
{
    if (!out_id)
        return GV_E_ARG;
    Rccl& r = rccl();
    if (!r.ok)
        return GV_E_RCCL;
    NcclId id{};
    if (r.GetUniqueId(&id) != 0)
        return GV_E_RCCL;
    memcpy(out_id, id.bytes, GV_EXCHANGE_ID_BYTES);
    return GV_OK;
}

int gv_exchange_init(GvCtx* ctx, const void* unique_id, int rank, int world_size)
{
    if (!ctx)
        return GV_E_ARG;
    if (!unique_id || world_size < 1 || world_size > (int)GV_EXCHANGE_MAX_RANKS || rank < 0 || rank >= world_size)
        return ctx->fail(GV_E_ARG, "gv_exchange_init: bad rank %d / world %d (at most %u ranks)", rank, world_size, GV_EXCHANGE_MAX_RANKS);
    Rccl& r = rccl();
    if (!r.ok)
        return ctx->fail(GV_E_RCCL, "gv_exchange_init: %s", r.why.c_str());
    GV_HIP(ctx, hipSetDevice(ctx->device));
    gv::exchange_release(ctx);
    NcclId id{};
    memcpy(id.bytes, unique_id, GV_EXCHANGE_ID_BYTES);
    ncclComm_t comm = nullptr;
    const int rc = r.CommInitRank(&comm, world_size, id, rank);
    if (rc != 0)
        return ctx->fail(GV_E_RCCL, "ncclCommInitRank: %s", r.GetErrorString(rc));
    ctx->exchange_comm = comm;
    ctx->exchange_rank = rank;
    ctx->exchange_world = world_size;
    GV_HIP(ctx, hipStreamCreateWithFlags(&ctx->exchange_stream, hipStreamNonBlocking));
    GV_HIP(ctx, hipEventCreateWithFlags(&ctx->exchange_in, hipEventDisableTiming));
    GV_HIP(ctx, hipEventCreateWithFlags(&ctx->exchange_out, hipEventDisableTiming));
    for (auto& slot : ctx->exchange_slots) {
        GV_HIP(ctx, hipEventCreateWithFlags(&slot.produced, hipEventDisableTiming));
        GV_HIP(ctx, hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    }
    if (const char* m = getenv("GV_EXCHANGE_MODE")) {
        if (!strcmp(m, "p2p"))
            ctx->exchange_mode = GV_EXCHANGE_P2P;
        else if (!strcmp(m, "broadcast"))
            ctx->exchange_mode = GV_EXCHANGE_BROADCAST;
        else if (!strcmp(m, "allgather"))
            ctx->exchange_mode = GV_EXCHANGE_ALLGATHER;
        else
            return ctx->fail(GV_E_ARG, "gv_exchange_init: GV_EXCHANGE_MODE=%s (allgather | p2p | broadcast)", m);
    }
    return GV_OK;
}

int gv_exchange_set_mode(GvCtx* ctx, uint32_t mode)
{
    if (!ctx)
        return GV_E_ARG;
    if (mode > GV_EXCHANGE_BROADCAST)
        return ctx->fail(GV_E_ARG, "gv_exchange_set_mode: unknown mode %u", mode);
    ctx->exchange_mode = mode;
    return GV_OK;
}

// ctx->d_shard of every rank into rows [rank * row_words ...) of gathered_device, by the configured pattern. travel[r] (NULL:
// row_words for all) = the leading words of rank r's row that matter: the direct patterns move exactly those, the equal-size
// all-gather always moves whole rows.
static int exchange_rows(GvCtx* ctx, size_t row_words, const uint32_t* travel, void* gathered_device, const char* what,
                         const uint32_t* shard = nullptr, hipStream_t stream = nullptr)
{
    Rccl& r = rccl();
    if (!shard)
        shard = ctx->d_shard.ptr;
    if (!stream)
        stream = ctx->stream;
    uint32_t* rows = static_cast<uint32_t*>(gathered_device);
    const int me = ctx->exchange_rank, world = ctx->exchange_world;
    auto words_of = [&](int rank) -> size_t { return travel ? std::min<size_t>(travel[rank], row_words) : row_words; };
    if (ctx->exchange_mode == GV_EXCHANGE_ALLGATHER) {
        const int nrc = r.AllGather(shard, gathered_device, row_words, kNcclUint32, ctx->exchange_comm, stream);
        if (nrc != 0)
            return ctx->fail(GV_E_RCCL, "%s: ncclAllGather: %s", what, r.GetErrorString(nrc));
        return GV_OK;
    }
    // the direct forms place this rank's own row with a device copy; the peers' rows arrive over the links
    GV_HIP(ctx, hipMemcpyAsync(rows + (size_t)me * row_words, shard, words_of(me) * sizeof(uint32_t), hipMemcpyDeviceToDevice,
                               stream));
    if (world == 1)
        return GV_OK;
    int nrc = r.GroupStart();
    if (ctx->exchange_mode == GV_EXCHANGE_P2P) {
        // one send/recv pair per peer inside one group: every shard crosses exactly one xGMI link, all links at once
        for (int d = 1; d < world && nrc == 0; d++) {
            const int to = (me + d) % world, from = (me - d + world) % world;
            nrc = r.Send(shard, words_of(me), kNcclUint32, to, ctx->exchange_comm, stream);
            if (nrc == 0)
                nrc = r.Recv(rows + (size_t)from * row_words, words_of(from), kNcclUint32, from, ctx->exchange_comm, stream);
        }
    } else {
        for (int root = 0; root < world && nrc == 0; root++)
            nrc = r.Broadcast(root == me ? shard : rows + (size_t)root * row_words, rows + (size_t)root * row_words, words_of(root),
                              kNcclUint32, root, ctx->exchange_comm, stream);
    }
    const int erc = r.GroupEnd();
    if (nrc == 0)
        nrc = erc;
    if (nrc != 0)
        return ctx->fail(GV_E_RCCL, "%s: %s exchange: %s", what, ctx->exchange_mode == GV_EXCHANGE_P2P ? "ncclSend/ncclRecv" : "ncclBroadcast",
                         r.GetErrorString(nrc));
    return GV_OK;
}

// the rank's staging shard, every byte defined (the collective reads all of it)
static int reserve_shard(GvCtx* ctx, size_t words)
{
    const uint32_t* before = ctx->d_shard.ptr;
    GV_HIP(ctx, ctx->d_shard.reserve(words));
    if (ctx->d_shard.ptr != before)
        GV_HIP(ctx, hipMemsetAsync(ctx->d_shard.ptr, 0, ctx->d_shard.cap * sizeof(uint32_t), ctx->stream));
    return GV_OK;
}

// The caller-owned forms promise their rows in the order of gv_stream(ctx). The collective itself still runs on the exchange stream
// — every operation of the communicator is issued on ONE stream, in the same order on every rank — between two hand-overs: the
// exchange stream waits for what gv_stream has produced, gv_stream waits for the rows.
static int hand_to_exchange_stream(GvCtx* ctx)
{
    GV_HIP(ctx, hipEventRecord(ctx->exchange_in, ctx->stream));
    GV_HIP(ctx, hipStreamWaitEvent(ctx->exchange_stream, ctx->exchange_in, 0));
    return GV_OK;
}
static int hand_back_from_exchange_stream(GvCtx* ctx)
{
    GV_HIP(ctx, hipEventRecord(ctx->exchange_out, ctx->exchange_stream));
    GV_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->exchange_out, 0));
    return GV_OK;
}

int gv_exchange_shards(GvCtx* ctx, uint32_t view_index, uint32_t capacity, const uint32_t* capacities, uint32_t index_base,
                       void* gathered_device)
{
    if (!ctx)
        return GV_E_ARG;
    if (!ctx->exchange_comm)
        return ctx->fail(GV_E_STATE, "gv_exchange_shards: gv_exchange_init has not run");
    if (!gathered_device || capacity == 0)
        return ctx->fail(GV_E_ARG, "gv_exchange_shards: NULL buffer or zero capacity");
    uint32_t travel[GV_EXCHANGE_MAX_RANKS];
    if (capacities)
        for (int r = 0; r < ctx->exchange_world; r++) {
            if (capacities[r] > capacity)
                return ctx->fail(GV_E_ARG, "gv_exchange_shards: capacities[%d] = %u above the row capacity %u", r, capacities[r], capacity);
            travel[r] = 1u + capacities[r];
        }
    GV_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = reserve_shard(ctx, (size_t)capacity + 1))
        return rc;
    const uint32_t own = capacities ? capacities[ctx->exchange_rank] : capacity;
    if (int rc = gv_results_copy_shard_device(ctx, view_index, ctx->d_shard.ptr, own, index_base))
        return rc;
    if (int rc = hand_to_exchange_stream(ctx))
        return rc;
    if (int rc = exchange_rows(ctx, (size_t)capacity + 1, capacities ? travel : nullptr, gathered_device, "gv_exchange_shards", ctx->d_shard.ptr,
                               ctx->exchange_stream))
        return rc;
    return hand_back_from_exchange_stream(ctx);
}

// ---- gv_exchange_visible: rows owned and sized by the library ----

// room for a list of `count` entries: count + max(count / 8, 1024), rounded up to 1024 words
static uint32_t room_for(uint32_t count)
{
    uint64_t c = (uint64_t)count + std::max<uint64_t>(count / 8u, 1024u);
    c = (c + 1023u) & ~1023ull;
    return (uint32_t)std::min<uint64_t>(c, 0xFFFFFC00u);
}

// waits until the headers of `slot`'s frame are on the host (written by exchange_headers_kernel behind the frame's collective)
static int wait_for_headers(GvCtx* ctx, gv::Context::ExchangeSlot& slot)
{
    const uint32_t seq = (uint32_t)(slot.frame + 1);
    volatile uint32_t* word = slot.hdr.ptr + ctx->exchange_world;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; *word != seq; spins++) {
        // (normally written two frames ago; a host that runs far ahead of the device waits here, which is what bounds it)
        if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {
            GV_HIP(ctx, hipStreamSynchronize(ctx->exchange_stream));
            if (*word != seq)
                return ctx->fail(GV_E_RCCL, "exchange frame %llu: the row headers never reached the host (sequence word %u, expected %u)",
                                 (unsigned long long)slot.frame, *word, seq);
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return GV_OK;
}

// The headers of `slot`'s frame decide the room the coming frames give each rank. Only gv_exchange_visible calls this, for the
// frame two before the one it is about to send: every rank applies every frame's headers at the same point of the same sequence,
// whatever else it asked in between (gv_exchange_counts reads headers early and decides nothing) — the row sizes of a collective
// must agree on all ranks.
static int retire_slot(GvCtx* ctx, gv::Context::ExchangeSlot& slot)
{
    if (!slot.in_flight)
        return GV_OK;
    if (int rc = wait_for_headers(ctx, slot))
        return rc;
    slot.in_flight = false;
    ctx->exchange_counts_frame = slot.frame;
    ctx->exchange_cut = 0;
    for (int r = 0; r < ctx->exchange_world; r++) {
        const uint32_t count = slot.hdr.ptr[r];
        ctx->exchange_counts[r] = count;
        if (count > slot.room[r]) {
            ctx->exchange_cut |= 1ull << r;
            ctx->exchange_need_exact = true;
        }
        const uint32_t want = room_for(count);
        if (want > ctx->exchange_room[r] || (uint64_t)want * 4u < (uint64_t)ctx->exchange_room[r] * 3u)
            ctx->exchange_room[r] = want;
    }
    return GV_OK;
}

int gv_exchange_visible(GvCtx* ctx, uint32_t view_index, uint32_t index_base, uint32_t flags, GvExchangeFrame* out)
{
    if (!ctx)
        return GV_E_ARG;
    if (!ctx->exchange_comm)
        return ctx->fail(GV_E_STATE, "gv_exchange_visible: gv_exchange_init has not run");
    if (!out || (flags & ~GV_EXCHANGE_EXACT))
        return ctx->fail(GV_E_ARG, "gv_exchange_visible: NULL frame or unknown flags 0x%x", flags);
    GV_HIP(ctx, hipSetDevice(ctx->device));
    Rccl& r = rccl();
    const int me = ctx->exchange_rank, world = ctx->exchange_world;
    const uint64_t frame = ctx->exchange_frame;
    gv::Context::ExchangeSlot& slot = ctx->exchange_slots[frame & 1u];
    if (int rc = retire_slot(ctx, slot))  // frame - 2: its rows may be overwritten now, its headers size this frame
        return rc;
    GvDeviceResult dres{};
    if (int rc = gv_pool_results_device(ctx, ctx->last_pool, view_index, &dres))
        return rc;
    if (!dres.visible_idx)
        return ctx->fail(GV_E_ARG, "gv_exchange_visible: view %u has no emitted records", view_index);
    const bool exact = ctx->exchange_need_exact || (flags & GV_EXCHANGE_EXACT);
    if (exact) {
        // this frame's own counts to every rank first (one word each), read on the host: the one synchronising step
        GV_HIP(ctx, ctx->d_xcounts.reserve(GV_EXCHANGE_MAX_RANKS));
        GV_HIP(ctx, ctx->h_xcounts.reserve(GV_EXCHANGE_MAX_RANKS));
        if (int rc = hand_to_exchange_stream(ctx))  // (the count is the emit's output, on the context's stream)
            return rc;
        const int nrc = r.AllGather(dres.draw_count, ctx->d_xcounts.ptr, 1, kNcclUint32, ctx->exchange_comm, ctx->exchange_stream);
        if (nrc != 0)
            return ctx->fail(GV_E_RCCL, "gv_exchange_visible: ncclAllGather of the counts: %s", r.GetErrorString(nrc));
        GV_HIP(ctx, hipMemcpyAsync(ctx->h_xcounts.ptr, ctx->d_xcounts.ptr, (size_t)world * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                   ctx->exchange_stream));
        GV_HIP(ctx, hipStreamSynchronize(ctx->exchange_stream));
        for (int k = 0; k < world; k++)
            ctx->exchange_room[k] = std::max(ctx->exchange_room[k], room_for(ctx->h_xcounts.ptr[k]));
        ctx->exchange_need_exact = false;
    }
    uint32_t widest = 0;
    for (int k = 0; k < world; k++)
        widest = std::max(widest, ctx->exchange_room[k]);
    const size_t row_words = (size_t)widest + 1;
    if ((size_t)world * row_words > slot.rows.cap || row_words > slot.shard.cap) {
        // grown by half again: a list that creeps up does not reallocate every time. (The slot's previous frame is retired — its
        // collective has run — but consumers of its rows may still be queued on the context's stream.)
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        GV_HIP(ctx, hipStreamSynchronize(ctx->exchange_stream));
        if ((size_t)world * row_words > slot.rows.cap)
            GV_HIP(ctx, slot.rows.reserve(std::max((size_t)world * row_words, slot.rows.cap + slot.rows.cap / 2)));
        if (row_words > slot.shard.cap) {
            GV_HIP(ctx, slot.shard.reserve(std::max(row_words, slot.shard.cap + slot.shard.cap / 2)));
            GV_HIP(ctx, hipMemsetAsync(slot.shard.ptr, 0, slot.shard.cap * sizeof(uint32_t), ctx->stream));  // (the all-gather reads whole rows)
        }
    }
    if (!slot.hdr.ptr) {
        GV_HIP(ctx, slot.hdr.reserve(GV_EXCHANGE_MAX_RANKS + 1));
        memset(slot.hdr.ptr, 0, (GV_EXCHANGE_MAX_RANKS + 1) * sizeof(uint32_t));
    }
    uint32_t travel[GV_EXCHANGE_MAX_RANKS];
    for (int k = 0; k < world; k++) {
        slot.room[k] = ctx->exchange_room[k];
        travel[k] = 1u + ctx->exchange_room[k];
    }
    // The shard is the last thing the context's stream does for this frame's list; the links are the exchange stream's business.
    // The next frame's pyramid and cull go on behind the shard copy at once, while this list is still travelling. (The slot's shard
    // and rows are free: retire_slot saw the headers of the frame that used them last, which its collective precedes; and work that
    // was enqueued on the context's stream to CONSUME those rows completes in front of `produced`, which the exchange stream waits for.)
    if (int rc = gv_results_copy_shard_device(ctx, view_index, slot.shard.ptr, ctx->exchange_room[me], index_base))
        return rc;
    GV_HIP(ctx, hipEventRecord(slot.produced, ctx->stream));
    GV_HIP(ctx, hipStreamWaitEvent(ctx->exchange_stream, slot.produced, 0));
    if (int rc = exchange_rows(ctx, row_words, travel, slot.rows.ptr, "gv_exchange_visible", slot.shard.ptr, ctx->exchange_stream))
        return rc;
    GV_HIP(ctx, gv::launch_exchange_headers(slot.rows.ptr, (uint32_t)row_words, (uint32_t)world, slot.hdr.ptr, (uint32_t)(frame + 1),
                                            ctx->exchange_stream));
    GV_HIP(ctx, hipEventRecord(slot.done, ctx->exchange_stream));
    slot.row_words = (uint32_t)row_words;
    slot.frame = frame;
    slot.in_flight = true;
    ctx->exchange_frame = frame + 1;

    memset(out, 0, sizeof(*out));
    out->gathered_device = slot.rows.ptr;
    out->row_words = (uint32_t)row_words;
    out->world_size = (uint32_t)world;
    out->frame = frame;
    for (int k = 0; k < world; k++) {
        out->room[k] = slot.room[k];
        out->travelled_words[k] = ctx->exchange_mode == GV_EXCHANGE_ALLGATHER ? (uint32_t)row_words : travel[k];
        out->counts[k] = ctx->exchange_counts[k];
    }
    out->counts_frame = ctx->exchange_counts_frame;
    out->cut_ranks = ctx->exchange_cut;
    out->exact = exact ? 1u : 0u;
    out->mode = ctx->exchange_mode;
    out->ready_event = slot.done;
    return GV_OK;
}

int gv_exchange_acquire(GvCtx* ctx, uint64_t frame)
{
    if (!ctx)
        return GV_E_ARG;
    if (!ctx->exchange_comm)
        return ctx->fail(GV_E_STATE, "gv_exchange_acquire: gv_exchange_init has not run");
    if (frame >= ctx->exchange_frame || frame + 2 < ctx->exchange_frame || ctx->exchange_slots[frame & 1u].frame != frame)
        return ctx->fail(GV_E_ARG, "gv_exchange_acquire: frame %llu is not one of the last two exchanged (next: %llu)",
                         (unsigned long long)frame, (unsigned long long)ctx->exchange_frame);
    GV_HIP(ctx, hipSetDevice(ctx->device));
    GV_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->exchange_slots[frame & 1u].done, 0));
    return GV_OK;
}

int gv_exchange_counts(GvCtx* ctx, uint64_t frame, uint32_t* counts, uint64_t* cut_ranks)
{
    if (!ctx)
        return GV_E_ARG;
    if (!ctx->exchange_comm)
        return ctx->fail(GV_E_STATE, "gv_exchange_counts: gv_exchange_init has not run");
    if (!counts || frame >= ctx->exchange_frame || frame + 2 < ctx->exchange_frame)
        return ctx->fail(GV_E_ARG, "gv_exchange_counts: frame %llu is not one of the last two exchanged (next: %llu)",
                         (unsigned long long)frame, (unsigned long long)ctx->exchange_frame);
    gv::Context::ExchangeSlot& slot = ctx->exchange_slots[frame & 1u];
    if (slot.frame != frame)
        return ctx->fail(GV_E_STATE, "gv_exchange_counts: frame %llu's rows have been reused", (unsigned long long)frame);
    if (slot.in_flight) {  // (not yet retired: its collective may still be running)
        GV_HIP(ctx, hipSetDevice(ctx->device));
        GV_HIP(ctx, hipStreamSynchronize(ctx->exchange_stream));
        if (int rc = wait_for_headers(ctx, slot))
            return rc;
    }
    uint64_t cut = 0;
    for (int k = 0; k < ctx->exchange_world; k++) {
        counts[k] = slot.hdr.ptr[k];
        if (counts[k] > slot.room[k])
            cut |= 1ull << k;
    }
    if (cut_ranks)
        *cut_ranks = cut;
    return GV_OK;
}

int gv_exchange_masks(GvCtx* ctx, uint32_t view_index, uint32_t word_count, void* gathered_device)
{
    if (!ctx)
        return GV_E_ARG;
    if (!ctx->exchange_comm)
        return ctx->fail(GV_E_STATE, "gv_exchange_masks: gv_exchange_init has not run");
    if (!gathered_device || word_count == 0)
        return ctx->fail(GV_E_ARG, "gv_exchange_masks: NULL buffer or zero word count");
    GV_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = reserve_shard(ctx, (size_t)word_count + 1))
        return rc;
    if (int rc = gv_results_copy_mask_device(ctx, view_index, ctx->d_shard.ptr, word_count))
        return rc;
    if (int rc = hand_to_exchange_stream(ctx))
        return rc;
    if (int rc = exchange_rows(ctx, (size_t)word_count + 1, nullptr, gathered_device, "gv_exchange_masks", ctx->d_shard.ptr, ctx->exchange_stream))
        return rc;
    return hand_back_from_exchange_stream(ctx);
}

int gv_exchange_shutdown(GvCtx* ctx)
{
    if (!ctx)
        return GV_E_ARG;
    GV_HIP(ctx, hipSetDevice(ctx->device));
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->exchange_stream)
        GV_HIP(ctx, hipStreamSynchronize(ctx->exchange_stream));
    gv::exchange_release(ctx);
    return GV_OK;
}

}  // extern "C"
