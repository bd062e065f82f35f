// gv_exchange.cpp — the multi-GPU exchange step of SURVEY.md §8e in the C-ABI, for hosts without torch.distributed
// (a C++ engine: one process per GPU, or one process whose one thread drives N contexts): every rank's compact visible list goes
// out as a shard [draw_count, global indices ...] and all ranks gather the shards into rows on the library's exchange stream (every
// operation of the communicator goes on that one stream, ordered against the context's stream by events). Counts are read from
// the row headers (pinned host words written behind the rows): in the steady state the host waits for nothing but the PREVIOUS
// frame's header words, which size the next frame's rows.
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
// gv_exchange_visible is the per-frame form the engine calls: the library owns the rows, predicts how much of each travels from
// the previous frame's headers (pinned memory, exchange_headers_kernel) and — the reference's gather never drops a record,
// mesh.cpp:177-183 — completes the rows whose list outgrew the prediction with a second, exactly sized exchange before the frame
// is handed out (settle_*, tails_*). The *_all forms drive the N contexts of one process from one thread inside one ncclGroup.
// gv_exchange_init_peers (GV_EXCHANGE_PEER): the N contexts of ONE process need no communicator at all — every rank's list is stored
// straight into its row of every rank's rows over xGMI (peer_scatter_kernel), ordered by events; rows are as wide as a rank's pools,
// so nothing is predicted and no frame is ever short (peer_collective below). RCCL is not even loaded then.
// See include/garden_vis.h.
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
    int (*CommAbort)(ncclComm_t) = nullptr;  // optional
    int (*CommGetAsyncError)(ncclComm_t, int*) = nullptr;  // optional
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
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(h, "ncclCommAbort"));
        r.CommGetAsyncError = reinterpret_cast<decltype(r.CommGetAsyncError)>(dlsym(h, "ncclCommGetAsyncError"));
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

// The communicator can no longer be trusted (a wait ran out, or RCCL reported an asynchronous error): a collective that will never
// finish must not keep the device — and with it every hipFree / synchronise of the process — waiting, so it is aborted on the spot.
int give_up(GvCtx* ctx, int code, const char* text)
{
    ctx->exchange_broken = true;
    if (ctx->exchange_comm) {  // (a peer group has none — and never loads RCCL)
        Rccl& r = rccl();
        if (r.CommAbort) {
            (void)r.CommAbort(ctx->exchange_comm);
            ctx->exchange_comm = nullptr;
        }
    }
    ctx->error = text;
    return code;
}

// what RCCL has to say about the collectives already enqueued (ncclCommGetAsyncError: a peer that died, a transport error)
int async_error(GvCtx* ctx)
{
    if (!ctx->exchange_comm)
        return 0;
    Rccl& r = rccl();
    int async = 0;
    if (!r.CommGetAsyncError || r.CommGetAsyncError(ctx->exchange_comm, &async) != 0)
        return 0;
    return async == 7 ? 0 : async;  // (ncclInProgress, rccl.h: a non-blocking communicator still at work — not an error)
}

// Waits — bounded, like every wait of the exchange — until everything queued on the exchange stream has run: what stands in front of
// a shutdown, a re-init or gv_destroy. A frame that was sent and never acquired may sit behind a peer that has left; synchronising
// the stream would then never return.
int drain_exchange_stream(GvCtx* ctx)
{
    if (!ctx->exchange_stream || ctx->exchange_broken)
        return GV_OK;
    (void)hipSetDevice(ctx->device);
    const auto t0 = std::chrono::steady_clock::now();
    char text[384];
    for (;;) {
        const hipError_t e = hipStreamQuery(ctx->exchange_stream);
        if (e == hipSuccess)
            return GV_OK;
        if (e != hipErrorNotReady)
            return ctx->hip_fail(e, "gv_exchange: draining the exchange stream");
        const auto waited = std::chrono::steady_clock::now() - t0;
        if (waited > std::chrono::milliseconds(ctx->exchange_timeout_ms)) {
            snprintf(text, sizeof(text), "exchange: the collectives still queued did not finish within %u ms (a frame that was sent and never acquired, "
                     "behind a peer rank that stalled or left); the communicator has been aborted", ctx->exchange_timeout_ms);
            return give_up(ctx, GV_E_TIMEOUT, text);
        }
        if (waited > std::chrono::milliseconds(2)) {
            if (const int async = async_error(ctx)) {
                snprintf(text, sizeof(text), "exchange: RCCL reports an asynchronous error behind the collectives still queued: %s; the communicator has been aborted",
                         rccl().GetErrorString(async));
                return give_up(ctx, GV_E_RCCL, text);
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
}

}  // namespace

namespace gv {

int exchange_drain(GvCtx* ctx) { return drain_exchange_stream(ctx); }

void exchange_release(GvCtx* ctx)
{
    // A peer group (gv_exchange_init_peers) goes as a whole: the other members' scatter kernels store into THIS context's rows, so
    // every member's exchange stream is drained before anything here is freed, and no member keeps a pointer to a context that may
    // be destroyed next (the others answer GV_E_STATE until they are initialised again; their buffers go with their own release).
    if (!ctx->exchange_peers.empty()) {
        const std::vector<GvCtx*> members = ctx->exchange_peers;
        for (GvCtx* m : members)
            if (m != ctx)
                (void)drain_exchange_stream(m);
        for (GvCtx* m : members)
            m->exchange_peers.clear();
        (void)hipSetDevice(ctx->device);
    }
    // what is still queued on the exchange stream goes first: gv_stream never waits for a frame's collective, so the communicator
    // would otherwise be destroyed under a queued all-gather. (After a timeout the stream may never drain: abort instead.)
    (void)drain_exchange_stream(ctx);  // (bounded: a timeout marks the communicator broken, and it is aborted below)
    bool hung = false;  // a collective that will never finish and cannot be aborted: its stream is left alone (and leaked)
    if (ctx->exchange_comm) {
        Rccl& r = rccl();
        if (r.ok) {
            if (ctx->exchange_broken && r.CommAbort)
                (void)r.CommAbort(ctx->exchange_comm);
            else if (ctx->exchange_broken)
                hung = true;
            else
                (void)r.CommDestroy(ctx->exchange_comm);
        }
        ctx->exchange_comm = nullptr;
    }
    if (hung) {
        // ... and so is everything the collective may still touch: hipFree waits for the device, i.e. for ever. Leaked, like the stream.
        ctx->exchange_stream = nullptr;
        ctx->d_shard.ptr = nullptr;
        ctx->d_shard.cap = 0;
        for (auto& slot : ctx->exchange_slots) {
            slot.rows.ptr = slot.shard.ptr = nullptr;
            slot.rows.cap = slot.shard.cap = 0;
            slot.d_items.ptr = nullptr;
            slot.d_items.cap = 0;
        }
    }
    if (ctx->exchange_stream)
        (void)hipStreamSynchronize(ctx->exchange_stream);
    ctx->d_shard.release();
    for (auto& slot : ctx->exchange_slots) {
        slot.rows.release();
        slot.shard.release();
        slot.hdr.release();
        slot.h_items.release();
        slot.d_items.release();
        slot.items = 0;
        slot.hdr_words = 1;
        slot.item_counts.clear();
        slot.items_uploaded.clear();
        if (slot.produced)
            (void)hipEventDestroy(slot.produced);
        if (slot.done)
            (void)hipEventDestroy(slot.done);
        for (hipEvent_t e : {slot.sent, slot.all_produced, slot.all_sent})
            if (e)
                (void)hipEventDestroy(e);
        slot.produced = slot.done = slot.sent = slot.all_produced = slot.all_sent = nullptr;
        slot.in_flight = slot.settled = false;
        slot.row_words = 0;
        slot.frame = 0;
        slot.cut = 0;
        memset(slot.room, 0, sizeof(slot.room));
        memset(slot.counts, 0, sizeof(slot.counts));
        memset(slot.tail_words, 0, sizeof(slot.tail_words));
        memset(slot.travelled, 0, sizeof(slot.travelled));
    }
    if (ctx->exchange_stream)
        (void)hipStreamDestroy(ctx->exchange_stream);
    ctx->exchange_stream = nullptr;
    if (ctx->exchange_in)
        (void)hipEventDestroy(ctx->exchange_in);
    if (ctx->exchange_out)
        (void)hipEventDestroy(ctx->exchange_out);
    ctx->exchange_in = ctx->exchange_out = nullptr;
    // a communicator starts from nothing: the rooms of an earlier one (other ranks, another history) must not size its rows —
    // ranks with different pasts would enter frame 0's collective with rows of different lengths
    ctx->exchange_frame = 0;
    memset(ctx->exchange_room, 0, sizeof(ctx->exchange_room));
    ctx->exchange_broken = false;
    ctx->exchange_by_group = false;
    ctx->exchange_rank = 0;
    ctx->exchange_world = 1;
    if (ctx->exchange_mode == GV_EXCHANGE_PEER)  // (the pattern of a peer group only: a communicator made next starts from the default)
        ctx->exchange_mode = GV_EXCHANGE_ALLGATHER;
}

}  // namespace gv

namespace {

using Slot = gv::Context::ExchangeSlot;

// streams, events, mode and timeout of a communicator about to be made — BEFORE the communicator exists, so that a failure here
// leaves no communicator behind that later calls would drive with a NULL stream
int exchange_setup(GvCtx* ctx, int rank, int world_size)
{
    GV_HIP(ctx, hipSetDevice(ctx->device));
    gv::exchange_release(ctx);
    ctx->exchange_rank = rank;
    ctx->exchange_world = world_size;
    GV_HIP(ctx, hipStreamCreateWithFlags(&ctx->exchange_stream, hipStreamNonBlocking));
    GV_HIP(ctx, hipEventCreateWithFlags(&ctx->exchange_in, hipEventDisableTiming));
    GV_HIP(ctx, hipEventCreateWithFlags(&ctx->exchange_out, hipEventDisableTiming));
    for (auto& slot : ctx->exchange_slots) {
        GV_HIP(ctx, hipEventCreateWithFlags(&slot.produced, hipEventDisableTiming));
        GV_HIP(ctx, hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
        GV_HIP(ctx, hipEventCreateWithFlags(&slot.sent, hipEventDisableTiming));
        GV_HIP(ctx, hipEventCreateWithFlags(&slot.all_produced, hipEventDisableTiming));
        GV_HIP(ctx, hipEventCreateWithFlags(&slot.all_sent, hipEventDisableTiming));
    }
    if (const char* m = getenv("GV_EXCHANGE_MODE")) {
        if (!strcmp(m, "p2p"))
            ctx->exchange_mode = GV_EXCHANGE_P2P;
        else if (!strcmp(m, "broadcast"))
            ctx->exchange_mode = GV_EXCHANGE_BROADCAST;
        else if (!strcmp(m, "allgather"))
            ctx->exchange_mode = GV_EXCHANGE_ALLGATHER;
        else
            return ctx->fail(GV_E_ARG, "gv_exchange_init: GV_EXCHANGE_MODE=%s (allgather | p2p | broadcast: a communicator's travel patterns)", m);
    }
    if (const char* t = getenv("GV_EXCHANGE_TIMEOUT_MS")) {
        const long ms = atol(t);
        if (ms > 0)
            ctx->exchange_timeout_ms = (uint32_t)std::min<long>(ms, 0x7FFFFFFFl);
    }
    return GV_OK;
}

int exchange_setup_failed(GvCtx* ctx, int rc)
{
    const std::string why = ctx->error;  // (the release resets nothing of the error text, but keep it explicit)
    gv::exchange_release(ctx);
    ctx->error = why;
    return rc;
}

}  // namespace

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
    if (int rc = exchange_setup(ctx, rank, world_size))
        return exchange_setup_failed(ctx, rc);
    NcclId id{};
    memcpy(id.bytes, unique_id, GV_EXCHANGE_ID_BYTES);
    ncclComm_t comm = nullptr;
    const int rc = r.CommInitRank(&comm, world_size, id, rank);
    if (rc != 0)
        return exchange_setup_failed(ctx, ctx->fail(GV_E_RCCL, "ncclCommInitRank: %s", r.GetErrorString(rc)));
    ctx->exchange_comm = comm;
    return GV_OK;
}

int gv_exchange_init_all(GvCtx* const* contexts, int world_size)
{
    if (!contexts || world_size < 1 || world_size > (int)GV_EXCHANGE_MAX_RANKS)
        return GV_E_ARG;
    for (int k = 0; k < world_size; k++) {
        if (!contexts[k])
            return GV_E_ARG;
        for (int j = 0; j < k; j++)
            if (contexts[j] == contexts[k])
                return contexts[k]->fail(GV_E_ARG, "gv_exchange_init_all: context %d is listed twice", k);
    }
    GvCtx* first = contexts[0];
    Rccl& r = rccl();
    if (!r.ok)
        return first->fail(GV_E_RCCL, "gv_exchange_init_all: %s", r.why.c_str());
    NcclId id{};
    if (r.GetUniqueId(&id) != 0)
        return first->fail(GV_E_RCCL, "gv_exchange_init_all: ncclGetUniqueId failed");
    int rc = GV_OK;
    for (int k = 0; k < world_size && rc == GV_OK; k++)
        rc = exchange_setup(contexts[k], k, world_size);
    ncclComm_t comms[GV_EXCHANGE_MAX_RANKS] = {};
    // every device is selected once BEFORE the group opens: nothing but ncclCommInitRank itself can fail between ncclGroupStart and
    // ncclGroupEnd (a group closed over fewer than world_size ranks would wait for the missing ones for ever)
    for (int k = 0; k < world_size && rc == GV_OK; k++)
        if (hipSetDevice(contexts[k]->device) != hipSuccess)
            rc = contexts[k]->fail(GV_E_HIP, "gv_exchange_init_all: hipSetDevice(%d)", contexts[k]->device);
    bool started = false;
    if (rc == GV_OK) {
        // one thread, N devices: the N ncclCommInitRank calls meet inside ONE group (each would otherwise wait for the others)
        int nrc = r.GroupStart();
        started = nrc == 0;
        for (int k = 0; k < world_size && nrc == 0; k++) {
            (void)hipSetDevice(contexts[k]->device);  // (checked above)
            nrc = r.CommInitRank(&comms[k], world_size, id, k);
        }
        const int erc = r.GroupEnd();
        if (nrc == 0)
            nrc = erc;
        if (nrc != 0)
            rc = first->fail(GV_E_RCCL, "gv_exchange_init_all: ncclCommInitRank: %s", r.GetErrorString(nrc));
    }
    if (rc != GV_OK && started)  // a group that failed half way: its communicators never met — aborted, not destroyed (a destroy may wait for peers)
        for (int k = 0; k < world_size; k++)
            if (comms[k]) {
                if (r.CommAbort)
                    (void)r.CommAbort(comms[k]);
                comms[k] = nullptr;
            }
    for (int k = 0; k < world_size; k++) {
        contexts[k]->exchange_comm = comms[k];  // (a failed start: released below, communicator included)
        contexts[k]->exchange_by_group = true;
    }
    if (rc != GV_OK) {
        const std::string why = first->error;
        for (int k = 0; k < world_size; k++)
            gv::exchange_release(contexts[k]);
        first->error = why;
    }
    return rc;
}

int gv_exchange_init_peers(GvCtx* const* contexts, int world_size)
{
    if (!contexts || world_size < 1 || world_size > (int)GV_EXCHANGE_MAX_RANKS)
        return GV_E_ARG;
    for (int k = 0; k < world_size; k++) {
        if (!contexts[k])
            return GV_E_ARG;
        for (int j = 0; j < k; j++)
            if (contexts[j] == contexts[k])
                return contexts[k]->fail(GV_E_ARG, "gv_exchange_init_peers: context %d is listed twice", k);
    }
    GvCtx* first = contexts[0];
    // every device must reach every other device's memory (xGMI within a node; contexts that share a device need nothing)
    for (int a = 0; a < world_size; a++)
        for (int b = 0; b < world_size; b++) {
            const int da = contexts[a]->device, db = contexts[b]->device;
            if (da == db)
                continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, da, db) != hipSuccess || !can) {
                (void)hipGetLastError();
                return first->fail(GV_E_STATE, "gv_exchange_init_peers: device %d cannot store into device %d's memory (no peer access): use gv_exchange_init_all", da, db);
            }
        }
    int rc = GV_OK;
    GvCtx* culprit = first;  // the context whose text says what went wrong
    for (int k = 0; k < world_size && rc == GV_OK; k++) {
        rc = exchange_setup(contexts[k], k, world_size);  // (releases what was there, a former group included)
        culprit = contexts[k];
    }
    for (int a = 0; a < world_size && rc == GV_OK; a++) {
        const int da = contexts[a]->device;
        culprit = contexts[a];
        if (hipSetDevice(da) != hipSuccess) {
            rc = contexts[a]->fail(GV_E_HIP, "gv_exchange_init_peers: hipSetDevice(%d)", da);
            break;
        }
        for (int b = 0; b < world_size && rc == GV_OK; b++) {
            const int db = contexts[b]->device;
            bool seen = db == da;
            for (int j = 0; j < b && !seen; j++)
                seen = contexts[j]->device == db;
            if (seen)
                continue;
            const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
            if (e == hipErrorPeerAccessAlreadyEnabled)
                (void)hipGetLastError();
            else if (e != hipSuccess)
                rc = contexts[a]->hip_fail(e, "gv_exchange_init_peers: hipDeviceEnablePeerAccess");
        }
    }
    if (rc != GV_OK) {
        const std::string why = culprit->error;
        for (int k = 0; k < world_size; k++)
            gv::exchange_release(contexts[k]);
        first->error = why;  // (callers of the *_all forms ask contexts[0])
        return rc;
    }
    const std::vector<GvCtx*> members(contexts, contexts + world_size);
    for (int k = 0; k < world_size; k++) {
        contexts[k]->exchange_peers = members;
        contexts[k]->exchange_by_group = true;
        contexts[k]->exchange_mode = GV_EXCHANGE_PEER;
    }
    return GV_OK;
}

int gv_exchange_set_mode(GvCtx* ctx, uint32_t mode)
{
    if (!ctx)
        return GV_E_ARG;
    if (mode > GV_EXCHANGE_PEER)
        return ctx->fail(GV_E_ARG, "gv_exchange_set_mode: unknown mode %u", mode);
    // a peer group travels by GV_EXCHANGE_PEER and nothing else (it has no communicator); a communicator by anything else
    if (!ctx->exchange_peers.empty() ? mode != GV_EXCHANGE_PEER : mode == GV_EXCHANGE_PEER)
        return ctx->fail(GV_E_ARG, "gv_exchange_set_mode: mode %u %s", mode, mode == GV_EXCHANGE_PEER ? "needs gv_exchange_init_peers" : "needs a communicator (this context belongs to a peer group)");
    ctx->exchange_mode = mode;
    return GV_OK;
}

int gv_exchange_set_timeout(GvCtx* ctx, uint32_t milliseconds)
{
    if (!ctx)
        return GV_E_ARG;
    ctx->exchange_timeout_ms = milliseconds ? milliseconds : 30000u;
    return GV_OK;
}

}  // extern "C"

namespace {

// `shard` of every rank into rows [rank * row_words ...) of gathered_device, by the configured pattern. travel[r] (NULL:
// row_words for all) = the leading words of rank r's row that matter: the direct patterns move exactly those, the equal-size
// all-gather always moves whole rows.
int exchange_rows(GvCtx* ctx, size_t row_words, const uint32_t* travel, void* gathered_device, const char* what, const uint32_t* shard,
                  hipStream_t stream)
{
    Rccl& r = rccl();
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

// the rank's staging shard of the caller-sized forms, every byte defined (the collective reads all of it)
int reserve_shard(GvCtx* ctx, size_t words)
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
int hand_to_exchange_stream(GvCtx* ctx)
{
    GV_HIP(ctx, hipEventRecord(ctx->exchange_in, ctx->stream));
    GV_HIP(ctx, hipStreamWaitEvent(ctx->exchange_stream, ctx->exchange_in, 0));
    return GV_OK;
}
int hand_back_from_exchange_stream(GvCtx* ctx)
{
    GV_HIP(ctx, hipEventRecord(ctx->exchange_out, ctx->exchange_stream));
    GV_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->exchange_out, 0));
    return GV_OK;
}

int usable(GvCtx* ctx, const char* what, bool by_group)
{
    if (ctx->exchange_broken)
        return ctx->fail(GV_E_STATE, "%s: the communicator timed out earlier and was aborted (gv_exchange_shutdown, then gv_exchange_init again)", what);
    if (!ctx->exchange_comm && ctx->exchange_peers.empty())
        return ctx->fail(GV_E_STATE, "%s: gv_exchange_init has not run", what);
    if (!by_group && ctx->exchange_by_group && ctx->exchange_world > 1)
        return ctx->fail(GV_E_STATE, "%s: this communicator was made by gv_exchange_init_all / _init_peers — one thread drives its %d ranks through the "
                         "*_all calls (a per-rank call would wait for ranks the same thread has not reached yet)", what, ctx->exchange_world);
    return GV_OK;
}

// ---- gv_exchange_visible: rows owned and sized by the library ----

// room for a list of `count` entries: count + max(count / 8, 1024), rounded up to 1024 words
uint32_t room_for(uint32_t count)
{
    uint64_t c = (uint64_t)count + std::max<uint64_t>(count / 8u, 1024u);
    c = (c + 1023u) & ~1023ull;
    return (uint32_t)std::min<uint64_t>(c, 0xFFFFFC00u);
}

// words between two rows that hold up to `entries` list entries behind their header: a multiple of 4 (16-byte rows)
size_t row_words_for(uint32_t entries)
{
    return ((size_t)entries + 1u + 3u) & ~(size_t)3u;
}

// Waits until the headers of `slot`'s frame are on the host (written by exchange_headers_kernel behind the frame's collective).
// Bounded: a peer that never enters the collective leaves it spinning on the device for ever — the host gets a status code. Rows
// behind a collective that RCCL reports as failed are not handed out either.
int wait_for_headers(GvCtx* ctx, Slot& slot)
{
    const uint32_t seq = (uint32_t)(slot.frame + 1);
    volatile uint32_t* word = slot.hdr.ptr;  // (a fixed place: whatever an earlier frame of another shape left there is an older number)
    const auto t0 = std::chrono::steady_clock::now();
    char text[384];
    for (uint32_t spins = 0; *word != seq; spins++) {
        if ((spins & 255u) != 255u)
            continue;
        const auto waited = std::chrono::steady_clock::now() - t0;
        if (waited > std::chrono::milliseconds(ctx->exchange_timeout_ms)) {
            snprintf(text, sizeof(text), "exchange frame %llu: the row headers did not reach the host within %u ms (sequence word %u, expected %u): "
                     "a peer rank stalled or left; the communicator has been aborted", (unsigned long long)slot.frame, ctx->exchange_timeout_ms, *word, seq);
            return give_up(ctx, GV_E_TIMEOUT, text);
        }
        if (waited > std::chrono::milliseconds(2)) {
            if (const int async = async_error(ctx)) {
                snprintf(text, sizeof(text), "exchange frame %llu: RCCL reports an asynchronous error while its rows travel: %s; the communicator has been aborted",
                         (unsigned long long)slot.frame, rccl().GetErrorString(async));
                return give_up(ctx, GV_E_RCCL, text);
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (const int async = async_error(ctx)) {  // (the stream went on behind a collective that did not complete: its rows are not the lists)
        snprintf(text, sizeof(text), "exchange frame %llu: RCCL reports an asynchronous error behind its rows: %s; the communicator has been aborted",
                 (unsigned long long)slot.frame, rccl().GetErrorString(async));
        return give_up(ctx, GV_E_RCCL, text);
    }
    return GV_OK;
}

// Reads the frame's own headers: which rows are short, and what room the NEXT frame gives each rank. Every rank holds the same
// headers and settles every frame exactly once, in front of the next frame's sizing: the sizes of a collective agree on all ranks.
int settle_read(GvCtx* ctx, Slot& slot)
{
    if (!slot.in_flight)
        return GV_OK;
    if (int rc = wait_for_headers(ctx, slot))
        return rc;
    slot.in_flight = false;
    slot.cut = 0;
    slot.item_counts.assign((size_t)ctx->exchange_world * slot.items, 0u);
    for (int r = 0; r < ctx->exchange_world; r++) {
        const uint32_t* row_head = slot.hdr.ptr + 1u + (size_t)r * slot.hdr_words;
        const uint32_t count = row_head[0];
        for (uint32_t i = 0; i < slot.items; i++)
            slot.item_counts[(size_t)r * slot.items + i] = row_head[1u + i];
        slot.counts[r] = count;
        slot.tail_words[r] = 0;
        if (slot.mode == GV_EXCHANGE_PEER) {  // rows as wide as the pools: what a list holds is what travelled, whole
            if (count > slot.room[r])
                return ctx->fail(GV_E_STATE, "gv_exchange: rank %d's header (%u words) exceeds its row (%u) in a peer frame", r, count, slot.room[r]);
            slot.travelled[r] = 1u + count;
            continue;
        }
        if (count > slot.room[r]) {
            slot.cut |= 1ull << r;
            slot.tail_words[r] = count - slot.room[r];
        }
        const uint32_t want = room_for(count);
        if (want > ctx->exchange_room[r] || (uint64_t)want * 4u < (uint64_t)ctx->exchange_room[r] * 3u)
            ctx->exchange_room[r] = want;
    }
    slot.settled = slot.cut == 0;
    if (slot.cut)
        ctx->stats.exchange_tail_rounds++;
    return GV_OK;
}

// A frame with short rows, step 1 of 3: rows wide enough for the longest list. (The exchange stream is idle: the frame's headers
// have been read, and nobody holds the rows — an unsettled frame has not been handed out.)
int tails_stage(GvCtx* ctx, Slot& slot)
{
    if (slot.settled)
        return GV_OK;
    GV_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t longest = 0;
    for (int r = 0; r < ctx->exchange_world; r++)
        longest = std::max(longest, std::max(slot.room[r], slot.counts[r]));
    const size_t need = row_words_for(longest), world = (size_t)ctx->exchange_world;
    if (need > slot.row_words) {
        gv::DeviceBuf<uint32_t> wider;
        GV_HIP(ctx, wider.reserve(world * need));
        hipError_t e = hipSuccess;
        for (size_t r = 0; r < world && e == hipSuccess; r++)
            e = hipMemcpyAsync(wider.ptr + r * need, slot.rows.ptr + r * slot.row_words, (size_t)slot.row_words * sizeof(uint32_t),
                               hipMemcpyDeviceToDevice, ctx->exchange_stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->exchange_stream);
        if (e != hipSuccess) {
            wider.release();
            return ctx->hip_fail(e, "gv_exchange: widening the rows of a frame with short rows");
        }
        std::swap(wider.ptr, slot.rows.ptr);
        std::swap(wider.cap, slot.rows.cap);
        wider.release();
        slot.row_words = (uint32_t)need;
    }
    return GV_OK;
}

// ... step 2: every short rank broadcasts the tail of its list — out of the staging shard, which holds the whole list until the
// slot's next frame — to its place in everybody's rows. Exactly sized: every rank knows every count.
int tails_collective(GvCtx* ctx, Slot& slot)
{
    if (slot.settled)
        return GV_OK;
    Rccl& r = rccl();
    const int me = ctx->exchange_rank;
    int nrc = r.GroupStart();
    for (int root = 0; root < ctx->exchange_world && nrc == 0; root++) {
        if (!((slot.cut >> root) & 1ull))
            continue;
        uint32_t* place = slot.rows.ptr + (size_t)root * slot.row_words + 1u + slot.room[root];
        nrc = r.Broadcast(root == me ? slot.shard.ptr + 1u + slot.room[root] : place, place, slot.tail_words[root], kNcclUint32, root,
                          ctx->exchange_comm, ctx->exchange_stream);
    }
    const int erc = r.GroupEnd();
    if (nrc == 0)
        nrc = erc;
    if (nrc != 0)
        return ctx->fail(GV_E_RCCL, "gv_exchange: completing the short rows of frame %llu: ncclBroadcast: %s", (unsigned long long)slot.frame,
                         r.GetErrorString(nrc));
    return GV_OK;
}

// ... step 3: the frame's ready event moves behind the tails.
int tails_finish(GvCtx* ctx, Slot& slot)
{
    if (slot.settled)
        return GV_OK;
    GV_HIP(ctx, hipSetDevice(ctx->device));
    GV_HIP(ctx, hipEventRecord(slot.done, ctx->exchange_stream));
    slot.settled = true;
    return GV_OK;
}

// ... and the host sees the tails arrive, with the bounded wait: a frame is handed out only behind collectives that are known to
// have finished (a peer that left between the headers and the tails would otherwise hang the first synchronisation of the context's
// stream, which the acquire orders behind `done`). Frames with short rows are host-synchronising anyway, and rare.
int tails_arrived(GvCtx* ctx, Slot& slot)
{
    GV_HIP(ctx, hipSetDevice(ctx->device));
    const auto t0 = std::chrono::steady_clock::now();
    char text[320];
    for (;;) {
        const hipError_t e = hipEventQuery(slot.done);
        if (e == hipSuccess)
            return GV_OK;
        if (e != hipErrorNotReady) {
            slot.settled = false;  // (rows whose tails are not known to have arrived are never handed out)
            return ctx->hip_fail(e, "gv_exchange: waiting for the tails of a frame with short rows");
        }
        const auto waited = std::chrono::steady_clock::now() - t0;
        if (waited > std::chrono::milliseconds(ctx->exchange_timeout_ms)) {
            snprintf(text, sizeof(text), "exchange frame %llu: the tails of its short rows did not arrive within %u ms: a peer rank stalled or left; the "
                     "communicator has been aborted", (unsigned long long)slot.frame, ctx->exchange_timeout_ms);
            slot.settled = false;
            return give_up(ctx, GV_E_TIMEOUT, text);
        }
        if (waited > std::chrono::milliseconds(2)) {
            if (const int async = async_error(ctx)) {
                snprintf(text, sizeof(text), "exchange frame %llu: RCCL reports an asynchronous error while the tails of its short rows travel: %s; the communicator "
                         "has been aborted", (unsigned long long)slot.frame, rccl().GetErrorString(async));
                slot.settled = false;
                return give_up(ctx, GV_E_RCCL, text);
            }
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }
}

// The three steps for all the contexts of a call (one for the per-rank forms): the collectives of all of them inside one group.
int settle(GvCtx* const* ctxs, int n, unsigned which)
{
    for (int k = 0; k < n; k++)
        if (int rc = settle_read(ctxs[k], ctxs[k]->exchange_slots[which])) {
            // a wait that ran out on one rank of the call: the collective hangs on all of them — none is left holding the device
            if (rc == GV_E_TIMEOUT || rc == GV_E_RCCL)
                for (int j = 0; j < n; j++)
                    if (j != k && !ctxs[j]->exchange_broken)
                        (void)give_up(ctxs[j], rc, ctxs[k]->error.c_str());
            return rc;
        }
    bool short_rows = false;
    for (int k = 0; k < n; k++) {
        Slot& slot = ctxs[k]->exchange_slots[which];
        short_rows = short_rows || !slot.settled;
        if (slot.settled != ctxs[0]->exchange_slots[which].settled)  // (every rank reads the same headers)
            return ctxs[k]->fail(GV_E_STATE, "gv_exchange: the ranks of one call disagree about frame %llu's headers", (unsigned long long)slot.frame);
    }
    if (!short_rows)
        return GV_OK;
    Rccl& r = rccl();
    for (int k = 0; k < n; k++)
        if (int rc = tails_stage(ctxs[k], ctxs[k]->exchange_slots[which]))
            return rc;
    int rc = GV_OK;
    (void)r.GroupStart();
    for (int k = 0; k < n && rc == GV_OK; k++) {
        if (hipSetDevice(ctxs[k]->device) != hipSuccess)
            rc = ctxs[k]->fail(GV_E_HIP, "hipSetDevice(%d)", ctxs[k]->device);
        else
            rc = tails_collective(ctxs[k], ctxs[k]->exchange_slots[which]);
    }
    const int erc = r.GroupEnd();
    if (rc == GV_OK && erc != 0)
        rc = ctxs[0]->fail(GV_E_RCCL, "gv_exchange: ncclGroupEnd: %s", r.GetErrorString(erc));
    for (int k = 0; k < n && rc == GV_OK; k++)
        rc = tails_finish(ctxs[k], ctxs[k]->exchange_slots[which]);
    for (int k = 0; k < n && rc == GV_OK; k++) {
        rc = tails_arrived(ctxs[k], ctxs[k]->exchange_slots[which]);
        if (rc == GV_E_TIMEOUT || rc == GV_E_RCCL)  // (as above: none of the call's ranks is left holding its device)
            for (int j = 0; j < n; j++)
                if (j != k && !ctxs[j]->exchange_broken)
                    (void)give_up(ctxs[j], rc, ctxs[k]->error.c_str());
    }
    return rc;
}

// A new frame, step 1 of 3: buffers, and this rank's whole list — or, batched (gv_exchange_views), ALL the frame's lists behind their
// count table — into the slot's staging shard on the context's stream. items / item_count: the frame's (pool, view) pairs;
// batched == false: one pair, no table (the single-list forms; pool GV_NONE = the pool of the most recent gv_cull).
// what the frame's whole shard can need on this rank — every list at its pool's occupancy — and the widest of its pools
int shard_need(GvCtx* ctx, const GvExchangeItem* items, uint32_t item_count, size_t* list_words, uint32_t* widest_pool)
{
    *list_words = 0;
    *widest_pool = 0;
    for (uint32_t i = 0; i < item_count; i++) {
        const uint32_t pool_id = items[i].pool_id == GV_NONE ? ctx->last_pool : items[i].pool_id;  // the view-indexed forms address the pool of the most recent gv_cull
        gv::ViewState* vs = pool_id < GV_MAX_POOLS ? gv::view_of(ctx, pool_id, items[i].view_index) : nullptr;
        if (!vs || !vs->emitted)
            return ctx->fail(GV_E_ARG, "gv_exchange_visible: pool %u view %u has no emitted records", pool_id, items[i].view_index);
        *list_words += vs->occupancy;
        *widest_pool = std::max(*widest_pool, vs->occupancy);
    }
    return GV_OK;
}

// peer_entries: GV_EXCHANGE_PEER — the entries every row has room for (the largest shard of the group); 0: rooms are predicted.
int frame_stage(GvCtx* ctx, const GvExchangeItem* items, uint32_t item_count, bool batched, Slot& slot, uint32_t peer_entries = 0)
{
    GV_HIP(ctx, hipSetDevice(ctx->device));
    const int world = ctx->exchange_world;
    size_t list_words = 0;
    uint32_t widest_pool = 0;
    if (int rc = shard_need(ctx, items, item_count, &list_words, &widest_pool))
        return rc;
    const uint32_t table_words = batched ? item_count : 0u;
    // (a row's count table always travels with its header: the direct patterns move 1 + room words, and the first frame of a
    // communicator has no history — room 0 — so the room is never smaller than the table)
    uint32_t rooms[GV_EXCHANGE_MAX_RANKS];
    uint32_t widest = 0;
    for (int k = 0; k < world; k++) {
        rooms[k] = peer_entries ? peer_entries : std::max(ctx->exchange_room[k], table_words);
        widest = std::max(widest, rooms[k]);
    }
    const size_t row_words = row_words_for(widest);
    // the shard holds the WHOLE list (what a short prediction leaves behind travels later, from here); the equal-size all-gather
    // reads row_words words of it whatever the list's length
    const size_t shard_words = std::max(row_words, list_words + table_words + 1u);
    if ((size_t)world * row_words > slot.rows.cap || shard_words > slot.shard.cap) {
        // grown by half again: a list that creeps up does not reallocate every time. (The slot's previous frame is settled — its
        // collectives have run — but consumers of its rows may still be queued on the context's stream.)
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        GV_HIP(ctx, hipStreamSynchronize(ctx->exchange_stream));
        if ((size_t)world * row_words > slot.rows.cap)
            GV_HIP(ctx, slot.rows.reserve(std::max((size_t)world * row_words, slot.rows.cap + slot.rows.cap / 2)));
        if (shard_words > slot.shard.cap) {
            GV_HIP(ctx, slot.shard.reserve(std::max(shard_words, slot.shard.cap + slot.shard.cap / 2)));
            GV_HIP(ctx, hipMemsetAsync(slot.shard.ptr, 0, slot.shard.cap * sizeof(uint32_t), ctx->stream));  // (the all-gather reads whole rows)
        }
    }
    if (!slot.hdr.ptr) {
        constexpr size_t kHdrWords = 1u + (size_t)GV_EXCHANGE_MAX_RANKS * (1u + GV_EXCHANGE_MAX_ITEMS);
        GV_HIP(ctx, slot.hdr.reserve(kHdrWords));
        memset(slot.hdr.ptr, 0, kHdrWords * sizeof(uint32_t));
    }
    for (int k = 0; k < world; k++) {
        slot.room[k] = rooms[k];
        slot.travelled[k] = ctx->exchange_mode == GV_EXCHANGE_ALLGATHER ? (uint32_t)row_words : peer_entries ? 0u : 1u + rooms[k];  // (peer: known once the headers are)
        slot.counts[k] = slot.tail_words[k] = 0;
    }
    slot.cut = 0;
    slot.row_words = (uint32_t)row_words;
    slot.mode = ctx->exchange_mode;
    slot.items = table_words;
    slot.hdr_words = 1u + table_words;
    slot.item_counts.clear();
    // The shard is the last thing the context's stream does for this frame's lists; the links are the exchange stream's business.
    // The next frame's pyramid and cull go on behind the shard copy at once, while this list is still travelling. (The slot's shard
    // and rows are free: the frame that used them last is settled, which its collectives precede; and work that was enqueued on the
    // context's stream to CONSUME those rows completes in front of `produced`, which the exchange stream waits for.)
    if (!batched) {
        const uint32_t pool_id = items[0].pool_id == GV_NONE ? ctx->last_pool : items[0].pool_id;
        if (int rc = gv::copy_shard_of_pool(ctx, pool_id, items[0].view_index, slot.shard.ptr, gv::view_of(ctx, pool_id, items[0].view_index)->occupancy,
                                            items[0].index_base))
            return rc;
    } else {
        if (int rc = gv::flush_sorts(ctx))  // (recorded culls and deferred sorts of every list first)
            return rc;
        GV_HIP(ctx, slot.h_items.reserve(GV_EXCHANGE_MAX_ITEMS));
        GV_HIP(ctx, slot.d_items.reserve(GV_EXCHANGE_MAX_ITEMS));
        // the descriptors change when a pool grows or a view's buffers move — rarely: a frame whose descriptors are what this slot's
        // device table already holds skips the copy (one small DMA less in front of every frame's shard)
        std::vector<gv::ShardItem>& want = slot.items_wanted;
        want.resize(item_count);
        for (uint32_t i = 0; i < item_count; i++) {
            const gv::ViewState& vs = *gv::view_of(ctx, items[i].pool_id, items[i].view_index);
            const gv::PoolState& pool = ctx->pools[vs.pool_id];
            want[i] = gv::ShardItem{vs.visible_idx.ptr, vs.draw_count.ptr, pool.index_map_count >= vs.occupancy ? pool.d_index_map.ptr : nullptr,
                                    items[i].index_base, vs.occupancy};
        }
        if (slot.items_uploaded.size() != want.size() || memcmp(slot.items_uploaded.data(), want.data(), want.size() * sizeof(gv::ShardItem)) != 0) {
            // (h_items of this slot is free: the copy that read it last ran in front of the slot's previous frame, which is settled)
            memcpy(slot.h_items.ptr, want.data(), want.size() * sizeof(gv::ShardItem));
            slot.items_uploaded.clear();  // (not known to be there until the copy has been enqueued)
            GV_HIP(ctx, hipMemcpyAsync(slot.d_items.ptr, slot.h_items.ptr, (size_t)item_count * sizeof(gv::ShardItem), hipMemcpyHostToDevice, ctx->stream));
            slot.items_uploaded = want;
        }
        GV_HIP(ctx, gv::launch_copy_shard_batch(slot.d_items.ptr, item_count, widest_pool, slot.shard.ptr, ctx->stream));
    }
    GV_HIP(ctx, hipEventRecord(slot.produced, ctx->stream));
    GV_HIP(ctx, hipStreamWaitEvent(ctx->exchange_stream, slot.produced, 0));
    return GV_OK;
}

// ... step 2: the predicted part of every row travels.
int frame_collective(GvCtx* ctx, Slot& slot)
{
    uint32_t travel[GV_EXCHANGE_MAX_RANKS];
    for (int k = 0; k < ctx->exchange_world; k++)
        travel[k] = 1u + slot.room[k];
    return exchange_rows(ctx, slot.row_words, travel, slot.rows.ptr, "gv_exchange_visible", slot.shard.ptr, ctx->exchange_stream);
}

// ... step 2 of a peer group's frame (GV_EXCHANGE_PEER): no collective — rank k's scatter kernel stores its list into row k of every
// member's rows. Two things order the ranks, both by events through rank 0's exchange stream (4 n waits per frame, not 2 n^2):
// nobody overwrites rows a consumer of the frame before last may still read (all_produced: every member's context stream has
// passed its `produced`, which follows whatever it enqueued to consume the slot's rows), and nobody's headers are read before every
// row has arrived (all_sent). Kernel boundaries carry the data: no flag is polled on a device, nothing can wait for ever.
static_assert(sizeof(gv::PeerRows::dst) / sizeof(uint32_t*) == GV_EXCHANGE_MAX_RANKS, "PeerRows holds one row pointer per rank");
int peer_collective(GvCtx* const* ctxs, int n, unsigned which)
{
    GvCtx* hub = ctxs[0];
    Slot& hs = hub->exchange_slots[which];
    auto on = [&](int k) -> int {
        if (hipSetDevice(ctxs[k]->device) != hipSuccess)
            return ctxs[k]->fail(GV_E_HIP, "hipSetDevice(%d)", ctxs[k]->device);
        return GV_OK;
    };
    // through the hub, every exchange stream waits for every `event_of(k)`; the hub's own stream order covers event 0
    auto meet = [&](hipEvent_t Slot::*arrived, hipEvent_t all) -> int {
        if (n == 1)
            return GV_OK;
        if (int rc = on(0))
            return rc;
        for (int k = 1; k < n; k++)
            GV_HIP(hub, hipStreamWaitEvent(hub->exchange_stream, ctxs[k]->exchange_slots[which].*arrived, 0));
        GV_HIP(hub, hipEventRecord(all, hub->exchange_stream));
        for (int k = 1; k < n; k++) {
            if (int rc = on(k))
                return rc;
            GV_HIP(ctxs[k], hipStreamWaitEvent(ctxs[k]->exchange_stream, all, 0));
        }
        return GV_OK;
    };
    if (int rc = meet(&Slot::produced, hs.all_produced))
        return rc;
    for (int k = 0; k < n; k++) {
        GvCtx* ctx = ctxs[k];
        Slot& slot = ctx->exchange_slots[which];
        if (int rc = on(k))
            return rc;
        gv::PeerRows rows{};
        for (int j = 0; j < n; j++)
            rows.dst[j] = ctxs[j]->exchange_slots[which].rows.ptr + (size_t)k * slot.row_words;
        GV_HIP(ctx, gv::launch_peer_scatter(slot.shard.ptr, slot.row_words, rows, (uint32_t)n, ctx->exchange_stream));
        if (n > 1)
            GV_HIP(ctx, hipEventRecord(slot.sent, ctx->exchange_stream));
    }
    return meet(&Slot::sent, hs.all_sent);
}

void describe(const GvCtx* ctx, const Slot& slot, GvExchangeFrame* out)
{
    if (!out)
        return;
    memset(out, 0, sizeof(*out));
    out->row_words = slot.row_words;
    out->world_size = (uint32_t)ctx->exchange_world;
    out->frame = slot.frame;
    out->mode = slot.mode;
    out->items = slot.items;
    for (int k = 0; k < ctx->exchange_world; k++) {
        out->room[k] = slot.room[k];
        out->travelled_words[k] = slot.travelled[k];
    }
    if (!slot.settled)
        return;  // (sent, not handed out: a completing exchange may still move the rows)
    out->gathered_device = slot.rows.ptr;
    out->ready_event = slot.done;
    out->item_counts = slot.items ? slot.item_counts.data() : nullptr;
    out->cut_ranks = slot.cut;
    out->complete = 1;
    for (int k = 0; k < ctx->exchange_world; k++) {
        out->counts[k] = slot.counts[k];
        out->tail_words[k] = slot.tail_words[k];
    }
}

// ... step 3: the headers reach the host behind the rows.
int frame_finish(GvCtx* ctx, Slot& slot, GvExchangeFrame* out)
{
    GV_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t frame = ctx->exchange_frame;
    GV_HIP(ctx, gv::launch_exchange_headers(slot.rows.ptr, slot.row_words, (uint32_t)ctx->exchange_world, slot.hdr_words, slot.hdr.ptr,
                                            (uint32_t)(frame + 1), ctx->exchange_stream));
    GV_HIP(ctx, hipEventRecord(slot.done, ctx->exchange_stream));
    slot.frame = frame;
    slot.in_flight = true;
    slot.settled = false;
    ctx->exchange_frame = frame + 1;
    ctx->stats.exchanges++;
    describe(ctx, slot, out);
    return GV_OK;
}

// items_of(k): the lists rank k's frame carries (batched: the same table-fronted row shape on every rank)
int exchange_all(GvCtx* const* ctxs, int n, const GvExchangeItem* const* items_of, uint32_t item_count, bool batched, GvExchangeFrame* outs, bool by_group)
{
    const uint64_t frame = ctxs[0]->exchange_frame;
    for (int k = 0; k < n; k++) {
        if (int rc = usable(ctxs[k], "gv_exchange_visible", by_group))
            return rc;
        if (ctxs[k]->exchange_mode != ctxs[0]->exchange_mode)
            return ctxs[k]->fail(GV_E_ARG, "gv_exchange_visible_all: contexts[%d] travels by mode %u, contexts[0] by mode %u (gv_exchange_set_mode: the same on every rank)", k,
                                 ctxs[k]->exchange_mode, ctxs[0]->exchange_mode);
        if (ctxs[k]->exchange_frame != frame || (by_group && (ctxs[k]->exchange_rank != k || ctxs[k]->exchange_world != n)))
            return ctxs[k]->fail(GV_E_ARG, "gv_exchange_visible_all: contexts[%d] is rank %d of %d at frame %llu (expected rank %d of %d at frame %llu)", k,
                                 ctxs[k]->exchange_rank, ctxs[k]->exchange_world, (unsigned long long)ctxs[k]->exchange_frame, k, n, (unsigned long long)frame);
    }
    // the previous frame first: its headers size this one, and rows it left short are completed in front of this frame's collective
    // — at the same point of the communicator's sequence on every rank, whether or not a rank acquired it in between
    if (frame > 0)
        if (int rc = settle(ctxs, n, (unsigned)((frame - 1) & 1u)))
            return rc;
    const unsigned which = (unsigned)(frame & 1u);
    if (ctxs[0]->exchange_mode == GV_EXCHANGE_PEER) {
        // the call names the whole group, in rank order: its members store into each other's rows
        for (int k = 0; k < n; k++)
            if ((int)ctxs[k]->exchange_peers.size() != n || ctxs[k]->exchange_peers[k] != ctxs[k] || ctxs[k]->exchange_peers != ctxs[0]->exchange_peers)
                return ctxs[k]->fail(GV_E_ARG, "gv_exchange_visible_all: contexts[%d] is not rank %d of the peer group of contexts[0] (%d members)", k, k,
                                     (int)ctxs[0]->exchange_peers.size());
        size_t widest_shard = 0;  // rows as wide as the largest shard any member can produce: nothing to predict, nothing ever short
        for (int k = 0; k < n; k++) {
            size_t list_words = 0;
            uint32_t widest_pool = 0;
            if (int rc = shard_need(ctxs[k], items_of[k], item_count, &list_words, &widest_pool))
                return rc;
            widest_shard = std::max(widest_shard, list_words + (batched ? item_count : 0u));
        }
        if (widest_shard > 0xFFFFFC00u)
            return ctxs[0]->fail(GV_E_ARG, "gv_exchange_visible_all: a frame of %zu words per rank", widest_shard);
        const uint32_t entries = std::max<uint32_t>((uint32_t)widest_shard, 3u);
        for (int k = 0; k < n; k++)
            if (int rc = frame_stage(ctxs[k], items_of[k], item_count, batched, ctxs[k]->exchange_slots[which], entries))
                return rc;
        if (int rc = peer_collective(ctxs, n, which))
            return rc;
        for (int k = 0; k < n; k++)
            if (int rc = frame_finish(ctxs[k], ctxs[k]->exchange_slots[which], outs ? outs + k : nullptr))
                return rc;
        return GV_OK;
    }
    for (int k = 0; k < n; k++)
        if (int rc = frame_stage(ctxs[k], items_of[k], item_count, batched, ctxs[k]->exchange_slots[which]))
            return rc;
    int rc = GV_OK;
    Rccl& r = rccl();
    (void)r.GroupStart();
    for (int k = 0; k < n && rc == GV_OK; k++) {
        if (hipSetDevice(ctxs[k]->device) != hipSuccess)
            rc = ctxs[k]->fail(GV_E_HIP, "hipSetDevice(%d)", ctxs[k]->device);
        else
            rc = frame_collective(ctxs[k], ctxs[k]->exchange_slots[which]);
    }
    const int erc = r.GroupEnd();
    if (rc == GV_OK && erc != 0)
        rc = ctxs[0]->fail(GV_E_RCCL, "gv_exchange_visible: ncclGroupEnd: %s", r.GetErrorString(erc));
    if (rc == GV_E_RCCL) {  // some ranks' collectives are queued, others' are not: nothing later on this communicator can match
        const std::string why = ctxs[0]->error;
        for (int k = 0; k < n; k++)
            if (!ctxs[k]->exchange_broken)
                (void)give_up(ctxs[k], rc, (ctxs[k]->error.empty() ? why : ctxs[k]->error).c_str());
        return rc;
    }
    for (int k = 0; k < n && rc == GV_OK; k++)
        rc = frame_finish(ctxs[k], ctxs[k]->exchange_slots[which], outs ? outs + k : nullptr);
    return rc;
}

int visible_all(GvCtx* const* ctxs, int n, uint32_t pool_id, const uint32_t* views, const uint32_t* bases, GvExchangeFrame* outs, bool by_group)
{
    GvExchangeItem single[GV_EXCHANGE_MAX_RANKS];
    const GvExchangeItem* items_of[GV_EXCHANGE_MAX_RANKS];
    for (int k = 0; k < n; k++) {
        single[k] = GvExchangeItem{pool_id, views[k], bases ? bases[k] : 0u};
        items_of[k] = &single[k];
    }
    return exchange_all(ctxs, n, items_of, 1, false, outs, by_group);
}

int views_all(GvCtx* const* ctxs, int n, const GvExchangeItem* items, uint32_t item_count, uint32_t flags, GvExchangeFrame* outs, bool by_group)
{
    if (!items || !outs || flags || item_count == 0 || item_count > GV_EXCHANGE_MAX_ITEMS)
        return ctxs[0]->fail(GV_E_ARG, "gv_exchange_views: NULL items / frames, flags 0x%x (none are defined) or %u items (1 .. %u)", flags, item_count,
                             GV_EXCHANGE_MAX_ITEMS);
    for (uint32_t i = 0; i < item_count; i++)
        if (items[i].pool_id >= GV_MAX_POOLS || items[i].view_index >= GV_MAX_VIEWS)
            return ctxs[0]->fail(GV_E_ARG, "gv_exchange_views: item %u names pool %u view %u", i, items[i].pool_id, items[i].view_index);
    const GvExchangeItem* items_of[GV_EXCHANGE_MAX_RANKS];
    for (int k = 0; k < n; k++)
        items_of[k] = items;
    return exchange_all(ctxs, n, items_of, item_count, true, outs, by_group);
}

int acquire_all(GvCtx* const* ctxs, int n, uint64_t frame, GvExchangeFrame* outs, bool by_group)
{
    const unsigned which = (unsigned)(frame & 1u);
    for (int k = 0; k < n; k++) {
        GvCtx* ctx = ctxs[k];
        if (int rc = usable(ctx, "gv_exchange_acquire", by_group))
            return rc;
        if (frame >= ctx->exchange_frame || frame + 2 < ctx->exchange_frame || ctx->exchange_slots[which].frame != frame)
            return ctx->fail(GV_E_ARG, "gv_exchange_acquire: frame %llu is not one of the last two exchanged (next: %llu)", (unsigned long long)frame,
                             (unsigned long long)ctx->exchange_frame);
    }
    if (int rc = settle(ctxs, n, which))
        return rc;
    for (int k = 0; k < n; k++) {
        GvCtx* ctx = ctxs[k];
        GV_HIP(ctx, hipSetDevice(ctx->device));
        GV_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->exchange_slots[which].done, 0));
        describe(ctx, ctx->exchange_slots[which], outs ? outs + k : nullptr);
    }
    return GV_OK;
}

int all_args(GvCtx* const* contexts, int world_size)
{
    if (!contexts || world_size < 1 || world_size > (int)GV_EXCHANGE_MAX_RANKS)
        return GV_E_ARG;
    for (int k = 0; k < world_size; k++)
        if (!contexts[k])
            return GV_E_ARG;
    return GV_OK;
}

}  // namespace

extern "C" {

int gv_exchange_shards(GvCtx* ctx, uint32_t view_index, uint32_t capacity, const uint32_t* capacities, uint32_t index_base,
                       void* gathered_device)
{
    if (!ctx)
        return GV_E_ARG;
    if (int rc = usable(ctx, "gv_exchange_shards", false))
        return rc;
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

int gv_exchange_visible(GvCtx* ctx, uint32_t view_index, uint32_t index_base, uint32_t flags, GvExchangeFrame* out)
{
    if (!ctx)
        return GV_E_ARG;
    if (!out || flags)
        return ctx->fail(GV_E_ARG, "gv_exchange_visible: NULL frame or flags 0x%x (none are defined)", flags);
    return visible_all(&ctx, 1, GV_NONE, &view_index, &index_base, out, false);
}

int gv_pool_exchange_visible(GvCtx* ctx, uint32_t pool_id, uint32_t view_index, uint32_t index_base, uint32_t flags, GvExchangeFrame* out)
{
    if (!ctx)
        return GV_E_ARG;
    if (!out || flags || pool_id >= GV_MAX_POOLS)
        return ctx->fail(GV_E_ARG, "gv_pool_exchange_visible: NULL frame, flags 0x%x (none are defined) or pool %u", flags, pool_id);
    return visible_all(&ctx, 1, pool_id, &view_index, &index_base, out, false);
}

int gv_exchange_visible_all(GvCtx* const* contexts, int world_size, const uint32_t* view_indices, const uint32_t* index_bases, uint32_t flags,
                            GvExchangeFrame* frames)
{
    if (int rc = all_args(contexts, world_size))
        return rc;
    if (!view_indices || !frames || flags)
        return contexts[0]->fail(GV_E_ARG, "gv_exchange_visible_all: NULL view indices / frames, or flags 0x%x (none are defined)", flags);
    return visible_all(contexts, world_size, GV_NONE, view_indices, index_bases, frames, true);
}

int gv_pool_exchange_visible_all(GvCtx* const* contexts, int world_size, uint32_t pool_id, const uint32_t* view_indices, const uint32_t* index_bases,
                                 uint32_t flags, GvExchangeFrame* frames)
{
    if (int rc = all_args(contexts, world_size))
        return rc;
    if (!view_indices || !frames || flags || pool_id >= GV_MAX_POOLS)
        return contexts[0]->fail(GV_E_ARG, "gv_pool_exchange_visible_all: NULL view indices / frames, flags 0x%x (none are defined) or pool %u", flags, pool_id);
    return visible_all(contexts, world_size, pool_id, view_indices, index_bases, frames, true);
}

int gv_exchange_views(GvCtx* ctx, const GvExchangeItem* items, uint32_t item_count, uint32_t flags, GvExchangeFrame* out)
{
    if (!ctx)
        return GV_E_ARG;
    return views_all(&ctx, 1, items, item_count, flags, out, false);
}

int gv_exchange_views_all(GvCtx* const* contexts, int world_size, const GvExchangeItem* items, uint32_t item_count, uint32_t flags,
                          GvExchangeFrame* frames)
{
    if (int rc = all_args(contexts, world_size))
        return rc;
    return views_all(contexts, world_size, items, item_count, flags, frames, true);
}

int gv_exchange_acquire(GvCtx* ctx, uint64_t frame, GvExchangeFrame* out)
{
    if (!ctx)
        return GV_E_ARG;
    return acquire_all(&ctx, 1, frame, out, false);
}

int gv_exchange_acquire_all(GvCtx* const* contexts, int world_size, uint64_t frame, GvExchangeFrame* frames)
{
    if (int rc = all_args(contexts, world_size))
        return rc;
    return acquire_all(contexts, world_size, frame, frames, true);
}

int gv_exchange_masks(GvCtx* ctx, uint32_t view_index, uint32_t word_count, void* gathered_device)
{
    if (!ctx)
        return GV_E_ARG;
    if (int rc = usable(ctx, "gv_exchange_masks", false))
        return rc;
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
    // the exchange stream first, with the bounded wait (gv_stream may be waiting for a frame's rows behind it); GV_E_TIMEOUT /
    // GV_E_RCCL: the communicator was aborted instead of drained — everything is released all the same
    const int rc = drain_exchange_stream(ctx);
    const std::string why = ctx->error;
    if (rc == GV_OK)
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    gv::exchange_release(ctx);
    if (rc != GV_OK)
        ctx->error = why;
    return rc;
}

}  // extern "C"
