// gv_exchange.cpp — the multi-GPU exchange step of SURVEY.md §8e in the C-ABI, for hosts without torch.distributed
// (a C++ engine, one process per GPU): every rank's compact visible list goes out as a fixed-capacity shard
// [draw_count, global indices ...] and all ranks gather the shards with ONE equal-size ncclAllGather enqueued on the
// context's stream — no host synchronisation; counts are read from the shard headers. Same wire format as
// garden_amd/multi.py::VisibleListExchange (which does this through torch.distributed in bench.py).
// The node's xGMI fabric is fully connected point to point, and a ring all-gather serialises world-1 hops over it, so
// two direct patterns sit beside the all-gather for an A/B on real hardware (gv_exchange_set_mode, or the environment
// variable GV_EXCHANGE_MODE = allgather | p2p | broadcast read at gv_exchange_init): one ncclGroup of send/recv pairs
// with every peer (each shard crosses exactly one link), or one ncclBroadcast per root. Same bytes in the same place.
//
// RCCL is bound at run time (dlopen): a process that already carries an RCCL — PyTorch bundles one — keeps using that
// copy, and libgarden_vis.so has no link-time dependency on it.
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
        for (const char* name : {"librccl.so", "librccl.so.1"}) {  // an already loaded copy first
            h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (h)
                break;
        }
        if (!h)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (h)
                    break;
            }
        if (!h) {
            r.why = std::string("librccl not loadable: ") + (dlerror() ? dlerror() : "?");
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
    if (!unique_id || world_size < 1 || rank < 0 || rank >= world_size)
        return ctx->fail(GV_E_ARG, "gv_exchange_init: bad rank %d / world %d", rank, world_size);
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

// ctx->d_shard[0 .. words) of every rank into rows [rank * words ...) of gathered_device, by the configured pattern
static int exchange_rows(GvCtx* ctx, size_t words, void* gathered_device, const char* what)
{
    Rccl& r = rccl();
    uint32_t* rows = static_cast<uint32_t*>(gathered_device);
    const int me = ctx->exchange_rank, world = ctx->exchange_world;
    if (ctx->exchange_mode == GV_EXCHANGE_ALLGATHER) {
        const int nrc = r.AllGather(ctx->d_shard.ptr, gathered_device, words, kNcclUint32, ctx->exchange_comm, ctx->stream);
        if (nrc != 0)
            return ctx->fail(GV_E_RCCL, "%s: ncclAllGather: %s", what, r.GetErrorString(nrc));
        return GV_OK;
    }
    // the direct forms place this rank's own row with a device copy; the peers' rows arrive over the links
    GV_HIP(ctx, hipMemcpyAsync(rows + (size_t)me * words, ctx->d_shard.ptr, words * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    int nrc = r.GroupStart();
    if (ctx->exchange_mode == GV_EXCHANGE_P2P) {
        // one send/recv pair per peer inside one group: every shard crosses exactly one xGMI link, all links at once
        for (int d = 1; d < world && nrc == 0; d++) {
            const int to = (me + d) % world, from = (me - d + world) % world;
            nrc = r.Send(ctx->d_shard.ptr, words, kNcclUint32, to, ctx->exchange_comm, ctx->stream);
            if (nrc == 0)
                nrc = r.Recv(rows + (size_t)from * words, words, kNcclUint32, from, ctx->exchange_comm, ctx->stream);
        }
    } else {
        for (int root = 0; root < world && nrc == 0; root++)
            nrc = r.Broadcast(root == me ? ctx->d_shard.ptr : rows + (size_t)root * words, rows + (size_t)root * words, words,
                              kNcclUint32, root, ctx->exchange_comm, ctx->stream);
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

int gv_exchange_shards(GvCtx* ctx, uint32_t view_index, uint32_t capacity, uint32_t index_base, void* gathered_device)
{
    if (!ctx)
        return GV_E_ARG;
    if (!ctx->exchange_comm)
        return ctx->fail(GV_E_STATE, "gv_exchange_shards: gv_exchange_init has not run");
    if (!gathered_device || capacity == 0)
        return ctx->fail(GV_E_ARG, "gv_exchange_shards: NULL buffer or zero capacity");
    GV_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = reserve_shard(ctx, (size_t)capacity + 1))
        return rc;
    if (int rc = gv_results_copy_shard_device(ctx, view_index, ctx->d_shard.ptr, capacity, index_base))
        return rc;
    return exchange_rows(ctx, (size_t)capacity + 1, gathered_device, "gv_exchange_shards");
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
    return exchange_rows(ctx, (size_t)word_count + 1, gathered_device, "gv_exchange_masks");
}

int gv_exchange_shutdown(GvCtx* ctx)
{
    if (!ctx)
        return GV_E_ARG;
    GV_HIP(ctx, hipSetDevice(ctx->device));
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    gv::exchange_release(ctx);
    return GV_OK;
}

}  // extern "C"
