// rccl_stub.cpp — TEST-ONLY transport: the ten RCCL entry points gv_exchange.cpp binds (garden_amd/csrc/gv_exchange.cpp),
// implemented over POSIX shared memory between the processes of ONE machine, so that the library's exchange logic — per-rank
// sizes, rows sized from earlier frames' headers, cut rows, the three travel patterns — runs with N ranks on a box with one
// GPU (RCCL refuses two ranks on one device) and, against tests/cpp/hip_stub, with no GPU at all. Selected with
// GV_RCCL_LIBRARY=<this library>; never linked into, or loaded by default by, the product. Nothing here is fast: every call
// synchronises the stream, stages through the host and meets the other ranks at a barrier.
//
// Wire: the unique id carries the name of a shared-memory object; rank 0 of ncclCommInitRank sizes and initialises it. A
// "round" moves, for every ordered pair (src, dst), at most kSlotBytes of one message through the pair's slot; all ranks
// first agree on the number of rounds (the largest message of the call, max over ranks).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

#include <hip/hip_runtime.h>

namespace {

constexpr int kMaxRanks = 16;
constexpr size_t kSlotBytes = 256u << 10;
constexpr uint32_t kReady = 0x52434331u;

struct Header {
    std::atomic<uint32_t> state;       // 0 fresh, 1 being initialised, kReady
    std::atomic<uint32_t> arrived;     // barrier: arrivals of the current generation
    std::atomic<uint32_t> generation;  // barrier generation
    std::atomic<uint32_t> failed;      // some rank gave up: everybody leaves their barriers with an error
    uint32_t world;
    uint64_t want[kMaxRanks];          // per rank: its largest message of the current call
    uint64_t sent[kMaxRanks][kMaxRanks];  // [from][to]: bytes of the current call's message (what the receiver must expect too)
};

struct Comm {
    Header* hdr = nullptr;
    uint8_t* slots = nullptr;  // [world][world][kSlotBytes]
    size_t bytes = 0;
    int rank = 0, world = 1;
    char name[64] = {};
    std::vector<uint8_t> bounce;
};

struct Message {
    bool send;
    void* buf;
    size_t bytes;
    int peer;
    hipStream_t stream;
};
struct Pending {
    Comm* comm = nullptr;
    std::vector<Message> msgs;
};
thread_local int group_depth = 0;
thread_local Pending pending;

size_t size_of(int datatype)
{
    switch (datatype) {  // ncclDataType_t
    case 0: case 1: return 1;                  // int8, uint8
    case 2: case 3: case 7: return 4;          // int32, uint32, float32
    case 4: case 5: case 8: return 8;          // int64, uint64, float64
    case 6: case 9: return 2;                  // float16, bfloat16
    default: return 0;
    }
}

uint8_t* slot(Comm* c, int src, int dst) { return c->slots + ((size_t)src * c->world + dst) * kSlotBytes; }

bool barrier(Comm* c)
{
    Header* h = c->hdr;
    const uint32_t gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->generation.store(gen + 1, std::memory_order_release);
        return h->failed.load() == 0;
    }
    const time_t t0 = time(nullptr);
    for (uint64_t spins = 0; h->generation.load(std::memory_order_acquire) == gen; spins++) {
        if (h->failed.load())
            return false;
        if ((spins & 0xFFF) == 0xFFF) {
            if (time(nullptr) - t0 > 120) {  // a rank died: do not hang the test tier
                h->failed.store(1);
                return false;
            }
            usleep(50);
        }
    }
    return h->failed.load() == 0;
}

// all queued messages of this rank, matched pair by pair with the peers' (one message per ordered pair and call)
int run(Comm* c, std::vector<Message>& msgs)
{
    if (msgs.empty())
        return 0;
    // RCCL_STUB_HANG_AT=n: the n-th transfer call of this process never returns (what a collective looks like when a peer never
    // enters it) — for the tests of what the callers do about that (bench.py's exchange watchdog)
    static const long hang_at = getenv("RCCL_STUB_HANG_AT") ? atol(getenv("RCCL_STUB_HANG_AT")) : 0;
    static std::atomic<long> calls{0};
    if (hang_at > 0 && ++calls == hang_at)
        for (;;)
            sleep(1);
    uint64_t mine = 0;
    bool seen_send[kMaxRanks] = {}, seen_recv[kMaxRanks] = {};
    for (const Message& m : msgs) {
        if (m.peer < 0 || m.peer >= c->world)
            return 4;  // ncclInvalidArgument
        bool& seen = m.send ? seen_send[m.peer] : seen_recv[m.peer];
        if (seen)
            return 5;  // ncclInvalidUsage: this transport carries one message per pair and call
        seen = true;
        mine = m.bytes > mine ? m.bytes : mine;
        if (hipStreamSynchronize(m.stream) != hipSuccess)
            return 1;
    }
    c->hdr->want[c->rank] = mine;
    for (int r = 0; r < c->world; r++)
        c->hdr->sent[c->rank][r] = 0;
    for (const Message& m : msgs)
        if (m.send)
            c->hdr->sent[c->rank][m.peer] = m.bytes;
    if (!barrier(c))
        return 3;
    // RCCL would hang or corrupt memory when the two ends of a transfer disagree about its size: here it is an error on the
    // receiving rank (ranks that sized a frame's rows differently are exactly what the exchange tests look for)
    bool sizes_agree = true;
    for (const Message& m : msgs)
        if (!m.send && c->hdr->sent[m.peer][c->rank] != m.bytes) {
            std::fprintf(stderr, "rccl_stub: rank %d expects %zu bytes from rank %d, which sends %llu\n", c->rank, m.bytes, m.peer,
                         (unsigned long long)c->hdr->sent[m.peer][c->rank]);
            sizes_agree = false;
        }
    uint64_t most = 0;
    for (int r = 0; r < c->world; r++)
        most = c->hdr->want[r] > most ? c->hdr->want[r] : most;
    const uint64_t rounds = (most + kSlotBytes - 1) / kSlotBytes;
    for (uint64_t round = 0; round < rounds; round++) {
        const size_t at = (size_t)round * kSlotBytes;
        for (const Message& m : msgs)
            if (m.send && at < m.bytes) {
                const size_t n = m.bytes - at < kSlotBytes ? m.bytes - at : kSlotBytes;
                if (hipMemcpy(slot(c, c->rank, m.peer), (const uint8_t*)m.buf + at, n, hipMemcpyDefault) != hipSuccess)
                    return 1;
            }
        if (!barrier(c))
            return 3;
        for (const Message& m : msgs)
            if (!m.send && at < m.bytes) {
                const size_t n = m.bytes - at < kSlotBytes ? m.bytes - at : kSlotBytes;
                if (hipMemcpy((uint8_t*)m.buf + at, slot(c, m.peer, c->rank), n, hipMemcpyDefault) != hipSuccess)
                    return 1;
            }
        if (!barrier(c))
            return 3;
    }
    // (a final barrier is implied: every round ends with one, and a call without rounds moved nothing)
    msgs.clear();
    return sizes_agree ? 0 : 4;  // ncclInvalidArgument, after the rounds: nobody is left waiting in a barrier
}

int submit(Comm* c, Message m)
{
    if (group_depth > 0) {
        if (pending.comm && pending.comm != c)
            return 5;
        pending.comm = c;
        pending.msgs.push_back(m);
        return 0;
    }
    std::vector<Message> one{m};
    return run(c, one);
}

}  // namespace

extern "C" {

struct ncclUniqueId {
    char internal[128];
};

int ncclGetUniqueId(ncclUniqueId* id)
{
    memset(id, 0, sizeof(*id));
    static std::atomic<uint32_t> serial{0};
    snprintf(id->internal, sizeof(id->internal), "/gv_rccl_stub_%d_%u_%ld", (int)getpid(), serial.fetch_add(1), (long)time(nullptr));
    return 0;
}

int ncclCommInitRank(void** comm, int world, ncclUniqueId id, int rank)
{
    if (!comm || world < 1 || world > kMaxRanks || rank < 0 || rank >= world)
        return 4;
    Comm* c = new Comm();
    c->rank = rank;
    c->world = world;
    memcpy(c->name, id.internal, sizeof(c->name) - 1);
    c->bytes = 4096 + (size_t)world * world * kSlotBytes;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) {
        delete c;
        return 2;  // ncclSystemError
    }
    void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        delete c;
        return 2;
    }
    c->hdr = static_cast<Header*>(p);
    c->slots = static_cast<uint8_t*>(p) + 4096;
    uint32_t fresh = 0;
    if (c->hdr->state.compare_exchange_strong(fresh, 1)) {  // (a new object is zero-filled) the first rank to arrive sets it up
        c->hdr->world = (uint32_t)world;
        c->hdr->arrived.store(0);
        c->hdr->generation.store(0);
        c->hdr->failed.store(0);
        c->hdr->state.store(kReady, std::memory_order_release);
    } else {
        const time_t t0 = time(nullptr);
        while (c->hdr->state.load(std::memory_order_acquire) != kReady) {
            if (time(nullptr) - t0 > 120) {
                delete c;
                return 2;
            }
            usleep(100);
        }
    }
    if (!barrier(c)) {  // everybody is attached: the name can go
        delete c;
        return 3;
    }
    if (rank == 0)
        shm_unlink(c->name);
    *comm = c;
    return 0;
}

int ncclCommDestroy(void* comm)
{
    Comm* c = static_cast<Comm*>(comm);
    if (!c)
        return 4;
    munmap(c->hdr, c->bytes);
    delete c;
    return 0;
}

int ncclGroupStart()
{
    group_depth++;
    return 0;
}

int ncclGroupEnd()
{
    if (group_depth <= 0)
        return 5;
    if (--group_depth > 0 || !pending.comm)
        return 0;
    Comm* c = pending.comm;
    pending.comm = nullptr;
    std::vector<Message> msgs;
    msgs.swap(pending.msgs);
    return run(c, msgs);
}

int ncclSend(const void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t stream)
{
    const size_t w = size_of(datatype);
    if (!comm || !w)
        return 4;
    return submit(static_cast<Comm*>(comm), Message{true, const_cast<void*>(buf), count * w, peer, stream});
}

int ncclRecv(void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t stream)
{
    const size_t w = size_of(datatype);
    if (!comm || !w)
        return 4;
    return submit(static_cast<Comm*>(comm), Message{false, buf, count * w, peer, stream});
}

// the collectives run as their own call even inside a group (a group of one broadcast per root would otherwise put several
// messages on one pair): ranks issue them in the same order, which is all this transport needs
int ncclAllGather(const void* send, void* recv, size_t count, int datatype, void* comm, hipStream_t stream)
{
    Comm* c = static_cast<Comm*>(comm);
    const size_t w = size_of(datatype);
    if (!c || !w)
        return 4;
    std::vector<Message> msgs;
    for (int r = 0; r < c->world; r++) {
        msgs.push_back(Message{true, const_cast<void*>(send), count * w, r, stream});
        msgs.push_back(Message{false, (uint8_t*)recv + (size_t)r * count * w, count * w, r, stream});
    }
    return run(c, msgs);
}

int ncclBroadcast(const void* send, void* recv, size_t count, int datatype, int root, void* comm, hipStream_t stream)
{
    Comm* c = static_cast<Comm*>(comm);
    const size_t w = size_of(datatype);
    if (!c || !w || root < 0 || root >= c->world)
        return 4;
    std::vector<Message> msgs;
    if (c->rank == root) {
        for (int r = 0; r < c->world; r++)
            if (r != root)
                msgs.push_back(Message{true, const_cast<void*>(send), count * w, r, stream});
        if (send != recv && hipMemcpy(recv, send, count * w, hipMemcpyDefault) != hipSuccess)
            return 1;
        if (msgs.empty())
            return 0;
    } else {
        msgs.push_back(Message{false, recv, count * w, root, stream});
    }
    return run(c, msgs);  // (every rank of the communicator takes part in a broadcast: the round count is agreed by all)
}

const char* ncclGetErrorString(int code)
{
    switch (code) {
    case 0: return "no error";
    case 1: return "unhandled HIP error (rccl_stub)";
    case 2: return "unhandled system error (rccl_stub: shared memory)";
    case 3: return "internal error (rccl_stub: a rank left or a barrier timed out)";
    case 4: return "invalid argument (rccl_stub)";
    case 5: return "invalid usage (rccl_stub)";
    default: return "unknown result code (rccl_stub)";
    }
}

}  // extern "C"
