// rccl_stub.cpp — TEST-ONLY transport: the RCCL entry points gv_exchange.cpp binds (garden_amd/csrc/gv_exchange.cpp), implemented
// over POSIX shared memory between the ranks of ONE machine, so that the library's exchange logic — per-rank sizes, rows predicted
// from the previous frame's headers, short rows completed, the three travel patterns, the events that order two streams — runs with N
// ranks on a box with one GPU (RCCL refuses two ranks on one device) and, against tests/cpp/hip_stub, with no GPU at all. Selected
// with GV_RCCL_LIBRARY=<this library>; never linked into, or loaded by default by, the product.
//
// Two transports behind the same entry points, agreed by all ranks when the communicator is made:
//   * DEVICE (one rank per process, built with hipcc): like RCCL itself, a call only ENQUEUES — one kernel on the caller's stream
//     that stages its messages through the shared segment (registered with HIP), meets the other ranks' kernels at barriers made of
//     system-scope words, and leaves; the host never waits. What the library orders with events and streams around a collective
//     (the shard's `produced`, the rows' `done`, the two alternating slots, the headers that reach pinned memory behind the rows) is
//     then really asynchronous with several ranks — an ordering mistake reads stale rows here as it would on xGMI.
//   * HOST (ranks that share a process — one thread driving N contexts, or N threads — and the CPU build): a worker thread per
//     communicator synchronises the stream, stages through the host and meets the other ranks at barriers; the call returns when
//     the transfer is done. ncclGroupStart / ncclGroupEnd hand the calls of all the thread's communicators to their workers at
//     once, which is what lets ONE thread drive N ranks (the group semantics gv_exchange_*_all relies on).
//
// Wire: the unique id carries the name of a shared-memory object; the first rank to arrive initialises it. A "round" moves, for
// every ordered pair (src, dst), at most kSlotBytes of one message through the pair's slot; all ranks first agree on the number
// of rounds (the largest message of the call, max over ranks). One message per ordered pair and call.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

namespace {

constexpr int kMaxRanks = 16;
constexpr int kMaxMsgs = 2 * kMaxRanks;
constexpr size_t kSlotBytes = 256u << 10;
constexpr size_t kHeaderBytes = 8192;
constexpr uint32_t kReady = 0x52434332u;

struct alignas(64) Tick {
    unsigned long long v;
};
struct Header {             // plain words: the host uses the __atomic builtins on them, the kernels system-scope atomics
    uint32_t state;         // 0 fresh, 1 being initialised, kReady
    uint32_t arrived;       // host barrier: arrivals of the current generation
    uint32_t generation;    // host barrier generation
    uint32_t failed;        // some rank gave up (or aborted): everybody leaves their barriers with an error
    uint32_t world;
    uint32_t device_votes;  // ranks able to run the device transport: all of them, or the communicator stages through the host
    unsigned long long want[kMaxRanks];            // per rank: its largest message of the current call
    unsigned long long sent[kMaxRanks][kMaxRanks];  // [from][to]: bytes of the current call's message (what the receiver must expect too)
    Tick tick[kMaxRanks];   // device barrier: rank r's kernels count their barriers here; only rank r writes tick[r]
};
static_assert(sizeof(Header) <= kHeaderBytes, "header page");

struct Message {
    bool send;
    void* buf;
    size_t bytes;
    int peer;
};
struct LocalCopy {
    void* dst;
    const void* src;
    size_t bytes;
};
struct Op {  // one call: a collective, or the send / recv pairs of one group
    std::vector<Message> msgs;
    std::vector<LocalCopy> copies;
    hipStream_t stream = nullptr;
};
struct Job {
    std::vector<Op> ops;
    int result = 0;
    bool done = false;
};

struct Comm {
    Header* hdr = nullptr;
    uint8_t* slots = nullptr;  // [world][world][kSlotBytes]
    size_t bytes = 0;
    int rank = 0, world = 1, device = 0;
    char name[64] = {};
    // device transport
    bool on_device = false, registered = false;
    Header* d_hdr = nullptr;
    uint8_t* d_slots = nullptr;
    unsigned long long* d_tick = nullptr;  // the kernels' own barrier count (device memory)
    uint32_t* h_err = nullptr;             // pinned: a kernel's verdict (3 a rank left / timed out, 4 sizes disagree), sticky
    // host transport
    std::thread worker;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::shared_ptr<Job>> queue;
    bool quit = false;
    std::atomic<bool> aborted{false};
};

std::atomic<int> g_live_comms{0};
std::atomic<long> g_calls{0};
long hang_at()
{
    // RCCL_STUB_HANG_AT=n: the n-th transfer of this process never completes (what a collective looks like when a peer never enters
    // it) — for the tests of what the callers do about that (bounded waits, bench.py's exchange watchdog)
    static const long n = getenv("RCCL_STUB_HANG_AT") ? atol(getenv("RCCL_STUB_HANG_AT")) : 0;
    return n;
}

uint32_t load32(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
void store32(uint32_t* p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }

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
    const uint32_t gen = load32(&h->generation);
    if (__atomic_add_fetch(&h->arrived, 1u, __ATOMIC_ACQ_REL) == (uint32_t)c->world) {
        __atomic_store_n(&h->arrived, 0u, __ATOMIC_RELAXED);
        store32(&h->generation, gen + 1);
        return load32(&h->failed) == 0;
    }
    const time_t t0 = time(nullptr);
    for (uint64_t spins = 0; load32(&h->generation) == gen; spins++) {
        if (load32(&h->failed))
            return false;
        if ((spins & 0xFFF) == 0xFFF) {
            if (time(nullptr) - t0 > 120) {  // a rank died: do not hang the test tier
                store32(&h->failed, 1);
                return false;
            }
            usleep(50);
        }
    }
    return load32(&h->failed) == 0;
}

int check(const Op& op, int world)
{
    bool seen_send[kMaxRanks] = {}, seen_recv[kMaxRanks] = {};
    if (op.msgs.size() > (size_t)kMaxMsgs)
        return 5;
    for (const Message& m : op.msgs) {
        if (m.peer < 0 || m.peer >= world)
            return 4;  // ncclInvalidArgument
        bool& seen = m.send ? seen_send[m.peer] : seen_recv[m.peer];
        if (seen)
            return 5;  // ncclInvalidUsage: this transport carries one message per pair and call
        seen = true;
    }
    return 0;
}

// ---- host transport: one op, staged through the host by this rank's worker ----
int run_on_host(Comm* c, const Op& op)
{
    if (hang_at() > 0 && ++g_calls == hang_at()) {
        while (!c->aborted.load() && !load32(&c->hdr->failed))
            usleep(1000);
        return 3;
    }
    if (int rc = check(op, c->world))
        return rc;
    if (hipStreamSynchronize(op.stream) != hipSuccess)
        return 1;
    for (const LocalCopy& k : op.copies)
        if (k.bytes && hipMemcpy(k.dst, k.src, k.bytes, hipMemcpyDefault) != hipSuccess)
            return 1;
    if (op.msgs.empty())
        return 0;  // (a broadcast in a world of one)
    unsigned long long mine = 0;
    for (const Message& m : op.msgs)
        mine = m.bytes > mine ? m.bytes : mine;
    c->hdr->want[c->rank] = mine;
    for (int r = 0; r < c->world; r++)
        c->hdr->sent[c->rank][r] = 0;
    for (const Message& m : op.msgs)
        if (m.send)
            c->hdr->sent[c->rank][m.peer] = m.bytes;
    if (!barrier(c))
        return 3;
    // RCCL would hang or corrupt memory when the two ends of a transfer disagree about its size: here it is an error on the
    // receiving rank (ranks that sized a frame's rows differently are exactly what the exchange tests look for)
    bool sizes_agree = true;
    for (const Message& m : op.msgs)
        if (!m.send && c->hdr->sent[m.peer][c->rank] != m.bytes) {
            std::fprintf(stderr, "rccl_stub: rank %d expects %zu bytes from rank %d, which sends %llu\n", c->rank, m.bytes, m.peer,
                         c->hdr->sent[m.peer][c->rank]);
            sizes_agree = false;
        }
    unsigned long long most = 0;
    for (int r = 0; r < c->world; r++)
        most = c->hdr->want[r] > most ? c->hdr->want[r] : most;
    if (!barrier(c))  // (everybody has read want / sent: the next call may overwrite them)
        return 3;
    const unsigned long long rounds = (most + kSlotBytes - 1) / kSlotBytes;
    for (unsigned long long round = 0; round < rounds; round++) {
        const size_t at = (size_t)round * kSlotBytes;
        for (const Message& m : op.msgs)
            if (m.send && at < m.bytes) {
                const size_t n = m.bytes - at < kSlotBytes ? m.bytes - at : kSlotBytes;
                if (hipMemcpy(slot(c, c->rank, m.peer), (const uint8_t*)m.buf + at, n, hipMemcpyDefault) != hipSuccess)
                    return 1;
            }
        if (!barrier(c))
            return 3;
        for (const Message& m : op.msgs)
            if (!m.send && at < m.bytes) {
                const size_t n = m.bytes - at < kSlotBytes ? m.bytes - at : kSlotBytes;
                if (hipMemcpy((uint8_t*)m.buf + at, slot(c, m.peer, c->rank), n, hipMemcpyDefault) != hipSuccess)
                    return 1;
            }
        if (!barrier(c))
            return 3;
    }
    return sizes_agree ? 0 : 4;  // ncclInvalidArgument, after the rounds: nobody is left waiting in a barrier
}

void worker_main(Comm* c)
{
    (void)hipSetDevice(c->device);
    for (;;) {
        std::shared_ptr<Job> job;
        {
            std::unique_lock<std::mutex> lock(c->m);
            c->cv.wait(lock, [&] { return c->quit || !c->queue.empty(); });
            if (c->queue.empty())
                return;
            job = c->queue.front();
            c->queue.pop_front();
        }
        int result = 0;
        for (const Op& op : job->ops) {
            const int rc = run_on_host(c, op);
            if (rc && !result)
                result = rc;
            if (rc == 1 || rc == 3)
                break;  // (a size disagreement still runs the remaining calls: the other ranks do)
        }
        {
            std::lock_guard<std::mutex> lock(c->m);
            job->result = result;
            job->done = true;
        }
        c->cv.notify_all();
    }
}

std::shared_ptr<Job> hand_to_worker(Comm* c, std::vector<Op> ops)
{
    auto job = std::make_shared<Job>();
    job->ops = std::move(ops);
    {
        std::lock_guard<std::mutex> lock(c->m);
        c->queue.push_back(job);
    }
    c->cv.notify_all();
    return job;
}

int wait_for(Comm* c, const std::shared_ptr<Job>& job)
{
    std::unique_lock<std::mutex> lock(c->m);
    c->cv.wait(lock, [&] { return job->done; });
    return job->result;
}

// ---- device transport: one op = one kernel on the caller's stream ----
#ifndef GV_HIP_STUB
struct DevMsg {
    void* buf;
    unsigned long long bytes;
    int peer, send;
};
struct DevOp {
    DevMsg msgs[kMaxMsgs];
    LocalCopy copies[2];
    int count, ncopies, hang;
};

__device__ void dev_copy(void* dst, const void* src, unsigned long long bytes)
{
    if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 3u) == 0) {
        uint32_t* d = (uint32_t*)dst;
        const uint32_t* s = (const uint32_t*)src;
        for (unsigned long long i = threadIdx.x; i < bytes / 4; i += blockDim.x)
            d[i] = s[i];
    } else {
        uint8_t* d = (uint8_t*)dst;
        const uint8_t* s = (const uint8_t*)src;
        for (unsigned long long i = threadIdx.x; i < bytes; i += blockDim.x)
            d[i] = s[i];
    }
}

// thread 0: everything this rank wrote is visible to the system, then its tick; then every rank's tick. false: a rank left.
__device__ bool dev_barrier(Header* h, int world, int me, unsigned long long t)
{
    __threadfence_system();
    __hip_atomic_store(&h->tick[me].v, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < world; r++)
        while (__hip_atomic_load(&h->tick[r].v, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < t) {
            if (__hip_atomic_load(&h->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))
                return false;
            if (wall_clock64() - t0 > 120ull * 100000000ull) {  // (100 MHz) a rank died: do not hold the GPU for ever
                __hip_atomic_store(&h->failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return false;
            }
            __builtin_amdgcn_s_sleep(32);
        }
    return true;
}

__global__ __launch_bounds__(256) void transport_kernel(Header* h, uint8_t* slots, unsigned long long* tick, uint32_t* err, DevOp op, int me, int world)
{
    __shared__ int ok;
    __shared__ unsigned long long rounds, t;
    for (int k = 0; k < op.ncopies; k++)
        dev_copy(op.copies[k].dst, op.copies[k].src, op.copies[k].bytes);
    if (op.count == 0)
        return;
    if (threadIdx.x == 0) {
        t = *tick;
        ok = 1;
        rounds = 0;
        if (op.hang) {  // never completes, until somebody aborts the communicator (or the bound of a barrier runs out)
            const unsigned long long t0 = wall_clock64();
            while (!__hip_atomic_load(&h->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) && wall_clock64() - t0 < 150ull * 100000000ull)
                __builtin_amdgcn_s_sleep(127);
            ok = 0;
        } else {
            unsigned long long mine = 0;
            for (int k = 0; k < op.count; k++)
                mine = op.msgs[k].bytes > mine ? op.msgs[k].bytes : mine;
            h->want[me] = mine;
            for (int r = 0; r < world; r++)
                h->sent[me][r] = 0;
            for (int k = 0; k < op.count; k++)
                if (op.msgs[k].send)
                    h->sent[me][op.msgs[k].peer] = op.msgs[k].bytes;
            ok = dev_barrier(h, world, me, ++t);
            if (ok) {
                unsigned long long most = 0;
                for (int r = 0; r < world; r++) {
                    const unsigned long long w = __hip_atomic_load(&h->want[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    most = w > most ? w : most;
                }
                for (int k = 0; k < op.count; k++)
                    if (!op.msgs[k].send &&
                        __hip_atomic_load(&h->sent[op.msgs[k].peer][me], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != op.msgs[k].bytes)
                        __hip_atomic_store(err, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // sizes disagree: reported by the next call
                rounds = (most + kSlotBytes - 1) / kSlotBytes;
                ok = dev_barrier(h, world, me, ++t);  // (everybody has read want / sent)
            }
        }
    }
    __syncthreads();
    for (unsigned long long round = 0; ok && round < rounds; round++) {
        const unsigned long long at = round * kSlotBytes;
        for (int k = 0; k < op.count; k++)
            if (op.msgs[k].send && at < op.msgs[k].bytes) {
                const unsigned long long n = op.msgs[k].bytes - at < kSlotBytes ? op.msgs[k].bytes - at : kSlotBytes;
                dev_copy(slots + ((size_t)me * world + op.msgs[k].peer) * kSlotBytes, (const uint8_t*)op.msgs[k].buf + at, n);
            }
        __syncthreads();
        if (threadIdx.x == 0)
            ok = dev_barrier(h, world, me, ++t);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // (every lane reads what the other ranks' kernels staged, not an earlier round's lines)
        if (!ok)
            break;
        for (int k = 0; k < op.count; k++)
            if (!op.msgs[k].send && at < op.msgs[k].bytes) {
                const unsigned long long n = op.msgs[k].bytes - at < kSlotBytes ? op.msgs[k].bytes - at : kSlotBytes;
                dev_copy((uint8_t*)op.msgs[k].buf + at, slots + ((size_t)op.msgs[k].peer * world + me) * kSlotBytes, n);
            }
        __syncthreads();
        if (threadIdx.x == 0)
            ok = dev_barrier(h, world, me, ++t);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *tick = t;
        if (!ok)
            __hip_atomic_store(err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int enqueue_on_device(Comm* c, const Op& op)
{
    if (int rc = check(op, c->world))
        return rc;
    if (op.copies.size() > 2)
        return 5;
    DevOp d{};
    d.count = (int)op.msgs.size();
    for (int k = 0; k < d.count; k++)
        d.msgs[k] = DevMsg{op.msgs[k].buf, (unsigned long long)op.msgs[k].bytes, op.msgs[k].peer, op.msgs[k].send ? 1 : 0};
    d.ncopies = (int)op.copies.size();
    for (int k = 0; k < d.ncopies; k++)
        d.copies[k] = op.copies[k];
    d.hang = hang_at() > 0 && ++g_calls == hang_at();
    hipLaunchKernelGGL(transport_kernel, dim3(1), dim3(256), 0, op.stream, c->d_hdr, c->d_slots, c->d_tick, c->h_err, d, c->rank, c->world);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

bool prepare_device_transport(Comm* c, void* segment)
{
    if (getenv("RCCL_STUB_HOST_TRANSPORT"))
        return false;
    if (hipHostRegister(segment, c->bytes, hipHostRegisterMapped) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    c->registered = true;
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, segment, 0) != hipSuccess || hipMalloc((void**)&c->d_tick, sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(c->d_tick, 0, sizeof(unsigned long long)) != hipSuccess || hipHostMalloc((void**)&c->h_err, sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    *c->h_err = 0;
    c->d_hdr = static_cast<Header*>(dev);
    c->d_slots = static_cast<uint8_t*>(dev) + kHeaderBytes;
    return true;
}
#else
int enqueue_on_device(Comm*, const Op&) { return 1; }
bool prepare_device_transport(Comm*, void*) { return false; }
#endif

void release_device_transport(Comm* c)
{
    if (c->registered)
        (void)hipHostUnregister(c->hdr);
    c->registered = false;
    if (c->d_tick)
        (void)hipFree(c->d_tick);
    if (c->h_err)
        (void)hipHostFree(c->h_err);
    c->d_tick = nullptr;
    c->h_err = nullptr;
}

// ---- groups: everything a thread issues between the outermost ncclGroupStart / ncclGroupEnd, per communicator ----
struct DeferredInit {
    void** out;
    int world, rank;
    char name[128];
};
struct Pending {
    std::vector<Comm*> comms;           // in order of first appearance
    std::vector<std::vector<Op>> ops;   // per communicator, in issue order
    std::vector<bool> open_pairs;       // the communicator's last op is a group of sends / recvs still being added to
    std::vector<DeferredInit> inits;
};
thread_local int group_depth = 0;
thread_local Pending pending;

std::vector<Op>& pending_ops(Comm* c, size_t* index)
{
    for (size_t k = 0; k < pending.comms.size(); k++)
        if (pending.comms[k] == c) {
            *index = k;
            return pending.ops[k];
        }
    pending.comms.push_back(c);
    pending.ops.emplace_back();
    pending.open_pairs.push_back(false);
    *index = pending.comms.size() - 1;
    return pending.ops.back();
}

int sticky_error(Comm* c)
{
    if (c->h_err) {
        const uint32_t e = __atomic_load_n(c->h_err, __ATOMIC_ACQUIRE);
        if (e)
            return (int)e;
    }
    return 0;
}

// hands the calls of every communicator to its transport; the host transport's are then waited for (all at once: the ranks
// of one thread meet each other inside them)
int dispatch(std::vector<Comm*>& comms, std::vector<std::vector<Op>>& ops)
{
    int result = 0;
    std::vector<std::pair<Comm*, std::shared_ptr<Job>>> jobs;
    for (size_t k = 0; k < comms.size(); k++) {
        Comm* c = comms[k];
        if (ops[k].empty())
            continue;
        if (c->on_device) {
            if (int e = sticky_error(c))
                result = result ? result : e;
            for (const Op& op : ops[k]) {
                const int rc = enqueue_on_device(c, op);
                if (rc && !result)
                    result = rc;
            }
        } else {
            jobs.emplace_back(c, hand_to_worker(c, std::move(ops[k])));
        }
    }
    for (auto& j : jobs) {
        const int rc = wait_for(j.first, j.second);
        if (rc && !result)
            result = rc;
    }
    return result;
}

int submit(Comm* c, Op op, bool pair_message)
{
    if (group_depth > 0) {
        size_t k = 0;
        std::vector<Op>& ops = pending_ops(c, &k);
        if (pair_message && pending.open_pairs[k] && ops.back().stream == op.stream) {
            ops.back().msgs.push_back(op.msgs[0]);
        } else {
            ops.push_back(std::move(op));
            pending.open_pairs[k] = pair_message;
        }
        return 0;
    }
    std::vector<Comm*> comms{c};
    std::vector<std::vector<Op>> ops(1);
    ops[0].push_back(std::move(op));
    return dispatch(comms, ops);
}

int init_rank(void** comm, int world, const char* name, int rank, bool shares_process)
{
    Comm* c = new Comm();
    c->rank = rank;
    c->world = world;
    (void)hipGetDevice(&c->device);
    snprintf(c->name, sizeof(c->name), "%s", name);
    c->bytes = kHeaderBytes + (size_t)world * world * kSlotBytes;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) {
        if (fd >= 0)
            close(fd);
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
    c->slots = static_cast<uint8_t*>(p) + kHeaderBytes;
    uint32_t fresh = 0;
    if (__atomic_compare_exchange_n(&c->hdr->state, &fresh, 1u, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
        c->hdr->world = (uint32_t)world;  // (a new object is zero-filled) the first rank to arrive sets it up
        store32(&c->hdr->state, kReady);
    } else {
        const time_t t0 = time(nullptr);
        while (load32(&c->hdr->state) != kReady) {
            if (time(nullptr) - t0 > 120) {
                munmap(p, c->bytes);
                delete c;
                return 2;
            }
            usleep(100);
        }
    }
    // (ranks that share a process share its hardware queues: a waiting kernel could sit in front of the one it waits for)
    const bool alone = g_live_comms.fetch_add(1) == 0;
    const bool able = alone && !shares_process && prepare_device_transport(c, p);
    if (able)
        __atomic_add_fetch(&c->hdr->device_votes, 1u, __ATOMIC_ACQ_REL);
    if (!barrier(c)) {  // everybody is attached: the name can go
        release_device_transport(c);
        munmap(p, c->bytes);
        delete c;
        g_live_comms--;
        return 3;
    }
    c->on_device = load32(&c->hdr->device_votes) == (uint32_t)world;
    if (!c->on_device)
        release_device_transport(c);
    if (rank == 0)
        shm_unlink(c->name);
    if (!c->on_device)
        c->worker = std::thread(worker_main, c);
    if (getenv("RCCL_STUB_VERBOSE") && rank == 0)
        std::fprintf(stderr, "rccl_stub: %d ranks, %s transport\n", world, c->on_device ? "device" : "host");
    *comm = c;
    return 0;
}

int destroy(Comm* c, bool abort)
{
    if (!c)
        return 4;
    if (abort) {
        c->aborted.store(true);
        store32(&c->hdr->failed, 1);  // every rank's barriers and kernels leave
    }
    if (c->worker.joinable()) {
        {
            std::lock_guard<std::mutex> lock(c->m);
            c->quit = true;
        }
        c->cv.notify_all();
        c->worker.join();
    }
    if (c->on_device)
        (void)hipDeviceSynchronize();  // (the kernels read the segment)
    release_device_transport(c);
    munmap(c->hdr, c->bytes);
    delete c;
    g_live_comms--;
    return 0;
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
    id.internal[sizeof(id.internal) - 1] = 0;
    if (group_depth > 0) {  // one thread starting several ranks: they meet each other at ncclGroupEnd
        DeferredInit d{comm, world, rank, {}};
        snprintf(d.name, sizeof(d.name), "%s", id.internal);
        pending.inits.push_back(d);
        return 0;
    }
    return init_rank(comm, world, id.internal, rank, false);
}

// what the kernels of the device transport have reported since the communicator was made (sticky), or that a rank has left
int ncclCommGetAsyncError(void* comm, int* async_error)
{
    Comm* c = static_cast<Comm*>(comm);
    if (!c || !async_error)
        return 4;
    *async_error = sticky_error(c);
    if (!*async_error && load32(&c->hdr->failed))
        *async_error = 3;
    return 0;
}

int ncclCommDestroy(void* comm) { return destroy(static_cast<Comm*>(comm), false); }
int ncclCommAbort(void* comm) { return destroy(static_cast<Comm*>(comm), true); }

int ncclGroupStart()
{
    group_depth++;
    return 0;
}

int ncclGroupEnd()
{
    if (group_depth <= 0)
        return 5;
    if (--group_depth > 0)
        return 0;
    int result = 0;
    if (!pending.inits.empty()) {
        std::vector<DeferredInit> inits;
        inits.swap(pending.inits);
        std::vector<int> rcs(inits.size(), 0);
        std::vector<int> devices(inits.size(), 0);
        std::vector<std::thread> threads;
        int device = 0;
        (void)hipGetDevice(&device);
        for (size_t k = 0; k < inits.size(); k++)
            threads.emplace_back([&, k] {
                (void)hipSetDevice(device);
                rcs[k] = init_rank(inits[k].out, inits[k].world, inits[k].name, inits[k].rank, true);
            });
        for (auto& t : threads)
            t.join();
        for (int rc : rcs)
            if (rc && !result)
                result = rc;
    }
    std::vector<Comm*> comms;
    std::vector<std::vector<Op>> ops;
    comms.swap(pending.comms);
    ops.swap(pending.ops);
    pending.open_pairs.clear();
    const int rc = dispatch(comms, ops);
    return result ? result : rc;
}

int ncclSend(const void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t stream)
{
    const size_t w = size_of(datatype);
    if (!comm || !w)
        return 4;
    Op op;
    op.stream = stream;
    op.msgs.push_back(Message{true, const_cast<void*>(buf), count * w, peer});
    return submit(static_cast<Comm*>(comm), std::move(op), true);
}

int ncclRecv(void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t stream)
{
    const size_t w = size_of(datatype);
    if (!comm || !w)
        return 4;
    Op op;
    op.stream = stream;
    op.msgs.push_back(Message{false, buf, count * w, peer});
    return submit(static_cast<Comm*>(comm), std::move(op), true);
}

// a collective is its own call also inside a group (a group of one broadcast per root would otherwise put several messages on
// one pair): ranks issue them in the same order, which is all this transport needs
int ncclAllGather(const void* send, void* recv, size_t count, int datatype, void* comm, hipStream_t stream)
{
    Comm* c = static_cast<Comm*>(comm);
    const size_t w = size_of(datatype);
    if (!c || !w)
        return 4;
    Op op;
    op.stream = stream;
    for (int r = 0; r < c->world; r++) {
        op.msgs.push_back(Message{true, const_cast<void*>(send), count * w, r});
        op.msgs.push_back(Message{false, (uint8_t*)recv + (size_t)r * count * w, count * w, r});
    }
    return submit(c, std::move(op), false);
}

int ncclBroadcast(const void* send, void* recv, size_t count, int datatype, int root, void* comm, hipStream_t stream)
{
    Comm* c = static_cast<Comm*>(comm);
    const size_t w = size_of(datatype);
    if (!c || !w || root < 0 || root >= c->world)
        return 4;
    Op op;
    op.stream = stream;
    if (c->rank == root) {
        for (int r = 0; r < c->world; r++)
            if (r != root)
                op.msgs.push_back(Message{true, const_cast<void*>(send), count * w, r});
        if (send != recv)
            op.copies.push_back(LocalCopy{recv, send, count * w});
    } else {
        op.msgs.push_back(Message{false, recv, count * w, root});
    }
    return submit(c, std::move(op), false);  // (every rank of the communicator takes part in a broadcast: the round count is agreed by all)
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
