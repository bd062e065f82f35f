// gv_workers.cpp — see gv_workers.hpp. Host-only translation unit.
#include "gv_workers.hpp"

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <pthread.h>
#include <vector>

namespace gv {
namespace {

struct WorkerPool {
    std::mutex run_lock;  // one run at a time
    std::mutex m;
    std::condition_variable wake, done;
    const std::function<void(uint32_t)>* job = nullptr;
    uint64_t generation = 0;
    uint32_t parts = 0;
    std::atomic<uint32_t> next{0};
    uint32_t finished = 0;  // parts completed in this generation (under m)
    uint32_t busy = 0;      // workers inside the current generation (under m)
    std::vector<std::thread> threads;

    void work()
    {
        uint32_t completed = 0;
        for (;;) {
            const uint32_t part = next.fetch_add(1, std::memory_order_relaxed);
            if (part >= parts)
                break;
            (*job)(part);
            completed++;
        }
        std::lock_guard<std::mutex> lock(m);
        finished += completed;
        if (finished == parts)
            done.notify_all();
    }

    void worker_main()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lock(m);
        for (;;) {
            wake.wait(lock, [&] { return generation != seen; });
            seen = generation;
            if (!job)
                continue;  // the run this wake-up belonged to is already over
            busy++;
            lock.unlock();
            work();
            lock.lock();
            busy--;
            if (busy == 0)
                done.notify_all();
        }
    }

    void ensure_threads(uint32_t want)
    {
        while (threads.size() < want) {
            threads.emplace_back([this] { worker_main(); });
            threads.back().detach();  // parked for the life of the process
        }
    }

    void run(uint32_t count, const std::function<void(uint32_t)>& fn)
    {
        std::lock_guard<std::mutex> serial(run_lock);
        {
            std::lock_guard<std::mutex> lock(m);
            ensure_threads(count - 1);
            job = &fn;
            parts = count;
            finished = 0;
            next.store(0, std::memory_order_relaxed);
            generation++;
        }
        wake.notify_all();
        work();  // the caller takes parts too (thread-pool.cpp:203-215: the waiting thread participates)
        std::unique_lock<std::mutex> lock(m);
        // all parts done AND no worker still inside this generation (it may be between its last fetch_add and its
        // bookkeeping): only then may `fn` and the counters be reused
        done.wait(lock, [&] { return finished == parts && busy == 0; });
        job = nullptr;
    }
};

std::atomic<WorkerPool*> g_pool{nullptr};
std::mutex g_pool_lock;

// A forked child inherits the pool object but none of its threads: start over with a fresh one on first use (the
// old object is leaked; its mutexes may have been held by threads that do not exist here).
void forget_pool_in_child() { g_pool.store(nullptr, std::memory_order_release); new (&g_pool_lock) std::mutex(); }

WorkerPool& pool()
{
    WorkerPool* p = g_pool.load(std::memory_order_acquire);
    if (!p) {
        std::lock_guard<std::mutex> lock(g_pool_lock);
        p = g_pool.load(std::memory_order_relaxed);
        if (!p) {
            static const int registered = pthread_atfork(nullptr, nullptr, forget_pool_in_child);
            (void)registered;
            p = new WorkerPool();  // never destroyed: its threads are parked on it until the process exits
            g_pool.store(p, std::memory_order_release);
        }
    }
    return *p;
}

}  // namespace

void run_parts(uint32_t parts, const std::function<void(uint32_t)>& job)
{
    if (parts <= 1) {
        if (parts == 1)
            job(0);
        return;
    }
    pool().run(parts, job);
}

}  // namespace gv
