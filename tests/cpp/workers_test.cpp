// TEST: the library's persistent host worker pool (garden_amd/csrc/gv_workers.*), host-only — built with
// -fsanitize=thread by tests/test_host_logic.py. Every item of every run is visited exactly once, runs from
// several caller threads serialise correctly, and small ranges stay on the calling thread.
#include "../../garden_amd/csrc/gv_workers.hpp"

#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include <sys/wait.h>
#include <unistd.h>

int main()
{
    int failures = 0;
    // 1. coverage: each index exactly once, across sizes around the inline / pooled threshold
    for (uint32_t n : {0u, 1u, 1000u, 32767u, 32768u, 100003u, 1u << 20}) {
        std::vector<uint8_t> hits(n, 0);
        std::atomic<uint32_t> calls{0};
        gv::parallel_ranges(7, n, [&](uint32_t lo, uint32_t hi) {
            calls++;
            for (uint32_t i = lo; i < hi; i++)
                hits[i - 7]++;
        });
        for (uint32_t i = 0; i < n; i++)
            if (hits[i] != 1) {
                failures++;
                break;
            }
        if (n < 32768 && calls.load() != 1) {
            printf("n=%u: expected one inline call, got %u\n", n, calls.load());
            failures++;
        }
    }
    // 2. many short runs back to back (a late waker must never touch a finished run's job)
    for (int round = 0; round < 2000; round++) {
        std::atomic<uint64_t> sum{0};
        gv::run_parts(8, [&](uint32_t part) { sum += part + 1; });
        if (sum.load() != 36) {
            printf("round %d: sum %llu\n", round, (unsigned long long)sum.load());
            failures++;
            break;
        }
    }
    // 3. concurrent callers (two contexts on two threads)
    std::vector<std::thread> callers;
    std::atomic<int> bad{0};
    for (int c = 0; c < 4; c++)
        callers.emplace_back([&, c] {
            for (int round = 0; round < 200; round++) {
                const uint32_t n = 40000 + 1000 * c;
                std::vector<uint32_t> out(n, 0);
                gv::parallel_ranges(0, n, [&](uint32_t lo, uint32_t hi) {
                    for (uint32_t i = lo; i < hi; i++)
                        out[i] = i * 3 + c;
                });
                for (uint32_t i = 0; i < n; i++)
                    if (out[i] != i * 3 + c) {
                        bad++;
                        break;
                    }
            }
        });
    for (auto& t : callers)
        t.join();
    failures += bad.load();
    // 4. a forked child has no worker threads: the pool must start over there instead of waiting for them
    {
        const pid_t child = fork();
        if (child == 0) {
            std::atomic<uint64_t> sum{0};
            gv::run_parts(6, [&](uint32_t part) { sum += part + 1; });
            _exit(sum.load() == 21 ? 0 : 1);
        }
        int status = 0;
        alarm(60);
        waitpid(child, &status, 0);
        alarm(0);
        if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) {
            printf("forked child: status %d\n", status);
            failures++;
        }
    }
    printf("{\"ok\": %s, \"failures\": %d}\n", failures == 0 ? "true" : "false", failures);
    return failures == 0 ? 0 : 1;
}
