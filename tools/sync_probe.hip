// Dev tool: how long after a kernel's last store (a flag in pinned host memory) does hipStreamSynchronize return?
//   hipcc --offload-arch=gfx950 -O3 tools/sync_probe.hip -o /tmp/sync_probe && /tmp/sync_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void work_then_flag(volatile unsigned* flag, unsigned value, unsigned spin)
{
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}  // ~spin * 10 ns of "work"
    __threadfence_system();
    *flag = value;
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv)
{
    // sync_probe [spin|yield|block]: hipSetDeviceFlags(hipDeviceSchedule...) before anything else touches the device;
    // sync_probe query: hipStreamQuery polled instead of hipStreamSynchronize
    const char* mode = argc > 1 ? argv[1] : "auto";
    if (mode[0] == 's') hipSetDeviceFlags(hipDeviceScheduleSpin);
    if (mode[0] == 'y') hipSetDeviceFlags(hipDeviceScheduleYield);
    if (mode[0] == 'b') hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
    const bool query = mode[0] == 'q';
    printf("mode %s\n", mode);
    unsigned* flag;
    hipHostMalloc(&flag, 4, hipHostMallocDefault);
    *flag = 0;
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (unsigned spin : {0u, 1000u, 2000u}) {  // 0, 10, 20 us kernels
        std::vector<double> launch_to_flag, flag_to_sync, sync_only;
        for (int it = 0; it < 2000; it++) {
            const unsigned v = it + 1;
            const double t0 = now_us();
            hipLaunchKernelGGL(work_then_flag, dim3(1), dim3(64), 0, s, flag, v, spin);
            while (*(volatile unsigned*)flag != v) {}
            const double t1 = now_us();
            if (query) { while (hipStreamQuery(s) == hipErrorNotReady) {} } else hipStreamSynchronize(s);
            const double t2 = now_us();
            launch_to_flag.push_back(t1 - t0);
            flag_to_sync.push_back(t2 - t1);
            // the same without the spin: launch + synchronize
            const double t3 = now_us();
            hipLaunchKernelGGL(work_then_flag, dim3(1), dim3(64), 0, s, flag, v, spin);
            if (query) { while (hipStreamQuery(s) == hipErrorNotReady) {} } else hipStreamSynchronize(s);
            sync_only.push_back(now_us() - t3);
        }
        auto med = [](std::vector<double>& a) { std::sort(a.begin(), a.end()); return a[a.size() / 2]; };
        printf("kernel ~%2u us: launch -> flag seen %.1f us; flag seen -> hipStreamSynchronize returns %.1f us; launch + hipStreamSynchronize %.1f us\n",
               spin / 100, med(launch_to_flag), med(flag_to_sync), med(sync_only));
    }
    return 0;
}
