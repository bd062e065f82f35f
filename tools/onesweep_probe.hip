// onesweep_probe.hip — DEV TOOL (not product, not shipped): where a pass of gv_sort's onesweep kernels spends its time.
// Includes the kernels with GV_SORT_TRACE (wall-clock stamps per tile and phase) and prints, per pass, the spread of
// tile start times and the median / max duration of each phase.
// Build (GPU box): hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DGV_SORT_TRACE \
//                  -Igarden_amd/csrc tools/onesweep_probe.hip -o /tmp/onesweep_probe
#include "../garden_amd/csrc/gv_sort.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 2124723u;
    const uint32_t capacity = argc > 2 ? (uint32_t)atoi(argv[2]) : 10000000u;
    std::vector<float> dist(n);
    srand(7);
    for (auto& d : dist) { const float r = 100.0f + 20000.0f * (float)rand() / RAND_MAX; d = r * r; }
    const uint32_t tiles = (capacity + 4095) / 4096;
    gv::SortBuffers b{};
    uint32_t *count, *idx_in, *idx_out, *hist;
    float *model_in, *model_out, *dist_in, *dist_out;
    CK(hipMalloc(&count, 4)); CK(hipMemcpy(count, &n, 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&idx_in, (size_t)capacity * 4)); CK(hipMalloc(&idx_out, (size_t)capacity * 4));
    CK(hipMalloc(&model_in, (size_t)capacity * 48)); CK(hipMalloc(&model_out, (size_t)capacity * 48));
    CK(hipMalloc(&dist_in, (size_t)capacity * 4)); CK(hipMalloc(&dist_out, (size_t)capacity * 4));
    CK(hipMemcpy(dist_in, dist.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(idx_in, 0, (size_t)capacity * 4)); CK(hipMemset(model_in, 0, (size_t)capacity * 48));
    for (int k = 0; k < 2; k++) { CK(hipMalloc(&b.keys[k], (size_t)capacity * 4)); CK(hipMalloc(&b.vals[k], (size_t)capacity * 4)); }
    const size_t words = 2 * 1280 + (size_t)4 * tiles * 256;
    CK(hipMalloc(&hist, words * 4)); CK(hipMemset(hist, 0, words * 4));
    b.count = count; b.idx_in = idx_in; b.model_in = model_in; b.dist_in = dist_in;
    b.idx_out = idx_out; b.model_out = model_out; b.dist_out = dist_out;
    for (int k = 0; k < 2; k++) { b.ghist[k] = hist + k * 1280; b.tile_counter[k] = hist + k * 1280 + 1024; }
    b.status = hist + 2 * 1280;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 12; it++) {
        b.parity = it & 1;
        CK(hipEventRecord(e0, st));
        CK(gv::launch_sort(b, capacity, false, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2) best = std::min(best, ms);
    }
    std::vector<float> out(n);
    CK(hipMemcpy(out.data(), dist_out, (size_t)n * 4, hipMemcpyDeviceToHost));
    bool sorted = std::is_sorted(out.begin(), out.end());
    printf("n=%u capacity=%u: gv_sort best %.1f us, sorted=%d\n", n, capacity, best * 1e3f, (int)sorted);
    const uint32_t live = (n + 4095) / 4096;
    static unsigned long long tr[4][8192][8];
    CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(gv::gv_sort_trace), sizeof(tr)));
    const char* names[5] = {"loads+rank", "look-back", "scans", "reorder(LDS)", "write-out"};
    for (int p = 0; p < 4; p++) {
        unsigned long long t0 = ~0ull, t5 = 0;
        for (uint32_t t = 0; t < live && t < 8192; t++) { t0 = std::min(t0, tr[p][t][0]); t5 = std::max(t5, tr[p][t][5]); }
        std::vector<double> start;
        for (uint32_t t = 0; t < live && t < 8192; t++) start.push_back((tr[p][t][0] - t0) * 0.01);
        std::sort(start.begin(), start.end());
        printf("pass %d: first start -> last end %.1f us; tile start spread: median %.1f max %.1f us\n", p, (t5 - t0) * 0.01,
               start[start.size() / 2], start.back());
        {
            std::vector<double> d;
            for (uint32_t t = 0; t < live && t < 8192; t++) d.push_back((double)(tr[p][t][6] - tr[p][t][0]) * 0.01);
            std::sort(d.begin(), d.end());
            printf("    %-14s median %6.2f  p90 %6.2f  max %6.2f us (part of loads+rank)\n", "loads only", d[d.size() / 2], d[d.size() * 9 / 10], d.back());
        }
        for (int k = 0; k < 5; k++) {
            std::vector<double> d;
            for (uint32_t t = 0; t < live && t < 8192; t++) d.push_back((double)(tr[p][t][k + 1] - tr[p][t][k]) * 0.01);
            std::sort(d.begin(), d.end());
            printf("    %-14s median %6.2f  p90 %6.2f  max %6.2f us\n", names[k], d[d.size() / 2], d[d.size() * 9 / 10], d.back());
        }
    }
    return 0;
}
