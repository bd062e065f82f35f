// onesweep_probe.hip — DEV TOOL (not product, not shipped): where a pass of gv_sort's radix kernels spends its time
// (named after the round-2 first form, one look-back kernel per digit; now a rank and a scatter kernel per digit).
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
    const uint32_t tiles = gv::sort_tile_count(capacity);
    gv::SortBuffers b{};
    uint32_t *count, *idx_in, *idx_out, *hist;
    uint16_t* ranks;
    float *model_in, *model_out, *dist_in, *dist_out;
    CK(hipMalloc(&count, 4)); CK(hipMemcpy(count, &n, 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&idx_in, (size_t)capacity * 4)); CK(hipMalloc(&idx_out, (size_t)capacity * 4));
    CK(hipMalloc(&model_in, (size_t)capacity * 48)); CK(hipMalloc(&model_out, (size_t)capacity * 48));
    CK(hipMalloc(&dist_in, (size_t)capacity * 4)); CK(hipMalloc(&dist_out, (size_t)capacity * 4));
    CK(hipMalloc(&ranks, (size_t)capacity * 2));
    CK(hipMemcpy(dist_in, dist.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(idx_in, 0, (size_t)capacity * 4)); CK(hipMemset(model_in, 0, (size_t)capacity * 48));
    for (int k = 0; k < 2; k++) { CK(hipMalloc(&b.keys[k], (size_t)capacity * 4)); CK(hipMalloc(&b.vals[k], (size_t)capacity * 4)); CK(hipMalloc(&b.slots[k], (size_t)capacity * 4)); }
    const size_t set_words = gv::sort_set_words(capacity);
    const size_t words = 2 * set_words + (size_t)tiles * 256;
    CK(hipMalloc(&hist, words * 4)); CK(hipMemset(hist, 0, words * 4));
    b.count = count; b.idx_in = idx_in; b.model_in = model_in; b.dist_in = dist_in;
    b.idx_out = idx_out; b.model_out = model_out; b.dist_out = dist_out; b.ranks = ranks;
    for (int k = 0; k < 2; k++) b.counters[k] = hist + k * set_words;
    b.tile_hist = hist + 2 * set_words;
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
#ifdef GV_SORT_TRACE
    const uint32_t live = std::min((n + 4095) / 4096, 8192u);
    static unsigned long long tr[4][8192][12];
    CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(gv::gv_sort_trace), sizeof(tr)));
    // slots: rank kernel 0 start, 1 keys loaded, 2 ranked, 3 end; scatter kernel 4 start, 5 keys + counts loaded, 6 bases, 7 reordered, 8 end
    struct Phase { const char* name; int from, to; };
    const Phase phases[] = {{"rank: loads", 0, 1}, {"rank: ranking", 1, 2}, {"rank: stores", 2, 3}, {"scatter: loads", 4, 5},
                            {"scatter: scans", 5, 6}, {"scatter: reorder", 6, 7}, {"scatter: write", 7, 8}};
    for (int p = 0; p < 4; p++) {
        unsigned long long a0 = ~0ull, a3 = 0, b4 = ~0ull, b8 = 0;
        for (uint32_t t = 0; t < live; t++) {
            a0 = std::min(a0, tr[p][t][0]); a3 = std::max(a3, tr[p][t][3]);
            b4 = std::min(b4, tr[p][t][4]); b8 = std::max(b8, tr[p][t][8]);
        }
        printf("pass %d: rank kernel first start -> last end %.1f us, gap to the scatter kernel's first start %.1f us, scatter kernel %.1f us\n", p,
               (a3 - a0) * 0.01, ((double)b4 - (double)a3) * 0.01, (b8 - b4) * 0.01);
        for (const Phase& ph : phases) {
            std::vector<double> d;
            for (uint32_t t = 0; t < live; t++) d.push_back((double)(tr[p][t][ph.to] - tr[p][t][ph.from]) * 0.01);
            std::sort(d.begin(), d.end());
            printf("    %-18s median %6.2f  p90 %6.2f  max %6.2f us\n", ph.name, d[d.size() / 2], d[d.size() * 9 / 10], d.back());
        }
    }
#endif
    return 0;
}
