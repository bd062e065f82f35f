// gv_hiz_kernels.hpp — launch interface of the any-size fused pyramid kernel in gv_hiz.hip (kept apart from gv_kernels.hpp:
// bench.py hashes that file to tell whether the cull kernels changed since the PMC counters in profiles/traffic.json were
// collected; the pyramid build of a frame size not divisible by 64 does not enter into that).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gv {

// Levels k+1, k+2, k+3 from level k in one launch, any sizes (hiz.frag:27-56 incl. the odd-size branches).
// w[0], h[0]: level k (the source: `depth` when k == 0, else `src_pairs`); w[l], h[l]: level k + l; dst[l - 1] receives it.
// Texels are float2, or packed binary16 pairs (4 bytes) when rg16f. Needs w[0] >= 2 && h[0] >= 2.
struct HizFused3Args {
    const float* depth;
    const float2* src_pairs;
    float2* dst[3];
    uint32_t w[4], h[4];
    uint32_t rule;
};
hipError_t launch_hiz_fused3(const HizFused3Args& args, bool rg16f, hipStream_t stream);

// The same with FOUR levels per launch: a workgroup owns 64 x 64 texels of level k+1 ... 8 x 8 of level k+4 (rims 75 / 37 / 18:
// 1.37x the level-k+1 arithmetic). For frames large enough to fill the GPU with such workgroups (1920 x 1080 and up): the
// one-workgroup tail kernel then starts a level later, where it has a quarter of the texels to read from memory.
struct HizFused4Args {
    const float* depth;
    const float2* src_pairs;
    float2* dst[4];
    uint32_t w[5], h[5];
    uint32_t rule;
};
hipError_t launch_hiz_fused4(const HizFused4Args& args, bool rg16f, hipStream_t stream);

}  // namespace gv
