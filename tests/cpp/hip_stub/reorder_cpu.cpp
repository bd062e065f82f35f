// TEST-ONLY (see hip/hip_runtime.h): host stand-ins for the launch functions of gv_reorder.hip and for launch_sort on bare keys, so
// that the device-side re-order's HOST half (gv_mirror.cpp reorder_*_device: scratch, downloads, table checks, buffer swaps) runs
// for real under the sanitizers — with kernels as no-ops it would only ever see zeroed tables. "Device" memory is host memory
// here. Never linked into the product library.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#include "gv_kernels.hpp"

namespace gv {

static uint32_t bits_of(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

hipError_t launch_sort(const SortBuffers& b, uint32_t capacity, bool descending, hipStream_t, SortMode)
{
    if (b.model_in || !b.idx_out)
        return hipSuccess;  // a sort of records: a no-op like every other kernel of the stub build
    const uint32_t n = std::min(*b.count, capacity);
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        const uint32_t kx = bits_of(b.dist_in[x]), ky = bits_of(b.dist_in[y]);
        return descending ? kx > ky : kx < ky;
    });
    for (uint32_t k = 0; k < n; k++)
        b.idx_out[k] = b.idx_in ? b.idx_in[order[k]] : order[k];
    return hipSuccess;
}

hipError_t launch_reorder_codes(const TransformMirror& xf, uint32_t* root, uint32_t*, float* code, hipStream_t)
{
    for (uint32_t j = 0; j < xf.count; j++) {
        uint32_t cur = j;
        for (uint32_t d = 0; xf.max_depth && d <= xf.max_depth; d++) {
            const uint32_t p = xf.parent[cur];
            if (p == kSlotNone || p >= xf.count)
                break;
            cur = p;
        }
        root[j] = cur;
    }
    for (uint32_t j = 0; j < xf.count; j++) {  // (a coarse key is as good as a Morton code for what is checked here: any key gives a permutation)
        uint32_t c = 0x3FFFFFFFu;
        if (xf.flags[j] & kXfLive) {
            const float x = xf.ab[root[j]].a.x;
            c = std::isfinite(x) ? (uint32_t)std::min(1023.0f, std::max(0.0f, std::fabs(x))) : 0u;
        }
        std::memcpy(&code[j], &c, 4);
    }
    return hipSuccess;
}

hipError_t launch_reorder_invert(const uint32_t* order, uint32_t n, uint32_t* newpos, hipStream_t)
{
    for (uint32_t k = 0; k < n; k++)
        newpos[order[k]] = k;
    return hipSuccess;
}

hipError_t launch_reorder_transforms(const uint32_t* order, const uint32_t* newpos, uint32_t n, const XfAB* ab_in, const float2* c_in,
                                     const uint8_t* flags_in, const uint32_t* parent_in, XfAB* ab_out, float2* c_out, uint8_t* flags_out,
                                     uint32_t* parent_out, hipStream_t)
{
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t j = order[k];
        ab_out[k] = ab_in[j];
        c_out[k] = c_in[j];
        flags_out[k] = flags_in[j];
        const uint32_t p = parent_in[j];
        parent_out[k] = (p == kSlotNone || p >= n) ? kSlotNone : newpos[p];
    }
    return hipSuccess;
}

hipError_t launch_reorder_remap(const uint32_t* table, uint32_t n, const uint32_t* newpos, uint32_t* out, uint32_t* inverse, hipStream_t)
{
    for (uint32_t s = 0; s < n; s++) {
        out[s] = newpos[table[s]];
        if (inverse)
            inverse[out[s]] = s;
    }
    return hipSuccess;
}

hipError_t launch_reorder_mesh_keys(const uint32_t* link, uint32_t n, const uint32_t* xnewpos, uint32_t xn, float* key, hipStream_t)
{
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t slot = link[i] & kSlotMask;
        uint32_t k = 0x3FFFFFFFu;
        if (slot != kSlotNone && slot < xn)
            k = xnewpos ? xnewpos[slot] : slot;
        std::memcpy(&key[i], &k, 4);
    }
    return hipSuccess;
}

hipError_t launch_reorder_meshes(const uint32_t* order, uint32_t n, const uint32_t* xnewpos, uint32_t xn, const float4* a_in, const float2* b_in,
                                 const uint32_t* link_in, const uint32_t* orig_in, float4* a_out, float2* b_out, uint32_t* link_out,
                                 uint32_t* orig_out, uint32_t* inv_out, uint32_t* unpaired, hipStream_t)
{
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t j = order[k];
        a_out[k] = a_in[j];
        b_out[k] = b_in[j];
        uint32_t link = link_in[j], slot = link & kSlotMask;
        if (xnewpos && slot != kSlotNone && slot < xn) {
            slot = xnewpos[slot];
            link = (link & ~kSlotMask) | slot;
        }
        link_out[k] = link;
        if ((link & kMeshCandidate) && slot != k)
            *unpaired = 1u;
        orig_out[k] = orig_in[j];
        inv_out[orig_in[j]] = k;
    }
    return hipSuccess;
}

}  // namespace gv
