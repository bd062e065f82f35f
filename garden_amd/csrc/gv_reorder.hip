// gv_reorder.hip — the spatial re-order of the device mirror ON the device. Entities created since the last
// full build sit in an unsorted tail of the mirror (gv_mirror.cpp grow_*); once the tail passes 1/8 of a pool the mirror is
// brought back into Morton order. The host form of that (build_transform_order + a gather of every component + a 73 B per
// entity upload) is 0.13-0.22 s at 10 M entities: a frame hitch in an engine that spawns entities (the reference's
// duplicate / destroy paths, source/system/transform.cpp:626-714). Everything the order needs is already in HBM, so here
// the codes are computed, sorted (gv_sort.hip's radix kernels on bare keys) and the mirror's streams permuted in place of
// that; the host only re-derives its slot <-> entry tables from the order it downloads.
#include "gv_device.hpp"

namespace gv {

namespace {

constexpr uint32_t kThreads = 256;

// order-preserving image of a float for integer atomicMin / atomicMax
__device__ __forceinline__ uint32_t ordered(float f)
{
    const uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float unordered(uint32_t k)
{
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

__device__ __forceinline__ uint32_t spread10(uint32_t v)  // 10 bits -> every third bit (as build_transform_order)
{
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// root[j] = the root ancestor of entry j; box = bounding box of the live roots' finite coordinates (ordered images)
__global__ __launch_bounds__(kThreads) void reorder_roots_kernel(const XfAB* __restrict__ ab, const uint8_t* __restrict__ flags,
                                                                 const uint32_t* __restrict__ parent, uint32_t n, uint32_t max_depth,
                                                                 uint32_t* __restrict__ root, uint32_t* __restrict__ box)
{
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t j = blockIdx.x * kThreads + threadIdx.x; j < n; j += gridDim.x * kThreads) {
        uint32_t cur = j;
        if (max_depth) {
            for (uint32_t d = 0; d <= max_depth; d++) {  // (the mirror's chains were validated: no cycles, at most max_depth links)
                const uint32_t p = parent[cur];
                if (p == kSlotNone || p >= n)
                    break;
                cur = p;
            }
        }
        root[j] = cur;
        if (cur == j && (flags[j] & kXfLive)) {
            const float4 a = ab[j].a;
            const float pos[3] = {a.x, a.y, a.z};
#pragma unroll
            for (int k = 0; k < 3; k++)
                if (isfinite(pos[k])) {
                    lo[k] = fminf(lo[k], pos[k]);
                    hi[k] = fmaxf(hi[k], pos[k]);
                }
        }
    }
    __shared__ float part[6][kThreads / 64];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float l = lo[k], h = hi[k];
#pragma unroll
        for (uint32_t d = 32; d >= 1; d >>= 1) {
            l = fminf(l, __shfl_xor(l, d, 64));
            h = fmaxf(h, __shfl_xor(h, d, 64));
        }
        if ((threadIdx.x & 63u) == 0) {
            part[k][threadIdx.x >> 6] = l;
            part[3 + k][threadIdx.x >> 6] = h;
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = part[threadIdx.x][0];
        for (uint32_t w = 1; w < kThreads / 64; w++)
            v = threadIdx.x < 3 ? fminf(v, part[threadIdx.x][w]) : fmaxf(v, part[threadIdx.x][w]);
        if (threadIdx.x < 3)
            atomicMin(&box[threadIdx.x], ordered(v));
        else
            atomicMax(&box[threadIdx.x], ordered(v));
    }
}

// the 30-bit Morton code of every entry's ROOT position inside the box (a whole tree shares one code); free entries last
__global__ __launch_bounds__(kThreads) void reorder_codes_kernel(const XfAB* __restrict__ ab, const uint8_t* __restrict__ flags,
                                                                 const uint32_t* __restrict__ root, const uint32_t* __restrict__ box, uint32_t n,
                                                                 float* __restrict__ code /* the bits of a small positive float: a radix key */)
{
    const uint32_t j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= n)
        return;
    uint32_t c = 0x3FFFFFFFu;
    if (flags[j] & kXfLive) {
        const float4 a = ab[root[j]].a;
        const float pos[3] = {a.x, a.y, a.z};
        uint32_t q[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float l = unordered(box[k]), ext = unordered(box[3 + k]) - l;
            const float f = (ext > 0.0f && isfinite(pos[k])) ? (pos[k] - l) / ext : 0.0f;
            q[k] = (uint32_t)fminf(1023.0f, fmaxf(0.0f, f * 1024.0f));
        }
        c = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    }
    code[j] = __uint_as_float(c);
}

// newpos[order[k]] = k
__global__ __launch_bounds__(kThreads) void reorder_invert_kernel(const uint32_t* __restrict__ order, uint32_t n, uint32_t* __restrict__ newpos)
{
    const uint32_t k = blockIdx.x * kThreads + threadIdx.x;
    if (k < n)
        newpos[order[k]] = k;
}

__global__ __launch_bounds__(kThreads) void reorder_transforms_kernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ newpos, uint32_t n,
                                                                      const XfAB* __restrict__ ab_in, const float2* __restrict__ c_in,
                                                                      const uint8_t* __restrict__ flags_in, const uint32_t* __restrict__ parent_in,
                                                                      XfAB* __restrict__ ab_out, float2* __restrict__ c_out, uint8_t* __restrict__ flags_out,
                                                                      uint32_t* __restrict__ parent_out)
{
    const uint32_t k = blockIdx.x * kThreads + threadIdx.x;
    if (k >= n)
        return;
    const uint32_t j = order[k];
    ab_out[k] = ab_in[j];
    c_out[k] = c_in[j];
    flags_out[k] = flags_in[j];
    const uint32_t p = parent_in[j];
    parent_out[k] = (p == kSlotNone || p >= n) ? kSlotNone : newpos[p];
}

// out[s] = newpos[table[s]] (slot -> entry tables), and its inverse entry -> slot when asked for
__global__ __launch_bounds__(kThreads) void reorder_remap_kernel(const uint32_t* __restrict__ table, uint32_t n, const uint32_t* __restrict__ newpos,
                                                                 uint32_t* __restrict__ out, uint32_t* __restrict__ inverse)
{
    const uint32_t s = blockIdx.x * kThreads + threadIdx.x;
    if (s >= n)
        return;
    const uint32_t j = newpos[table[s]];
    out[s] = j;
    if (inverse)
        inverse[j] = s;
}

// a mesh entry sorts by the (new) mirror entry of its transform; entries without one last, in their old order
__global__ __launch_bounds__(kThreads) void reorder_mesh_keys_kernel(const uint32_t* __restrict__ link, uint32_t n, const uint32_t* __restrict__ xnewpos,
                                                                     uint32_t xn, float* __restrict__ key)
{
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t slot = link[i] & kSlotMask;
    uint32_t k = 0x3FFFFFFFu;
    if (slot != kSlotNone && slot < xn)
        k = xnewpos ? xnewpos[slot] : slot;
    key[i] = __uint_as_float(k);
}

__global__ __launch_bounds__(kThreads) void reorder_meshes_kernel(const uint32_t* __restrict__ order, uint32_t n, const uint32_t* __restrict__ xnewpos,
                                                                  uint32_t xn, const float4* __restrict__ a_in, const float2* __restrict__ b_in,
                                                                  const uint32_t* __restrict__ link_in, const uint32_t* __restrict__ orig_in,
                                                                  float4* __restrict__ a_out, float2* __restrict__ b_out, uint32_t* __restrict__ link_out,
                                                                  uint32_t* __restrict__ orig_out, uint32_t* __restrict__ inv_out,
                                                                  uint32_t* __restrict__ unpaired)
{
    const uint32_t k = blockIdx.x * kThreads + threadIdx.x;
    if (k >= n)
        return;
    const uint32_t j = order[k];
    a_out[k] = a_in[j];
    b_out[k] = b_in[j];
    uint32_t link = link_in[j];
    uint32_t slot = link & kSlotMask;
    if (xnewpos && slot != kSlotNone && slot < xn) {
        slot = xnewpos[slot];
        link = (link & ~kSlotMask) | slot;
    }
    link_out[k] = link;
    if ((link & kMeshCandidate) && slot != k)
        *unpaired = 1u;  // (every writer stores the same value)
    const uint32_t s = orig_in[j];
    orig_out[k] = s;
    inv_out[s] = k;
}

inline dim3 grid_for(uint32_t n) { return dim3((n + kThreads - 1) / kThreads); }

// ---- scattered dirty slots: one packet, one launch (gv_sort_kernels.hpp) ----
__global__ __launch_bounds__(kThreads) void scatter_xf_packets_kernel(const XfPacket* __restrict__ packets, uint32_t count, XfAB* __restrict__ ab,
                                                                      float2* __restrict__ c, uint8_t* __restrict__ flags, uint32_t* __restrict__ parent,
                                                                      unsigned long long* __restrict__ active_bits, uint8_t* __restrict__ world_dirty,
                                                                      const BlockFlagTargets blocks)
{
    const uint32_t t = blockIdx.x * kThreads + threadIdx.x;
    if (t >= count)
        return;
    const XfPacket p = packets[t];
    const uint32_t e = p.entry;
    XfAB rec;
    rec.a = p.a;
    rec.b = p.b;
    ab[e] = rec;
    c[e] = p.c;
    flags[e] = (uint8_t)p.flags;
    parent[e] = p.parent;
    if (world_dirty)
        world_dirty[e] = 1;
    // the entry's bit of the active bit-plane (pack_active_kernel's rule), without a pass over the pool
    const unsigned long long bit = 1ull << (e & 63u);
    if (p.flags & kXfActive)
        atomicOr(&active_bits[e >> 6], bit);
    else
        atomicAnd(&active_bits[e >> 6], ~bit);
#pragma unroll
    for (uint32_t k = 0; k < kMaxFlaggedPools; k++)
        if (blocks.flags[k] && e < blocks.occupancy[k])
            blocks.flags[k][e / kCullBlock] = 1;
}

__global__ __launch_bounds__(kThreads) void scatter_mesh_packets_kernel(const MeshPacket* __restrict__ packets, uint32_t count, float4* __restrict__ a,
                                                                        float2* __restrict__ b, uint32_t* __restrict__ link,
                                                                        uint8_t* __restrict__ block_flags)
{
    const uint32_t t = blockIdx.x * kThreads + threadIdx.x;
    if (t >= count)
        return;
    const MeshPacket p = packets[t];
    a[p.entry] = p.a;
    b[p.entry] = p.b;
    link[p.entry] = p.link;
    if (block_flags)
        block_flags[p.entry / kCullBlock] = 1;
}

}  // namespace

hipError_t launch_scatter_xf_packets(const XfPacket* packets, uint32_t count, XfAB* ab, float2* c, uint8_t* flags, uint32_t* parent,
                                     unsigned long long* active_bits, uint8_t* world_dirty, const BlockFlagTargets& blocks, hipStream_t stream)
{
    if (count)
        hipLaunchKernelGGL(scatter_xf_packets_kernel, grid_for(count), dim3(kThreads), 0, stream, packets, count, ab, c, flags, parent, active_bits,
                           world_dirty, blocks);
    return hipGetLastError();
}

hipError_t launch_scatter_mesh_packets(const MeshPacket* packets, uint32_t count, float4* a, float2* b, uint32_t* link, uint8_t* block_flags,
                                       hipStream_t stream)
{
    if (count)
        hipLaunchKernelGGL(scatter_mesh_packets_kernel, grid_for(count), dim3(kThreads), 0, stream, packets, count, a, b, link, block_flags);
    return hipGetLastError();
}

namespace {
}  // namespace

hipError_t launch_reorder_codes(const TransformMirror& xf, uint32_t* root, uint32_t* box, float* code, hipStream_t stream)
{
    if (xf.count == 0)
        return hipSuccess;
    static const uint32_t init[6] = {0xFF800000u, 0xFF800000u, 0xFF800000u, 0x007FFFFFu, 0x007FFFFFu, 0x007FFFFFu};  // ordered(+inf) x3, ordered(-inf) x3
    hipError_t e = hipMemcpyAsync(box, init, sizeof(init), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess)
        return e;
    const uint32_t blocks = std::min<uint32_t>(2048u, (xf.count + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(reorder_roots_kernel, dim3(blocks), dim3(kThreads), 0, stream, xf.ab, xf.flags, xf.parent, xf.count, xf.max_depth, root, box);
    hipLaunchKernelGGL(reorder_codes_kernel, grid_for(xf.count), dim3(kThreads), 0, stream, xf.ab, xf.flags, root, box, xf.count, code);
    return hipGetLastError();
}

hipError_t launch_reorder_invert(const uint32_t* order, uint32_t n, uint32_t* newpos, hipStream_t stream)
{
    if (n)
        hipLaunchKernelGGL(reorder_invert_kernel, grid_for(n), dim3(kThreads), 0, stream, order, n, newpos);
    return hipGetLastError();
}

hipError_t launch_reorder_transforms(const uint32_t* order, const uint32_t* newpos, uint32_t n, const XfAB* ab_in, const float2* c_in,
                                     const uint8_t* flags_in, const uint32_t* parent_in, XfAB* ab_out, float2* c_out, uint8_t* flags_out,
                                     uint32_t* parent_out, hipStream_t stream)
{
    if (n)
        hipLaunchKernelGGL(reorder_transforms_kernel, grid_for(n), dim3(kThreads), 0, stream, order, newpos, n, ab_in, c_in, flags_in, parent_in, ab_out,
                           c_out, flags_out, parent_out);
    return hipGetLastError();
}

hipError_t launch_reorder_remap(const uint32_t* table, uint32_t n, const uint32_t* newpos, uint32_t* out, uint32_t* inverse, hipStream_t stream)
{
    if (n)
        hipLaunchKernelGGL(reorder_remap_kernel, grid_for(n), dim3(kThreads), 0, stream, table, n, newpos, out, inverse);
    return hipGetLastError();
}

hipError_t launch_reorder_mesh_keys(const uint32_t* link, uint32_t n, const uint32_t* xnewpos, uint32_t xn, float* key, hipStream_t stream)
{
    if (n)
        hipLaunchKernelGGL(reorder_mesh_keys_kernel, grid_for(n), dim3(kThreads), 0, stream, link, n, xnewpos, xn, key);
    return hipGetLastError();
}

hipError_t launch_reorder_meshes(const uint32_t* order, uint32_t n, const uint32_t* xnewpos, uint32_t xn, const float4* a_in, const float2* b_in,
                                 const uint32_t* link_in, const uint32_t* orig_in, float4* a_out, float2* b_out, uint32_t* link_out,
                                 uint32_t* orig_out, uint32_t* inv_out, uint32_t* unpaired, hipStream_t stream)
{
    if (n)
        hipLaunchKernelGGL(reorder_meshes_kernel, grid_for(n), dim3(kThreads), 0, stream, order, n, xnewpos, xn, a_in, b_in, link_in, orig_in, a_out,
                           b_out, link_out, orig_out, inv_out, unpaired);
    return hipGetLastError();
}

}  // namespace gv
