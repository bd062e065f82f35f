// gv_context.cpp — host side of libgarden_vis.so: context, AoS -> SoA device mirror, per-frame
// dispatch, results. Implements include/garden_vis.h.
//
// Replaces (reference paths): the scratch sizing / dispatch / wait of MeshRenderSystem::prepareMeshes
// (source/system/render/mesh.cpp:331-553), the per-entity Manager::tryGet lookup (mesh.cpp:149) —
// resolved once into the mirror — and HizRenderSystem::downsampleHiz's per-mip pass loop
// (source/system/render/hiz.cpp:148-164).
//
// There is NO CPU fallback: without a gfx950 device gv_create fails with GV_E_NODEVICE.
#include <hip/hip_runtime.h>
#include <rocprofiler-sdk-roctx/roctx.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/garden_vis.h"
#include "gv_kernels.hpp"

using namespace gv;

namespace {

thread_local std::string g_create_error;

template <typename T>
struct DeviceBuf {  // grow-only device allocation (scratch vectors grow, never shrink: mesh.cpp:377-395)
    T* ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= cap)
            return hipSuccess;
        if (ptr)
            (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T));
        if (e == hipSuccess)
            cap = n;
        return e;
    }
    // like reserve, but the first `keep` elements survive (pool growth: the mirror is appended to, not rebuilt);
    // capacity grows by half so that steady appends do not reallocate every frame
    hipError_t grow(size_t n, size_t keep, hipStream_t stream)
    {
        if (n <= cap)
            return hipSuccess;
        const size_t want = std::max(n, cap + cap / 2);
        T* fresh = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&fresh), want * sizeof(T));
        if (e != hipSuccess)
            return e;
        if (ptr && keep) {
            e = hipMemcpyAsync(fresh, ptr, std::min(keep, cap) * sizeof(T), hipMemcpyDeviceToDevice, stream);
            if (e == hipSuccess)
                e = hipStreamSynchronize(stream);
            if (e != hipSuccess) {
                (void)hipFree(fresh);
                return e;
            }
        }
        if (ptr)
            (void)hipFree(ptr);
        ptr = fresh;
        cap = want;
        return hipSuccess;
    }
    void release()
    {
        if (ptr)
            (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

template <typename T>
struct PinnedBuf {
    T* ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= cap)
            return hipSuccess;
        if (ptr)
            (void)hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess)
            cap = n;
        return e;
    }
    hipError_t grow(size_t n, size_t keep)  // the caller has drained every async copy that reads this buffer
    {
        if (n <= cap)
            return hipSuccess;
        const size_t want = std::max(n, cap + cap / 2);
        T* fresh = nullptr;
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&fresh), want * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess)
            return e;
        if (ptr && keep)
            memcpy(fresh, ptr, std::min(keep, cap) * sizeof(T));
        if (ptr)
            (void)hipHostFree(ptr);
        ptr = fresh;
        cap = want;
        return hipSuccess;
    }
    void release()
    {
        if (ptr)
            (void)hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

struct DirtyRange {
    uint32_t lo = UINT32_MAX, hi = 0;
    bool any() const { return lo < hi; }
    void add(uint32_t first, uint32_t count)
    {
        lo = std::min(lo, first);
        hi = std::max(hi, first + count);
    }
    void clear()
    {
        lo = UINT32_MAX;
        hi = 0;
    }
};

// One field of a bound pool: element i lives at ptr + i * stride. An AoS pool binds every field with the component
// stride and its offset folded into ptr; column (SoA) storage binds each field with its own array and element size.
struct Column {
    const uint8_t* ptr = nullptr;
    size_t stride = 0;
    const uint8_t* at(size_t i) const { return ptr + i * stride; }
    uint32_t u32(size_t i) const
    {
        uint32_t v;
        memcpy(&v, ptr + i * stride, 4);
        return v;
    }
    const float* f32(size_t i) const { return reinterpret_cast<const float*>(ptr + i * stride); }
    uint8_t u8(size_t i) const { return ptr[i * stride]; }
};

struct TransformBinding {
    Column entity, parent, position, scale, rotation, self_active, ancestors_active, model_with_ancestors;
    uint32_t occupancy = 0;
    const uint32_t* entity_to_transform = nullptr;
    uint32_t entity_capacity = 0;
    bool bound = false;
};

struct PoolState {
    Column entity, is_enabled, aabb_min, aabb_max;
    uint8_t* is_visible = nullptr;  // write-back target (NULL: none), element i at is_visible + i * is_visible_stride
    size_t is_visible_stride = 0;
    uint32_t occupancy = 0;
    bool bound = false;
    bool need_full = false;
    uint32_t mapping = kMapGeneral;  // MeshMapping, chosen at full gather (kMapExact may only be demoted afterwards)
    DirtyRange dirty;
    // spatial mirror order (empty = slot order): perm[j] = pool slot held by mirror entry j, inv = its inverse
    std::vector<uint32_t> perm, inv;
    DeviceBuf<uint32_t> d_orig;  // perm on the device: emit reports original pool slots
    // GV_CONFIG_BLOCK_BOUNDS: per-workgroup world boxes, valid for (bounds_xf_epoch, bounds_epoch)
    DeviceBuf<float4> d_blk_lo, d_blk_hi;
    uint32_t mirrored = 0, appended = 0;  // entries the mirror holds / of those, appended (unsorted) since the last full build
    uint64_t epoch = 1, bounds_epoch = 0, bounds_xf_epoch = 0;  // epoch: bumped whenever this pool's mirror changes
    uint64_t seen_epoch = 0, seen_xf_epoch = 0;                 // state at this pool's previous gv_cull
    bool changed_prev = false;                                  // ... and whether it had changed then too (dynamic pool)
    // device mirror + pinned staging
    DeviceBuf<float4> d_a;
    DeviceBuf<float2> d_b;
    DeviceBuf<uint32_t> d_link;
    PinnedBuf<float4> h_a;
    PinnedBuf<float2> h_b;
    PinnedBuf<uint32_t> h_link;
};

struct ViewState {
    DeviceBuf<unsigned long long> mask;
    DeviceBuf<uint32_t> chunk_count, chunk_count2, chunk_offset, draw_count;
    uint32_t count_parity = 0;  // which totals buffer the next cull adds into (see launch_emit self_prefix)
    uint32_t stale_chunks[2] = {0, 0};  // entries of each totals buffer that may be non-zero right now

    DeviceBuf<uint8_t> is_visible;
    DeviceBuf<uint32_t> visible_idx;
    DeviceBuf<float> baked_model, distance_sq;
    // gv_sort: alternate record set + radix-sort scratch (allocated on first use)
    DeviceBuf<uint32_t> alt_idx, sort_keys[2], sort_vals[2], sort_hist;
    DeviceBuf<float> alt_model, alt_dist;
    PinnedBuf<uint32_t> h_visible_idx, h_draw_count;
    PinnedBuf<float> h_baked_model, h_distance_sq;
    PinnedBuf<uint8_t> h_is_visible, h_is_visible_mirror;
    uint32_t pool_id = 0, occupancy = 0;
    bool main_pass = false, emitted = false, valid = false;
};

struct PendingEvent {
    hipEvent_t start, stop;
    int kernel;
};

}  // namespace

struct GvCtx {
    GvConfig config{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;

    TransformBinding xf;
    bool xf_need_full = false;
    bool sweep_with_cull_mfma = true;
    bool hiz_level1_virtual = false;  // decided in gv_hiz_build: even sizes whose first six levels take the fused kernel
    bool hiz_level1_stored = false;   // ... and whether gv_hiz_read_level has materialised it since the last build
    uint64_t xf_epoch = 1;  // bumped whenever the transform mirror changes
    uint32_t xf_mirrored = 0, xf_appended = 0;  // as PoolState::mirrored / appended, for the transform pool
    DeviceBuf<uint8_t> d_raw;      // raw AoS bytes of a dirty slot range (device-side gather path)
    DirtyRange staging_stale;      // slots whose host staging entries lag behind the device (written by that path)
    bool device_gather = getenv("GV_NO_DEVICE_GATHER") == nullptr;  // turned off after a failed page-lock, or by the env
    DeviceBuf<uint8_t> d_examined;  // block-bounds statistics of the LAST bounded cull: 1 byte per workgroup
    uint64_t bounds_blocks_total = 0;
    bool sweep_with_cull = false;  // GV_SWEEP_WITH_CULL requested: the next gv_cull also writes the world matrices
    bool xf_links_dirty = false;  // a ranged GV_DIRTY_HIERARCHY: parent links changed -> re-validate depth / cycles
    DirtyRange xf_dirty;
    DeviceBuf<float4> d_xa, d_xb;
    DeviceBuf<float2> d_xc;
    DeviceBuf<uint8_t> d_xflags;
    DeviceBuf<unsigned long long> d_xactive;  // bit-plane of kXfActive, derived on the device
    DeviceBuf<uint32_t> d_xparent;
    PinnedBuf<float4> h_xa, h_xb;
    PinnedBuf<float2> h_xc;
    PinnedBuf<uint8_t> h_xflags;
    PinnedBuf<uint32_t> h_xparent;
    uint32_t max_depth = 0;
    // spatial mirror order of the transform pool (empty = slot order)
    std::vector<uint32_t> xperm, xinv;
    DeviceBuf<uint32_t> d_xinv;  // slot -> mirror entry, for gv_get_world
    // scratch of the scattered (dirty-range) upload path
    PinnedBuf<uint32_t> sc_idx, sc_u32;
    PinnedBuf<float4> sc_a, sc_b;
    PinnedBuf<float2> sc_c;
    PinnedBuf<uint8_t> sc_u8;
    DeviceBuf<uint32_t> dsc_idx, dsc_u32;
    DeviceBuf<float4> dsc_a, dsc_b;
    DeviceBuf<float2> dsc_c;
    DeviceBuf<uint8_t> dsc_u8;

    PoolState pools[GV_MAX_POOLS];
    ViewState views[GV_MAX_VIEWS];

    // world-matrix cache
    DeviceBuf<float4> d_world;
    bool world_valid = false;

    // Hi-Z
    DeviceBuf<float> d_depth;
    const float* depth_ptr = nullptr;  // d_depth.ptr or caller's device memory
    DeviceBuf<float2> d_mips;
    DeviceBuf<uint64_t> d_mip_offset;
    uint32_t hiz_w = 0, hiz_h = 0, hiz_mips = 0;
    uint32_t mip_w[GV_MAX_MIPS]{}, mip_h[GV_MAX_MIPS]{};
    uint64_t mip_off[GV_MAX_MIPS]{};
    bool hiz_valid = false;
    bool hiz_nested = false;  // every level bounds all the texels it covers (see HizDevice::nested)

    // profiling
    std::vector<PendingEvent> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
    GvStats stats{};

    int fail(int code, const char* fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        error = buf;
        return code;
    }
    int hip_fail(hipError_t e, const char* what)
    {
        return fail(e == hipErrorOutOfMemory ? GV_E_OOM : GV_E_HIP, "%s: %s", what, hipGetErrorString(e));
    }
};

#define GV_HIP(ctx, call)                                   \
    do {                                                    \
        hipError_t e__ = (call);                            \
        if (e__ != hipSuccess)                              \
            return (ctx)->hip_fail(e__, #call);             \
    } while (0)

namespace {

// roctx ranges named after the reference's profiler zones (SET_CPU_ZONE_SCOPED / SET_GPU_DEBUG_LABEL:
// "Meshes Prepare" source/system/render/mesh.cpp:334, "Meshes Sort" :267, "HiZ Downsample" hiz.cpp:146), so a
// rocprofv3 --marker-trace timeline reads like the engine's Tracy capture.
struct ZoneScope {
    explicit ZoneScope(const char* name) { roctxRangePushA(name); }
    ~ZoneScope() { roctxRangePop(); }
};

// ---- profiling events ----
struct KernelTimer {
    GvCtx* ctx;
    int kernel;
    hipEvent_t start = nullptr, stop = nullptr;
    bool on = false;
    KernelTimer(GvCtx* c, int k) : ctx(c), kernel(k)
    {
        ctx->stats.launches[k]++;
        if (!(ctx->config.flags & GV_CONFIG_PROFILE_EVENTS))
            return;
        if ((ctx->config.flags & GV_CONFIG_PROFILE_CULL_ONLY) && k != GV_K_CULL)
            return;
        if (!ctx->free_events.empty()) {
            start = ctx->free_events.back().first;
            stop = ctx->free_events.back().second;
            ctx->free_events.pop_back();
        } else if (hipEventCreate(&start) != hipSuccess || hipEventCreate(&stop) != hipSuccess) {
            return;
        }
        on = hipEventRecord(start, ctx->stream) == hipSuccess;
    }
    ~KernelTimer()
    {
        if (!on)
            return;
        (void)hipEventRecord(stop, ctx->stream);
        ctx->pending.push_back({start, stop, kernel});
    }
};

void drain_events(GvCtx* ctx)
{
    for (auto& p : ctx->pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess)
            ctx->stats.device_ms[p.kernel] += ms;
        ctx->free_events.emplace_back(p.start, p.stop);
    }
    ctx->pending.clear();
}

// ---- host gather: AoS component pools -> SoA staging ----
template <typename F>
void parallel_ranges(uint32_t first, uint32_t count, F&& fn)
{
    const uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
    // 16 threads: measured on the 256-thread box, 64 made the 10 M gather slower (60 vs 34 ms)
    const uint32_t threads = count < (1u << 16) ? 1u : std::min(hw, 16u);
    if (threads == 1) {
        fn(first, first + count);
        return;
    }
    const uint32_t per = (count + threads - 1) / threads;
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < threads; t++) {
        const uint32_t lo = first + std::min(count, per * t), hi = first + std::min(count, per * (t + 1));
        if (lo < hi)
            pool.emplace_back([=, &fn] { fn(lo, hi); });
    }
    fn(first, first + std::min(count, per));
    for (auto& th : pool)
        th.join();
}

inline uint32_t entity_slot(const TransformBinding& xf, uint32_t entity)
{
    if (entity == 0 || entity >= xf.entity_capacity)
        return kSlotNone;
    const uint32_t s = xf.entity_to_transform[entity];
    return (s == GV_NONE || s >= xf.occupancy) ? kSlotNone : s;
}

// ---- spatial mirror order ---------------------------------------------------------------------------
// The mirror does not have to keep pool order. At a full rebuild the transform entries are ordered by the
// Morton code of their ROOT ancestor's position (a whole tree shares one code and stays contiguous, ancestors
// before descendants when the pool had them so), and every mesh pool follows its transforms. Neighbouring lanes
// then see neighbouring pieces of screen: the Hi-Z texel gathers and the emit gather hit the same sectors
// (measured on a pre-sorted scene: cull 180 -> 157 us, emit 34 -> 22 us at 10 M entities). Every output goes
// back through the permutation (visible_idx, isVisible, gv_get_world), so callers only ever see pool slots.
inline uint32_t xslot_to_mirror(const GvCtx* ctx, uint32_t slot)
{
    return (slot == kSlotNone || ctx->xinv.empty()) ? slot : ctx->xinv[slot];
}

// stable LSD radix sort of `order` by 30-bit keys[order[k]] (3 passes of 10 bits), multi-threaded: every thread
// owns one contiguous chunk of the input, histograms it, and scatters it to offsets derived from the
// (digit, thread) prefix — stable because chunks keep their relative order inside each digit.
void radix_order(const std::vector<uint32_t>& keys, std::vector<uint32_t>& order)
{
    const size_t n = order.size();
    const uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
    const uint32_t threads = n < (1u << 16) ? 1u : std::min(hw, 16u);
    const size_t per = (n + threads - 1) / threads;
    std::vector<uint32_t> tmp(n);
    std::vector<size_t> hist((size_t)threads * 1024);
    auto run = [&](auto&& fn) {
        std::vector<std::thread> pool;
        for (uint32_t t = 1; t < threads; t++)
            pool.emplace_back([&, t] { fn(t); });
        fn(0u);
        for (auto& th : pool)
            th.join();
    };
    for (int pass = 0; pass < 3; pass++) {
        const int shift = pass * 10;
        std::fill(hist.begin(), hist.end(), 0);
        run([&](uint32_t t) {
            size_t* h = hist.data() + (size_t)t * 1024;
            for (size_t k = std::min(n, per * t), e = std::min(n, per * (t + 1)); k < e; k++)
                h[(keys[order[k]] >> shift) & 1023u]++;
        });
        size_t sum = 0;
        for (uint32_t d = 0; d < 1024; d++)
            for (uint32_t t = 0; t < threads; t++) {
                const size_t c = hist[(size_t)t * 1024 + d];
                hist[(size_t)t * 1024 + d] = sum;
                sum += c;
            }
        run([&](uint32_t t) {
            size_t* h = hist.data() + (size_t)t * 1024;
            for (size_t k = std::min(n, per * t), e = std::min(n, per * (t + 1)); k < e; k++) {
                const uint32_t v = order[k];
                tmp[h[(keys[v] >> shift) & 1023u]++] = v;
            }
        });
        order.swap(tmp);
    }
}

int build_transform_order(GvCtx* ctx)
{
    const TransformBinding& xf = ctx->xf;
    const uint32_t n = xf.occupancy;
    ctx->xperm.clear();
    ctx->xinv.clear();
    if ((ctx->config.flags & GV_CONFIG_KEEP_SLOT_ORDER) || n < 2)
        return GV_OK;
    // root ancestor of every slot. Fast path: every slot walks its own chain on the gather threads (chains are short);
    // a chain longer than kWalkCap — very deep, or a cycle — sends the whole pool through the serial memoised walk,
    // which also reports cycles (transform.cpp:137-143).
    std::vector<uint32_t> root(n, UINT32_MAX);
    constexpr uint32_t kWalkCap = 1u << 12;
    std::atomic<bool> capped{false};
    parallel_ranges(0, n, [&](uint32_t a, uint32_t b) {
        for (uint32_t s = a; s < b && !capped.load(std::memory_order_relaxed); s++) {
            uint32_t cur = s, steps = 0;
            for (;;) {
                const uint32_t ps = entity_slot(xf, xf.parent.u32(cur));
                if (ps == kSlotNone)
                    break;
                cur = ps;
                if (++steps > kWalkCap) {
                    capped.store(true, std::memory_order_relaxed);
                    break;
                }
            }
            root[s] = cur;
        }
    });
    if (capped) {
        std::fill(root.begin(), root.end(), UINT32_MAX);
        std::vector<uint32_t> path;
        for (uint32_t s = 0; s < n; s++) {
            if (root[s] != UINT32_MAX)
                continue;
            path.clear();
            uint32_t cur = s;
            for (;;) {
                if (root[cur] != UINT32_MAX) {
                    cur = root[cur];
                    break;
                }
                path.push_back(cur);
                if (path.size() > n)
                    return ctx->fail(GV_E_ARG, "transform hierarchy has a cycle through slot %u", s);
                const uint32_t ps = entity_slot(xf, xf.parent.u32(cur));
                if (ps == kSlotNone)
                    break;
                cur = ps;
            }
            for (uint32_t v : path)
                root[v] = cur;
        }
    }
    // bounding box of the live roots (per-thread partial boxes; min / max are order-independent)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    {
        std::mutex merge;
        parallel_ranges(0, n, [&](uint32_t a, uint32_t b) {
            float tl[3] = {INFINITY, INFINITY, INFINITY}, th[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t s = a; s < b; s++) {
                if (root[s] != s || !xf.entity.u32(s))
                    continue;
                const float* pos = xf.position.f32(s);
                for (int k = 0; k < 3; k++)
                    if (std::isfinite(pos[k])) {
                        tl[k] = std::min(tl[k], pos[k]);
                        th[k] = std::max(th[k], pos[k]);
                    }
            }
            std::lock_guard<std::mutex> g(merge);
            for (int k = 0; k < 3; k++) {
                lo[k] = std::min(lo[k], tl[k]);
                hi[k] = std::max(hi[k], th[k]);
            }
        });
    }
    auto spread = [](uint32_t v) {  // 10 bits -> every third bit
        v = (v | (v << 16)) & 0x030000FFu;
        v = (v | (v << 8)) & 0x0300F00Fu;
        v = (v | (v << 4)) & 0x030C30C3u;
        v = (v | (v << 2)) & 0x09249249u;
        return v;
    };
    std::vector<uint32_t> code(n);
    parallel_ranges(0, n, [&](uint32_t a, uint32_t b) {
        for (uint32_t s = a; s < b; s++) {
            if (!xf.entity.u32(s)) {
                code[s] = 0x3FFFFFFFu;  // free slots last
                continue;
            }
            const float* pos = xf.position.f32(root[s]);
            uint32_t q[3];
            for (int k = 0; k < 3; k++) {
                const float ext = hi[k] - lo[k];
                const float f = (ext > 0.0f && std::isfinite(pos[k])) ? (pos[k] - lo[k]) / ext : 0.0f;
                q[k] = (uint32_t)std::min(1023.0f, std::max(0.0f, f * 1024.0f));
            }
            code[s] = spread(q[0]) | (spread(q[1]) << 1) | (spread(q[2]) << 2);
        }
    });
    ctx->xperm.resize(n);
    for (uint32_t s = 0; s < n; s++)
        ctx->xperm[s] = s;
    radix_order(code, ctx->xperm);
    ctx->xinv.resize(n);
    parallel_ranges(0, n, [&](uint32_t a, uint32_t b) {
        for (uint32_t j = a; j < b; j++)
            ctx->xinv[ctx->xperm[j]] = j;  // a permutation: every write lands on its own element
    });
    return GV_OK;
}

void build_mesh_order(GvCtx* ctx, PoolState& p)
{
    p.perm.clear();
    p.inv.clear();
    const uint32_t n = p.occupancy;
    if (ctx->xinv.empty() || n < 2)
        return;
    // 1:1 pools (mesh slot i <-> transform slot i, or no transform at all): the transform permutation IS the mesh
    // permutation — entries without a transform sort last on both sides, in slot order — so skip the second sort
    if (n == ctx->xf.occupancy) {
        std::atomic<bool> paired{true};
        parallel_ranges(0, n, [&](uint32_t a, uint32_t b) {
            for (uint32_t i = a; i < b && paired.load(std::memory_order_relaxed); i++) {
                const uint32_t slot = entity_slot(ctx->xf, p.entity.u32(i));
                const bool free_xf = !ctx->xf.entity.u32(i);
                if (!(slot == i || (slot == kSlotNone && free_xf)))
                    paired.store(false, std::memory_order_relaxed);
            }
        });
        if (paired) {
            p.perm = ctx->xperm;
            p.inv = ctx->xinv;
            return;
        }
    }
    std::vector<uint32_t> key(n);
    parallel_ranges(0, n, [&](uint32_t a, uint32_t b) {
        for (uint32_t i = a; i < b; i++) {
            const uint32_t slot = entity_slot(ctx->xf, p.entity.u32(i));
            key[i] = slot == kSlotNone ? 0x3FFFFFFFu : ctx->xinv[slot];  // < 2^28: fits the 30-bit sort key
        }
    });
    p.perm.resize(n);
    for (uint32_t i = 0; i < n; i++)
        p.perm[i] = i;
    radix_order(key, p.perm);
    p.inv.resize(n);
    for (uint32_t j = 0; j < n; j++)
        p.inv[p.perm[j]] = j;
}

// pool slot s -> SoA staging at mirror entry j
inline void gather_transform(GvCtx* ctx, uint32_t s, uint32_t j)
{
    const TransformBinding& xf = ctx->xf;
    const float* pos = xf.position.f32(s);
    const float* scl = xf.scale.f32(s);
    const float* rot = xf.rotation.f32(s);
    const uint32_t entity = xf.entity.u32(s);
    uint8_t flags = 0;
    if (entity)
        flags |= kXfLive;
    if (xf.self_active.u8(s) && xf.ancestors_active.u8(s))
        flags |= kXfActive;
    if (xf.model_with_ancestors.u8(s))
        flags |= kXfWithAncestors;
    ctx->h_xa.ptr[j] = make_float4(pos[0], pos[1], pos[2], scl[0]);
    ctx->h_xb.ptr[j] = make_float4(rot[0], rot[1], rot[2], rot[3]);
    ctx->h_xc.ptr[j] = make_float2(scl[1], scl[2]);
    ctx->h_xflags.ptr[j] = flags;
    ctx->h_xparent.ptr[j] = xslot_to_mirror(ctx, entity_slot(xf, xf.parent.u32(s)));
}

// AoS slots [lo, hi) -> SoA staging at their mirror entries
void gather_transforms(GvCtx* ctx, uint32_t lo, uint32_t hi)
{
    parallel_ranges(lo, hi - lo, [&](uint32_t a, uint32_t b) {
        for (uint32_t s = a; s < b; s++)
            gather_transform(ctx, s, xslot_to_mirror(ctx, s));
    });
}

void gather_meshes(GvCtx* ctx, PoolState& p, uint32_t lo, uint32_t hi)
{
    const TransformBinding& xf = ctx->xf;
    std::atomic<bool> demoted{false};
    parallel_ranges(lo, hi - lo, [&](uint32_t a, uint32_t b) {
        for (uint32_t i = a; i < b; i++) {
            const float* mn = p.aabb_min.f32(i);
            const float* mx = p.aabb_max.f32(i);
            const uint32_t entity = p.entity.u32(i);
            const uint32_t slot = xslot_to_mirror(ctx, entity_slot(xf, entity));  // Manager::tryGet<TransformComponent>  mesh.cpp:149
            const bool candidate = entity && p.is_enabled.u8(i) && slot != kSlotNone;
            const uint32_t j = p.inv.empty() ? i : p.inv[i];
            // A non-candidate entry (free slot, disabled, no transform) carries an empty box: the all(size <= 0)
            // filter (mesh.cpp:140-142) then rejects it without the kernel having to read link[] (kMapExact).
            p.h_a.ptr[j] = candidate ? make_float4(mn[0], mn[1], mn[2], mx[0]) : make_float4(0, 0, 0, 0);
            p.h_b.ptr[j] = candidate ? make_float2(mx[1], mx[2]) : make_float2(0, 0);
            p.h_link.ptr[j] = slot | (candidate ? kMeshCandidate : 0u);
            if (candidate && slot != j && p.mapping == kMapExact)
                demoted = true;  // an edited mesh no longer pairs with its own index
        }
    });
    if (demoted)
        p.mapping = kMapSpeculate;
}

// Longest parent chain (mirror entries); a cycle (the reference asserts against it, transform.cpp:137-143) is an error.
int compute_max_depth(GvCtx* ctx, uint32_t* out_depth)
{
    const uint32_t n = ctx->xf.occupancy;
    std::vector<uint32_t> depth(n, UINT32_MAX);
    uint32_t max_depth = 0;
    std::vector<uint32_t> stack;
    for (uint32_t s = 0; s < n; s++) {
        if (depth[s] != UINT32_MAX)
            continue;
        stack.clear();
        uint32_t cur = s;
        while (cur != kSlotNone && depth[cur] == UINT32_MAX) {
            stack.push_back(cur);
            if (stack.size() > n)
                return ctx->fail(GV_E_ARG, "transform hierarchy has a cycle through slot %u", s);
            depth[cur] = UINT32_MAX - 1;  // on the current path
            cur = ctx->h_xparent.ptr[cur];
            if (cur != kSlotNone && depth[cur] == UINT32_MAX - 1)
                return ctx->fail(GV_E_ARG, "transform hierarchy has a cycle through slot %u", cur);
        }
        uint32_t d = cur == kSlotNone ? 0 : depth[cur] + 1;
        for (size_t k = stack.size(); k-- > 0;) {
            depth[stack[k]] = d;
            max_depth = std::max(max_depth, d);
            d++;
        }
    }
    *out_depth = max_depth;
    return GV_OK;
}

// contiguous mirror entries [lo, hi)
int upload_transforms(GvCtx* ctx, uint32_t lo, uint32_t hi)
{
    const size_t n = hi - lo;
    GV_HIP(ctx, hipMemcpyAsync(ctx->d_xa.ptr + lo, ctx->h_xa.ptr + lo, n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    GV_HIP(ctx, hipMemcpyAsync(ctx->d_xb.ptr + lo, ctx->h_xb.ptr + lo, n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    GV_HIP(ctx, hipMemcpyAsync(ctx->d_xc.ptr + lo, ctx->h_xc.ptr + lo, n * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
    GV_HIP(ctx, hipMemcpyAsync(ctx->d_xflags.ptr + lo, ctx->h_xflags.ptr + lo, n, hipMemcpyHostToDevice, ctx->stream));
    GV_HIP(ctx, hipMemcpyAsync(ctx->d_xparent.ptr + lo, ctx->h_xparent.ptr + lo, n * 4, hipMemcpyHostToDevice, ctx->stream));
    ctx->stats.upload_bytes += n * 45;
    return GV_OK;
}

// The bound transform fields as one array of structs, if that is what they are: equal strides, every field inside
// one stride-sized window. (Column bindings with separate arrays are gathered on the host.)
bool aos_transform_layout(const TransformBinding& xf, const uint8_t** base, AosTransformLayout* L)
{
    const Column* cols[7] = {&xf.entity, &xf.position, &xf.scale, &xf.rotation, &xf.self_active, &xf.ancestors_active,
                             &xf.model_with_ancestors};
    const uint32_t width[7] = {4, 12, 12, 16, 1, 1, 1};
    const size_t stride = xf.entity.stride;
    const uint8_t* lo = xf.entity.ptr;
    for (const Column* c : cols) {
        if (c->stride != stride || !c->ptr)
            return false;
        lo = std::min(lo, c->ptr);
    }
    uint32_t off[7];
    for (int k = 0; k < 7; k++) {
        const size_t o = (size_t)(cols[k]->ptr - lo);
        if (o + width[k] > stride)
            return false;
        off[k] = (uint32_t)o;
    }
    *base = lo;
    *L = AosTransformLayout{(uint32_t)stride, off[0], off[1], off[2], off[3], off[4], off[5], off[6]};
    return true;
}

// GV_DIRTY_TRANSFORM over slots [lo, hi) of an AoS pool, device side: page-lock just that span of the caller's pool for
// the duration of the copy (measured on the MI355X box: 7 ms per 800 MB to lock, then 57 GB/s instead of 10 GB/s
// from pageable memory), copy the raw components, gather on the device. The host staging of those slots goes stale
// and is refreshed only if a host path needs it later. Returns GV_E_STATE when the path is not applicable.
int upload_transforms_device(GvCtx* ctx, uint32_t lo, uint32_t hi)
{
    const uint8_t* base = nullptr;
    AosTransformLayout L{};
    if (!ctx->device_gather || !aos_transform_layout(ctx->xf, &base, &L))
        return GV_E_STATE;
    const uint32_t count = hi - lo;
    const size_t bytes = (size_t)count * L.stride;
    void* span = const_cast<uint8_t*>(base) + (size_t)lo * L.stride;
    if (ctx->d_raw.reserve(bytes) != hipSuccess)
        return GV_E_STATE;
    bool locked_here = true;
    const hipError_t lock = hipHostRegister(span, bytes, hipHostRegisterDefault);
    if (lock == hipErrorHostMemoryAlreadyRegistered) {
        (void)hipGetLastError();
        locked_here = false;  // the caller keeps its pools in pinned memory already: copy straight from it
    } else if (lock != hipSuccess) {
        (void)hipGetLastError();
        ctx->device_gather = false;  // no page-locking here (memlock limit, exotic memory): host gathers from now on
        return GV_E_STATE;
    }
    hipError_t e = hipMemcpyAsync(ctx->d_raw.ptr, span, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = launch_aos_transforms(ctx->d_raw.ptr, L, lo, count, ctx->xinv.empty() ? nullptr : ctx->d_xinv.ptr, ctx->d_xa.ptr,
                                  ctx->d_xb.ptr, ctx->d_xc.ptr, ctx->d_xflags.ptr, ctx->stream);
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);  // the span is unlocked (and may be freed by its owner) after this
    if (locked_here)
        (void)hipHostUnregister(span);
    if (e != hipSuccess || e2 != hipSuccess)
        return ctx->fail(GV_E_HIP, "device-side transform gather: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    ctx->staging_stale.add(lo, count);
    ctx->stats.upload_bytes += bytes;
    return GV_OK;
}

// Host paths read the staging arrays: bring stale entries (written on the device only) up to date first.
void refresh_stale_staging(GvCtx* ctx)
{
    if (!ctx->staging_stale.any())
        return;
    const uint32_t lo = ctx->staging_stale.lo, hi = std::min(ctx->staging_stale.hi, ctx->xf.occupancy);
    if (lo < hi)
        gather_transforms(ctx, lo, hi);
    ctx->staging_stale.clear();
}

// Dense re-mirror of the dirty slots [lo, hi) of a pool whose mirror is mostly dirty: walk the MIRROR in chunks, gather
// the dirty entries of a chunk, enqueue the chunk's upload, go on gathering — the DMA of one chunk runs under the host
// gather of the next (gather-all-then-upload-all costs their sum).
int regather_transforms_pipelined(GvCtx* ctx, uint32_t lo, uint32_t hi)
{
    const uint32_t n = ctx->xf.occupancy;
    constexpr uint32_t kChunk = 1u << 19;
    for (uint32_t j0 = 0; j0 < n; j0 += kChunk) {
        const uint32_t j1 = std::min(n, j0 + kChunk);
        parallel_ranges(j0, j1 - j0, [&](uint32_t a, uint32_t b) {
            for (uint32_t j = a; j < b; j++) {
                const uint32_t s = ctx->xperm.empty() ? j : ctx->xperm[j];
                if (s >= lo && s < hi)
                    gather_transform(ctx, s, j);
            }
        });
        const int rc = upload_transforms(ctx, j0, j1);
        if (rc != GV_OK)
            return rc;
    }
    return GV_OK;
}

int upload_meshes(GvCtx* ctx, PoolState& p, uint32_t lo, uint32_t hi)
{
    const size_t n = hi - lo;
    GV_HIP(ctx, hipMemcpyAsync(p.d_a.ptr + lo, p.h_a.ptr + lo, n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    GV_HIP(ctx, hipMemcpyAsync(p.d_b.ptr + lo, p.h_b.ptr + lo, n * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
    GV_HIP(ctx, hipMemcpyAsync(p.d_link.ptr + lo, p.h_link.ptr + lo, n * 4, hipMemcpyHostToDevice, ctx->stream));
    ctx->stats.upload_bytes += n * 28;
    return GV_OK;
}

int reserve_scatter(GvCtx* ctx, size_t n)
{
    GV_HIP(ctx, ctx->sc_idx.reserve(n));
    GV_HIP(ctx, ctx->sc_u32.reserve(n));
    GV_HIP(ctx, ctx->sc_a.reserve(n));
    GV_HIP(ctx, ctx->sc_b.reserve(n));
    GV_HIP(ctx, ctx->sc_c.reserve(n));
    GV_HIP(ctx, ctx->sc_u8.reserve(n));
    GV_HIP(ctx, ctx->dsc_idx.reserve(n));
    GV_HIP(ctx, ctx->dsc_u32.reserve(n));
    GV_HIP(ctx, ctx->dsc_a.reserve(n));
    GV_HIP(ctx, ctx->dsc_b.reserve(n));
    GV_HIP(ctx, ctx->dsc_c.reserve(n));
    GV_HIP(ctx, ctx->dsc_u8.reserve(n));
    return GV_OK;
}

// one stream of a scattered packet: host packet -> device packet -> dst[idx[k]] = packet[k]
template <typename T>
int scatter_stream(GvCtx* ctx, const T* host_packet, T* device_packet, T* dst, uint32_t n)
{
    GV_HIP(ctx, hipMemcpyAsync(device_packet, host_packet, (size_t)n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    GV_HIP(ctx, launch_scatter(ctx->dsc_idx.ptr, n, device_packet, dst, (uint32_t)sizeof(T), ctx->stream));
    return GV_OK;
}

// Dirty pool slots [lo, hi) of a permuted mirror land on scattered entries: ship them as one compact packet
// {entry, record} and scatter on the device.
int upload_transforms_scattered(GvCtx* ctx, uint32_t lo, uint32_t hi)
{
    const uint32_t n = hi - lo;
    int rc = reserve_scatter(ctx, n);
    if (rc != GV_OK)
        return rc;
    parallel_ranges(0, n, [&](uint32_t a, uint32_t b) {  // random reads of the staging arrays: spread over the cores
        for (uint32_t k = a; k < b; k++) {
            const uint32_t j = ctx->xinv[lo + k];
            ctx->sc_idx.ptr[k] = j;
            ctx->sc_a.ptr[k] = ctx->h_xa.ptr[j];
            ctx->sc_b.ptr[k] = ctx->h_xb.ptr[j];
            ctx->sc_c.ptr[k] = ctx->h_xc.ptr[j];
            ctx->sc_u8.ptr[k] = ctx->h_xflags.ptr[j];
            ctx->sc_u32.ptr[k] = ctx->h_xparent.ptr[j];
        }
    });
    GV_HIP(ctx, hipMemcpyAsync(ctx->dsc_idx.ptr, ctx->sc_idx.ptr, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = scatter_stream(ctx, ctx->sc_a.ptr, ctx->dsc_a.ptr, ctx->d_xa.ptr, n)) != GV_OK) return rc;
    if ((rc = scatter_stream(ctx, ctx->sc_b.ptr, ctx->dsc_b.ptr, ctx->d_xb.ptr, n)) != GV_OK) return rc;
    if ((rc = scatter_stream(ctx, ctx->sc_c.ptr, ctx->dsc_c.ptr, ctx->d_xc.ptr, n)) != GV_OK) return rc;
    if ((rc = scatter_stream(ctx, ctx->sc_u8.ptr, ctx->dsc_u8.ptr, ctx->d_xflags.ptr, n)) != GV_OK) return rc;
    if ((rc = scatter_stream(ctx, ctx->sc_u32.ptr, ctx->dsc_u32.ptr, ctx->d_xparent.ptr, n)) != GV_OK) return rc;
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the packet buffers are reused by the next dirty range
    ctx->stats.upload_bytes += (size_t)n * (4 + 45);
    return GV_OK;
}

int upload_meshes_scattered(GvCtx* ctx, PoolState& p, uint32_t lo, uint32_t hi)
{
    const uint32_t n = hi - lo;
    int rc = reserve_scatter(ctx, n);
    if (rc != GV_OK)
        return rc;
    parallel_ranges(0, n, [&](uint32_t a, uint32_t b) {
        for (uint32_t k = a; k < b; k++) {
            const uint32_t j = p.inv[lo + k];
            ctx->sc_idx.ptr[k] = j;
            ctx->sc_a.ptr[k] = p.h_a.ptr[j];
            ctx->sc_c.ptr[k] = p.h_b.ptr[j];
            ctx->sc_u32.ptr[k] = p.h_link.ptr[j];
        }
    });
    GV_HIP(ctx, hipMemcpyAsync(ctx->dsc_idx.ptr, ctx->sc_idx.ptr, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = scatter_stream(ctx, ctx->sc_a.ptr, ctx->dsc_a.ptr, p.d_a.ptr, n)) != GV_OK) return rc;
    if ((rc = scatter_stream(ctx, ctx->sc_c.ptr, ctx->dsc_c.ptr, p.d_b.ptr, n)) != GV_OK) return rc;
    if ((rc = scatter_stream(ctx, ctx->sc_u32.ptr, ctx->dsc_u32.ptr, p.d_link.ptr, n)) != GV_OK) return rc;
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stats.upload_bytes += (size_t)n * (4 + 28);
    return GV_OK;
}

// Pool growth (entities created since the last sync): the new slots [n0, n1) are appended to the mirror as entries
// [n0, n1) — identity on the tail of the permutation — instead of rebuilding it; they stay outside the spatial order
// until the next full build, which sync_mirror schedules once the unsorted tail passes 1/8 of the pool.
int grow_transforms(GvCtx* ctx, uint32_t n0, uint32_t n1)
{
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GV_HIP(ctx, ctx->d_xa.grow(n1, n0, ctx->stream));
    GV_HIP(ctx, ctx->d_xb.grow(n1, n0, ctx->stream));
    GV_HIP(ctx, ctx->d_xc.grow(n1, n0, ctx->stream));
    GV_HIP(ctx, ctx->d_xflags.grow(n1, n0, ctx->stream));
    GV_HIP(ctx, ctx->d_xparent.grow(n1, n0, ctx->stream));
    GV_HIP(ctx, ctx->d_xactive.grow((size_t)n1 / 64 + 1, 0, ctx->stream));  // re-derived below
    GV_HIP(ctx, ctx->h_xa.grow(n1, n0));
    GV_HIP(ctx, ctx->h_xb.grow(n1, n0));
    GV_HIP(ctx, ctx->h_xc.grow(n1, n0));
    GV_HIP(ctx, ctx->h_xflags.grow(n1, n0));
    GV_HIP(ctx, ctx->h_xparent.grow(n1, n0));
    if (!ctx->xperm.empty()) {
        ctx->xperm.resize(n1);
        ctx->xinv.resize(n1);
        for (uint32_t s = n0; s < n1; s++)
            ctx->xperm[s] = ctx->xinv[s] = s;
        GV_HIP(ctx, ctx->d_xinv.grow(n1, n0, ctx->stream));
        GV_HIP(ctx, hipMemcpyAsync(ctx->d_xinv.ptr + n0, ctx->xinv.data() + n0, (size_t)(n1 - n0) * 4, hipMemcpyHostToDevice, ctx->stream));
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pageable source
    }
    gather_transforms(ctx, n0, n1);
    bool chained = ctx->max_depth != 0;
    for (uint32_t j = n0; j < n1 && !chained; j++)
        chained = ctx->h_xparent.ptr[j] != kSlotNone;
    if (chained) {  // new slots with parents (or a pool that already has chains): depth / cycle check over the links
        uint32_t depth = 0;
        const int rc = compute_max_depth(ctx, &depth);
        if (rc != GV_OK) {
            ctx->xf_need_full = true;
            return rc;
        }
        ctx->max_depth = depth;
    }
    const int rc = upload_transforms(ctx, n0, n1);
    if (rc != GV_OK)
        return rc;
    GV_HIP(ctx, launch_pack_active(ctx->d_xflags.ptr, n1, ctx->d_xactive.ptr, ctx->stream));
    ctx->xf_mirrored = n1;
    ctx->xf_appended += n1 - n0;
    ctx->world_valid = false;
    ctx->xf_epoch++;
    return GV_OK;
}

int grow_meshes(GvCtx* ctx, PoolState& p, uint32_t n0, uint32_t n1)
{
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GV_HIP(ctx, p.d_a.grow(n1, n0, ctx->stream));
    GV_HIP(ctx, p.d_b.grow(n1, n0, ctx->stream));
    GV_HIP(ctx, p.d_link.grow(n1, n0, ctx->stream));
    GV_HIP(ctx, p.h_a.grow(n1, n0));
    GV_HIP(ctx, p.h_b.grow(n1, n0));
    GV_HIP(ctx, p.h_link.grow(n1, n0));
    if (!p.perm.empty()) {
        p.perm.resize(n1);
        p.inv.resize(n1);
        for (uint32_t i = n0; i < n1; i++)
            p.perm[i] = p.inv[i] = i;
        GV_HIP(ctx, p.d_orig.grow(n1, n0, ctx->stream));
        GV_HIP(ctx, hipMemcpyAsync(p.d_orig.ptr + n0, p.perm.data() + n0, (size_t)(n1 - n0) * 4, hipMemcpyHostToDevice, ctx->stream));
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pageable source
    }
    gather_meshes(ctx, p, n0, n1);
    const int rc = upload_meshes(ctx, p, n0, n1);
    if (rc != GV_OK)
        return rc;
    p.mirrored = n1;
    p.appended += n1 - n0;
    p.epoch++;
    return GV_OK;
}

struct PhaseTimer {  // GV_DEBUG_TIMING=1: prints the host phases of a mirror build
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    bool on = getenv("GV_DEBUG_TIMING") != nullptr;
    void lap(const char* what)
    {
        if (!on)
            return;
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[gv] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

int sync_mirror(GvCtx* ctx)
{
    PhaseTimer phase;
    if (!ctx->xf.bound)
        return ctx->fail(GV_E_STATE, "gv_sync: no transform pool bound");
    GV_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t n = ctx->xf.occupancy;
    bool staged = false;
    const bool spatial = !(ctx->config.flags & GV_CONFIG_KEEP_SLOT_ORDER);
    if (!ctx->xf_need_full && n > ctx->xf_mirrored && spatial &&
        ((uint64_t)ctx->xf_appended + (n - ctx->xf_mirrored)) * 8 > n && n >= 1024)
        ctx->xf_need_full = true;  // too much of the pool sits in the unsorted tail: re-order everything
    if (ctx->xf_need_full) {
        // staging is about to be rewritten: make sure earlier async uploads have drained
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        staged = true;
        const size_t cap = std::max<size_t>(n, 1);
        GV_HIP(ctx, ctx->d_xa.reserve(cap));
        GV_HIP(ctx, ctx->d_xb.reserve(cap));
        GV_HIP(ctx, ctx->d_xc.reserve(cap));
        GV_HIP(ctx, ctx->d_xflags.reserve(cap));
        GV_HIP(ctx, ctx->d_xactive.reserve(cap / 64 + 1));
        GV_HIP(ctx, ctx->d_xparent.reserve(cap));
        GV_HIP(ctx, ctx->h_xa.reserve(cap));
        GV_HIP(ctx, ctx->h_xb.reserve(cap));
        GV_HIP(ctx, ctx->h_xc.reserve(cap));
        GV_HIP(ctx, ctx->h_xflags.reserve(cap));
        GV_HIP(ctx, ctx->h_xparent.reserve(cap));
        phase.lap("reserve transforms");
        int rc = build_transform_order(ctx);
        if (rc != GV_OK)
            return rc;
        phase.lap("transform order");
        if (n) {
            gather_transforms(ctx, 0, n);
            phase.lap("gather transforms");
            uint32_t depth = 0;
            rc = compute_max_depth(ctx, &depth);
            if (rc != GV_OK)
                return rc;
            ctx->max_depth = depth;
            phase.lap("max depth");
            rc = upload_transforms(ctx, 0, n);
            if (rc != GV_OK)
                return rc;
            if (!ctx->xinv.empty()) {
                GV_HIP(ctx, ctx->d_xinv.reserve(cap));
                GV_HIP(ctx, hipMemcpyAsync(ctx->d_xinv.ptr, ctx->xinv.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
                GV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pageable source
            }
        } else {
            ctx->max_depth = 0;
        }
        if (n)
            GV_HIP(ctx, launch_pack_active(ctx->d_xflags.ptr, n, ctx->d_xactive.ptr, ctx->stream));
        ctx->xf_need_full = false;
        ctx->xf_links_dirty = false;
        ctx->xf_dirty.clear();
        ctx->staging_stale.clear();  // everything was gathered afresh
        ctx->world_valid = false;
        ctx->xf_epoch++;
        ctx->xf_mirrored = n;
        ctx->xf_appended = 0;
        // transform entries may have moved: every mesh pool's slot column must be re-resolved
        for (auto& p : ctx->pools)
            if (p.bound)
                p.need_full = true;
    } else {
      if (n > ctx->xf_mirrored) {
        staged = true;
        const int rc = grow_transforms(ctx, ctx->xf_mirrored, n);
        if (rc != GV_OK)
            return rc;
      }
      if (ctx->xf_dirty.any()) {
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        staged = true;
        const uint32_t lo = ctx->xf_dirty.lo, hi = std::min(ctx->xf_dirty.hi, n);
        if (lo < hi) {
            int rc = GV_E_STATE;
            if (!ctx->xf_links_dirty && hi - lo >= 2048)
                rc = upload_transforms_device(ctx, lo, hi);  // raw AoS span + device gather (falls through if not applicable)
            if (rc == GV_OK) {
                // done on the device
            } else if (rc != GV_E_STATE) {
                return rc;
            } else if ((size_t)(hi - lo) * 2 > n) {
                refresh_stale_staging(ctx);  // this path re-uploads every entry from the staging arrays
                rc = regather_transforms_pipelined(ctx, lo, hi);  // most of the pool: dense, chunked, DMA under gather
            } else {
                gather_transforms(ctx, lo, hi);
                rc = ctx->xinv.empty() ? upload_transforms(ctx, lo, hi) : upload_transforms_scattered(ctx, lo, hi);
            }
            if (rc != GV_OK)
                return rc;
            GV_HIP(ctx, launch_pack_active(ctx->d_xflags.ptr, n, ctx->d_xactive.ptr, ctx->stream));
            if (ctx->xf_links_dirty) {  // setParent (transform.cpp:130-195): chains changed length, maybe closed a cycle
                uint32_t depth = 0;
                rc = compute_max_depth(ctx, &depth);
                if (rc != GV_OK) {
                    ctx->xf_need_full = true;  // the mirror now holds a cyclic link: rebuild once the caller has fixed it
                    return rc;
                }
                ctx->max_depth = depth;
            }
        }
        ctx->xf_links_dirty = false;
        ctx->xf_dirty.clear();
        ctx->world_valid = false;
        ctx->xf_epoch++;
      }
    }
    for (auto& p : ctx->pools) {
        if (!p.bound)
            continue;
        if (!p.need_full && p.occupancy > p.mirrored && spatial &&
            ((uint64_t)p.appended + (p.occupancy - p.mirrored)) * 8 > p.occupancy && p.occupancy >= 1024)
            p.need_full = true;
        if (!p.need_full && p.occupancy > p.mirrored) {
            staged = true;
            const int rc = grow_meshes(ctx, p, p.mirrored, p.occupancy);
            if (rc != GV_OK)
                return rc;
        }
        if (p.need_full) {
            if (!staged) {
                GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
                staged = true;
            }
            const size_t cap = std::max<size_t>(p.occupancy, 1);
            GV_HIP(ctx, p.d_a.reserve(cap));
            GV_HIP(ctx, p.d_b.reserve(cap));
            GV_HIP(ctx, p.d_link.reserve(cap));
            GV_HIP(ctx, p.h_a.reserve(cap));
            GV_HIP(ctx, p.h_b.reserve(cap));
            GV_HIP(ctx, p.h_link.reserve(cap));
            phase.lap("upload transforms + reserve");
            build_mesh_order(ctx, p);
            phase.lap("mesh order");
            p.mapping = kMapGeneral;
            if (p.occupancy) {
                gather_meshes(ctx, p, 0, p.occupancy);
                phase.lap("gather meshes");
                // how do mesh entries pair with transform entries? (speed only: every mapping is handled)
                size_t candidates = 0, own = 0;
                for (uint32_t i = 0; i < p.occupancy; i++) {
                    const uint32_t link = p.h_link.ptr[i];
                    if (link & kMeshCandidate) {
                        candidates++;
                        own += (link & kSlotMask) == i;
                    }
                }
                p.mapping = own == candidates ? kMapExact : (own * 10 >= candidates * 9 ? kMapSpeculate : kMapGeneral);
                int rc = upload_meshes(ctx, p, 0, p.occupancy);
                if (rc != GV_OK)
                    return rc;
                if (!p.perm.empty()) {
                    GV_HIP(ctx, p.d_orig.reserve(cap));
                    GV_HIP(ctx, hipMemcpyAsync(p.d_orig.ptr, p.perm.data(), (size_t)p.occupancy * 4, hipMemcpyHostToDevice, ctx->stream));
                    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pageable source
                }
            }
            phase.lap("mapping + upload meshes");
            p.need_full = false;
            p.dirty.clear();
            p.epoch++;
            p.mirrored = p.occupancy;
            p.appended = 0;
        } else if (p.dirty.any()) {
            if (!staged) {
                GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
                staged = true;
            }
            const uint32_t lo = p.dirty.lo, hi = std::min(p.dirty.hi, p.occupancy);
            if (lo < hi) {
                gather_meshes(ctx, p, lo, hi);
                int rc;
                if (p.inv.empty())
                    rc = upload_meshes(ctx, p, lo, hi);
                else if ((size_t)(hi - lo) * 2 > p.occupancy)
                    rc = upload_meshes(ctx, p, 0, p.occupancy);
                else
                    rc = upload_meshes_scattered(ctx, p, lo, hi);
                if (rc != GV_OK)
                    return rc;
            }
            p.dirty.clear();
            p.epoch++;
        }
    }
    return GV_OK;
}

TransformMirror xf_mirror(const GvCtx* ctx)
{
    TransformMirror m;
    m.a = ctx->d_xa.ptr;
    m.b = ctx->d_xb.ptr;
    m.c = ctx->d_xc.ptr;
    m.flags = ctx->d_xflags.ptr;
    m.active_bits = ctx->d_xactive.ptr;
    m.parent = ctx->d_xparent.ptr;
    m.count = ctx->xf.occupancy;
    m.max_depth = ctx->max_depth;
    return m;
}

// Frustum(viewProj) (mesh.cpp:815,867,869,900,902) — host side, once per view: Gribb-Hartmann rows of
// the column-major matrix for a [0,1] clip depth, normalised; degenerate planes (|n|^2 < 1e-12, the
// z >= 0 plane of the infinite reversed-Z projection) are dropped. DESIGN.md §"Canonical arithmetic".
void build_view_params(const GvView& v, ViewParams* out)
{
    float row[4][4];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++)
            row[r][c] = v.view_proj[c * 4 + r];
    float p[6][4];
    for (int c = 0; c < 4; c++) {
        p[0][c] = row[3][c] + row[0][c];
        p[1][c] = row[3][c] - row[0][c];
        p[2][c] = row[3][c] + row[1][c];
        p[3][c] = row[3][c] - row[1][c];
        p[4][c] = row[2][c];
        p[5][c] = row[3][c] - row[2][c];
    }
    memset(out, 0, sizeof(*out));
    for (int i = 0; i < 6; i++) {
        const float len2 = std::fmaf(p[i][2], p[i][2], std::fmaf(p[i][1], p[i][1], p[i][0] * p[i][0]));
        if (!(len2 >= 1e-12f))
            continue;
        const float inv = 1.0f / std::sqrt(len2);
        float* q = out->planes[out->plane_count++];
        for (int c = 0; c < 4; c++)
            q[c] = p[i][c] * inv;
    }
    for (int c = 0; c < 3; c++) {
        out->cam[c] = v.camera_position[c];
        out->cam_offset[c] = v.camera_offset[c];
    }
    memcpy(out->vp, v.view_proj, sizeof(out->vp));
    out->write_is_visible = v.shadow_pass < 0 ? 1u : 0u;  // isNotShadowPass  mesh.cpp:121
    out->use_hiz = v.use_hiz ? 1u : 0u;
    out->distance_2d = v.distance_2d ? 1u : 0u;
}

int reserve_view(GvCtx* ctx, ViewState& vs, uint32_t occupancy, bool emit)
{
    const size_t n = std::max<uint32_t>(occupancy, 1);
    const size_t blocks = (n + kCullBlock - 1) / kCullBlock;
    const size_t chunks = (n + kEmitChunk - 1) / kEmitChunk;
    GV_HIP(ctx, vs.mask.reserve(blocks * (kCullBlock / 64)));
    if (chunks > vs.chunk_count.cap) {
        GV_HIP(ctx, vs.chunk_count.reserve(chunks));
        GV_HIP(ctx, vs.chunk_count2.reserve(chunks));
        // the cull workgroups add into the totals; scan (or the self-prefixing emit) re-zeroes them; fresh ones start at zero
        GV_HIP(ctx, hipMemsetAsync(vs.chunk_count.ptr, 0, vs.chunk_count.cap * sizeof(uint32_t), ctx->stream));
        GV_HIP(ctx, hipMemsetAsync(vs.chunk_count2.ptr, 0, vs.chunk_count2.cap * sizeof(uint32_t), ctx->stream));
        vs.count_parity = 0;
        vs.stale_chunks[0] = vs.stale_chunks[1] = 0;
    }
    GV_HIP(ctx, vs.chunk_offset.reserve(chunks));
    GV_HIP(ctx, vs.draw_count.reserve(4));
    GV_HIP(ctx, vs.is_visible.reserve(n));
    GV_HIP(ctx, vs.h_draw_count.reserve(4));
    if (emit) {
        GV_HIP(ctx, vs.visible_idx.reserve(n));
        GV_HIP(ctx, vs.baked_model.reserve(n * 12));
        GV_HIP(ctx, vs.distance_sq.reserve(n));
    }
    return GV_OK;
}

ViewBuffers view_buffers(ViewState& vs)
{
    ViewBuffers b;
    b.mask = vs.mask.ptr;
    b.chunk_count = vs.count_parity ? vs.chunk_count2.ptr : vs.chunk_count.ptr;
    b.chunk_count_next = vs.count_parity ? vs.chunk_count.ptr : vs.chunk_count2.ptr;
    b.chunk_offset = vs.chunk_offset.ptr;
    b.draw_count = vs.draw_count.ptr;
    b.is_visible = vs.is_visible.ptr;
    b.visible_idx = vs.visible_idx.ptr;
    b.baked_model = vs.baked_model.ptr;
    b.distance_sq = vs.distance_sq.ptr;
    return b;
}

int hiz_reduce(GvCtx* ctx)
{
    ctx->hiz_level1_stored = false;
    ZoneScope zone("HiZ Downsample");
    KernelTimer timer(ctx, GV_K_HIZ);
    uint32_t k = 1;
    while (k < ctx->hiz_mips) {
        const uint32_t sw = ctx->mip_w[k - 1], sh = ctx->mip_h[k - 1];
        const float* src_d = k == 1 ? ctx->depth_ptr : nullptr;
        const float2* src_p = k == 1 ? nullptr : ctx->d_mips.ptr + ctx->mip_off[k - 1];
        if (sw % 64 == 0 && sh % 64 == 0 && k + 5 < ctx->hiz_mips) {
            HizFusedDst dst;
            for (int l = 0; l < 6; l++)
                dst.level[l] = ctx->d_mips.ptr + ctx->mip_off[k + l];
            if (k == 1 && ctx->hiz_level1_virtual)
                dst.level[0] = nullptr;  // not written: 3/4 of the pyramid's bytes (gv_hiz_read_level materialises it on demand)
            GV_HIP(ctx, launch_hiz_fused(src_d, src_p, dst, sw, sh, ctx->stream));
            k += 6;
        } else {
            GV_HIP(ctx, launch_hiz_level(src_d, src_p, ctx->d_mips.ptr + ctx->mip_off[k], sw, sh, ctx->mip_w[k],
                                         ctx->mip_h[k], ctx->config.hiz_rule, ctx->stream));
            k += 1;
        }
    }
    return GV_OK;
}

}  // namespace

// ================================================================================================
// C-ABI
// ================================================================================================
extern "C" {

uint32_t gv_abi_version(void) { return GV_ABI_VERSION; }

const char* gv_last_error(const GvCtx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int gv_create(const GvConfig* config, GvCtx** out_ctx)
{
    if (!out_ctx) {
        g_create_error = "gv_create: out_ctx is NULL";
        return GV_E_ARG;
    }
    *out_ctx = nullptr;
    GvConfig cfg{};
    cfg.struct_size = sizeof(GvConfig);
    if (config) {
        if (config->struct_size != sizeof(GvConfig)) {
            g_create_error = "gv_create: GvConfig.struct_size mismatch (ABI)";
            return GV_E_ARG;
        }
        cfg = *config;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_error = std::string("gv_create: no HIP device (") + hipGetErrorString(e) +
                         "); libgarden_vis has no CPU fallback";
        return GV_E_NODEVICE;
    }
    if (cfg.device < 0 || cfg.device >= count) {
        g_create_error = "gv_create: device ordinal out of range";
        return GV_E_ARG;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, cfg.device);
    if (e != hipSuccess) {
        g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        return GV_E_HIP;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("gv_create: device is ") + prop.gcnArchName +
                         ", kernels are built for gfx950 only (no fallback)";
        return GV_E_NODEVICE;
    }
    GvCtx* ctx = new (std::nothrow) GvCtx();
    if (!ctx) {
        g_create_error = "gv_create: out of host memory";
        return GV_E_OOM;
    }
    ctx->config = cfg;
    ctx->device = cfg.device;
    if ((e = hipSetDevice(cfg.device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_error = std::string("gv_create: ") + hipGetErrorString(e);
        delete ctx;
        return GV_E_HIP;
    }
    *out_ctx = ctx;
    return GV_OK;
}

void gv_destroy(GvCtx* ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    drain_events(ctx);
    for (auto& ev : ctx->free_events) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    ctx->d_xa.release(); ctx->d_xb.release(); ctx->d_xc.release(); ctx->d_xflags.release(); ctx->d_xactive.release(); ctx->d_xparent.release();
    ctx->h_xa.release(); ctx->h_xb.release(); ctx->h_xc.release(); ctx->h_xflags.release(); ctx->h_xparent.release();
    for (auto& p : ctx->pools) {
        p.d_a.release(); p.d_b.release(); p.d_link.release(); p.h_a.release(); p.h_b.release(); p.h_link.release(); p.d_orig.release();
    }
    for (auto& v : ctx->views) {
        v.mask.release(); v.chunk_count.release(); v.chunk_count2.release(); v.chunk_offset.release(); v.draw_count.release();
        v.is_visible.release(); v.visible_idx.release(); v.baked_model.release(); v.distance_sq.release();
        v.alt_idx.release(); v.alt_model.release(); v.alt_dist.release(); v.sort_hist.release();
        for (int k = 0; k < 2; k++) { v.sort_keys[k].release(); v.sort_vals[k].release(); }
        v.h_visible_idx.release(); v.h_draw_count.release(); v.h_baked_model.release();
        v.h_distance_sq.release(); v.h_is_visible.release(); v.h_is_visible_mirror.release();
    }
    ctx->d_world.release();
    ctx->d_xinv.release(); ctx->sc_idx.release(); ctx->sc_u32.release(); ctx->sc_a.release(); ctx->sc_b.release(); ctx->sc_c.release(); ctx->sc_u8.release();
    ctx->dsc_idx.release(); ctx->dsc_u32.release(); ctx->dsc_a.release(); ctx->dsc_b.release(); ctx->dsc_c.release(); ctx->dsc_u8.release();
    ctx->d_depth.release(); ctx->d_mips.release(); ctx->d_mip_offset.release();
    if (ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int gv_transform_bind(GvCtx* ctx, const void* base, size_t stride, uint32_t occupancy,
                      const GvTransformLayout* layout, const uint32_t* entity_to_transform,
                      uint32_t entity_capacity)
{
    if (!ctx)
        return GV_E_ARG;
    if (!layout || (occupancy && !base) || (entity_capacity && !entity_to_transform))
        return ctx->fail(GV_E_ARG, "gv_transform_bind: NULL argument");
    if (occupancy >= kSlotNone)
        return ctx->fail(GV_E_ARG, "gv_transform_bind: occupancy %u exceeds the 28-bit slot range", occupancy);
    const uint32_t need = std::max({layout->position, layout->scale, layout->rotation}) + 16u;
    if (occupancy && stride < need)
        return ctx->fail(GV_E_ARG, "gv_transform_bind: stride %zu smaller than layout (%u)", stride, need);
    GvTransformColumns c{};
    const uint8_t* b = static_cast<const uint8_t*>(base);
    auto col = [&](uint32_t offset) { return GvColumn{b ? b + offset : nullptr, (uint32_t)stride}; };
    c.entity = col(layout->entity);
    c.parent = col(layout->parent);
    c.position = col(layout->position);
    c.scale = col(layout->scale);
    c.rotation = col(layout->rotation);
    c.self_active = col(layout->self_active);
    c.ancestors_active = col(layout->ancestors_active);
    c.model_with_ancestors = col(layout->model_with_ancestors);
    return gv_transform_bind_columns(ctx, &c, occupancy, entity_to_transform, entity_capacity);
}

int gv_transform_bind_columns(GvCtx* ctx, const GvTransformColumns* columns, uint32_t occupancy,
                              const uint32_t* entity_to_transform, uint32_t entity_capacity)
{
    if (!ctx)
        return GV_E_ARG;
    if (!columns || (entity_capacity && !entity_to_transform))
        return ctx->fail(GV_E_ARG, "gv_transform_bind_columns: NULL argument");
    if (occupancy >= kSlotNone)
        return ctx->fail(GV_E_ARG, "gv_transform_bind_columns: occupancy %u exceeds the 28-bit slot range", occupancy);
    const GvColumn* all[8] = {&columns->entity, &columns->parent, &columns->position, &columns->scale, &columns->rotation,
                              &columns->self_active, &columns->ancestors_active, &columns->model_with_ancestors};
    const uint32_t width[8] = {4, 4, 12, 12, 16, 1, 1, 1};
    for (int k = 0; k < 8; k++)
        if (occupancy && (!all[k]->data || all[k]->stride < width[k]))
            return ctx->fail(GV_E_ARG, "gv_transform_bind_columns: column %d is NULL or its stride is below %u bytes", k, width[k]);
    auto col = [](const GvColumn& g) { return Column{static_cast<const uint8_t*>(g.data), g.stride}; };
    // first bind or a pool that shrank: rebuild. A pool that GREW keeps its mirror; gv_sync appends the new slots.
    const bool moved = !ctx->xf.bound || occupancy < ctx->xf.occupancy;
    ctx->xf.entity = col(columns->entity);
    ctx->xf.parent = col(columns->parent);
    ctx->xf.position = col(columns->position);
    ctx->xf.scale = col(columns->scale);
    ctx->xf.rotation = col(columns->rotation);
    ctx->xf.self_active = col(columns->self_active);
    ctx->xf.ancestors_active = col(columns->ancestors_active);
    ctx->xf.model_with_ancestors = col(columns->model_with_ancestors);
    ctx->xf.occupancy = occupancy;
    ctx->xf.entity_to_transform = entity_to_transform;
    ctx->xf.entity_capacity = entity_capacity;
    ctx->xf.bound = true;
    if (moved)
        ctx->xf_need_full = true;  // first bind or pool resized: rebuild the mirror
    return GV_OK;
}

int gv_pool_bind(GvCtx* ctx, uint32_t pool_id, void* base, size_t stride, uint32_t occupancy,
                 const GvMeshLayout* layout)
{
    if (!ctx)
        return GV_E_ARG;
    if (pool_id >= GV_MAX_POOLS || !layout || (occupancy && !base))
        return ctx->fail(GV_E_ARG, "gv_pool_bind: bad argument (pool_id %u)", pool_id);
    if (occupancy >= kSlotNone)
        return ctx->fail(GV_E_ARG, "gv_pool_bind: occupancy %u exceeds the 28-bit slot range", occupancy);
    const uint32_t need = std::max(layout->aabb_min, layout->aabb_max) + 12u;
    if (occupancy && stride < need)
        return ctx->fail(GV_E_ARG, "gv_pool_bind: stride %zu smaller than layout (%u)", stride, need);
    GvMeshColumns c{};
    uint8_t* b = static_cast<uint8_t*>(base);
    auto col = [&](uint32_t offset) { return GvColumn{b ? b + offset : nullptr, (uint32_t)stride}; };
    c.entity = col(layout->entity);
    c.is_enabled = col(layout->is_enabled);
    c.aabb_min = col(layout->aabb_min);
    c.aabb_max = col(layout->aabb_max);
    c.is_visible = b ? b + layout->is_visible : nullptr;
    c.is_visible_stride = (uint32_t)stride;
    return gv_pool_bind_columns(ctx, pool_id, &c, occupancy);
}

int gv_pool_bind_columns(GvCtx* ctx, uint32_t pool_id, const GvMeshColumns* columns, uint32_t occupancy)
{
    if (!ctx)
        return GV_E_ARG;
    if (pool_id >= GV_MAX_POOLS || !columns)
        return ctx->fail(GV_E_ARG, "gv_pool_bind_columns: bad argument (pool_id %u)", pool_id);
    if (occupancy >= kSlotNone)
        return ctx->fail(GV_E_ARG, "gv_pool_bind_columns: occupancy %u exceeds the 28-bit slot range", occupancy);
    const GvColumn* all[4] = {&columns->entity, &columns->is_enabled, &columns->aabb_min, &columns->aabb_max};
    const uint32_t width[4] = {4, 1, 12, 12};
    for (int k = 0; k < 4; k++)
        if (occupancy && (!all[k]->data || all[k]->stride < width[k]))
            return ctx->fail(GV_E_ARG, "gv_pool_bind_columns: column %d is NULL or its stride is below %u bytes", k, width[k]);
    if (columns->is_visible && columns->is_visible_stride == 0)
        return ctx->fail(GV_E_ARG, "gv_pool_bind_columns: is_visible_stride is 0");
    auto col = [](const GvColumn& g) { return Column{static_cast<const uint8_t*>(g.data), g.stride}; };
    PoolState& p = ctx->pools[pool_id];
    const bool moved = !p.bound || occupancy < p.occupancy;
    p.entity = col(columns->entity);
    p.is_enabled = col(columns->is_enabled);
    p.aabb_min = col(columns->aabb_min);
    p.aabb_max = col(columns->aabb_max);
    p.is_visible = static_cast<uint8_t*>(columns->is_visible);
    p.is_visible_stride = columns->is_visible_stride;
    p.occupancy = occupancy;
    p.bound = true;
    if (moved)
        p.need_full = true;
    return GV_OK;
}

int gv_mark_dirty(GvCtx* ctx, uint32_t kind, uint32_t first, uint32_t count)
{
    if (!ctx)
        return GV_E_ARG;
    switch (kind) {
    case GV_DIRTY_TRANSFORM:
        ctx->xf_dirty.add(first, count);
        return GV_OK;
    case GV_DIRTY_HIERARCHY:
        if (count == 0) {
            ctx->xf_need_full = true;  // everything: entities came or went; the mirror is re-ordered too
        } else {
            ctx->xf_dirty.add(first, count);  // re-parented slots: links re-gathered in place, order kept
            ctx->xf_links_dirty = true;
        }
        return GV_OK;
    case GV_DIRTY_MESH: {
        const uint32_t pool = first >> 28, lo = first & kSlotMask;
        if (pool >= GV_MAX_POOLS)
            return ctx->fail(GV_E_ARG, "gv_mark_dirty: pool id %u", pool);
        ctx->pools[pool].dirty.add(lo, count);
        return GV_OK;
    }
    default:
        return ctx->fail(GV_E_ARG, "gv_mark_dirty: unknown kind %u", kind);
    }
}

int gv_hierarchy_rebuild(GvCtx* ctx)
{
    if (!ctx)
        return GV_E_ARG;
    ctx->xf_need_full = true;
    return sync_mirror(ctx);
}

int gv_sync(GvCtx* ctx)
{
    if (!ctx)
        return GV_E_ARG;
    return sync_mirror(ctx);
}

int gv_cull(GvCtx* ctx, uint32_t pool_id, const GvView* views, uint32_t view_count)
{
    if (!ctx)
        return GV_E_ARG;
    if (pool_id >= GV_MAX_POOLS || !views || view_count == 0 || view_count > GV_MAX_VIEWS)
        return ctx->fail(GV_E_ARG, "gv_cull: bad argument (pool %u, %u views)", pool_id, view_count);
    ZoneScope zone("Meshes Prepare");
    PoolState& p = ctx->pools[pool_id];
    if (!p.bound || !ctx->xf.bound)
        return ctx->fail(GV_E_STATE, "gv_cull: pools not bound");
    int rc = sync_mirror(ctx);
    if (rc != GV_OK)
        return rc;
    GV_HIP(ctx, hipSetDevice(ctx->device));
    const MeshMirror mesh{p.d_a.ptr, p.d_b.ptr, p.d_link.ptr, p.occupancy, p.mapping, p.perm.empty() ? nullptr : p.d_orig.ptr};
    const TransformMirror xf = xf_mirror(ctx);
    HizDevice hz{};
    for (uint32_t v = 0; v < view_count; v++) {
        if (views[v].use_hiz && !ctx->hiz_valid)
            return ctx->fail(GV_E_STATE, "gv_cull: view %u asks for Hi-Z but gv_hiz_build has not run", v);
    }
    if (ctx->hiz_valid) {
        hz.depth = ctx->depth_ptr;
        hz.mips = ctx->d_mips.ptr;
        hz.mip_offset = ctx->d_mip_offset.ptr;
        hz.width = ctx->hiz_w;
        hz.height = ctx->hiz_h;
        hz.mip_count = ctx->hiz_mips;
        hz.nested = ctx->hiz_nested ? 1u : 0u;
        hz.level1_virtual = ctx->hiz_level1_virtual ? 1u : 0u;
    }
    // Views that share cameraPosition (the main camera and its shadow cascades: mesh.cpp:809-843 passes the same
    // cameraPosition to every prepareMeshes) are culled in ONE pass over the streams; Hi-Z only on view 0.
    bool batched = view_count > 1 && view_count <= kMaxBatchViews && p.occupancy > 0;
    for (uint32_t v = 1; v < view_count && batched; v++)
        batched = memcmp(views[v].camera_position, views[0].camera_position, 12) == 0 && !views[v].use_hiz;
    ViewParams vps[GV_MAX_VIEWS];
    ViewBuffers vbs[GV_MAX_VIEWS];
    const uint32_t chunks = (p.occupancy + kEmitChunk - 1) / kEmitChunk;
    for (uint32_t v = 0; v < view_count; v++) {
        ViewState& vs = ctx->views[v];
        const bool emit = views[v].emit_records != 0;
        rc = reserve_view(ctx, vs, p.occupancy, emit);
        if (rc != GV_OK)
            return rc;
        vs.pool_id = pool_id;
        vs.occupancy = p.occupancy;
        vs.main_pass = views[v].shadow_pass < 0;
        vs.emitted = emit;
        vs.valid = true;
        build_view_params(views[v], &vps[v]);
        vbs[v] = view_buffers(vs);
        if (p.occupancy == 0)
            GV_HIP(ctx, hipMemsetAsync(vs.draw_count.ptr, 0, 4, ctx->stream));
    }
    // GV_SWEEP_WITH_CULL: an exactly paired pool takes the fused MFMA sweep + cull for its first view; anything else
    // gets the same results from the plain MFMA sweep followed by the ordinary cull
    const bool sweep_requested = ctx->sweep_with_cull;
    ctx->sweep_with_cull = false;
    const bool fused = sweep_requested && !batched && p.occupancy != 0 && mesh.mapping == kMapExact && mesh.count <= xf.count;
    if (sweep_requested) {
        GV_HIP(ctx, ctx->d_world.reserve((size_t)std::max(ctx->xf.occupancy, 1u) * 3));
        if (!fused) {
            KernelTimer t(ctx, GV_K_SWEEP);
            if (ctx->sweep_with_cull_mfma)
                GV_HIP(ctx, launch_sweep_mfma(xf, ctx->d_world.ptr, ctx->stream));
            else
                GV_HIP(ctx, launch_sweep_valu(xf, ctx->d_world.ptr, ctx->stream));
        }
        ctx->world_valid = true;
    }
    // GV_CONFIG_BLOCK_BOUNDS: workgroup boxes are (re)built when the mirror of this pool is clean, or has just changed
    // after a quiet frame; a pool that changes frame after frame (dynamic scene) is culled without them
    BlockBounds bounds;
    bool use_bounds = false;
    if ((ctx->config.flags & GV_CONFIG_BLOCK_BOUNDS) && p.occupancy != 0 && !fused) {
        const bool changed = p.seen_epoch != p.epoch || p.seen_xf_epoch != ctx->xf_epoch;
        bool current = p.bounds_epoch == p.epoch && p.bounds_xf_epoch == ctx->xf_epoch;
        if (!current && !(changed && p.changed_prev)) {
            const size_t nb = (p.occupancy + kCullBlock - 1) / kCullBlock;
            GV_HIP(ctx, p.d_blk_lo.reserve(nb));
            GV_HIP(ctx, p.d_blk_hi.reserve(nb));
            KernelTimer t(ctx, GV_K_SWEEP);  // accounted with the other per-change passes
            GV_HIP(ctx, launch_block_bounds(mesh, xf, p.d_blk_lo.ptr, p.d_blk_hi.ptr, ctx->stream));
            p.bounds_epoch = p.epoch;
            p.bounds_xf_epoch = ctx->xf_epoch;
            current = true;
        }
        p.changed_prev = changed;
        p.seen_epoch = p.epoch;
        p.seen_xf_epoch = ctx->xf_epoch;
        if (current) {
            GV_HIP(ctx, ctx->d_examined.reserve((p.occupancy + kCullBlock - 1) / kCullBlock));
            use_bounds = true;
        }
    }
    if (p.occupancy != 0) {
        if (use_bounds) {
            bounds.lo = p.d_blk_lo.ptr;
            bounds.hi = p.d_blk_hi.ptr;
            bounds.examined = ctx->d_examined.ptr;
            ctx->bounds_blocks_total = (p.occupancy + kCullBlock - 1) / kCullBlock;
        }
        if (batched) {
            KernelTimer t(ctx, GV_K_CULL);
            GV_HIP(ctx, launch_cull_multi(mesh, xf, hz, vps, vbs, view_count, ctx->stream, use_bounds ? &bounds : nullptr));
        }
        for (uint32_t v = 0; v < view_count; v++) {
            if (!batched) {
                KernelTimer t(ctx, GV_K_CULL);
                if (fused && v == 0)
                    GV_HIP(ctx, launch_sweep_cull(mesh, xf, hz, vps[v], vbs[v], ctx->d_world.ptr, ctx->sweep_with_cull_mfma, ctx->stream));
                else {
                    GV_HIP(ctx, launch_cull(mesh, xf, hz, vps[v], vbs[v], ctx->stream, use_bounds ? &bounds : nullptr));
                }
            }
            static const uint32_t self_max = getenv("GV_DEBUG_SELF_PREFIX_MAX") ? (uint32_t)atoi(getenv("GV_DEBUG_SELF_PREFIX_MAX")) : kSelfPrefixMaxChunks;
            if (ctx->views[v].emitted && chunks <= self_max) {
                // no scan launch: emit derives the chunk bases itself and leaves THIS totals buffer as it is; the
                // next cull of this view adds into the other one, which this emit has cleared
                ViewState& vs = ctx->views[v];
                const uint32_t cur = vs.count_parity, other = cur ^ 1u;
                KernelTimer t(ctx, GV_K_EMIT);
                GV_HIP(ctx, launch_emit(mesh, xf, vps[v], vbs[v], ctx->stream, true, std::max(chunks, vs.stale_chunks[other])));
                vs.stale_chunks[other] = 0;
                vs.stale_chunks[cur] = chunks;
                vs.count_parity = other;
                continue;
            }
            {
                KernelTimer t(ctx, GV_K_SCAN);
                GV_HIP(ctx, launch_scan(vbs[v], chunks, ctx->stream));
            }
            if (ctx->views[v].emitted) {
                KernelTimer t(ctx, GV_K_EMIT);
                GV_HIP(ctx, launch_emit(mesh, xf, vps[v], vbs[v], ctx->stream));
            }
        }
    }
    for (uint32_t v = view_count; v < GV_MAX_VIEWS; v++)
        ctx->views[v].valid = false;
    return GV_OK;
}

int gv_wait(GvCtx* ctx)
{
    if (!ctx)
        return GV_E_ARG;
    GV_HIP(ctx, hipSetDevice(ctx->device));
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drain_events(ctx);
    return GV_OK;
}

int gv_result_count(GvCtx* ctx, uint32_t view_index, uint32_t* draw_count)
{
    if (!ctx)
        return GV_E_ARG;
    if (view_index >= GV_MAX_VIEWS || !draw_count || !ctx->views[view_index].valid)
        return ctx->fail(GV_E_ARG, "gv_result_count: view %u has no results", view_index);
    ViewState& vs = ctx->views[view_index];
    GV_HIP(ctx, hipSetDevice(ctx->device));
    GV_HIP(ctx, hipMemcpyAsync(vs.h_draw_count.ptr, vs.draw_count.ptr, 4, hipMemcpyDeviceToHost, ctx->stream));
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drain_events(ctx);
    *draw_count = vs.h_draw_count.ptr[0];
    return GV_OK;
}

int gv_results_fetch(GvCtx* ctx, uint32_t view_index, int write_back, GvResult* out)
{
    if (!ctx)
        return GV_E_ARG;
    if (!out)
        return ctx->fail(GV_E_ARG, "gv_results_fetch: out is NULL");
    uint32_t count = 0;
    int rc = gv_result_count(ctx, view_index, &count);
    if (rc != GV_OK)
        return rc;
    ViewState& vs = ctx->views[view_index];
    memset(out, 0, sizeof(*out));
    out->draw_count = vs.emitted ? count : count;
    out->instance_count = count;  // default getReadyMeshesAsync returns 0/1 (render/mesh.hpp:142-146)
    if (vs.emitted && count) {
        GV_HIP(ctx, vs.h_visible_idx.reserve(vs.occupancy));
        GV_HIP(ctx, vs.h_baked_model.reserve((size_t)vs.occupancy * 12));
        GV_HIP(ctx, vs.h_distance_sq.reserve(vs.occupancy));
        GV_HIP(ctx, hipMemcpyAsync(vs.h_visible_idx.ptr, vs.visible_idx.ptr, (size_t)count * 4, hipMemcpyDeviceToHost, ctx->stream));
        GV_HIP(ctx, hipMemcpyAsync(vs.h_baked_model.ptr, vs.baked_model.ptr, (size_t)count * 48, hipMemcpyDeviceToHost, ctx->stream));
        GV_HIP(ctx, hipMemcpyAsync(vs.h_distance_sq.ptr, vs.distance_sq.ptr, (size_t)count * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    PoolState& pool = ctx->pools[vs.pool_id];
    const bool permuted = !pool.perm.empty() && pool.perm.size() == vs.occupancy;
    if (vs.main_pass && vs.occupancy) {
        GV_HIP(ctx, vs.h_is_visible.reserve(vs.occupancy));
        uint8_t* dst = vs.h_is_visible.ptr;
        if (permuted) {
            GV_HIP(ctx, vs.h_is_visible_mirror.reserve(vs.occupancy));
            dst = vs.h_is_visible_mirror.ptr;
        }
        GV_HIP(ctx, hipMemcpyAsync(dst, vs.is_visible.ptr, vs.occupancy, hipMemcpyDeviceToHost, ctx->stream));
    }
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (vs.main_pass && vs.occupancy && permuted) {  // mirror order -> pool-slot order
        const uint8_t* src = vs.h_is_visible_mirror.ptr;
        uint8_t* out_vis = vs.h_is_visible.ptr;
        const uint32_t* perm = pool.perm.data();
        parallel_ranges(0, vs.occupancy, [&](uint32_t a, uint32_t b) {
            for (uint32_t j = a; j < b; j++)
                out_vis[perm[j]] = src[j];
        });
    }
    if (vs.emitted && count) {
        out->visible_idx = vs.h_visible_idx.ptr;
        out->baked_model = vs.h_baked_model.ptr;
        out->distance_sq = vs.h_distance_sq.ptr;
    }
    if (vs.main_pass && vs.occupancy) {
        out->is_visible = vs.h_is_visible.ptr;
        if (write_back) {  // meshRenderView->isVisible = ...  mesh.cpp:144,152,161,166
            PoolState& p = ctx->pools[vs.pool_id];
            if (!p.bound || p.occupancy != vs.occupancy)
                return ctx->fail(GV_E_STATE, "gv_results_fetch: pool %u rebound since gv_cull", vs.pool_id);
            const uint8_t* src = vs.h_is_visible.ptr;
            if (p.is_visible)
                parallel_ranges(0, vs.occupancy, [&](uint32_t a, uint32_t b) {
                    for (uint32_t i = a; i < b; i++)
                        p.is_visible[(size_t)i * p.is_visible_stride] = src[i];
                });
        }
    }
    return GV_OK;
}

int gv_results_device(GvCtx* ctx, uint32_t view_index, GvDeviceResult* out)
{
    if (!ctx)
        return GV_E_ARG;
    if (view_index >= GV_MAX_VIEWS || !out || !ctx->views[view_index].valid)
        return ctx->fail(GV_E_ARG, "gv_results_device: view %u has no results", view_index);
    ViewState& vs = ctx->views[view_index];
    out->visible_idx = vs.emitted ? vs.visible_idx.ptr : nullptr;
    out->baked_model = vs.emitted ? vs.baked_model.ptr : nullptr;
    out->distance_sq = vs.emitted ? vs.distance_sq.ptr : nullptr;
    out->is_visible = vs.main_pass ? vs.is_visible.ptr : nullptr;
    out->draw_count = vs.draw_count.ptr;
    return GV_OK;
}

int gv_results_copy_idx_device(GvCtx* ctx, uint32_t view_index, void* dst_device, uint32_t capacity,
                               uint32_t index_base)
{
    if (!ctx)
        return GV_E_ARG;
    if (view_index >= GV_MAX_VIEWS || !dst_device || !ctx->views[view_index].valid || !ctx->views[view_index].emitted)
        return ctx->fail(GV_E_ARG, "gv_results_copy_idx_device: view %u has no emitted records", view_index);
    ViewState& vs = ctx->views[view_index];
    GV_HIP(ctx, hipSetDevice(ctx->device));
    GV_HIP(ctx, launch_copy_idx(vs.visible_idx.ptr, vs.draw_count.ptr, static_cast<uint32_t*>(dst_device), capacity,
                                index_base, ctx->stream));
    return GV_OK;
}

int gv_results_copy_shard_device(GvCtx* ctx, uint32_t view_index, void* dst_device, uint32_t capacity,
                                 uint32_t index_base)
{
    if (!ctx)
        return GV_E_ARG;
    if (view_index >= GV_MAX_VIEWS || !dst_device || !ctx->views[view_index].valid || !ctx->views[view_index].emitted)
        return ctx->fail(GV_E_ARG, "gv_results_copy_shard_device: view %u has no emitted records", view_index);
    ViewState& vs = ctx->views[view_index];
    GV_HIP(ctx, hipSetDevice(ctx->device));
    GV_HIP(ctx, launch_copy_shard(vs.visible_idx.ptr, vs.draw_count.ptr, static_cast<uint32_t*>(dst_device), capacity,
                                  index_base, ctx->stream));
    return GV_OK;
}

int gv_sort(GvCtx* ctx, uint32_t view_index, int descending)
{
    if (!ctx)
        return GV_E_ARG;
    if (view_index >= GV_MAX_VIEWS || !ctx->views[view_index].valid || !ctx->views[view_index].emitted)
        return ctx->fail(GV_E_ARG, "gv_sort: view %u has no emitted records", view_index);
    ZoneScope zone("Meshes Sort");
    ViewState& vs = ctx->views[view_index];
    if (vs.occupancy == 0)
        return GV_OK;
    GV_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = vs.occupancy;  // upper bound of draw_count, known without a readback
    const size_t nblocks = (n + 4095) / 4096;
    GV_HIP(ctx, vs.alt_idx.reserve(n));
    GV_HIP(ctx, vs.alt_model.reserve(n * 12));
    GV_HIP(ctx, vs.alt_dist.reserve(n));
    for (int k = 0; k < 2; k++) {
        GV_HIP(ctx, vs.sort_keys[k].reserve(n));
        GV_HIP(ctx, vs.sort_vals[k].reserve(n));
    }
    GV_HIP(ctx, vs.sort_hist.reserve(256 * nblocks + 256));
    SortBuffers b;
    b.count = vs.draw_count.ptr;
    b.idx_in = vs.visible_idx.ptr;
    b.model_in = vs.baked_model.ptr;
    b.dist_in = vs.distance_sq.ptr;
    b.idx_out = vs.alt_idx.ptr;
    b.model_out = vs.alt_model.ptr;
    b.dist_out = vs.alt_dist.ptr;
    for (int k = 0; k < 2; k++) {
        b.keys[k] = vs.sort_keys[k].ptr;
        b.vals[k] = vs.sort_vals[k].ptr;
    }
    b.hist = vs.sort_hist.ptr + 256;
    b.bin_total = vs.sort_hist.ptr;
    {
        KernelTimer t(ctx, GV_K_SORT);
        GV_HIP(ctx, launch_sort(b, (uint32_t)n, descending != 0, ctx->stream));
    }
    // the sorted records now live in the alternate set: swap it in
    std::swap(vs.visible_idx, vs.alt_idx);
    std::swap(vs.baked_model, vs.alt_model);
    std::swap(vs.distance_sq, vs.alt_dist);
    return GV_OK;
}

int gv_sweep(GvCtx* ctx, uint32_t mode)
{
    if (!ctx)
        return GV_E_ARG;
    if (mode == GV_SWEEP_WITH_CULL || mode == GV_SWEEP_WITH_CULL_VALU) {  // nothing to launch now: the next gv_cull carries the sweep
        ctx->sweep_with_cull = true;
        ctx->sweep_with_cull_mfma = mode == GV_SWEEP_WITH_CULL;
        return GV_OK;
    }
    if (mode != GV_SWEEP_VALU && mode != GV_SWEEP_MFMA)
        return ctx->fail(GV_E_ARG, "gv_sweep: unknown mode %u", mode);
    int rc = sync_mirror(ctx);
    if (rc != GV_OK)
        return rc;
    GV_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t n = ctx->xf.occupancy;
    GV_HIP(ctx, ctx->d_world.reserve((size_t)std::max(n, 1u) * 3));
    {
        KernelTimer t(ctx, GV_K_SWEEP);
        if (mode == GV_SWEEP_MFMA)
            GV_HIP(ctx, launch_sweep_mfma(xf_mirror(ctx), ctx->d_world.ptr, ctx->stream));
        else
            GV_HIP(ctx, launch_sweep_valu(xf_mirror(ctx), ctx->d_world.ptr, ctx->stream));
    }
    ctx->world_valid = true;
    return GV_OK;
}

int gv_get_world(GvCtx* ctx, uint32_t first, uint32_t count, float* out12)
{
    if (!ctx)
        return GV_E_ARG;
    if (!ctx->world_valid)
        return ctx->fail(GV_E_STATE, "gv_get_world: gv_sweep has not run since the last transform change");
    if (!out12 || (uint64_t)first + count > ctx->xf.occupancy)
        return ctx->fail(GV_E_ARG, "gv_get_world: range [%u, +%u) outside the pool", first, count);
    GV_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->xinv.empty()) {
        GV_HIP(ctx, hipMemcpyAsync(out12, ctx->d_world.ptr + (size_t)first * 3, (size_t)count * 48, hipMemcpyDeviceToHost, ctx->stream));
    } else if (count) {  // the cache is in mirror order: gather the requested pool slots on the device first
        GV_HIP(ctx, ctx->dsc_a.reserve((size_t)count * 3));
        GV_HIP(ctx, launch_gather_world(ctx->d_world.ptr, ctx->d_xinv.ptr, first, count, ctx->dsc_a.ptr, ctx->stream));
        GV_HIP(ctx, hipMemcpyAsync(out12, ctx->dsc_a.ptr, (size_t)count * 48, hipMemcpyDeviceToHost, ctx->stream));
    }
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drain_events(ctx);
    return GV_OK;
}

int gv_hiz_build(GvCtx* ctx, const float* depth, uint32_t width, uint32_t height, uint32_t mem_kind)
{
    if (!ctx)
        return GV_E_ARG;
    if (!depth || width == 0 || height == 0 || width > 32768 || height > 32768)
        return ctx->fail(GV_E_ARG, "gv_hiz_build: bad depth image %ux%u", width, height);
    GV_HIP(ctx, hipSetDevice(ctx->device));
    // calcMipCount(frameSize) hiz.cpp:27; sizes max(size / 2, 1) hiz.cpp:55
    uint32_t mips = 0;
    for (uint32_t m = std::max(width, height); m; m >>= 1)
        mips++;
    if (mips > GV_MAX_MIPS)
        return ctx->fail(GV_E_ARG, "gv_hiz_build: %u mips exceed GV_MAX_MIPS", mips);
    uint64_t off = 0;
    uint32_t cw = width, ch = height;
    for (uint32_t k = 0; k < mips; k++) {
        ctx->mip_w[k] = cw;
        ctx->mip_h[k] = ch;
        ctx->mip_off[k] = off;
        if (k >= 1)
            off += (uint64_t)cw * ch;
        cw = std::max(cw / 2, 1u);
        ch = std::max(ch / 2, 1u);
    }
    ctx->hiz_w = width;
    ctx->hiz_h = height;
    ctx->hiz_mips = mips;
    // The reference rule (hiz.frag:49-55) skips one texel of the extra row on odd heights, so a level is only
    // guaranteed to bound everything below it when no source level is odd, or under the conservative rule.
    ctx->hiz_nested = ctx->config.hiz_rule == GV_HIZ_RULE_CONSERVATIVE;
    if (!ctx->hiz_nested) {
        ctx->hiz_nested = true;
        for (uint32_t k = 0; k + 1 < mips; k++)
            if ((ctx->mip_w[k] > 1 && (ctx->mip_w[k] & 1u)) || (ctx->mip_h[k] > 1 && (ctx->mip_h[k] & 1u)))
                ctx->hiz_nested = false;
    }
    // level 1 stays virtual when the first six levels come from the fused kernel (sizes divisible by 64: plain 2x2 rule)
    ctx->hiz_level1_virtual = width % 64 == 0 && height % 64 == 0 && mips > 6 && getenv("GV_DEBUG_STORE_HIZ_LEVEL1") == nullptr;
    GV_HIP(ctx, ctx->d_mips.reserve(std::max<uint64_t>(off, 1)));
    GV_HIP(ctx, ctx->d_mip_offset.reserve(GV_MAX_MIPS));
    GV_HIP(ctx, hipMemcpyAsync(ctx->d_mip_offset.ptr, ctx->mip_off, sizeof(uint64_t) * GV_MAX_MIPS, hipMemcpyHostToDevice, ctx->stream));
    if (mem_kind == GV_MEM_DEVICE) {
        ctx->depth_ptr = depth;
    } else {
        GV_HIP(ctx, ctx->d_depth.reserve((size_t)width * height));
        GV_HIP(ctx, hipMemcpyAsync(ctx->d_depth.ptr, depth, (size_t)width * height * 4, hipMemcpyHostToDevice, ctx->stream));
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // caller's (pageable) buffer may go away
        ctx->depth_ptr = ctx->d_depth.ptr;
    }
    ctx->hiz_valid = true;
    return hiz_reduce(ctx);
}

int gv_hiz_rebuild(GvCtx* ctx)
{
    if (!ctx)
        return GV_E_ARG;
    if (!ctx->hiz_valid)
        return ctx->fail(GV_E_STATE, "gv_hiz_rebuild: no depth image resident");
    GV_HIP(ctx, hipSetDevice(ctx->device));
    return hiz_reduce(ctx);
}

int gv_hiz_mip_count(GvCtx* ctx, uint32_t* mip_count)
{
    if (!ctx)
        return GV_E_ARG;
    if (!ctx->hiz_valid || !mip_count)
        return ctx->fail(GV_E_STATE, "gv_hiz_mip_count: no pyramid");
    *mip_count = ctx->hiz_mips;
    return GV_OK;
}

int gv_hiz_read_level(GvCtx* ctx, uint32_t level, float* out_pairs, uint32_t* w, uint32_t* h)
{
    if (!ctx)
        return GV_E_ARG;
    if (!ctx->hiz_valid)
        return ctx->fail(GV_E_STATE, "gv_hiz_read_level: no pyramid");
    if (level == 0 || level >= ctx->hiz_mips || !out_pairs)
        return ctx->fail(GV_E_ARG, "gv_hiz_read_level: level %u (valid 1..%u)", level, ctx->hiz_mips - 1);
    GV_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)ctx->mip_w[level] * ctx->mip_h[level];
    if (level == 1 && ctx->hiz_level1_virtual && !ctx->hiz_level1_stored) {  // on demand: one generic level pass
        GV_HIP(ctx, launch_hiz_level(ctx->depth_ptr, nullptr, ctx->d_mips.ptr + ctx->mip_off[1], ctx->mip_w[0], ctx->mip_h[0],
                                     ctx->mip_w[1], ctx->mip_h[1], ctx->config.hiz_rule, ctx->stream));
        ctx->hiz_level1_stored = true;
    }
    GV_HIP(ctx, hipMemcpyAsync(out_pairs, ctx->d_mips.ptr + ctx->mip_off[level], n * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drain_events(ctx);
    if (w)
        *w = ctx->mip_w[level];
    if (h)
        *h = ctx->mip_h[level];
    return GV_OK;
}

int gv_stats(GvCtx* ctx, GvStats* out)
{
    if (!ctx || !out)
        return GV_E_ARG;
    if (!ctx->pending.empty()) {
        GV_HIP(ctx, hipSetDevice(ctx->device));
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        drain_events(ctx);
    }
    ctx->stats.bounds_blocks_total = ctx->bounds_blocks_total;
    ctx->stats.bounds_blocks_examined = 0;
    if (ctx->d_examined.ptr && ctx->bounds_blocks_total) {
        std::vector<uint8_t> flags(ctx->bounds_blocks_total);
        GV_HIP(ctx, hipSetDevice(ctx->device));
        GV_HIP(ctx, hipMemcpyAsync(flags.data(), ctx->d_examined.ptr, flags.size(), hipMemcpyDeviceToHost, ctx->stream));
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (uint8_t f : flags)
            ctx->stats.bounds_blocks_examined += f;
    }
    ctx->stats.max_depth = ctx->max_depth;
    ctx->stats.transform_count = ctx->xf.occupancy;
    for (uint32_t i = 0; i < GV_MAX_POOLS; i++)
        ctx->stats.mesh_count[i] = ctx->pools[i].bound ? ctx->pools[i].occupancy : 0;
    *out = ctx->stats;
    return GV_OK;
}

int gv_stats_reset(GvCtx* ctx)
{
    if (!ctx)
        return GV_E_ARG;
    if (!ctx->pending.empty()) {
        GV_HIP(ctx, hipSetDevice(ctx->device));
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        drain_events(ctx);
    }
    memset(ctx->stats.launches, 0, sizeof(ctx->stats.launches));
    memset(ctx->stats.device_ms, 0, sizeof(ctx->stats.device_ms));
    ctx->stats.upload_bytes = 0;
    ctx->bounds_blocks_total = 0;
    return GV_OK;
}

void* gv_stream(GvCtx* ctx) { return ctx ? static_cast<void*>(ctx->stream) : nullptr; }

}  // extern "C"
