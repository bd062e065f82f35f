// gv_mirror.cpp — the device mirror of the bound component pools: spatial order, host gathers (AoS or columns -> SoA
// staging), dense / scattered / device-side uploads of dirty ranges, pool growth, and sync_mirror() which picks among
// them. Replaces the per-entity Manager::tryGet / Manager::get lookups of the reference (mesh.cpp:149,
// transform.hpp:206) with slot indices resolved once per change.
#include "gv_ctx.hpp"

namespace gv {

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

namespace {

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
    const uint32_t threads = worker_parts(n);
    const size_t per = (n + threads - 1) / threads;
    std::vector<uint32_t> tmp;
    tmp.reserve(order.capacity());  // (the buffers swap roles: the caller's head-room survives)
    tmp.resize(n);
    std::vector<size_t> hist((size_t)threads * 1024);
    auto run = [&](auto&& fn) { run_parts(threads, fn); };
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
    ctx->xperm.reserve((size_t)n + n / 4);  // (head-room for appended slots, like the staging arrays)
    ctx->xinv.reserve((size_t)n + n / 4);
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
            p.perm.reserve((size_t)n + n / 4);
            p.inv.reserve((size_t)n + n / 4);
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
    p.perm.reserve((size_t)n + n / 4);
    p.inv.reserve((size_t)n + n / 4);
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
    ctx->h_xab.ptr[j].a = make_float4(pos[0], pos[1], pos[2], scl[0]);
    ctx->h_xab.ptr[j].b = make_float4(rot[0], rot[1], rot[2], rot[3]);
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
            // a mesh that is not ready (ready count 0: resources still loading) ends like a disabled one: readyCount == 0 ->
            // isVisible = false, no record (mesh.cpp:158-165)
            const bool candidate = entity && p.is_enabled.u8(i) && slot != kSlotNone && p.ready_count(i) != 0;
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
    GV_HIP(ctx, hipMemcpyAsync(ctx->d_xab.ptr + lo, ctx->h_xab.ptr + lo, n * sizeof(XfAB), hipMemcpyHostToDevice, ctx->stream));
    GV_HIP(ctx, hipMemcpyAsync(ctx->d_xc.ptr + lo, ctx->h_xc.ptr + lo, n * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
    GV_HIP(ctx, hipMemcpyAsync(ctx->d_xflags.ptr + lo, ctx->h_xflags.ptr + lo, n, hipMemcpyHostToDevice, ctx->stream));
    GV_HIP(ctx, hipMemcpyAsync(ctx->d_xparent.ptr + lo, ctx->h_xparent.ptr + lo, n * 4, hipMemcpyHostToDevice, ctx->stream));
    ctx->stats.upload_bytes += n * 45;
    return GV_OK;
}

// The bound transform fields as one array of structs, if that is what they are: equal strides, every field inside
// one stride-sized window. (Column bindings with separate arrays are gathered on the host.)
// *extent = bytes from `base` to the end of the last bound field of one element (<= stride): the last element of a span
// is only read that far (base is the LOWEST bound field, not necessarily the start of the component).
bool aos_transform_layout(const TransformBinding& xf, const uint8_t** base, AosTransformLayout* L, uint32_t* extent)
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
    *extent = 0;
    for (int k = 0; k < 7; k++) {
        const size_t o = (size_t)(cols[k]->ptr - lo);
        if (o + width[k] > stride)
            return false;
        off[k] = (uint32_t)o;
        *extent = std::max(*extent, off[k] + width[k]);
    }
    *base = lo;
    *L = AosTransformLayout{(uint32_t)stride, off[0], off[1], off[2], off[3], off[4], off[5], off[6]};
    return true;
}

// GV_DIRTY_TRANSFORM over slots [lo, hi) of an AoS pool, device side: the raw components of the span travel through the
// library's own pinned chunks (the caller's memory is never page-locked, see below) and a device kernel does the
// AoS -> SoA gather. Only the bytes between the first and the last bound field are read: (count - 1) strides + the
// extent of one element, so a layout whose first field sits above offset 0 never reads past the caller's pool. The
// host staging of those slots goes stale and is refreshed only if a host path needs it later. Returns GV_E_STATE when
// the path is not applicable.
// The world-matrix cache (gv_sweep) survives a dirty range when every re-mirrored entry is flagged for the subtree-scoped
// sweep. True while the cache is valid and the flag array covers the mirror.
bool track_world_dirty(GvCtx* ctx)
{
    return ctx->world_valid && ctx->d_xdirty.ptr && ctx->d_xdirty.cap >= ctx->xf.occupancy;
}

// `bytes` of the caller's (pageable) memory at `span` into ctx->d_raw, through two pinned chunks of the library's own: worker
// threads fill chunk k while chunk k-1 is on the wire. The caller's memory is never page-locked: transient hipHostRegister /
// hipHostUnregister of application memory was a third faster but left this stack aborting in LATER pageable copies that
// touched the same addresses (3 of 10 runs of the GPU suite; 0 of 10 without it). GV_E_STATE: no room on the device.
int stream_raw_span(GvCtx* ctx, const uint8_t* span, size_t bytes)
{
    if (bytes > ctx->d_raw.cap) {
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // an earlier gather may still be reading the buffer about to go
        if (ctx->d_raw.reserve(bytes) != hipSuccess)
            return GV_E_STATE;
    }
    // chunk size: at least four chunks so that copying into a chunk overlaps the previous one's DMA, 1 .. 32 MB each
    const size_t chunk_bytes = std::min<size_t>((size_t)32 << 20, std::max<size_t>((size_t)1 << 20, ((bytes / 4 + 65535) >> 16) << 16));
    const size_t chunk_cap = std::min(bytes, chunk_bytes);
    for (int k = 0; k < 2; k++) {
        if (!ctx->raw_done[k])
            GV_HIP(ctx, hipEventCreateWithFlags(&ctx->raw_done[k], hipEventDisableTiming));
        if (chunk_cap > ctx->h_raw[k].cap) {
            GV_HIP(ctx, hipEventSynchronize(ctx->raw_done[k]));
            GV_HIP(ctx, ctx->h_raw[k].reserve(chunk_cap));
        }
    }
    uint32_t turn = 0;
    for (size_t off = 0; off < bytes; off += chunk_bytes, turn ^= 1u) {
        const size_t n = std::min(chunk_bytes, bytes - off);
        GV_HIP(ctx, hipEventSynchronize(ctx->raw_done[turn]));  // (a never-recorded event is complete)
        uint8_t* stage = ctx->h_raw[turn].ptr;
        const uint8_t* src = span + off;
        const uint32_t parts = n >= ((size_t)1 << 20) ? worker_parts((size_t)1 << 30) : 1u;  // workers from 1 MB up
        const size_t per = ((n + parts - 1) / parts + 63) & ~(size_t)63;
        run_parts(parts, [&](uint32_t t) {
            const size_t a = std::min(n, per * t), b = std::min(n, per * (t + 1));
            if (a < b)
                memcpy(stage + a, src + a, b - a);
        });
        GV_HIP(ctx, hipMemcpyAsync(ctx->d_raw.ptr + off, stage, n, hipMemcpyHostToDevice, ctx->stream));
        GV_HIP(ctx, hipEventRecord(ctx->raw_done[turn], ctx->stream));
    }
    return GV_OK;
}

int upload_transforms_device(GvCtx* ctx, uint32_t lo, uint32_t hi)
{
    const uint8_t* base = nullptr;
    AosTransformLayout L{};
    uint32_t extent = 0;
    if (hi <= lo || !aos_transform_layout(ctx->xf, &base, &L, &extent))
        return GV_E_STATE;
    const uint32_t count = hi - lo;
    const size_t bytes = (size_t)(count - 1) * L.stride + extent;
    const uint8_t* span = base + (size_t)lo * L.stride;
    if (int rc = stream_raw_span(ctx, span, bytes))
        return rc;
    GV_HIP(ctx, launch_aos_transforms(ctx->d_raw.ptr, L, lo, count, ctx->xinv.empty() ? nullptr : ctx->d_xinv.ptr, ctx->d_xab.ptr,
                                      ctx->d_xc.ptr, ctx->d_xflags.ptr, track_world_dirty(ctx) ? ctx->d_xdirty.ptr : nullptr,
                                      ctx->stream));
    ctx->staging_stale.add(lo, count);
    ctx->stats.upload_bytes += bytes;
    return GV_OK;
}

// Field offsets of a mesh pool bound as an array of structs (every column shares one stride and lies inside one element).
bool aos_mesh_layout(const PoolState& p, const uint8_t** base, AosMeshLayout* L, uint32_t* extent)
{
    const Column* cols[4] = {&p.entity, &p.is_enabled, &p.aabb_min, &p.aabb_max};
    const uint32_t width[4] = {4, 1, 12, 12};
    const size_t stride = p.entity.stride;
    const uint8_t* lo = p.entity.ptr;
    for (const Column* c : cols) {
        if (c->stride != stride || !c->ptr)
            return false;
        lo = std::min(lo, c->ptr);
    }
    uint32_t off[4];
    *extent = 0;
    for (int k = 0; k < 4; k++) {
        const size_t o = (size_t)(cols[k]->ptr - lo);
        if (o + width[k] > stride)
            return false;
        off[k] = (uint32_t)o;
        *extent = std::max(*extent, off[k] + width[k]);
    }
    *base = lo;
    *L = AosMeshLayout{(uint32_t)stride, off[0], off[1], off[2], off[3]};
    return true;
}

// GV_DIRTY_MESH over slots [lo, hi) of an AoS pool, device side (round 3; the transform side has had this since round 1): the
// raw components travel through the pinned chunks, aos_meshes_kernel resolves each mesh's transform through a device copy of
// the entity -> slot table (refreshed here: the table is the caller's and may have changed) and writes the mirror entries.
// Worth it when the span outweighs that table: ranges of at least 2048 slots and 1/12 of the entity capacity (48 raw bytes per
// slot against 4 per entity). Not applicable (GV_E_STATE -> the host gather) to column binds and to pools with a ready column.
int upload_meshes_device(GvCtx* ctx, PoolState& p, uint32_t lo, uint32_t hi)
{
    const uint8_t* base = nullptr;
    AosMeshLayout L{};
    uint32_t extent = 0;
    const uint32_t count = hi > lo ? hi - lo : 0;
    if (count < 2048 || (uint64_t)count * 12 < ctx->xf.entity_capacity || p.ready.ptr ||
        !aos_mesh_layout(p, &base, &L, &extent))
        return GV_E_STATE;
    const size_t bytes = (size_t)(count - 1) * L.stride + extent;
    if (int rc = stream_raw_span(ctx, base + (size_t)lo * L.stride, bytes))
        return rc;
    const uint32_t cap = ctx->xf.entity_capacity;
    if (ctx->d_e2t.reserve(std::max<size_t>(cap, 1)) != hipSuccess || ctx->d_flag.reserve(4) != hipSuccess || ctx->h_flag.reserve(4) != hipSuccess)
        return GV_E_STATE;
    if (cap)
        GV_HIP(ctx, hipMemcpyAsync(ctx->d_e2t.ptr, ctx->xf.entity_to_transform, (size_t)cap * 4, hipMemcpyHostToDevice, ctx->stream));
    GV_HIP(ctx, hipMemsetAsync(ctx->d_flag.ptr, 0, 4, ctx->stream));
    GV_HIP(ctx, launch_aos_meshes(ctx->d_raw.ptr, L, lo, count, p.inv.empty() ? nullptr : p.d_inv.ptr, ctx->d_e2t.ptr, cap, ctx->xf.occupancy,
                                  ctx->xinv.empty() ? nullptr : ctx->d_xinv.ptr, p.d_a.ptr, p.d_b.ptr, p.d_link.ptr, ctx->d_flag.ptr,
                                  ctx->stream));
    GV_HIP(ctx, hipMemcpyAsync(ctx->h_flag.ptr, ctx->d_flag.ptr, 4, hipMemcpyDeviceToHost, ctx->stream));
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the mapping has to be known before the next cull is chosen
    if (ctx->h_flag.ptr[0] && p.mapping == kMapExact)
        p.mapping = kMapSpeculate;  // an edited mesh no longer pairs with its own index
    p.staging_stale.add(lo, count);
    ctx->stats.upload_bytes += bytes + (size_t)cap * 4;
    return GV_OK;
}

// (mesh side of refresh_stale_staging)
void refresh_stale_mesh_staging(GvCtx* ctx, PoolState& p)
{
    if (!p.staging_stale.any())
        return;
    const uint32_t lo = p.staging_stale.lo, hi = std::min(p.staging_stale.hi, p.occupancy);
    if (lo < hi)
        gather_meshes(ctx, p, lo, hi);
    p.staging_stale.clear();
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

// The pinned staging arrays and packet buffers are read by asynchronous copies: before the host rewrites them it waits for the
// copies of the previous sync — an event recorded behind them — not for the whole stream: the previous frame's cull and emit, queued
// behind those copies, may still be running (a scene in which something moves every frame would otherwise run host and device one
// after the other: 0.20 instead of 0.12 ms per frame at 10 M entities with ten movers, tools/moving_bench.py).
int wait_uploads(GvCtx* ctx)
{
    if (ctx->upload_pending) {
        GV_HIP(ctx, hipEventSynchronize(ctx->upload_done));
        ctx->upload_pending = false;
    }
    return GV_OK;
}
int record_uploads(GvCtx* ctx)
{
    if (!ctx->upload_done)
        GV_HIP(ctx, hipEventCreateWithFlags(&ctx->upload_done, hipEventDisableTiming));
    GV_HIP(ctx, hipEventRecord(ctx->upload_done, ctx->stream));
    ctx->upload_pending = true;
    return GV_OK;
}

bool track_world_dirty(GvCtx* ctx);

// Dirty pool slots of a permuted mirror land on scattered entries: ALL the dirty ranges of a sync travel as one packet — one
// copy, one launch (scatter_xf_packets_kernel: records, world-cache dirty bytes, active bits, block flags), one event.
constexpr size_t kRangedCopyMaxRanges = 32;  // dirty ranges of a slot-order mirror sent as plain copies; more go as one scattered packet
static_assert(kMaxFlaggedPools == GV_MAX_POOLS, "BlockFlagTargets has a slot per pool");

template <typename Packet>
int reserve_packets(GvCtx* ctx, PinnedBuf<Packet>& host, DeviceBuf<Packet>& dev, size_t n)
{
    if (n > host.cap || n > dev.cap) {  // (buffers about to be replaced: nothing queued may still use the old ones)
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const size_t want = std::max<size_t>(n + n / 2, 1024);
        GV_HIP(ctx, host.reserve(want));
        GV_HIP(ctx, dev.reserve(want));
    }
    return wait_uploads(ctx);  // the previous packet's copy has read the pinned buffer
}

int upload_transforms_scattered(GvCtx* ctx, const std::vector<DirtyRanges::R>& ranges, const BlockFlagTargets& blocks)
{
    std::vector<uint32_t> start(ranges.size() + 1, 0);
    for (size_t k = 0; k < ranges.size(); k++)
        start[k + 1] = start[k] + (ranges[k].hi - ranges[k].lo);
    const uint32_t n = start.back();
    if (n == 0)
        return GV_OK;
    if (int rc = reserve_packets(ctx, ctx->sc_xf, ctx->dsc_xf, n))
        return rc;
    for (size_t q = 0; q < ranges.size(); q++) {
        const uint32_t lo = ranges[q].lo, base = start[q];
        parallel_ranges(0, ranges[q].hi - lo, [&](uint32_t a, uint32_t b) {  // random reads of the staging arrays: spread over the cores
            for (uint32_t k = a; k < b; k++) {
                const uint32_t j = ctx->xinv.empty() ? lo + k : ctx->xinv[lo + k];  // (a mirror in slot order: the identity)
                XfPacket& pk = ctx->sc_xf.ptr[base + k];
                pk.a = ctx->h_xab.ptr[j].a;
                pk.b = ctx->h_xab.ptr[j].b;
                pk.c = ctx->h_xc.ptr[j];
                pk.entry = j;
                pk.flags = ctx->h_xflags.ptr[j];
                pk.parent = ctx->h_xparent.ptr[j];
                pk.pad[0] = pk.pad[1] = pk.pad[2] = 0;
            }
        });
    }
    GV_HIP(ctx, hipMemcpyAsync(ctx->dsc_xf.ptr, ctx->sc_xf.ptr, (size_t)n * sizeof(XfPacket), hipMemcpyHostToDevice, ctx->stream));
    GV_HIP(ctx, launch_scatter_xf_packets(ctx->dsc_xf.ptr, n, ctx->d_xab.ptr, ctx->d_xc.ptr, ctx->d_xflags.ptr, ctx->d_xparent.ptr, ctx->d_xactive.ptr,
                                          track_world_dirty(ctx) ? ctx->d_xdirty.ptr : nullptr, blocks, ctx->stream));
    if (int rc = record_uploads(ctx))  // the packet buffer is reused by the next sync: it waits for this
        return rc;
    ctx->stats.upload_bytes += (size_t)n * sizeof(XfPacket);
    return GV_OK;
}

// all dirty ranges of a pool as ONE packet (see upload_transforms_scattered)
int upload_meshes_scattered(GvCtx* ctx, PoolState& p, const std::vector<DirtyRanges::R>& ranges, uint8_t* block_flags)
{
    std::vector<uint32_t> start(ranges.size() + 1, 0);
    for (size_t k = 0; k < ranges.size(); k++)
        start[k + 1] = start[k] + (ranges[k].hi - ranges[k].lo);
    const uint32_t n = start.back();
    if (n == 0)
        return GV_OK;
    if (int rc = reserve_packets(ctx, ctx->sc_mesh, ctx->dsc_mesh, n))
        return rc;
    for (size_t q = 0; q < ranges.size(); q++) {
        const uint32_t lo = ranges[q].lo, base = start[q];
        parallel_ranges(0, ranges[q].hi - lo, [&](uint32_t a, uint32_t b) {
            for (uint32_t k = a; k < b; k++) {
                const uint32_t j = p.inv.empty() ? lo + k : p.inv[lo + k];
                MeshPacket& pk = ctx->sc_mesh.ptr[base + k];
                pk.a = p.h_a.ptr[j];
                pk.b = p.h_b.ptr[j];
                pk.entry = j;
                pk.link = p.h_link.ptr[j];
            }
        });
    }
    GV_HIP(ctx, hipMemcpyAsync(ctx->dsc_mesh.ptr, ctx->sc_mesh.ptr, (size_t)n * sizeof(MeshPacket), hipMemcpyHostToDevice, ctx->stream));
    GV_HIP(ctx, launch_scatter_mesh_packets(ctx->dsc_mesh.ptr, n, p.d_a.ptr, p.d_b.ptr, p.d_link.ptr, block_flags, ctx->stream));
    if (int rc = record_uploads(ctx))
        return rc;
    ctx->stats.upload_bytes += (size_t)n * sizeof(MeshPacket);
    return GV_OK;
}

// Pool growth (entities created since the last sync): the new slots [n0, n1) are appended to the mirror as entries
// [n0, n1) — identity on the tail of the permutation — instead of rebuilding it; they stay outside the spatial order
// until the next full build, which sync_mirror schedules once the unsorted tail passes 1/8 of the pool.
int grow_transforms(GvCtx* ctx, uint32_t n0, uint32_t n1, PhaseTimer& phase)
{
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GV_HIP(ctx, ctx->d_xab.grow(n1, n0, ctx->stream));
    GV_HIP(ctx, ctx->d_xc.grow(n1, n0, ctx->stream));
    GV_HIP(ctx, ctx->d_xflags.grow(n1, n0, ctx->stream));
    GV_HIP(ctx, ctx->d_xparent.grow(n1, n0, ctx->stream));
    GV_HIP(ctx, ctx->d_xactive.grow((size_t)n1 / 64 + 1, 0, ctx->stream));  // re-derived below
    phase.lap("  append transforms: device streams");
    GV_HIP(ctx, ctx->h_xab.grow(n1, n0));
    GV_HIP(ctx, ctx->h_xc.grow(n1, n0));
    GV_HIP(ctx, ctx->h_xflags.grow(n1, n0));
    GV_HIP(ctx, ctx->h_xparent.grow(n1, n0));
    phase.lap("  append transforms: pinned staging");
    if (!ctx->xperm.empty()) {
        ctx->xperm.resize(n1);
        ctx->xinv.resize(n1);
        for (uint32_t s = n0; s < n1; s++)
            ctx->xperm[s] = ctx->xinv[s] = s;
        GV_HIP(ctx, ctx->d_xinv.grow(n1, n0, ctx->stream));
        GV_HIP(ctx, hipMemcpyAsync(ctx->d_xinv.ptr + n0, ctx->xinv.data() + n0, (size_t)(n1 - n0) * 4, hipMemcpyHostToDevice, ctx->stream));
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pageable source
    }
    phase.lap("  append transforms: tables");
    gather_transforms(ctx, n0, n1);
    phase.lap("  append transforms: gather");
    bool chained = ctx->max_depth != 0;
    for (uint32_t j = n0; j < n1 && !chained; j++)
        chained = ctx->h_xparent.ptr[j] != kSlotNone;
    if (chained) {  // new slots with parents (or a pool that already has chains): depth / cycle check over the links
        // Only the appended entries can have changed the longest chain: the links of the entries already mirrored are what they
        // were (a re-parented old slot is a GV_DIRTY_HIERARCHY mark, which re-validates everything). Every new entry walks to its
        // root — on the gather threads, 1.25 M entries appended to 8.75 M: 25 ms for the serial pass over the whole pool -> < 1 ms;
        // a walk longer than any chain can be (the old maximum + every new entry) is a cycle among the new links: the full pass
        // then names it.
        uint32_t depth = ctx->max_depth;
        const uint64_t bound = (uint64_t)ctx->max_depth + (n1 - n0) + 1;
        std::atomic<uint32_t> deepest{depth};
        std::atomic<bool> cyclic{false};
        parallel_ranges(n0, n1 - n0, [&](uint32_t a, uint32_t b) {
            uint32_t local = 0;
            for (uint32_t j = a; j < b && !cyclic.load(std::memory_order_relaxed); j++) {
                uint32_t cur = j;
                uint64_t steps = 0;
                while ((cur = ctx->h_xparent.ptr[cur]) != kSlotNone)
                    if (++steps > bound) {
                        cyclic.store(true, std::memory_order_relaxed);
                        break;
                    }
                local = std::max<uint32_t>(local, (uint32_t)std::min<uint64_t>(steps, UINT32_MAX));
            }
            uint32_t seen = deepest.load(std::memory_order_relaxed);
            while (local > seen && !deepest.compare_exchange_weak(seen, local, std::memory_order_relaxed)) {
            }
        });
        depth = deepest.load();
        const int rc = cyclic ? compute_max_depth(ctx, &depth) : GV_OK;
        if (rc != GV_OK) {
            ctx->xf_need_full = true;
            return rc;
        }
        ctx->max_depth = depth;
    }
    phase.lap("  append transforms: depth check");
    const int rc = upload_transforms(ctx, n0, n1);
    if (rc != GV_OK)
        return rc;
    GV_HIP(ctx, launch_pack_active(ctx->d_xflags.ptr, n1, ctx->d_xactive.ptr, ctx->stream));
    ctx->xf_mirrored = n1;
    for (auto& q : ctx->pools) {
        q.patch_valid = false;  // (appended entries: the blocks change; rebuilt once the pools are at rest)
        q.small_streak = 0;
    }
    ctx->xf_appended += n1 - n0;
    ctx->world_valid = false;  // (d_world is sized at the next sweep; appended entries have no matrix yet)
    ctx->world_partial = false;
    GV_HIP(ctx, ctx->d_xdirty.grow(n1, n0, ctx->stream));
    GV_HIP(ctx, hipMemsetAsync(ctx->d_xdirty.ptr + n0, 0, ctx->d_xdirty.cap - n0, ctx->stream));
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
        GV_HIP(ctx, p.d_inv.grow(n1, n0, ctx->stream));
        GV_HIP(ctx, hipMemcpyAsync(p.d_orig.ptr + n0, p.perm.data() + n0, (size_t)(n1 - n0) * 4, hipMemcpyHostToDevice, ctx->stream));
        GV_HIP(ctx, hipMemcpyAsync(p.d_inv.ptr + n0, p.inv.data() + n0, (size_t)(n1 - n0) * 4, hipMemcpyHostToDevice, ctx->stream));
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pageable source
    }
    gather_meshes(ctx, p, n0, n1);
    const int rc = upload_meshes(ctx, p, n0, n1);
    if (rc != GV_OK)
        return rc;
    p.mirrored = n1;
    p.appended += n1 - n0;
    p.epoch++;
    p.order_epoch++;
    p.patch_valid = false;
    return GV_OK;
}


// ---- the spatial re-order on the device (gv_reorder.hip) ---------------------------------------------
// What a full rebuild does on the host and ships over PCIe (order, gather of every component, 73 B per entity of upload:
// 0.13-0.22 s at 10 M entities) the device does from what it already holds: Morton codes of the roots, gv_sort's radix
// kernels on the bare codes (stable: trees stay contiguous, ancestors stay in front of their descendants), one permuting
// pass per stream. The host downloads the new slot <-> entry tables; its staging arrays go stale as a whole and are
// re-gathered only if a dense host path ever needs them (refresh_stale_*).
struct KeySorter {  // scratch of launch_sort on bare keys (~30 B per key; lives for one re-order)
    DeviceBuf<uint32_t> keys[2], vals[2], slots[2], hist, count;
    DeviceBuf<uint16_t> ranks;
    int sort(GvCtx* ctx, const float* key_bits, uint32_t n, uint32_t* order_out)
    {
        for (int k = 0; k < 2; k++) {
            GV_HIP(ctx, keys[k].reserve(n));
            GV_HIP(ctx, vals[k].reserve(n));
            GV_HIP(ctx, slots[k].reserve(n));
        }
        GV_HIP(ctx, ranks.reserve(n));
        GV_HIP(ctx, count.reserve(4));
        const size_t set_words = sort_set_words(n), tiles = sort_tile_count(n);
        GV_HIP(ctx, hist.reserve(2 * set_words + tiles * 256));
        GV_HIP(ctx, hipMemsetAsync(hist.ptr, 0, 2 * set_words * sizeof(uint32_t), ctx->stream));
        GV_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(count.ptr), (int)n, 1, ctx->stream));
        SortBuffers b{};
        b.count = count.ptr;
        b.dist_in = key_bits;
        b.idx_out = order_out;
        b.ranks = ranks.ptr;
        for (int k = 0; k < 2; k++) {
            b.keys[k] = keys[k].ptr;
            b.vals[k] = vals[k].ptr;
            b.slots[k] = slots[k].ptr;
            b.counters[k] = hist.ptr + k * set_words;
        }
        b.tile_hist = hist.ptr + 2 * set_words;
        GV_HIP(ctx, launch_sort(b, n, false, ctx->stream, kSortRadixOnly));
        return GV_OK;
    }
    void release()
    {
        for (int k = 0; k < 2; k++)
            keys[k].release(), vals[k].release(), slots[k].release();
        hist.release(), count.release(), ranks.release();
    }
};

// `words` 32-bit words from the device into a std::vector through a pinned bounce buffer (enqueue only)
inline int download_words(GvCtx* ctx, const uint32_t* dev, uint32_t* bounce, size_t words)
{
    GV_HIP(ctx, hipMemcpyAsync(bounce, dev, words * 4, hipMemcpyDeviceToHost, ctx->stream));
    return GV_OK;
}
inline void copy_words(std::vector<uint32_t>& dst, const uint32_t* src, size_t words)
{
    dst.resize(words);
    parallel_ranges(0, (uint32_t)words, [&](uint32_t a, uint32_t b) { memcpy(dst.data() + a, src + a, (size_t)(b - a) * 4); });
}
// a slot <-> entry pair as downloaded: both tables in range and inverse of each other where it is cheap to see (every
// 257th entry) — a table that is not would send later host gathers out of bounds
inline bool tables_agree(const uint32_t* perm, const uint32_t* inv, size_t n)
{
    std::atomic<bool> ok{true};
    parallel_ranges(0, (uint32_t)n, [&](uint32_t a, uint32_t b) {
        bool good = true;
        for (uint32_t j = a; j < b; j++)
            good &= perm[j] < n && inv[j] < n;
        for (uint32_t j = a; j < b && good; j += 257)
            good &= inv[perm[j]] == j;
        if (!good)
            ok.store(false, std::memory_order_relaxed);
    });
    return ok;
}

// xnewpos (out): old transform entry -> new entry, for the mesh pools' links
int reorder_transforms_device(GvCtx* ctx, KeySorter& ks, DeviceBuf<uint32_t>& xnewpos, PhaseTimer& phase)
{
    const uint32_t n = ctx->xf_mirrored;
    DeviceBuf<uint32_t> root, order, box, perm, xinv_new;
    DeviceBuf<float> code;
    DeviceBuf<XfAB> ab;
    DeviceBuf<float2> c;
    DeviceBuf<uint8_t> flags;
    DeviceBuf<uint32_t> parent;
    auto drop = [&] { root.release(), order.release(), box.release(), perm.release(), xinv_new.release(), code.release(), ab.release(), c.release(), flags.release(), parent.release(); };
    struct Guard { decltype(drop)& f; ~Guard() { f(); } } guard{drop};
    GV_HIP(ctx, root.reserve(n));
    GV_HIP(ctx, order.reserve(n));
    GV_HIP(ctx, perm.reserve(n));
    GV_HIP(ctx, xinv_new.reserve(ctx->d_xinv.cap));
    GV_HIP(ctx, box.reserve(8));
    GV_HIP(ctx, code.reserve(n));
    GV_HIP(ctx, xnewpos.reserve(n));
    // the fresh streams keep the old capacities (head-room of a growing pool)
    GV_HIP(ctx, ab.reserve(ctx->d_xab.cap));
    GV_HIP(ctx, c.reserve(ctx->d_xc.cap));
    GV_HIP(ctx, flags.reserve(ctx->d_xflags.cap));
    GV_HIP(ctx, parent.reserve(ctx->d_xparent.cap));
    phase.lap("  re-order: scratch");
    GV_HIP(ctx, launch_reorder_codes(xf_mirror(ctx), root.ptr, box.ptr, code.ptr, ctx->stream));
    if (int rc = ks.sort(ctx, code.ptr, n, order.ptr))
        return rc;
    GV_HIP(ctx, launch_reorder_invert(order.ptr, n, xnewpos.ptr, ctx->stream));
    GV_HIP(ctx, launch_reorder_transforms(order.ptr, xnewpos.ptr, n, ctx->d_xab.ptr, ctx->d_xc.ptr, ctx->d_xflags.ptr, ctx->d_xparent.ptr, ab.ptr, c.ptr,
                                          flags.ptr, parent.ptr, ctx->stream));
    GV_HIP(ctx, launch_reorder_remap(ctx->d_xinv.ptr, n, xnewpos.ptr, xinv_new.ptr, perm.ptr, ctx->stream));  // (nothing of the old mirror is written: a failure below leaves it whole)
    // the new tables and parent links come home; the staging records double as the pinned bounce buffer (stale from here on)
    uint32_t* bounce = reinterpret_cast<uint32_t*>(ctx->h_xab.ptr);  // 8 words per entry
    ctx->staging_stale.add(0, n);  // (from the first word written: an early return below must not leave the records looking current)
    if (int rc = download_words(ctx, xinv_new.ptr, bounce, n)) return rc;
    if (int rc = download_words(ctx, perm.ptr, bounce + n, n)) return rc;
    if (int rc = download_words(ctx, parent.ptr, bounce + 2 * (size_t)n, n)) return rc;
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    phase.lap("  re-order: device + downloads");
    if (!tables_agree(bounce + n, bounce, n))
        return ctx->fail(GV_E_HIP, "device re-order of the transform mirror returned tables that are not a permutation");
    copy_words(ctx->xinv, bounce, n);  // (in place: the vectors already have this size)
    copy_words(ctx->xperm, bounce + n, n);
    parallel_ranges(0, n, [&](uint32_t a, uint32_t b) { memcpy(ctx->h_xparent.ptr + a, bounce + 2 * (size_t)n + a, (size_t)(b - a) * 4); });
    phase.lap("  re-order: host tables");
    std::swap(ctx->d_xab, ab);
    std::swap(ctx->d_xc, c);
    std::swap(ctx->d_xflags, flags);
    std::swap(ctx->d_xparent, parent);
    std::swap(ctx->d_xinv, xinv_new);
    GV_HIP(ctx, launch_pack_active(ctx->d_xflags.ptr, n, ctx->d_xactive.ptr, ctx->stream));
    ctx->staging_stale.add(0, n);
    ctx->world_valid = false;  // (the cache is in the old order; the next sweep rebuilds it)
    ctx->world_partial = false;
    if (ctx->d_xdirty.ptr)
        GV_HIP(ctx, hipMemsetAsync(ctx->d_xdirty.ptr, 0, ctx->d_xdirty.cap, ctx->stream));
    ctx->xdirty_set = false;
    ctx->xf_epoch++;
    ctx->xf_appended = 0;
    ctx->stats.mirror_reorders++;
    return GV_OK;
}

// one mesh pool follows its transforms (xnewpos: they have just moved; NULL: only this pool's tail is out of order).
// GV_E_STATE: not applicable (the pool has no order table) -> the caller schedules a host rebuild of the pool
int reorder_meshes_device(GvCtx* ctx, PoolState& p, KeySorter& ks, const uint32_t* xnewpos, PhaseTimer& phase)
{
    const uint32_t n = p.mirrored;
    if (p.perm.empty() || n < 2 || !p.d_orig.ptr || !p.d_inv.ptr)
        return GV_E_STATE;
    DeviceBuf<uint32_t> order, link, orig, inv;
    DeviceBuf<float> key;
    DeviceBuf<float4> a;
    DeviceBuf<float2> b;
    auto drop = [&] { order.release(), link.release(), orig.release(), inv.release(), key.release(), a.release(), b.release(); };
    struct Guard { decltype(drop)& f; ~Guard() { f(); } } guard{drop};
    GV_HIP(ctx, order.reserve(n));
    GV_HIP(ctx, key.reserve(n));
    GV_HIP(ctx, a.reserve(p.d_a.cap));
    GV_HIP(ctx, b.reserve(p.d_b.cap));
    GV_HIP(ctx, link.reserve(p.d_link.cap));
    GV_HIP(ctx, orig.reserve(p.d_orig.cap));
    GV_HIP(ctx, inv.reserve(p.d_inv.cap));
    GV_HIP(ctx, ctx->d_flag.reserve(4));
    GV_HIP(ctx, ctx->h_flag.reserve(4));
    GV_HIP(ctx, hipMemsetAsync(ctx->d_flag.ptr, 0, 4, ctx->stream));
    phase.lap("  re-order: scratch");
    GV_HIP(ctx, launch_reorder_mesh_keys(p.d_link.ptr, n, xnewpos, ctx->xf_mirrored, key.ptr, ctx->stream));
    if (int rc = ks.sort(ctx, key.ptr, n, order.ptr))
        return rc;
    GV_HIP(ctx, launch_reorder_meshes(order.ptr, n, xnewpos, ctx->xf_mirrored, p.d_a.ptr, p.d_b.ptr, p.d_link.ptr, p.d_orig.ptr, a.ptr, b.ptr, link.ptr,
                                      orig.ptr, inv.ptr, ctx->d_flag.ptr, ctx->stream));
    uint32_t* bounce = reinterpret_cast<uint32_t*>(p.h_a.ptr);  // 4 words per entry
    p.staging_stale.add(0, n);  // (an early return below must not leave the staging records looking current)
    if (int rc = download_words(ctx, orig.ptr, bounce, n)) return rc;
    if (int rc = download_words(ctx, inv.ptr, bounce + n, n)) return rc;
    GV_HIP(ctx, hipMemcpyAsync(ctx->h_flag.ptr, ctx->d_flag.ptr, 4, hipMemcpyDeviceToHost, ctx->stream));
    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    phase.lap("  re-order: device + downloads");
    if (!tables_agree(bounce, bounce + n, n))
        return ctx->fail(GV_E_HIP, "device re-order of a mesh mirror returned tables that are not a permutation");
    copy_words(p.perm, bounce, n);
    copy_words(p.inv, bounce + n, n);
    phase.lap("  re-order: host tables");
    std::swap(p.d_a, a);
    std::swap(p.d_b, b);
    std::swap(p.d_link, link);
    std::swap(p.d_orig, orig);
    std::swap(p.d_inv, inv);
    if (ctx->h_flag.ptr[0] && p.mapping == kMapExact)
        p.mapping = kMapSpeculate;
    p.staging_stale.add(0, n);
    p.epoch++;
    p.order_epoch++;
    p.patch_valid = false;
    p.appended = 0;
    const uint32_t pool_id = (uint32_t)(&p - ctx->pools);
    for (auto& vs : ctx->views[pool_id]) {  // per-entry outputs of earlier culls are in the old order
        vs.vis_flags_current = false;
        vs.ballots_current = false;
    }
    return GV_OK;
}


// The blocks of pool `p` that hold an entry of the slot ranges `ranges` get their flag set (PoolState::d_blk_dirty): what the
// next cull has to re-derive of the pool's block bounds and emit seeds. inv: slot -> mirror entry of those slots on the device, a
// table of `slots` elements (the transform pool's for transform-side ranges, the mesh pool's own otherwise).
int mark_dirty_blocks(GvCtx* ctx, PoolState& p, const std::vector<DirtyRanges::R>& ranges, const uint32_t* inv, uint32_t slots)
{
    const uint32_t nr = (uint32_t)ranges.size();
    if (nr == 0 || !p.d_blk_dirty.ptr)
        return GV_OK;
    if (2 * (size_t)nr + 1 > ctx->h_ranges.cap || 2 * (size_t)nr + 1 > ctx->d_ranges.cap) {
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (buffers about to be replaced)
        GV_HIP(ctx, ctx->h_ranges.reserve(std::max<size_t>(2 * (size_t)nr + 1, 1024)));
        GV_HIP(ctx, ctx->d_ranges.reserve(std::max<size_t>(2 * (size_t)nr + 1, 1024)));
    }
    if (int rc = wait_uploads(ctx))  // (an earlier call's copy may still be reading the pinned words)
        return rc;
    uint32_t* start = ctx->h_ranges.ptr;
    uint32_t* first = start + nr + 1;
    uint32_t total = 0;
    for (uint32_t k = 0; k < nr; k++) {
        start[k] = total;
        first[k] = ranges[k].lo;
        total += ranges[k].hi - ranges[k].lo;
    }
    start[nr] = total;
    GV_HIP(ctx, hipMemcpyAsync(ctx->d_ranges.ptr, ctx->h_ranges.ptr, (2 * (size_t)nr + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    GV_HIP(ctx, launch_mark_dirty_blocks(ctx->d_ranges.ptr, ctx->d_ranges.ptr + nr + 1, nr, total, inv, slots, p.occupancy, p.d_blk_dirty.ptr, ctx->stream));
    return record_uploads(ctx);
}

}  // namespace

int sync_mirror(GvCtx* ctx)
{
    PhaseTimer phase;
    if (!ctx->xf.bound)
        return ctx->fail(GV_E_STATE, "gv_sync: no transform pool bound");
    GV_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t n = ctx->xf.occupancy;
    bool staged = false;
    const bool spatial = !(ctx->config.flags & GV_CONFIG_KEEP_SLOT_ORDER);
    // Too much of a pool sits in the unsorted tail: back into spatial order — on the device (the new slots are appended first, like
    // any growth, then the mirror is permuted where it lies: reorder_*_device)
    bool reorder_xf = false;
    if (!ctx->xf_need_full && n > ctx->xf_mirrored && spatial &&
        ((uint64_t)ctx->xf_appended + (n - ctx->xf_mirrored)) * 8 > n && n >= 1024) {
        if (!ctx->xperm.empty())
            reorder_xf = true;
        else
            ctx->xf_need_full = true;
    }
    if (ctx->xf_need_full) {
        // staging is about to be rewritten: make sure earlier async uploads have drained
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        staged = true;
        const size_t cap = std::max<size_t>(n, 1);
        GV_HIP(ctx, ctx->d_xab.reserve(cap));
        GV_HIP(ctx, ctx->d_xc.reserve(cap));
        GV_HIP(ctx, ctx->d_xflags.reserve(cap));
        GV_HIP(ctx, ctx->d_xactive.reserve(cap / 64 + 1));
        GV_HIP(ctx, ctx->d_xparent.reserve(cap));
        // pinned staging and the host tables get a quarter of head-room: page-locking is what an append of created entities
        // would otherwise wait for (76 of 188 ms when 1.25 M slots were appended to 8.75 M, tools/reorder_bench.py), and the
        // mirror is re-ordered in place (no new allocation) by the time the tail reaches an eighth of the pool
        const size_t hcap = cap + cap / 4;
        GV_HIP(ctx, ctx->h_xab.reserve(hcap));
        GV_HIP(ctx, ctx->h_xc.reserve(hcap));
        GV_HIP(ctx, ctx->h_xflags.reserve(hcap));
        GV_HIP(ctx, ctx->h_xparent.reserve(hcap));
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
        ctx->world_partial = false;
        GV_HIP(ctx, ctx->d_xdirty.reserve(cap));
        GV_HIP(ctx, hipMemsetAsync(ctx->d_xdirty.ptr, 0, ctx->d_xdirty.cap, ctx->stream));
        ctx->xdirty_set = false;
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
        const int rc = grow_transforms(ctx, ctx->xf_mirrored, n, phase);
        if (rc != GV_OK)
            return rc;
      }
      if (ctx->xf_dirty.any()) {
        if (int wrc = wait_uploads(ctx))  // (staging is about to be rewritten: the copies that read it have to be through)
            return wrc;
        staged = true;
        ctx->xf_dirty.normalise(n, 0);  // exact: a re-mirrored slot is a flagged slot (its whole subtree is re-swept)
        const std::vector<DirtyRanges::R> ranges = ctx->xf_dirty.items;
        const uint64_t total = ctx->xf_dirty.total();
        if (total) {
            const uint32_t lo = ranges.front().lo, hi = ranges.back().hi;  // the covering range
            int rc = GV_OK;
            const bool dense = total * 2 > n;
            // Pools that keep block bounds / emit seeds current across changes (flat, exactly paired: entry i of the pool is
            // transform entry i) get the blocks these transforms sit in flagged; anything else falls back to "rebuilt once the pool
            // is at rest". ... while the re-mirrored entries are few: slots that are neighbours in the pool are scattered over the
            // spatially ordered mirror, so 10^5 of them touch nearly every block of a 10 M pool and the patch becomes a full rebuild
            // (measured: 426 us against 145 us for culling that frame without boxes; 1000 scattered movers: 37 us).
            BlockFlagTargets targets{};
            bool flagging = false;
            for (uint32_t k = 0; k < GV_MAX_POOLS; k++) {
                PoolState& q = ctx->pools[k];
                if (!q.bound)
                    continue;
                const uint64_t nblocks = (q.occupancy + kCullBlock - 1) / kCullBlock;
                const bool few = !dense && !ctx->xf_links_dirty && total * 16 <= nblocks + 16 * 64;
                q.small_streak = few ? std::min(q.small_streak + 1u, 1000u) : 0u;
                if (!q.patch_valid)
                    continue;
                if (!few || ctx->max_depth != 0 || q.mapping != kMapExact || !q.d_blk_dirty.ptr) {
                    q.patch_valid = false;
                } else {
                    targets.flags[k] = q.d_blk_dirty.ptr;
                    targets.occupancy[k] = q.occupancy;
                    flagging = true;
                }
            }
            std::vector<DirtyRanges::R> unflagged;  // ranges whose upload path does not flag blocks itself
            bool bits_current = false;              // the active bit-plane was kept up to date entry by entry (the packet path)
            if (dense) {
                // most of the pool: one pass over the covering range (re-mirroring a clean slot is harmless)
                rc = GV_E_STATE;
                if (!ctx->xf_links_dirty)
                    rc = upload_transforms_device(ctx, lo, hi);  // raw AoS span + device gather (GV_E_STATE: not applicable)
                if (rc == GV_E_STATE) {
                    refresh_stale_staging(ctx);  // this path re-uploads every entry from the staging arrays
                    rc = regather_transforms_pipelined(ctx, lo, hi);  // dense, chunked, DMA under gather
                }
            } else {
                // itemised: large ranges take the device-side gather, everything else travels as ONE scattered packet
                // (or as plain ranged copies when the mirror is in slot order)
                std::vector<DirtyRanges::R> host;
                for (const auto& r : ranges) {
                    int one = GV_E_STATE;
                    if (!ctx->xf_links_dirty && r.hi - r.lo >= 2048)
                        one = upload_transforms_device(ctx, r.lo, r.hi);
                    if (one == GV_E_STATE)
                        host.push_back(r);
                    else if (one != GV_OK)
                        return one;
                    else
                        unflagged.push_back(r);
                }
                for (const auto& r : host)
                    gather_transforms(ctx, r.lo, r.hi);
                // a mirror in slot order takes plain ranged copies — five small copies and a memset per range — while the ranges
                // are few; a frame that moved thousands of scattered entities (up to 16 384 ranges) would queue ~10^5 tiny copies
                // that way: it travels as the one scattered packet too
                if (ctx->xinv.empty() && host.size() <= kRangedCopyMaxRanges) {
                    for (const auto& r : host) {
                        if ((rc = upload_transforms(ctx, r.lo, r.hi)) != GV_OK)
                            break;
                        if (track_world_dirty(ctx))
                            GV_HIP(ctx, hipMemsetAsync(ctx->d_xdirty.ptr + r.lo, 1, r.hi - r.lo, ctx->stream));
                        unflagged.push_back(r);
                    }
                } else {
                    rc = upload_transforms_scattered(ctx, host, targets);
                    bits_current = unflagged.empty();
                }
            }
            if (rc != GV_OK)
                return rc;
            if (!bits_current)
                GV_HIP(ctx, launch_pack_active(ctx->d_xflags.ptr, n, ctx->d_xactive.ptr, ctx->stream));
            if (flagging && !unflagged.empty())
                for (uint32_t k = 0; k < GV_MAX_POOLS; k++)
                    if (targets.flags[k])
                        if (int mrc = mark_dirty_blocks(ctx, ctx->pools[k], unflagged, ctx->xinv.empty() ? nullptr : ctx->d_xinv.ptr, ctx->xf.occupancy))
                            return mrc;
            if (ctx->xf_links_dirty) {  // setParent (transform.cpp:130-195): chains changed length, maybe closed a cycle
                uint32_t depth = 0;
                rc = compute_max_depth(ctx, &depth);
                if (rc != GV_OK) {
                    ctx->xf_need_full = true;  // the mirror now holds a cyclic link: rebuild once the caller has fixed it
                    return rc;
                }
                ctx->max_depth = depth;
            }
            if (track_world_dirty(ctx) && !dense) {
                ctx->xdirty_set = true;     // (the upload paths above flagged what they wrote)
                ctx->world_partial = true;  // the cache stays, minus the chains through the flagged entries
            } else {
                if (track_world_dirty(ctx))
                    ctx->xdirty_set = true;
                ctx->world_valid = false;   // most of the pool moved: a full sweep is cheaper than walking flags
            }
            ctx->xf_epoch++;
        }
        ctx->xf_links_dirty = false;
        ctx->xf_dirty.clear();
      }
    }
    bool reorder_pool[GV_MAX_POOLS] = {};
    for (auto& p : ctx->pools) {
        if (!p.bound)
            continue;
        if (!p.need_full && p.occupancy > p.mirrored && spatial &&
            ((uint64_t)p.appended + (p.occupancy - p.mirrored)) * 8 > p.occupancy && p.occupancy >= 1024) {
            if (!p.perm.empty())
                reorder_pool[&p - ctx->pools] = true;
            else
                p.need_full = true;
        }
        if (!p.need_full && p.occupancy > p.mirrored) {
            staged = true;
            phase.lap("transforms up to date");
            const int rc = grow_meshes(ctx, p, p.mirrored, p.occupancy);
            if (rc != GV_OK)
                return rc;
            phase.lap("append meshes");
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
            GV_HIP(ctx, p.h_a.reserve(cap + cap / 4));  // (head-room: see the transform staging above)
            GV_HIP(ctx, p.h_b.reserve(cap + cap / 4));
            GV_HIP(ctx, p.h_link.reserve(cap + cap / 4));
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
                    GV_HIP(ctx, p.d_inv.reserve(cap));
                    GV_HIP(ctx, hipMemcpyAsync(p.d_orig.ptr, p.perm.data(), (size_t)p.occupancy * 4, hipMemcpyHostToDevice, ctx->stream));
                    GV_HIP(ctx, hipMemcpyAsync(p.d_inv.ptr, p.inv.data(), (size_t)p.occupancy * 4, hipMemcpyHostToDevice, ctx->stream));
                    GV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pageable source
                }
            }
            phase.lap("mapping + upload meshes");
            p.need_full = false;
            p.patch_valid = false;
            p.dirty.clear();
            p.staging_stale.clear();  // (every entry has just been gathered)
            p.epoch++;
            p.order_epoch++;
            p.mirrored = p.occupancy;
            p.appended = 0;
        } else if (p.dirty.any()) {
            if (int wrc = wait_uploads(ctx))
                return wrc;
            staged = true;
            p.dirty.normalise(p.occupancy, 0);
            const std::vector<DirtyRanges::R> ranges = p.dirty.items;
            const uint64_t total = p.dirty.total();
            if (total) {
                int rc = GV_OK;
                // large ranges of an AoS pool: raw span + device-side gather (upload_meshes_device says when it applies);
                // `left` is what remains for the host paths
                std::vector<DirtyRanges::R> left;
                uint64_t left_total = 0;
                bool packet_flagged = false;  // the scattered packet flagged the blocks of `left` itself
                for (const auto& r : ranges) {
                    const int one = upload_meshes_device(ctx, p, r.lo, r.hi);
                    if (one == GV_E_STATE) {
                        left.push_back(r);
                        left_total += r.hi - r.lo;
                    } else if (one != GV_OK) {
                        return one;
                    }
                }
                if (left.empty()) {
                } else if (!p.inv.empty() && left_total * 2 > p.occupancy) {  // most of a permuted pool: one dense upload
                    refresh_stale_mesh_staging(ctx, p);  // (it re-uploads every entry from the staging arrays)
                    gather_meshes(ctx, p, left.front().lo, left.back().hi);
                    rc = upload_meshes(ctx, p, 0, p.occupancy);
                } else {
                    for (const auto& r : left)
                        gather_meshes(ctx, p, r.lo, r.hi);
                    if (p.inv.empty() && left.size() <= kRangedCopyMaxRanges) {
                        for (const auto& r : left)
                            if ((rc = upload_meshes(ctx, p, r.lo, r.hi)) != GV_OK)
                                break;
                    } else {
                        const uint64_t nb = (p.occupancy + kCullBlock - 1) / kCullBlock;
                        const bool flag_here = p.patch_valid && p.d_blk_dirty.ptr && total * 16 <= nb + 16 * 64;
                        rc = upload_meshes_scattered(ctx, p, left, flag_here ? p.d_blk_dirty.ptr : nullptr);
                        packet_flagged = flag_here;
                    }
                }
                if (rc != GV_OK)
                    return rc;
                const uint64_t nblocks = (p.occupancy + kCullBlock - 1) / kCullBlock;
                if (total * 16 > nblocks + 16 * 64)
                    p.small_streak = 0;  // (small mesh edits leave the streak to the transform side: no double count)
                if (p.patch_valid) {  // (see the transform side)
                    const bool most = !left.empty() && !p.inv.empty() && left_total * 2 > p.occupancy;
                    if (most || p.mapping != kMapExact || total * 16 > nblocks + 16 * 64) {
                        p.patch_valid = false;
                    } else {
                        std::vector<DirtyRanges::R> unflagged;  // what did not travel in a packet that flags blocks itself
                        if (!packet_flagged) {
                            unflagged = ranges;
                        } else {
                            for (const auto& r : ranges)
                                if (std::find_if(left.begin(), left.end(), [&](const DirtyRanges::R& l) { return l.lo == r.lo && l.hi == r.hi; }) == left.end())
                                    unflagged.push_back(r);
                        }
                        if (!unflagged.empty())
                            if (int mrc = mark_dirty_blocks(ctx, p, unflagged, p.inv.empty() ? nullptr : p.d_inv.ptr, p.occupancy))
                                return mrc;
                    }
                }
            }
            p.dirty.clear();
            p.epoch++;
        }
    }
    if (staged)  // whatever copies this sync queued from the pinned staging arrays: the next sync waits for them before it rewrites those
        if (int rrc = record_uploads(ctx))
            return rrc;
    // the re-order itself, behind everything that brought the mirror up to date in its old order
    bool any_reorder = reorder_xf;
    for (bool b : reorder_pool)
        any_reorder = any_reorder || b;
    if (any_reorder) {
        phase.lap("appended slots + dirty ranges");
        KeySorter ks;
        DeviceBuf<uint32_t> xnewpos;
        struct Scratch { KeySorter& ks; DeviceBuf<uint32_t>& x; ~Scratch() { ks.release(); x.release(); } } scratch{ks, xnewpos};
        bool again = false;
        if (reorder_xf) {
            if (int rc = reorder_transforms_device(ctx, ks, xnewpos, phase))
                return rc;
            phase.lap("device re-order: transforms");
        }
        for (auto& p : ctx->pools) {
            if (!p.bound || p.need_full || !(reorder_xf || reorder_pool[&p - ctx->pools]))
                continue;  // (a pool whose transforms moved follows them: its links name transform entries)
            const int rc = p.occupancy ? reorder_meshes_device(ctx, p, ks, reorder_xf ? xnewpos.ptr : nullptr, phase) : GV_OK;
            if (rc == GV_E_STATE) {
                p.need_full = true;
                again = true;
            } else if (rc != GV_OK) {
                if (reorder_xf) {
                    // the transform mirror is already in its new order: this pool and every pool behind it still name OLD transform
                    // entries in their links. Nothing re-fires the re-order, so they are rebuilt from the bound pools at the next
                    // sync (the next gv_cull must not run on stale links after a failed scratch allocation)
                    for (PoolState* q = &p; q != ctx->pools + GV_MAX_POOLS; q++)
                        if (q->bound && !q->need_full) {
                            q->need_full = true;
                            q->order_epoch++;
                        }
                }
                return rc;
            }
        }
        phase.lap("device re-order: mesh pools");
        GV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the old streams are freed with the scratch
        ks.release();
        xnewpos.release();
        phase.lap("device re-order: scratch freed");
        if (again)
            return sync_mirror(ctx);  // (pools without an order table: rebuilt on the host from the new transform tables)
    }
    return GV_OK;
}

TransformMirror xf_mirror(const GvCtx* ctx)
{
    TransformMirror m;
    m.ab = ctx->d_xab.ptr;
    m.c = ctx->d_xc.ptr;
    m.flags = ctx->d_xflags.ptr;
    m.active_bits = ctx->d_xactive.ptr;
    m.parent = ctx->d_xparent.ptr;
    m.count = ctx->xf.occupancy;
    m.max_depth = ctx->max_depth;
    return m;
}

}  // namespace gv
