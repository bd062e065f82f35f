// Host orchestration of libgarden_vis under AddressSanitizer + UndefinedBehaviorSanitizer (CPU tier, no GPU).
//
// gv_context.cpp, gv_mirror.cpp and gv_exchange.cpp are normally only ever compiled as HIP; here they are built as plain C++
// against tests/cpp/hip_stub (device memory = zeroed host memory, copies = memcpy, kernels = no-ops) and driven through the
// C-ABI of include/garden_vis.h: binds, mirror builds in both orders, every dirty-range path (ranged copies, the one scattered
// packet, the device-side gather through the pinned chunks and the worker threads, link changes), pools that grow / shrink /
// move, column binds, ready columns, record layouts and targets, batched ticks, sorts, Hi-Z builds of odd sizes, sweeps,
// scene ingest and tile extraction, error paths. Kernels do nothing, so RESULTS are not checked here (the GPU tier does
// that against the oracle) — what is checked is that every staging buffer, index table and copy the host side makes stays
// inside what it allocated, and that the status codes are the documented ones. TEST-ONLY: the product library still
// returns GV_E_NODEVICE without a gfx950 device.
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/garden_vis.h"

struct Transform {  // include/garden/system/transform.hpp:31-61 (80 bytes in a release build)
    uint32_t entity, parent;
    uint64_t uid;
    float position[4], scale[4], rotation[4];
    void* childs;
    uint8_t selfActive, ancestorsActive, modelWithAncestors, pad[5];
};
static_assert(sizeof(Transform) == 80, "TransformComponent layout");
struct Mesh {  // include/garden/system/render/mesh.hpp:45-55
    uint32_t entity;
    uint8_t reserved[3], isEnabled, isVisible, pad[7];
    float aabbMin[4], aabbMax[4];
};
static_assert(sizeof(Mesh) == 48, "MeshRenderComponent layout");
struct BigMesh : Mesh {  // a derived component: larger stride
    float extra[12];
};

#define CHECK(call)                                                                                     \
    do {                                                                                                \
        const int rc__ = (call);                                                                        \
        if (rc__ != GV_OK) {                                                                            \
            std::fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc__, gv_last_error(ctx)); \
            std::exit(1);                                                                               \
        }                                                                                               \
    } while (0)
#define EXPECT(call, code)                                                                              \
    do {                                                                                                \
        const int rc__ = (call);                                                                        \
        if (rc__ != (code)) {                                                                           \
            std::fprintf(stderr, "%s:%d: %s -> %d, expected %d\n", __FILE__, __LINE__, #call, rc__, (int)(code)); \
            std::exit(1);                                                                               \
        }                                                                                               \
    } while (0)

struct World {
    std::vector<Transform> xf;
    std::vector<Mesh> meshes;
    std::vector<BigMesh> big;
    std::vector<uint32_t> e2t;  // entity id -> transform slot
    std::mt19937 rng{12345};

    void build(uint32_t n, uint32_t depth)
    {
        std::uniform_real_distribution<float> u(-1.0f, 1.0f);
        xf.assign(n, Transform{});
        meshes.assign(n, Mesh{});
        e2t.assign(n + 1, GV_NONE);
        for (uint32_t i = 0; i < n; i++) {
            Transform& t = xf[i];
            if (i % 97 == 5)
                continue;  // a free slot
            t.entity = i + 1;
            e2t[t.entity] = i;
            // a forest: slot i hangs under an earlier slot of the previous "level" band
            const uint32_t level = depth ? (i * (depth + 1)) / n : 0;
            if (level > 0) {
                const uint32_t band = n / (depth + 1);
                uint32_t p = (level - 1) * band + rng() % band;
                if (xf[p].entity != 0)
                    t.parent = xf[p].entity;
            }
            t.uid = 1000u + i;
            for (int k = 0; k < 3; k++) {
                t.position[k] = 3000.0f * u(rng);
                t.scale[k] = 1.0f + 0.5f * u(rng);
            }
            float q[4] = {u(rng), u(rng), u(rng), u(rng)}, len = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]) + 1e-6f;
            for (int k = 0; k < 4; k++)
                t.rotation[k] = q[k] / len;
            t.selfActive = i % 101 != 3;
            t.ancestorsActive = 1;
            t.modelWithAncestors = i % 13 != 0;
            Mesh& m = meshes[i];
            m.entity = t.entity;
            m.isEnabled = i % 89 != 1;
            for (int k = 0; k < 3; k++) {
                m.aabbMin[k] = -0.5f - 0.5f * std::fabs(u(rng));
                m.aabbMax[k] = 0.5f + 0.5f * std::fabs(u(rng));
            }
        }
        big.assign(n / 3 + 7, BigMesh{});
        for (uint32_t j = 0; j + 7 < big.size(); j++) {  // every third entity, in another order; the tail stays free
            const uint32_t i = ((j * 7919u) % (n / 3)) * 3;
            static_cast<Mesh&>(big[j]) = meshes[i];
        }
    }
};

static GvTransformLayout transform_layout()
{
    return GvTransformLayout{(uint32_t)offsetof(Transform, entity), (uint32_t)offsetof(Transform, parent), (uint32_t)offsetof(Transform, position),
                             (uint32_t)offsetof(Transform, scale), (uint32_t)offsetof(Transform, rotation), (uint32_t)offsetof(Transform, selfActive),
                             (uint32_t)offsetof(Transform, ancestorsActive), (uint32_t)offsetof(Transform, modelWithAncestors)};
}
static GvMeshLayout mesh_layout()
{
    return GvMeshLayout{(uint32_t)offsetof(Mesh, entity), (uint32_t)offsetof(Mesh, isEnabled), (uint32_t)offsetof(Mesh, isVisible),
                        (uint32_t)offsetof(Mesh, aabbMin), (uint32_t)offsetof(Mesh, aabbMax)};
}

static GvView make_view(int8_t shadow_pass, uint8_t hiz, uint8_t emit, float cx = 0.0f)
{
    GvView v{};
    const float f = 1.0f, aspect = 16.0f / 9.0f, zn = 0.01f;  // infinite reversed-Z perspective, camera looking down -z
    v.view_proj[0] = f / aspect;
    v.view_proj[5] = f;
    v.view_proj[11] = -1.0f;
    v.view_proj[14] = zn;
    v.camera_position[0] = cx;
    v.shadow_pass = shadow_pass;
    v.use_hiz = hiz;
    v.emit_records = emit;
    return v;
}

static void bind_all(GvCtx* ctx, World& w)
{
    const GvTransformLayout tl = transform_layout();
    const GvMeshLayout ml = mesh_layout();
    CHECK(gv_transform_bind(ctx, w.xf.data(), sizeof(Transform), (uint32_t)w.xf.size(), &tl, w.e2t.data(), (uint32_t)w.e2t.size()));
    CHECK(gv_pool_bind(ctx, 0, w.meshes.data(), sizeof(Mesh), (uint32_t)w.meshes.size(), &ml));
    CHECK(gv_pool_bind(ctx, 1, w.big.data(), sizeof(BigMesh), (uint32_t)w.big.size(), &ml));
}

static void check_permutation(GvCtx* ctx, uint32_t pool, uint32_t n)
{
    std::vector<uint32_t> slots(n, GV_NONE);
    CHECK(gv_pool_mirror_slots(ctx, pool, slots.data(), n));
    std::vector<uint8_t> seen(n, 0);
    for (uint32_t s : slots) {
        if (s >= n || seen[s]) {
            std::fprintf(stderr, "pool %u: the mirror order is not a permutation of its %u slots\n", pool, n);
            std::exit(1);
        }
        seen[s] = 1;
    }
}

static void frame(GvCtx* ctx, uint32_t pools, bool hiz, bool sort)
{
    GvView views[3] = {make_view(-1, hiz ? 1 : 0, 1), make_view(0, 0, 1), make_view(1, 0, 1)};
    for (uint32_t p = 0; p < pools; p++) {
        CHECK(gv_cull(ctx, p, views, p == 0 ? 3u : 1u));
        if (sort)
            CHECK(gv_pool_sort(ctx, p, 0, p & 1));
    }
    for (uint32_t p = 0; p < pools; p++) {
        GvResult r{};
        CHECK(gv_pool_results_fetch(ctx, p, 0, 1, &r));
        if (r.draw_count != 0) {  // kernels are no-ops and device memory is zeroed
            std::fprintf(stderr, "draw_count %u from no-op kernels\n", r.draw_count);
            std::exit(1);
        }
        uint32_t count = 7;
        CHECK(gv_pool_result_count(ctx, p, 0, &count));
        GvDeviceResult d{};
        CHECK(gv_pool_results_device(ctx, p, 0, &d));
    }
}

static void exercise(uint32_t config_flags, uint32_t n, uint32_t depth)
{
    GvConfig cfg{(uint32_t)sizeof(GvConfig), 0, GV_HIZ_RULE_REFERENCE, config_flags};
    GvCtx* ctx = nullptr;
    if (gv_create(&cfg, &ctx) != GV_OK) {
        std::fprintf(stderr, "gv_create: %s\n", gv_last_error(nullptr));
        std::exit(1);
    }
    World w;
    w.build(n, depth);
    const GvView one = make_view(-1, 0, 1);
    // ---- calls out of order / bad arguments ----
    EXPECT(gv_cull(ctx, 0, &one, 1), GV_E_STATE);
    EXPECT(gv_cull(ctx, GV_MAX_POOLS, &one, 1), GV_E_ARG);
    EXPECT(gv_hiz_rebuild(ctx), GV_E_STATE);
    {
        GvTransformLayout tl = transform_layout();
        EXPECT(gv_transform_bind(ctx, w.xf.data(), 8, (uint32_t)w.xf.size(), &tl, w.e2t.data(), (uint32_t)w.e2t.size()), GV_E_ARG);
        GvMeshLayout ml = mesh_layout();
        EXPECT(gv_pool_bind(ctx, 0, w.meshes.data(), 16, (uint32_t)w.meshes.size(), &ml), GV_E_ARG);
        EXPECT(gv_mark_dirty(ctx, 9, 0, 1), GV_E_ARG);
    }
    // ---- first build, both pools ----
    bind_all(ctx, w);
    CHECK(gv_hierarchy_rebuild(ctx));
    check_permutation(ctx, 0, (uint32_t)w.meshes.size());
    check_permutation(ctx, 1, (uint32_t)w.big.size());
    frame(ctx, 2, false, false);
    // ---- Hi-Z builds: sizes that take every kernel path's host logic, host and "device" memory ----
    for (const auto& size : {std::pair<uint32_t, uint32_t>{256, 128}, {135, 77}, {1, 1}, {1920, 1080}, {64, 4096}}) {
        std::vector<float> depth_image((size_t)size.first * size.second, 0.25f);
        CHECK(gv_hiz_build(ctx, depth_image.data(), size.first, size.second, GV_MEM_HOST));
        uint32_t mips = 0;
        CHECK(gv_hiz_mip_count(ctx, &mips));
        for (uint32_t level = 1; level < mips; level++) {
            uint32_t lw = 0, lh = 0;
            std::vector<float> pairs((size_t)std::max(size.first >> level, 1u) * std::max(size.second >> level, 1u) * 2);
            CHECK(gv_hiz_read_level(ctx, level, pairs.data(), &lw, &lh));
        }
        CHECK(gv_hiz_rebuild(ctx));
        frame(ctx, 1, true, true);
    }
    EXPECT(gv_hiz_build(ctx, nullptr, 4, 4, GV_MEM_HOST), GV_E_ARG);
    // ---- sweeps and the world-matrix cache ----
    for (uint32_t mode : {GV_SWEEP_VALU, GV_SWEEP_MFMA, GV_SWEEP_INCREMENTAL, GV_SWEEP_WITH_CULL, GV_SWEEP_WITH_CULL_VALU}) {
        CHECK(gv_sweep(ctx, mode));
        frame(ctx, 2, false, false);
    }
    CHECK(gv_sweep(ctx, GV_SWEEP_VALU));
    {
        std::vector<float> world((size_t)1000 * 12);
        CHECK(gv_get_world(ctx, n / 2, 1000, world.data()));
        EXPECT(gv_get_world(ctx, n - 10, 1000, world.data()), GV_E_ARG);
    }
    // ---- dirty ranges: few, thousands of scattered ones, large (device-side gather), everything; meshes too ----
    std::mt19937 rng(99);
    for (uint32_t count : {3u, 40u, 3000u}) {
        for (uint32_t k = 0; k < count; k++) {
            const uint32_t s = rng() % n;
            w.xf[s].position[0] += 1.0f;
            CHECK(gv_mark_dirty(ctx, GV_DIRTY_TRANSFORM, s, 1 + k % 3 < n - s ? 1 + k % 3 : 1));
            const uint32_t ms = rng() % n;
            w.meshes[ms].aabbMax[1] += 0.1f;
            CHECK(gv_mark_dirty(ctx, GV_DIRTY_MESH, ms, 1));
            CHECK(gv_mark_dirty(ctx, GV_DIRTY_MESH, (1u << 28) | (rng() % (uint32_t)w.big.size()), 1));
        }
        CHECK(gv_sweep(ctx, GV_SWEEP_INCREMENTAL));
        frame(ctx, 2, false, false);
    }
    CHECK(gv_mark_dirty(ctx, GV_DIRTY_TRANSFORM, n / 4, n / 3));       // one large range: raw AoS span + device gather
    CHECK(gv_mark_dirty(ctx, GV_DIRTY_TRANSFORM, n - 5, 100));          // reaches past the pool: clamped
    CHECK(gv_mark_dirty(ctx, GV_DIRTY_TRANSFORM, 0xFFFFFFF0u, 0x40u));  // wraps: saturated, then clamped
    CHECK(gv_sync(ctx));
    CHECK(gv_mark_dirty(ctx, GV_DIRTY_TRANSFORM, 0, n));                // most of the pool: the dense forms
    CHECK(gv_mark_dirty(ctx, GV_DIRTY_MESH, 0, n));
    frame(ctx, 2, false, true);
    // ---- re-parenting (ranged GV_DIRTY_HIERARCHY), a cycle (rejected), its repair ----
    for (uint32_t k = 0; k < 50; k++) {
        const uint32_t s = n / 2 + rng() % (n / 2), p = rng() % (n / 2);
        if (w.xf[s].entity && w.xf[p].entity) {
            w.xf[s].parent = w.xf[p].entity;
            CHECK(gv_mark_dirty(ctx, GV_DIRTY_HIERARCHY, s, 1));
        }
    }
    CHECK(gv_sync(ctx));
    {
        uint32_t a = 10, b = 11;
        while (!w.xf[a].entity || !w.xf[b].entity)
            a += 2, b += 2;
        const uint32_t pa = w.xf[a].parent, pb = w.xf[b].parent;
        w.xf[a].parent = w.xf[b].entity;
        w.xf[b].parent = w.xf[a].entity;
        CHECK(gv_mark_dirty(ctx, GV_DIRTY_HIERARCHY, a, 2));
        EXPECT(gv_sync(ctx), GV_E_ARG);  // a cycle
        w.xf[a].parent = pa;
        w.xf[b].parent = pb;
        CHECK(gv_hierarchy_rebuild(ctx));
    }
    // ---- pools that grow (appended to the mirror), move, and shrink (rebuilt) ----
    for (int round = 0; round < 6; round++) {
        const uint32_t old = (uint32_t)w.xf.size(), grown = old + old / 20 + 3;
        World bigger;
        bigger.rng.seed(round);
        bigger.build(grown, depth);
        for (uint32_t i = 0; i < old; i++) {  // the old slots keep their contents (at new addresses)
            bigger.xf[i] = w.xf[i];
            bigger.meshes[i] = w.meshes[i];
        }
        bigger.e2t.assign(grown + 1, GV_NONE);
        for (uint32_t i = 0; i < grown; i++)
            if (bigger.xf[i].entity) {
                bigger.xf[i].entity = i + 1;
                bigger.meshes[i].entity = i + 1;
                if (i >= old)
                    bigger.xf[i].parent = 0;
                bigger.e2t[i + 1] = i;
            }
        w.xf.swap(bigger.xf);
        w.meshes.swap(bigger.meshes);
        w.e2t.swap(bigger.e2t);
        bind_all(ctx, w);
        CHECK(gv_mark_dirty(ctx, GV_DIRTY_TRANSFORM, old, grown - old));
        CHECK(gv_mark_dirty(ctx, GV_DIRTY_MESH, old, grown - old));
        frame(ctx, 2, false, round & 1);
        check_permutation(ctx, 0, (uint32_t)w.meshes.size());
    }
    {   // six rounds of +5 %: the unsorted tail has passed 1/8 of the pools on the way — a spatially ordered mirror was re-ordered
        // where it lies (gv_mirror.cpp reorder_*_device; the stub build runs tests/cpp/hip_stub/reorder_cpu.cpp for its kernels)
        GvStats st{};
        CHECK(gv_stats(ctx, &st));
        if ((st.mirror_reorders != 0) == ((config_flags & GV_CONFIG_KEEP_SLOT_ORDER) != 0)) {
            std::fprintf(stderr, "mirror_reorders = %llu with config flags %#x\n", (unsigned long long)st.mirror_reorders, config_flags);
            std::exit(1);
        }
        // dirty marks after the re-order land on the new entries; a hierarchy mark re-validates the downloaded links
        CHECK(gv_mark_dirty(ctx, GV_DIRTY_TRANSFORM, 7, 3000));
        CHECK(gv_mark_dirty(ctx, GV_DIRTY_TRANSFORM, 3, 2));
        CHECK(gv_mark_dirty(ctx, GV_DIRTY_MESH, 11, 2500));
        CHECK(gv_mark_dirty(ctx, GV_DIRTY_HIERARCHY, 40, 10));
        frame(ctx, 2, false, false);
        check_permutation(ctx, 0, (uint32_t)w.meshes.size());
    }
    w.build(n / 2, depth);  // shrinks: rebuilt
    bind_all(ctx, w);
    frame(ctx, 2, false, false);
    check_permutation(ctx, 0, (uint32_t)w.meshes.size());
    // ---- ready columns ----
    {
        std::vector<uint8_t> ready8(w.meshes.size(), 1);
        std::vector<uint32_t> ready32(w.meshes.size(), 2);
        ready8[5] = 0;
        CHECK(gv_pool_bind_ready(ctx, 0, ready8.data(), 1, 1));
        frame(ctx, 1, false, false);
        CHECK(gv_pool_bind_ready(ctx, 0, ready32.data(), 4, 4));
        frame(ctx, 1, false, false);
        EXPECT(gv_pool_bind_ready(ctx, 0, ready32.data(), 4, 2), GV_E_ARG);
        CHECK(gv_pool_bind_ready(ctx, 0, nullptr, 0, 0));
    }
    // ---- records in the engine's struct; the engine's own array as the target: set, grow + replace, too small, removed ----
    {
        const GvRecordLayout layout{64, 0, 8, 56, GV_NONE, (uint32_t)sizeof(Mesh), 0};
        CHECK(gv_pool_set_record_layout(ctx, 0, &layout));
        const GvRecordLayout bad{60, 0, 8, 56, GV_NONE, (uint32_t)sizeof(Mesh), 0};
        EXPECT(gv_pool_set_record_layout(ctx, 0, &bad), GV_E_ARG);
        std::vector<uint8_t> a(w.meshes.size() * 64 + 16), b(w.meshes.size() * 64 * 2 + 16), small(w.meshes.size() * 64 - 64 + 16);
        auto aligned = [](std::vector<uint8_t>& v) { return reinterpret_cast<void*>(((uintptr_t)v.data() + 15) & ~(uintptr_t)15); };
        CHECK(gv_pool_set_record_target(ctx, 0, 0, aligned(a), a.size() - 16));
        frame(ctx, 1, false, true);
        const void* records = nullptr;
        uint32_t count = 1;
        CHECK(gv_pool_results_records(ctx, 0, 0, &records, &count));
        const uint32_t* bases = nullptr;
        CHECK(gv_pool_results_instance_bases(ctx, 0, 0, &bases, &count));
        CHECK(gv_pool_set_record_target(ctx, 0, 0, aligned(b), b.size() - 16));  // replaced while the old one is still allocated
        frame(ctx, 1, false, false);
        CHECK(gv_pool_set_record_target(ctx, 0, 0, aligned(small), small.size() - 16));
        GvView v = make_view(-1, 0, 1);
        CHECK(gv_cull(ctx, 0, &v, 1));
        GvResult r{};
        EXPECT(gv_pool_results_fetch(ctx, 0, 0, 0, &r), GV_E_ARG);  // smaller than occupancy * stride
        EXPECT(gv_pool_set_record_target(ctx, 0, 0, reinterpret_cast<uint8_t*>(aligned(a)) + 4, 64), GV_E_ARG);  // misaligned
        CHECK(gv_pool_set_record_target(ctx, 0, 0, nullptr, 0));
        CHECK(gv_pool_set_record_layout(ctx, 0, nullptr));
        frame(ctx, 1, false, true);
    }
    // ---- exchange helpers: index map, device copies into caller memory ----
    {
        std::vector<uint32_t> global_ids(w.meshes.size());
        for (size_t i = 0; i < global_ids.size(); i++)
            global_ids[i] = (uint32_t)(i * 3);
        CHECK(gv_pool_set_index_map(ctx, 0, global_ids.data(), (uint32_t)global_ids.size()));
        GvView v = make_view(-1, 0, 1);
        CHECK(gv_cull(ctx, 0, &v, 1));
        std::vector<uint32_t> dst(w.meshes.size() + 1), words((w.meshes.size() + 31) / 32 + 1);
        CHECK(gv_results_copy_idx_device(ctx, 0, dst.data(), (uint32_t)w.meshes.size(), 100));
        CHECK(gv_results_copy_shard_device(ctx, 0, dst.data(), (uint32_t)w.meshes.size(), 100));
        CHECK(gv_results_copy_mask_device(ctx, 0, words.data(), (uint32_t)words.size() - 1));
        EXPECT(gv_results_copy_mask_device(ctx, 0, words.data(), 1), GV_E_ARG);
        CHECK(gv_pool_set_index_map(ctx, 0, nullptr, 0));
        CHECK(gv_wait(ctx));
    }
    // ---- column (SoA) binds: every field its own array, padded strides ----
    {
        const uint32_t m = (uint32_t)w.xf.size();
        std::vector<uint32_t> entity(m), parent(m);
        std::vector<float> pos((size_t)m * 4), scl((size_t)m * 3), rot((size_t)m * 4), mn((size_t)m * 3), mx((size_t)m * 5);
        std::vector<uint8_t> sa(m), aa(m), wa(m), en(m), vis(m);
        for (uint32_t i = 0; i < m; i++) {
            entity[i] = w.xf[i].entity;
            parent[i] = w.xf[i].parent;
            for (int k = 0; k < 3; k++) {
                pos[(size_t)i * 4 + k] = w.xf[i].position[k];
                scl[(size_t)i * 3 + k] = w.xf[i].scale[k];
                mn[(size_t)i * 3 + k] = w.meshes[i].aabbMin[k];
                mx[(size_t)i * 5 + k] = w.meshes[i].aabbMax[k];
            }
            for (int k = 0; k < 4; k++)
                rot[(size_t)i * 4 + k] = w.xf[i].rotation[k];
            sa[i] = w.xf[i].selfActive, aa[i] = w.xf[i].ancestorsActive, wa[i] = w.xf[i].modelWithAncestors, en[i] = w.meshes[i].isEnabled;
        }
        GvTransformColumns tc{{entity.data(), 4}, {parent.data(), 4}, {pos.data(), 16}, {scl.data(), 12}, {rot.data(), 16}, {sa.data(), 1}, {aa.data(), 1}, {wa.data(), 1}};
        GvMeshColumns mc{{entity.data(), 4}, {en.data(), 1}, {mn.data(), 12}, {mx.data(), 20}, vis.data(), 1};
        CHECK(gv_transform_bind_columns(ctx, &tc, m, w.e2t.data(), (uint32_t)w.e2t.size()));
        CHECK(gv_pool_bind_columns(ctx, 0, &mc, m));
        CHECK(gv_hierarchy_rebuild(ctx));
        CHECK(gv_mark_dirty(ctx, GV_DIRTY_TRANSFORM, m / 3, m / 2));  // columns: the host gather, whatever the size
        frame(ctx, 1, false, false);
        GvTransformColumns broken = tc;
        broken.rotation.stride = 8;
        EXPECT(gv_transform_bind_columns(ctx, &broken, m, w.e2t.data(), (uint32_t)w.e2t.size()), GV_E_ARG);
        bind_all(ctx, w);
        CHECK(gv_hierarchy_rebuild(ctx));
    }
    // ---- a tick of engine-sized pools: recorded culls, changes inside the batch, deferred sorts ----
    {
        World small;
        small.build(9000, depth);
        const GvTransformLayout tl = transform_layout();
        const GvMeshLayout ml = mesh_layout();
        CHECK(gv_transform_bind(ctx, small.xf.data(), sizeof(Transform), (uint32_t)small.xf.size(), &tl, small.e2t.data(), (uint32_t)small.e2t.size()));
        for (uint32_t p = 0; p < 5; p++)
            CHECK(gv_pool_bind(ctx, p, small.meshes.data(), sizeof(Mesh), (uint32_t)small.meshes.size() - p * 1000, &ml));
        for (int tick = 0; tick < 4; tick++) {
            CHECK(gv_cull_batch_begin(ctx));
            GvView views[4] = {make_view(-1, 0, 1), make_view(0, 0, 1), make_view(1, 0, 1), make_view(2, 0, 1)};
            for (uint32_t p = 0; p < 5; p++) {
                CHECK(gv_cull(ctx, p, views, 1 + p % 4));
                for (uint32_t v = 0; v < 1 + p % 4; v++)
                    CHECK(gv_pool_sort(ctx, p, v, v & 1));
                if (tick == 2 && p == 2) {  // a change in the middle of the batch
                    CHECK(gv_mark_dirty(ctx, GV_DIRTY_TRANSFORM, 100, 50));
                    CHECK(gv_pool_bind(ctx, 1, small.meshes.data(), sizeof(Mesh), 4000, &ml));
                }
            }
            for (uint32_t p = 0; p < 5; p++)
                for (uint32_t v = 0; v < 1 + p % 4; v++) {
                    GvResult r{};
                    if (tick == 2 && p == 1 && v == 0)  // its pool was re-bound (smaller) after the cull: no write-back into it
                        EXPECT(gv_pool_results_fetch(ctx, p, v, 1, &r), GV_E_STATE);
                    CHECK(gv_pool_results_fetch(ctx, p, v, (tick != 2 || p != 1) && v == 0, &r));
                }
            CHECK(gv_cull_batch_end(ctx));
        }
        for (uint32_t p = 2; p < 5; p++)  // (these systems go away with `small`: an empty pool needs no memory)
            CHECK(gv_pool_bind(ctx, p, nullptr, sizeof(Mesh), 0, &ml));
        bind_all(ctx, w);
        CHECK(gv_hierarchy_rebuild(ctx));
    }
    // ---- a frame of MID-SIZED pools (beyond the one-launch publish and the one-launch batch sort): their sorts wait for the first read
    //      and go out together (launch_sort_batch); structs, three arrays and a record target side by side ----
    {
        World large;
        large.build(40000, depth);
        const GvTransformLayout tl = transform_layout();
        const GvMeshLayout ml = mesh_layout();
        CHECK(gv_transform_bind(ctx, large.xf.data(), sizeof(Transform), (uint32_t)large.xf.size(), &tl, large.e2t.data(), (uint32_t)large.e2t.size()));
        for (uint32_t p = 0; p < 3; p++)
            CHECK(gv_pool_bind(ctx, p, large.meshes.data(), sizeof(Mesh), (uint32_t)large.meshes.size() - p * 2000, &ml));
        const GvRecordLayout layout = {64, 0, 8, 56, GV_NONE, sizeof(Mesh), 0};
        CHECK(gv_pool_set_record_layout(ctx, 1, &layout));  // pool 1 delivers structs, the others the three arrays
        CHECK(gv_pool_set_record_layout(ctx, 0, nullptr));
        CHECK(gv_pool_set_record_layout(ctx, 2, nullptr));
        std::vector<uint8_t> own_records((size_t)40000 * 64 + 16);
        void* target = reinterpret_cast<void*>(((uintptr_t)own_records.data() + 15) & ~(uintptr_t)15);
        for (int tick = 0; tick < 3; tick++) {
            GvView views[3] = {make_view(-1, 0, 1), make_view(0, 0, 1), make_view(1, 0, 1)};
            CHECK(gv_pool_set_record_target(ctx, 1, 0, tick == 1 ? target : nullptr, tick == 1 ? (size_t)40000 * 64 : 0));
            for (uint32_t p = 0; p < 3; p++) {
                CHECK(gv_cull(ctx, p, views, 1 + p));
                for (uint32_t v = 0; v < 1 + p; v++)
                    CHECK(gv_pool_sort(ctx, p, v, v & 1));
            }
            if (tick == 2)
                CHECK(gv_pool_sort(ctx, 2, 1, 0));  // asked for twice before anyone reads: the later request stands
            for (uint32_t p = 0; p < 3; p++)
                for (uint32_t v = 0; v < 1 + p; v++) {
                    GvResult r{};
                    CHECK(gv_pool_results_fetch(ctx, p, v, v == 0, &r));
                    const void* records = nullptr;
                    uint32_t count = 0;
                    if (p == 1)
                        CHECK(gv_pool_results_records(ctx, p, v, &records, &count));
                }
        }
        CHECK(gv_pool_set_record_target(ctx, 1, 0, nullptr, 0));
        CHECK(gv_pool_set_record_layout(ctx, 1, nullptr));
        CHECK(gv_pool_bind(ctx, 2, nullptr, sizeof(Mesh), 0, &ml));
        bind_all(ctx, w);
        CHECK(gv_hierarchy_rebuild(ctx));
    }
    // ---- scene ingest -> columns -> bind; tiles ----
    {
        std::string text = "{\"version\":\"0.0.1\",\"entities\":[";
        for (int e = 0; e < 300; e++) {
            char buf[512];
            std::snprintf(buf, sizeof(buf),
                          "%s{\"components\":[{\".type\":\"Transform\",\"uid\":\"AAAAAAAAA%02d\",\"position\":{\"x\":%d.5,\"y\":%d.25,\"z\":-%d.0}%s},"
                          "{\".type\":\"Model\",\"aabb\":{\"min\":{\"x\":-1.0,\"y\":-1.0,\"z\":-1.0},\"max\":{\"x\":1.0,\"y\":2.0,\"z\":1.0}}}]}",
                          e ? "," : "", e % 64, e * 7 % 900 - 450, e * 3 % 700 - 350, e * 11 % 800,
                          e % 5 == 4 ? ",\"parent\":\"AAAAAAAAA00\"" : "");
            text += buf;
        }
        text += "]}";
        const GvScenePool pools[1] = {{"Model", 0}};
        GvScene* scene = nullptr;
        char error[256];
        const int rc = gv_scene_parse_json(text.c_str(), text.size(), pools, 1, GV_SCENE_ADD_ROOT_ENTITY, &scene, error, sizeof(error));
        if (rc != GV_OK) {
            std::fprintf(stderr, "scene: %s\n", error);
            std::exit(1);
        }
        CHECK(gv_scene_bind(ctx, scene));
        frame(ctx, 1, false, true);
        const uint32_t grid[3] = {2, 2, 1};
        for (uint32_t t = 0; t < 4; t++) {
            GvScene* tile = nullptr;
            CHECK(gv_scene_extract_tile(scene, grid, 1000.0, t, &tile));
            CHECK(gv_scene_bind(ctx, tile));
            frame(ctx, 1, false, false);
            gv_scene_destroy(tile);
            CHECK(gv_scene_bind(ctx, scene));
        }
        GvScene* none = nullptr;
        EXPECT(gv_scene_parse_json(text.c_str(), text.size() / 2, pools, 1, 0, &none, error, sizeof(error)), GV_E_ARG);
        bind_all(ctx, w);
        CHECK(gv_hierarchy_rebuild(ctx));
        gv_scene_destroy(scene);
    }
    GvStats stats{};
    CHECK(gv_stats(ctx, &stats));
    CHECK(gv_stats_reset(ctx));
    gv_destroy(ctx);
}

// ---- the exchange step with several ranks over tests/cpp/rccl_stub (GV_RCCL_LIBRARY): one context per thread, or ONE thread that
// drives all the contexts through the *_all forms ----
// Kernels are no-ops here, so every rank WRITES the list it wants to exchange into its view's result buffers ("device" memory
// is host memory in this build): rank r's list in frame f has count(r, f) entries value(r, f, k). Lists creep, jump (the
// prediction is short: the frame is completed by a second exchange before it is handed out) and collapse (room is given back);
// every rank checks that EVERY row of EVERY frame holds its rank's whole list, under all three travel patterns.
static std::atomic<int> g_cut_frames{0}, g_tail_words{0};  // (summed over the ranks of a run, for the log)
static int g_list_seed = 0;  // 0: the scripted sequence below; otherwise lists that jump at random between empty and the whole pool
static uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
static uint32_t list_count(int rank, int frame, uint32_t n)
{
    if (g_list_seed) {
        // a scale that holds for a few frames (shared by the ranks: a camera), a rank's own share of it, a per-frame wobble;
        // now and then nothing at all, or everything
        const uint32_t era = mix32((uint32_t)g_list_seed * 977u + (uint32_t)(frame / 3));
        const uint32_t scales[6] = {0u, 40u, 700u, 2600u, n / 2u, n};
        const uint32_t scale = scales[era % 6u];
        const uint32_t share = 50u + mix32((uint32_t)g_list_seed * 31u + (uint32_t)rank) % 100u;                    // 50 .. 149 %
        const uint32_t wobble = 90u + mix32((uint32_t)g_list_seed * 131071u + (uint32_t)rank * 8191u + (uint32_t)frame) % 21u;  // 90 .. 110 %
        return (uint32_t)std::min<uint64_t>(n, (uint64_t)scale * share * wobble / 10000u);
    }
    const uint32_t base = 200u + 150u * (uint32_t)rank;
    if (frame >= 14)
        return base / 8u;                                       // collapse: room is given back
    if (frame >= 7)
        return std::min(n, base * 9u + 37u * (uint32_t)frame);  // jump: every rank's list outgrows its room
    return base + 11u * (uint32_t)frame * (uint32_t)(rank + 1);  // creep
}
static uint32_t list_value(int rank, int frame, uint32_t k) { return (uint32_t)rank * 1000003u + (uint32_t)frame * 7919u + k; }

struct ExchangeRank {
    World w;
    GvCtx* ctx = nullptr;
    int rank = 0, ranks = 1, failures = 0, cut_frames = 0;
    uint64_t tail_words = 0;
    static constexpr uint32_t n = 6000;
    void fail(const char* what, int frame, int row)
    {
        std::fprintf(stderr, "exchange rank %d frame %d row %d: %s\n", rank, frame, row, what);
        ++failures;
    }
    bool create(int rank_, int ranks_)
    {
        rank = rank_;
        ranks = ranks_;
        w.rng.seed(777u + (uint32_t)rank);
        w.build(n, 0);
        GvConfig config{};
        config.struct_size = sizeof(config);
        if (gv_create(&config, &ctx) != GV_OK) {
            fail("gv_create", -1, -1);
            return false;
        }
        bind_all(ctx, w);
        return true;
    }
    // the frame's list, written where the (no-op) emit would have left it
    void produce(int frame, uint32_t mode)
    {
        GvView v = make_view(-1, 0, 1);
        CHECK(gv_exchange_set_mode(ctx, mode));
        CHECK(gv_cull(ctx, 0, &v, 1));
        GvDeviceResult dr{};
        CHECK(gv_results_device(ctx, 0, &dr));
        const uint32_t mine = list_count(rank, frame, n);
        *(uint32_t*)dr.draw_count = mine;
        for (uint32_t k = 0; k < mine; k++)
            ((uint32_t*)dr.visible_idx)[k] = list_value(rank, frame, k);
    }
    // another mesh system culled AFTER the list was produced (gv_results_device and the view-indexed exchange now address pool 1)
    static constexpr uint32_t other_count = 7;
    void cull_other_pool()
    {
        GvView v = make_view(-1, 0, 1);
        CHECK(gv_cull(ctx, 1, &v, 1));
        GvDeviceResult dr{};
        CHECK(gv_results_device(ctx, 0, &dr));
        *(uint32_t*)dr.draw_count = other_count;
        for (uint32_t k = 0; k < other_count; k++)
            ((uint32_t*)dr.visible_idx)[k] = 4200u + (uint32_t)rank + k;
    }
    void check_sent(const GvExchangeFrame& xf, int frame, uint32_t mode)
    {
        if (xf.world_size != (uint32_t)ranks || xf.frame != (uint64_t)frame || xf.mode != mode || xf.complete || xf.gathered_device || xf.ready_event)
            fail("fields of a frame that was sent, not acquired", frame, -1);
        if (xf.row_words % 4u)
            fail("row stride is not a multiple of 4 words", frame, -1);
        for (int r = 0; r < ranks; r++) {
            if (xf.travelled_words[r] != (mode == GV_EXCHANGE_ALLGATHER ? xf.row_words : xf.room[r] + 1) || xf.room[r] + 1 > xf.row_words)
                fail("travelled words", frame, r);
            if (frame >= 1 && xf.room[r] < list_count(r, frame - 1, n))  // (rooms follow the previous frame's headers, never below them)
                fail("room below the previous frame's list", frame, r);
        }
    }
    // an acquired frame: complete, whatever the prediction was
    void check_acquired(const GvExchangeFrame& sent, const GvExchangeFrame& xf, int frame)
    {
        if (!xf.complete || !xf.gathered_device || !xf.ready_event || xf.frame != (uint64_t)frame || xf.world_size != (uint32_t)ranks || xf.mode != sent.mode ||
            xf.row_words % 4u || xf.row_words < sent.row_words)
            fail("fields of an acquired frame", frame, -1);
        const uint32_t* rows = (const uint32_t*)xf.gathered_device;
        bool cut = false;
        for (int r = 0; r < ranks; r++) {
            const uint32_t* row = rows + (size_t)r * xf.row_words;
            const uint32_t count = list_count(r, frame, n);
            if (xf.room[r] != sent.room[r] || xf.travelled_words[r] != sent.travelled_words[r])
                fail("an acquired frame reports other rooms than it was sent with", frame, r);
            if (row[0] != count || xf.counts[r] != count)
                fail("header / count", frame, r);
            if ((size_t)count + 1 > xf.row_words)
                fail("a list longer than its row", frame, r);
            for (uint32_t k = 0; k < count; k++)  // the WHOLE list: no row of an acquired frame is short
                if (row[1 + k] != list_value(r, frame, k)) {
                    fail("entry", frame, r);
                    break;
                }
            const bool short_row = count > xf.room[r];
            if (short_row != (((xf.cut_ranks >> r) & 1u) != 0) || xf.tail_words[r] != (short_row ? count - xf.room[r] : 0u))
                fail("cut statistics", frame, r);
            cut = cut || short_row;
            tail_words += xf.tail_words[r];
        }
        cut_frames += cut ? 1 : 0;
    }
    void check_shards()
    {
        GvView v = make_view(-1, 0, 1);
        const uint32_t capacity = n;
        std::vector<uint32_t> rows((size_t)ranks * (capacity + 1), 0xDEADBEEFu);
        uint32_t caps[GV_EXCHANGE_MAX_RANKS];
        for (int r = 0; r < ranks; r++)
            caps[r] = 100u + 40u * (uint32_t)r;
        for (uint32_t mode = 0; mode < 3; mode++) {
            CHECK(gv_exchange_set_mode(ctx, mode));
            CHECK(gv_cull(ctx, 0, &v, 1));
            GvDeviceResult dr{};
            CHECK(gv_results_device(ctx, 0, &dr));
            const uint32_t mine = 120u + 30u * (uint32_t)rank;
            *(uint32_t*)dr.draw_count = mine;
            for (uint32_t k = 0; k < mine; k++)
                ((uint32_t*)dr.visible_idx)[k] = list_value(rank, 99, k);
            CHECK(gv_exchange_shards(ctx, 0, capacity, caps, 5, rows.data()));
            for (int r = 0; r < ranks; r++) {
                const uint32_t* row = rows.data() + (size_t)r * (capacity + 1);
                const uint32_t count = 120u + 30u * (uint32_t)r;
                if (row[0] != count)
                    fail("shards header", (int)mode, r);
                for (uint32_t k = 0; k < std::min(count, caps[r]); k++)
                    if (row[1 + k] != list_value(r, 99, k) + 5u) {
                        fail("shards entry", (int)mode, r);
                        break;
                    }
            }
        }
        caps[0] = capacity + 1;
        EXPECT(gv_exchange_shards(ctx, 0, capacity, caps, 0, rows.data()), GV_E_ARG);
    }
};

static void exchange_rank_thread(int rank, int ranks, const unsigned char* id, int* failures)
{
    std::unique_ptr<ExchangeRank> owner(new ExchangeRank());
    if (!owner->create(rank, ranks)) {
        ++*failures;
        return;
    }
    GvCtx* ctx = owner->ctx;
    ExchangeRank& x = *owner;
    CHECK(gv_exchange_init(ctx, id, rank, ranks));
    GvExchangeFrame previous{};
    for (int frame = 0; frame < 20; frame++) {
        const uint32_t mode = (uint32_t)(frame % 3);
        x.produce(frame, mode);
        GvExchangeFrame sent, got;
        CHECK(gv_exchange_visible(ctx, 0, 0, 0, &sent));
        x.check_sent(sent, frame, mode);
        // ranks acquire at different points — at once, or a frame late (the next gv_exchange_visible has then completed the
        // frame already) — the same on every rank for a given frame: acquiring is a collective where rows were short
        if (frame % 4 != 1) {
            CHECK(gv_exchange_acquire(ctx, (uint64_t)frame, &got));
            x.check_acquired(sent, got, frame);
            CHECK(gv_exchange_acquire(ctx, (uint64_t)frame, &got));  // (again: nothing left to do, the same answer)
            if (!got.complete)
                x.fail("second acquire", frame, -1);
        }
        if (frame >= 1 && (frame - 1) % 4 == 1) {
            CHECK(gv_exchange_acquire(ctx, (uint64_t)frame - 1, &got));
            x.check_acquired(previous, got, frame - 1);
        }
        if (frame >= 3)
            EXPECT(gv_exchange_acquire(ctx, (uint64_t)frame - 3, nullptr), GV_E_ARG);  // rows long since reused
        previous = sent;
    }
    g_cut_frames += x.cut_frames;
    g_tail_words += (int)x.tail_words;
    if (x.cut_frames < 1 && !g_list_seed)
        x.fail("the jump of the scripted sequence never outgrew a prediction", -1, -1);
    x.check_shards();
    GvExchangeFrame none;
    EXPECT(gv_exchange_visible(ctx, 0, 0, 0x80, &none), GV_E_ARG);
    EXPECT(gv_exchange_visible(ctx, 0, 0, 1, &none), GV_E_ARG);  // (round 4's GV_EXCHANGE_EXACT: gone, not ignored)
    CHECK(gv_exchange_shutdown(ctx));
    EXPECT(gv_exchange_visible(ctx, 0, 0, 0, &none), GV_E_STATE);
    EXPECT(gv_exchange_acquire(ctx, 0, &none), GV_E_STATE);
    // a second communicator on the same context starts from nothing: no room of the first one sizes its rows (ranks whose
    // contexts have different pasts — this one keeps its history, rank 1 below is given a FRESH context — must agree on frame 0)
    int carried = x.failures;
    if (ranks > 1 && rank == 1) {
        gv_destroy(ctx);
        owner.reset(new ExchangeRank());
        if (!owner->create(rank, ranks)) {
            ++*failures;
            return;
        }
        ctx = owner->ctx;
    } else {
        owner->failures = 0;
    }
    ExchangeRank& y = *owner;
    unsigned char id2[GV_EXCHANGE_ID_BYTES];
    memcpy(id2, id + GV_EXCHANGE_ID_BYTES, GV_EXCHANGE_ID_BYTES);
    CHECK(gv_exchange_init(ctx, id2, rank, ranks));
    for (int frame = 0; frame < 3; frame++) {
        y.produce(frame + 5, GV_EXCHANGE_P2P);
        GvExchangeFrame sent, got;
        CHECK(gv_exchange_visible(ctx, 0, 0, 0, &sent));
        for (int r = 0; r < ranks; r++)
            if (frame == 0 && sent.room[r] != 0)
                y.fail("frame 0 of a new communicator is sized from an earlier one's rooms", frame, r);
        CHECK(gv_exchange_acquire(ctx, (uint64_t)frame, &got));
        if (!got.complete)
            y.fail("re-init: acquire", frame, -1);
        const uint32_t* rows = (const uint32_t*)got.gathered_device;
        for (int r = 0; r < ranks; r++)
            if (rows[(size_t)r * got.row_words] != list_count(r, frame + 5, ExchangeRank::n))
                y.fail("re-init: header", frame, r);
    }
    CHECK(gv_exchange_shutdown(ctx));
    *failures += carried + y.failures;
    gv_destroy(ctx);
}

static void exchange_in_threads(int ranks, int list_seed = 0)
{
    g_list_seed = list_seed;
    g_cut_frames = 0;
    g_tail_words = 0;
    if (!std::getenv("GV_RCCL_LIBRARY")) {
        std::printf("exchange over the stub transport: skipped (GV_RCCL_LIBRARY not set)\n");
        return;
    }
    unsigned char id[2 * GV_EXCHANGE_ID_BYTES];  // (two communicators, one after the other)
    if (gv_exchange_unique_id(id) != GV_OK || gv_exchange_unique_id(id + GV_EXCHANGE_ID_BYTES) != GV_OK) {
        std::fprintf(stderr, "gv_exchange_unique_id failed\n");
        std::exit(1);
    }
    std::vector<int> failures(ranks, 0);
    std::vector<std::thread> threads;
    for (int r = 0; r < ranks; r++)
        threads.emplace_back(exchange_rank_thread, r, ranks, id, &failures[r]);
    for (auto& t : threads)
        t.join();
    for (int r = 0; r < ranks; r++)
        if (failures[r]) {
            std::fprintf(stderr, "exchange with %d ranks: rank %d reported %d failures\n", ranks, r, failures[r]);
            std::exit(1);
        }
    std::printf("exchange over the stub transport, %d ranks, list sequence %d: ok — every row of every frame complete (per rank: %d of 20 frames needed a "
                "second exchange, %d words in tails)\n", ranks, list_seed, g_cut_frames.load() / ranks, g_tail_words.load() / ranks);
}

// ONE thread, N contexts (the reference's shape: one process, one Manager — source/editor/entry.cpp:135): gv_exchange_init_all /
// _visible_all / _acquire_all put the ranks' collectives inside one group, so that a single thread can issue all of them.
static void exchange_in_one_thread(int ranks, int list_seed)
{
    g_list_seed = list_seed;
    if (!std::getenv("GV_RCCL_LIBRARY")) {
        std::printf("exchange driven by one thread: skipped (GV_RCCL_LIBRARY not set)\n");
        return;
    }
    std::vector<ExchangeRank> xs(ranks);
    std::vector<GvCtx*> ctxs;
    for (int r = 0; r < ranks; r++) {
        if (!xs[r].create(r, ranks))
            std::exit(1);
        ctxs.push_back(xs[r].ctx);
    }
    GvCtx* ctx = ctxs[0];  // (CHECK prints its error text)
    CHECK(gv_exchange_init_all(ctxs.data(), ranks));
    std::vector<uint32_t> views(ranks, 0u);
    std::vector<GvExchangeFrame> sent(ranks), got(ranks);
    GvExchangeFrame one;
    if (ranks > 1)  // a per-rank call on a communicator one thread drives would wait for ranks the thread has not reached
        EXPECT(gv_exchange_visible(ctxs[0], 0, 0, 0, &one), GV_E_STATE);
    int cut_frames = 0;
    for (int frame = 0; frame < 12; frame++) {
        const uint32_t mode = (uint32_t)(frame % 3);
        for (int r = 0; r < ranks; r++)
            xs[r].produce(frame, mode);
        CHECK(gv_exchange_visible_all(ctxs.data(), ranks, views.data(), nullptr, 0, sent.data()));
        for (int r = 0; r < ranks; r++)
            xs[r].check_sent(sent[r], frame, mode);
        if (frame % 3 != 2) {  // (every third frame is left to the next gv_exchange_visible_all to complete)
            CHECK(gv_exchange_acquire_all(ctxs.data(), ranks, (uint64_t)frame, got.data()));
            for (int r = 0; r < ranks; r++)
                xs[r].check_acquired(sent[r], got[r], frame);
        }
    }
    // several mesh systems per frame: pool 0's list is exchanged AFTER pool 1 was culled (gv_pool_exchange_visible_all names the
    // pool; the view-indexed form then carries pool 1's list, the pool of the most recent cull)
    for (int frame = 12; frame < 14; frame++) {
        for (int r = 0; r < ranks; r++) {
            xs[r].produce(frame, GV_EXCHANGE_P2P);
            xs[r].cull_other_pool();
        }
        if (frame == 12) {
            CHECK(gv_pool_exchange_visible_all(ctxs.data(), ranks, 0, views.data(), nullptr, 0, sent.data()));
            CHECK(gv_exchange_acquire_all(ctxs.data(), ranks, (uint64_t)frame, got.data()));
            for (int r = 0; r < ranks; r++)
                xs[r].check_acquired(sent[r], got[r], frame);
        } else {
            CHECK(gv_exchange_visible_all(ctxs.data(), ranks, views.data(), nullptr, 0, sent.data()));
            CHECK(gv_exchange_acquire_all(ctxs.data(), ranks, (uint64_t)frame, got.data()));
            for (int r = 0; r < ranks; r++) {
                const uint32_t* rows = (const uint32_t*)got[r].gathered_device;
                for (int q = 0; q < ranks; q++) {
                    const uint32_t* row = rows + (size_t)q * got[r].row_words;
                    if (!got[r].complete || row[0] != ExchangeRank::other_count || row[1] != 4200u + (uint32_t)q || row[ExchangeRank::other_count] != 4200u + (uint32_t)q + ExchangeRank::other_count - 1)
                        xs[r].fail("the view-indexed exchange after a cull of pool 1 does not carry pool 1's list", frame, q);
                }
            }
        }
    }
    // ONE exchange for ALL the lists of a frame (gv_exchange_views_all): pool 0's list and pool 1's (index_base 5) behind a count
    // table, in one row per rank; the lists alternate between short and long, so that batched frames are completed by a second
    // exchange too
    {
        const GvExchangeItem items[2] = {{0u, 0u, 0u}, {1u, 0u, 5u}};
        GvStats before{};
        CHECK(gv_stats(ctxs[0], &before));
        int batched = 0, batched_cut = 0;
        for (int frame = 14; frame < 22; frame++) {
            const int list = frame % 2 ? 9 : 3;  // (list_count's scripted sequence: 3 = creeping, 9 = after the jump)
            for (int r = 0; r < ranks; r++) {
                xs[r].produce(list, (uint32_t)(frame % 3));
                xs[r].cull_other_pool();
            }
            CHECK(gv_exchange_views_all(ctxs.data(), ranks, items, 2, 0, sent.data()));
            if (frame % 4 == 3)
                continue;  // (left to the next exchange to complete)
            CHECK(gv_exchange_acquire_all(ctxs.data(), ranks, (uint64_t)frame, got.data()));
            batched++;
            for (int r = 0; r < ranks; r++) {
                const GvExchangeFrame& f = got[r];
                if (!f.complete || !f.gathered_device || f.items != 2 || !f.item_counts || f.frame != (uint64_t)frame || sent[r].items != 2 || sent[r].item_counts)
                    xs[r].fail("fields of a batched frame", frame, -1);
                const uint32_t* rows = (const uint32_t*)f.gathered_device;
                for (int q = 0; q < ranks && f.complete; q++) {
                    const uint32_t* row = rows + (size_t)q * f.row_words;
                    const uint32_t c0 = list_count(q, list, ExchangeRank::n), c1 = ExchangeRank::other_count;
                    if (row[0] != 2 + c0 + c1 || f.counts[q] != row[0] || row[1] != c0 || row[2] != c1 || f.item_counts[q * 2] != c0 || f.item_counts[q * 2 + 1] != c1)
                        xs[r].fail("header / count table of a batched row", frame, q);
                    for (uint32_t k = 0; k < c0; k++)
                        if (row[3 + k] != list_value(q, list, k)) {
                            xs[r].fail("entry of the first list of a batched row", frame, q);
                            break;
                        }
                    for (uint32_t k = 0; k < c1; k++)
                        if (row[3 + c0 + k] != 4200u + (uint32_t)q + k + 5u) {
                            xs[r].fail("entry of the second list of a batched row (index_base 5)", frame, q);
                            break;
                        }
                }
                if (r == 0 && f.cut_ranks)
                    batched_cut++;
            }
        }
        GvStats after{};
        CHECK(gv_stats(ctxs[0], &after));
        if (after.exchanges - before.exchanges != 8)
            xs[0].fail("GvStats::exchanges does not count one per gv_exchange_views_all", -1, -1);
        if (ranks > 1 && g_list_seed == 0 && batched_cut == 0)  // (the scripted sequence: list 9 is nine times list 3)
            xs[0].fail("no batched frame needed a second exchange: the tail path of the batched form was not exercised", -1, -1);
        EXPECT(gv_exchange_views_all(ctxs.data(), ranks, items, 0, 0, sent.data()), GV_E_ARG);
        EXPECT(gv_exchange_views_all(ctxs.data(), ranks, items, GV_EXCHANGE_MAX_ITEMS + 1, 0, sent.data()), GV_E_ARG);
        EXPECT(gv_exchange_views_all(ctxs.data(), ranks, nullptr, 2, 0, sent.data()), GV_E_ARG);
        const GvExchangeItem bad[1] = {{GV_MAX_POOLS, 0u, 0u}};
        EXPECT(gv_exchange_views_all(ctxs.data(), ranks, bad, 1, 0, sent.data()), GV_E_ARG);
        std::printf("batched exchange (gv_exchange_views_all), %d ranks: %d frames acquired, %d of them completed by a second exchange\n", ranks, batched, batched_cut);
    }
    EXPECT(gv_pool_exchange_visible_all(ctxs.data(), ranks, GV_MAX_POOLS, views.data(), nullptr, 0, sent.data()), GV_E_ARG);
    EXPECT(gv_pool_exchange_visible_all(ctxs.data(), ranks, 0, views.data(), nullptr, 2, sent.data()), GV_E_ARG);
    EXPECT(gv_pool_exchange_visible_all(ctxs.data(), ranks, 5, views.data(), nullptr, 0, sent.data()), GV_E_ARG);  // a pool never bound / culled
    if (ranks == 1) {
        xs[0].produce(14, GV_EXCHANGE_P2P);
        xs[0].cull_other_pool();
        CHECK(gv_pool_exchange_visible(ctxs[0], 0, 0, 0, 0, &sent[0]));
        CHECK(gv_exchange_acquire(ctxs[0], 22, &got[0]));
        got[0].frame = sent[0].frame = 14;  // (check_acquired derives the expected list from the frame number: list 14 travelled as frame 22)
        xs[0].check_acquired(sent[0], got[0], 14);
        const GvExchangeItem alone[1] = {{0u, 0u, 0u}};
        CHECK(gv_exchange_views(ctxs[0], alone, 1, 0, &sent[0]));  // the per-rank form of the batched exchange
        CHECK(gv_exchange_acquire(ctxs[0], 23, &got[0]));
        if (!got[0].complete || got[0].items != 1 || ((const uint32_t*)got[0].gathered_device)[1] != list_count(0, 14, ExchangeRank::n))
            xs[0].fail("gv_exchange_views of one rank", 23, 0);
        EXPECT(gv_pool_exchange_visible(ctxs[0], GV_MAX_POOLS, 0, 0, 0, &sent[0]), GV_E_ARG);
    }
    int failures = 0;
    for (int r = 0; r < ranks; r++) {
        failures += xs[r].failures;
        cut_frames += xs[r].cut_frames;
        ctx = xs[r].ctx;
        CHECK(gv_exchange_shutdown(ctx));
        gv_destroy(ctx);
    }
    if (failures) {
        std::fprintf(stderr, "exchange driven by one thread, %d ranks: %d failures\n", ranks, failures);
        std::exit(1);
    }
    std::printf("exchange driven by ONE thread, %d ranks, list sequence %d: ok — every acquired frame complete (%d frames per rank needed a second exchange)\n",
                ranks, list_seed, cut_frames / ranks);
}

// ONE thread, N contexts, NO communicator (gv_exchange_init_peers, GV_EXCHANGE_PEER): every rank's list is stored straight into its row
// of every member's rows. No transport library is loaded; rows are as wide as the pools, so no frame is ever short whatever the
// lists do — every acquired frame complete with cut_ranks == 0, travelled_words == 1 + count.
static void exchange_by_peers(int ranks, int list_seed)
{
    g_list_seed = list_seed;
    std::vector<ExchangeRank> xs(ranks);
    std::vector<GvCtx*> ctxs;
    for (int r = 0; r < ranks; r++) {
        if (!xs[r].create(r, ranks))
            std::exit(1);
        ctxs.push_back(xs[r].ctx);
    }
    GvCtx* ctx = ctxs[0];
    std::vector<uint32_t> views(ranks, 0u);
    std::vector<GvExchangeFrame> sent(ranks), got(ranks);
    GvExchangeFrame one;
    EXPECT(gv_exchange_visible_all(ctxs.data(), ranks, views.data(), nullptr, 0, sent.data()), GV_E_STATE);  // nothing initialised yet
    EXPECT(gv_exchange_set_mode(ctxs[0], GV_EXCHANGE_PEER), GV_E_ARG);                                        // not without a peer group
    CHECK(gv_exchange_init_peers(ctxs.data(), ranks));
    EXPECT(gv_exchange_set_mode(ctxs[0], GV_EXCHANGE_ALLGATHER), GV_E_ARG);  // a peer group has no communicator to travel by
    CHECK(gv_exchange_set_mode(ctxs[0], GV_EXCHANGE_PEER));
    if (ranks > 1) {
        EXPECT(gv_exchange_visible(ctxs[0], 0, 0, 0, &one), GV_E_STATE);  // one thread drives the group through the *_all calls
        std::vector<GvCtx*> swapped = ctxs;
        std::swap(swapped[0], swapped[1]);
        for (int r = 0; r < ranks; r++)
            xs[r].produce(0, GV_EXCHANGE_PEER);
        EXPECT(gv_exchange_visible_all(swapped.data(), ranks, views.data(), nullptr, 0, sent.data()), GV_E_ARG);  // not in rank order
    }
    auto check_rows = [&](int frame, int list) {
        for (int r = 0; r < ranks; r++) {
            const GvExchangeFrame& f = got[r];
            if (!f.complete || !f.gathered_device || !f.ready_event || f.frame != (uint64_t)frame || f.mode != GV_EXCHANGE_PEER || f.cut_ranks || f.row_words % 4u ||
                f.world_size != (uint32_t)ranks)
                xs[r].fail("fields of an acquired peer frame", frame, -1);
            const uint32_t* rows = (const uint32_t*)f.gathered_device;
            for (int q = 0; q < ranks; q++) {
                const uint32_t* row = rows + (size_t)q * f.row_words;
                const uint32_t count = list_count(q, list, ExchangeRank::n);
                if (row[0] != count || f.counts[q] != count || f.tail_words[q] || f.travelled_words[q] != 1u + count || f.room[q] + 1u > f.row_words || count > f.room[q])
                    xs[r].fail("header / count / statistics of a peer row", frame, q);
                for (uint32_t k = 0; k < count; k++)
                    if (row[1 + k] != list_value(q, list, k)) {
                        xs[r].fail("entry of a peer row", frame, q);
                        break;
                    }
            }
        }
    };
    int frame = 0;
    for (; frame < 18; frame++) {  // list_count's scripted sequence creeps, jumps ninefold at 7 and collapses at 14
        for (int r = 0; r < ranks; r++)
            xs[r].produce(frame, GV_EXCHANGE_PEER);
        CHECK(gv_exchange_visible_all(ctxs.data(), ranks, views.data(), nullptr, 0, sent.data()));
        for (int r = 0; r < ranks; r++)
            if (sent[r].complete || sent[r].gathered_device || sent[r].mode != GV_EXCHANGE_PEER || sent[r].frame != (uint64_t)frame)
                xs[r].fail("fields of a peer frame that was sent, not acquired", frame, -1);
        if (frame % 3 == 2)
            continue;  // (settled by the next exchange; acquired a frame late below)
        if (frame % 3 == 0 && frame > 0) {
            CHECK(gv_exchange_acquire_all(ctxs.data(), ranks, (uint64_t)frame - 1, got.data()));
            check_rows(frame - 1, frame - 1);
        }
        CHECK(gv_exchange_acquire_all(ctxs.data(), ranks, (uint64_t)frame, got.data()));
        check_rows(frame, frame);
    }
    // ONE exchange for two lists per rank (gv_exchange_views_all): count table behind the header, lists back to back
    const GvExchangeItem items[2] = {{0u, 0u, 0u}, {1u, 0u, 5u}};
    for (; frame < 24; frame++) {
        const int list = frame % 2 ? 9 : 3;
        for (int r = 0; r < ranks; r++) {
            xs[r].produce(list, GV_EXCHANGE_PEER);
            xs[r].cull_other_pool();
        }
        CHECK(gv_exchange_views_all(ctxs.data(), ranks, items, 2, 0, sent.data()));
        CHECK(gv_exchange_acquire_all(ctxs.data(), ranks, (uint64_t)frame, got.data()));
        for (int r = 0; r < ranks; r++) {
            const GvExchangeFrame& f = got[r];
            if (!f.complete || f.items != 2 || !f.item_counts || f.cut_ranks)
                xs[r].fail("fields of a batched peer frame", frame, -1);
            const uint32_t* rows = (const uint32_t*)f.gathered_device;
            for (int q = 0; q < ranks && f.complete; q++) {
                const uint32_t* row = rows + (size_t)q * f.row_words;
                const uint32_t c0 = list_count(q, list, ExchangeRank::n), c1 = ExchangeRank::other_count;
                bool ok = row[0] == 2 + c0 + c1 && f.counts[q] == row[0] && row[1] == c0 && row[2] == c1 && f.item_counts[q * 2] == c0 && f.item_counts[q * 2 + 1] == c1 &&
                          f.travelled_words[q] == 1u + row[0];
                for (uint32_t k = 0; k < c0 && ok; k++)
                    ok = row[3 + k] == list_value(q, list, k);
                for (uint32_t k = 0; k < c1 && ok; k++)
                    ok = row[3 + c0 + k] == 4200u + (uint32_t)q + k + 5u;
                if (!ok)
                    xs[r].fail("a batched peer row", frame, q);
            }
        }
    }
    // one member leaves: the group is dissolved — the others answer GV_E_STATE until they are initialised again
    ctx = ctxs[ranks - 1];
    CHECK(gv_exchange_shutdown(ctx));
    ctx = ctxs[0];
    EXPECT(gv_exchange_visible_all(ctxs.data(), ranks, views.data(), nullptr, 0, sent.data()), GV_E_STATE);
    CHECK(gv_exchange_init_peers(ctxs.data(), ranks));  // ... and a new group starts from frame 0
    for (int r = 0; r < ranks; r++)
        xs[r].produce(5, GV_EXCHANGE_PEER);
    CHECK(gv_exchange_visible_all(ctxs.data(), ranks, views.data(), nullptr, 0, sent.data()));
    CHECK(gv_exchange_acquire_all(ctxs.data(), ranks, 0, got.data()));
    check_rows(0, 5);
    if (std::getenv("GV_RCCL_LIBRARY")) {  // the same contexts under a communicator afterwards: the default pattern, not the group's
        CHECK(gv_exchange_init_all(ctxs.data(), ranks));
        for (int r = 0; r < ranks; r++)
            xs[r].produce(6, GV_EXCHANGE_P2P);
        CHECK(gv_exchange_visible_all(ctxs.data(), ranks, views.data(), nullptr, 0, sent.data()));
        CHECK(gv_exchange_acquire_all(ctxs.data(), ranks, 0, got.data()));
        for (int r = 0; r < ranks; r++) {
            got[r].frame = sent[r].frame = 6;  // (check_acquired derives the expected list from the frame number: list 6 travelled as frame 0)
            xs[r].check_acquired(sent[r], got[r], 6);
        }
    }
    int failures = 0;
    for (int r = 0; r < ranks; r++) {
        failures += xs[r].failures;
        gv_destroy(xs[r].ctx);  // (no shutdown first: destroying a member drains and dissolves the group itself)
    }
    if (failures) {
        std::fprintf(stderr, "exchange by peer stores, %d ranks: %d failures\n", ranks, failures);
        std::exit(1);
    }
    std::printf("exchange by peer stores (no communicator), ONE thread, %d ranks, list sequence %d: ok — 24 frames, none short\n", ranks, list_seed);
}

// The FIRST frame of a communicator carries several lists (gv_exchange_views_all) under a direct travel pattern: there is no history,
// every room is 0 — the count tables must still arrive with the headers (round 6: they travelled with the tails, and the frame
// reported the counts of whatever the rows had held before).
static void batched_first_frame(int ranks, uint32_t mode)
{
    g_list_seed = 0;
    if (!std::getenv("GV_RCCL_LIBRARY"))
        return;
    std::vector<ExchangeRank> xs(ranks);
    std::vector<GvCtx*> ctxs;
    for (int r = 0; r < ranks; r++) {
        if (!xs[r].create(r, ranks))
            std::exit(1);
        ctxs.push_back(xs[r].ctx);
    }
    GvCtx* ctx = ctxs[0];
    CHECK(gv_exchange_init_all(ctxs.data(), ranks));
    const GvExchangeItem items[2] = {{0u, 0u, 0u}, {1u, 0u, 0u}};
    std::vector<GvExchangeFrame> sent(ranks), got(ranks);
    for (int frame = 0; frame < 2; frame++) {
        for (int r = 0; r < ranks; r++) {
            xs[r].produce(frame, mode);
            xs[r].cull_other_pool();
        }
        CHECK(gv_exchange_views_all(ctxs.data(), ranks, items, 2, 0, sent.data()));
        CHECK(gv_exchange_acquire_all(ctxs.data(), ranks, (uint64_t)frame, got.data()));
        for (int r = 0; r < ranks; r++)
            for (int q = 0; q < ranks; q++) {
                const uint32_t* row = (const uint32_t*)got[r].gathered_device + (size_t)q * got[r].row_words;
                const uint32_t c0 = list_count(q, frame, ExchangeRank::n), c1 = ExchangeRank::other_count;
                if (!got[r].complete || got[r].items != 2 || row[0] != 2 + c0 + c1 || row[1] != c0 || row[2] != c1 || got[r].item_counts[q * 2] != c0 ||
                    got[r].item_counts[q * 2 + 1] != c1 || row[3] != list_value(q, frame, 0) || row[3 + c0] != 4200u + (uint32_t)q)
                    xs[r].fail("the first batched frame of a communicator under a direct travel pattern", frame, q);
            }
    }
    int failures = 0;
    for (int r = 0; r < ranks; r++) {
        failures += xs[r].failures;
        ctx = xs[r].ctx;
        CHECK(gv_exchange_shutdown(ctx));
        gv_destroy(ctx);
    }
    if (failures)
        std::exit(1);
    std::printf("first batched frame of a communicator, %d ranks, travel pattern %u: ok\n", ranks, mode);
}

// ---- random schedules over the held-back mechanisms: the text tests/schedules.py generates (the GPU tier replays the same
// schedules against the oracle, tests/test_gpu_fuzz.py). Kernels do nothing here; what is checked is that every call returns
// GV_OK and that the host orchestration behind it — recorded culls, deferred sorts, published views, flushes forced by dirty
// marks / re-binds / pyramid builds — stays inside its buffers (ASan) whatever the order of calls. ----
static void replay_schedule(const std::string& path)
{
    auto aligned = [](std::vector<uint8_t>& v) { return reinterpret_cast<void*>(((uintptr_t)v.data() + 15) & ~(uintptr_t)15); };
    std::ifstream in(path);
    std::string line;
    if (!std::getline(in, line)) {
        std::fprintf(stderr, "schedule %s: empty\n", path.c_str());
        std::exit(1);
    }
    std::istringstream head(line);
    std::string word;
    uint32_t n_xf = 0;
    head >> word >> n_xf;
    std::vector<uint32_t> sizes;
    bool exchange = false;
    for (std::string t; head >> t;) {
        if (t == "x")
            exchange = true;
        else
            sizes.push_back((uint32_t)std::stoul(t));
    }
    exchange = exchange && std::getenv("GV_RCCL_LIBRARY") != nullptr;  // (the shared-memory transport of the tests)
    World w;
    w.build(n_xf, 0);
    std::vector<std::vector<Mesh>> pools;
    for (uint32_t n : sizes)
        pools.emplace_back(w.meshes.begin(), w.meshes.begin() + n);
    std::vector<Transform> xf = w.xf;
    GvConfig config{};
    config.struct_size = sizeof(config);
    GvCtx* ctx = nullptr;
    if (gv_create(&config, &ctx) != GV_OK) {
        std::fprintf(stderr, "schedule %s: gv_create\n", path.c_str());
        std::exit(1);
    }
    const GvTransformLayout tl = transform_layout();
    const GvMeshLayout ml = mesh_layout();
    const GvRecordLayout rl = {64, 0, 8, 56, GV_NONE, (uint32_t)sizeof(Mesh), 0};
    CHECK(gv_transform_bind(ctx, xf.data(), sizeof(Transform), n_xf, &tl, w.e2t.data(), (uint32_t)w.e2t.size()));
    for (uint32_t p = 0; p < pools.size(); p++) {
        CHECK(gv_pool_bind(ctx, p, pools[p].data(), sizeof(Mesh), (uint32_t)pools[p].size(), &ml));
        if (p % 2 == 1)
            CHECK(gv_pool_set_record_layout(ctx, p, &rl));
    }
    std::vector<std::vector<uint8_t>> ready(pools.size());
    for (uint32_t p = 0; p < pools.size(); p++)
        if (p % 4 == 2) {
            ready[p].assign(pools[p].size(), 1);
            for (size_t i = 0; i < ready[p].size(); i += 7)
                ready[p][i] = (uint8_t)(i % 4);
            CHECK(gv_pool_bind_ready(ctx, p, ready[p].data(), 1, 1));
        }
    CHECK(gv_hierarchy_rebuild(ctx));
    if (exchange) {
        unsigned char id[GV_EXCHANGE_ID_BYTES];
        if (gv_exchange_unique_id(id) != GV_OK) {
            std::fprintf(stderr, "schedule %s: gv_exchange_unique_id\n", path.c_str());
            std::exit(1);
        }
        CHECK(gv_exchange_init(ctx, id, 0, 1));
    }
    std::vector<std::vector<uint8_t>> targets(pools.size() * GV_MAX_VIEWS);  // the caller's record arrays, [pool * GV_MAX_VIEWS + view]
    uint32_t last_pool = 0;
    std::vector<bool> count_only(pools.size(), false);  // the pool's last cull was a count-only view (no records to ask for)
    std::vector<float> depth;
    std::vector<uint32_t> scratch;
    while (std::getline(in, line)) {
        std::istringstream ops(line);
        std::string op;
        ops >> op;
        std::vector<std::string> a;
        for (std::string t; ops >> t;)
            a.push_back(t);
        auto num = [&](size_t k) { return (uint32_t)std::stoul(a.at(k)); };
        if (op == "begin") {
            CHECK(gv_cull_batch_begin(ctx));
        } else if (op == "end") {
            CHECK(gv_cull_batch_end(ctx));
        } else if (op == "wait") {
            CHECK(gv_wait(ctx));
        } else if (op == "sync") {
            CHECK(gv_sync(ctx));
        } else if (op == "rebuild") {
            CHECK(gv_hierarchy_rebuild(ctx));
        } else if (op == "reparent") {
            for (uint32_t s = num(0); s < num(0) + num(1); s++)
                xf[s].parent = xf[s / 2].entity;  // a lower slot: no cycles
            CHECK(gv_mark_dirty(ctx, GV_DIRTY_HIERARCHY, num(0), num(1)));
        } else if (op == "cull") {
            GvView views[GV_MAX_VIEWS];
            uint32_t nv = 0;
            for (size_t k = 1; k < a.size(); k++) {
                const char kind = a[k][0];
                views[nv] = make_view(kind == 's' ? (int8_t)(k - 1) : (int8_t)-1, kind == 'h', kind != 'c');
                views[nv].distance_2d = kind == 'u';
                nv++;
            }
            CHECK(gv_cull(ctx, num(0), views, nv));
            last_pool = num(0);
            count_only[num(0)] = a[1][0] == 'c';
        } else if (op == "sort") {
            CHECK(gv_pool_sort(ctx, num(0), num(1), (int)num(2)));
        } else if (op == "dirty_xf") {
            for (uint32_t s = num(0); s < num(0) + num(1); s++)
                xf[s].position[0] += 1.0f;
            CHECK(gv_mark_dirty(ctx, GV_DIRTY_TRANSFORM, num(0), num(1)));
        } else if (op == "dirty_mesh") {
            for (uint32_t s = num(1); s < num(1) + num(2); s++)
                pools[num(0)][s].aabbMax[0] += 0.25f;
            CHECK(gv_mark_dirty(ctx, GV_DIRTY_MESH, (num(0) << 28) | num(1), num(2)));
        } else if (op == "move") {
            std::vector<Mesh> moved(pools[num(0)]);
            pools[num(0)].swap(moved);  // (the old storage is freed at the end of this scope: the library must not touch it again)
            CHECK(gv_pool_bind(ctx, num(0), pools[num(0)].data(), sizeof(Mesh), (uint32_t)pools[num(0)].size(), &ml));
        } else if (op == "grow") {
            std::vector<Mesh> grown(pools[num(0)]);
            for (uint32_t k = 0; k < num(1); k++)
                grown.push_back(w.meshes[grown.size()]);
            pools[num(0)].swap(grown);
            for (uint32_t v = 0; v < GV_MAX_VIEWS; v++)  // record targets are too small for the grown pool: let them go first
                if (!targets[num(0) * GV_MAX_VIEWS + v].empty()) {
                    CHECK(gv_pool_set_record_target(ctx, num(0), v, nullptr, 0));
                    targets[num(0) * GV_MAX_VIEWS + v].clear();
                }
            CHECK(gv_pool_bind(ctx, num(0), pools[num(0)].data(), sizeof(Mesh), (uint32_t)pools[num(0)].size(), &ml));
            if (!ready[num(0)].empty()) {
                ready[num(0)].resize(pools[num(0)].size(), 1);
                CHECK(gv_pool_bind_ready(ctx, num(0), ready[num(0)].data(), 1, 1));
            }
        } else if (op == "move_xf") {
            std::vector<Transform> moved(xf);
            xf.swap(moved);
            CHECK(gv_transform_bind(ctx, xf.data(), sizeof(Transform), n_xf, &tl, w.e2t.data(), (uint32_t)w.e2t.size()));
        } else if (op == "hiz") {
            depth.assign((size_t)num(0) * num(1), 0.25f);
            CHECK(gv_hiz_build(ctx, depth.data(), num(0), num(1), GV_MEM_HOST));
        } else if (op == "hiz_rebuild") {
            CHECK(gv_hiz_rebuild(ctx));
        } else if (op == "sweep") {
            CHECK(gv_sweep(ctx, num(0)));
            float world[12 * 8];
            if (num(0) != GV_SWEEP_WITH_CULL && num(0) != GV_SWEEP_WITH_CULL_VALU)  // (those are deferred to the next cull)
                CHECK(gv_get_world(ctx, 0, 8, world));
        } else if (op == "fetch" || op == "records") {
            GvResult r{};
            CHECK(gv_pool_results_fetch(ctx, num(0), num(1), op == "fetch" ? (int)num(2) : 0, &r));
            if (num(0) % 2 == 1 && !count_only[num(0)]) {
                const void* records = nullptr;
                uint32_t count = 0;
                CHECK(gv_pool_results_records(ctx, num(0), num(1), &records, &count));
            }
        } else if (op == "count") {
            uint32_t count = 0;
            CHECK(gv_pool_result_count(ctx, num(0), num(1), &count));
        } else if (op == "device") {
            GvDeviceResult d{};
            CHECK(gv_pool_results_device(ctx, num(0), num(1), &d));
        } else if (op == "bases") {
            const uint32_t* bases = nullptr;
            uint32_t count = 0;
            CHECK(gv_pool_results_instance_bases(ctx, num(0), num(1), &bases, &count));
        } else if (op == "ready") {
            for (uint32_t k = num(1); k < num(1) + num(2); k++)
                ready[num(0)][k] = (uint8_t)((k * 7u) % 4u);
            CHECK(gv_mark_dirty(ctx, GV_DIRTY_MESH, (num(0) << 28) | num(1), num(2)));
        } else if (op == "target") {
            std::vector<uint8_t>& t = targets[num(0) * GV_MAX_VIEWS + num(1)];
            if (num(2)) {
                std::vector<uint8_t> fresh(pools[num(0)].size() * 64 + 16);
                CHECK(gv_pool_set_record_target(ctx, num(0), num(1), aligned(fresh), pools[num(0)].size() * 64));
                t.swap(fresh);  // (the previous array is let go only now: the call above replaced it as the target)
            } else {
                CHECK(gv_pool_set_record_target(ctx, num(0), num(1), nullptr, 0));
                t.clear();
            }
        } else if (op == "exch" || op == "exchp") {
            if (exchange) {
                GvExchangeFrame xf;
                if (op == "exchp")  // a named pool's view 0, whatever was culled since
                    CHECK(gv_pool_exchange_visible(ctx, num(0), 0, 11, 0, &xf));
                else
                    CHECK(gv_exchange_visible(ctx, 0, 11, 0, &xf));
                GvExchangeFrame got;
                CHECK(gv_exchange_acquire(ctx, xf.frame, &got));
                if (!got.complete || !got.gathered_device) {
                    std::fprintf(stderr, "schedule %s: an acquired frame is not complete\n", path.c_str());
                    std::exit(1);
                }
            }
        } else if (op == "shard" || op == "mask") {
            const size_t n = pools[last_pool].size();
            scratch.assign(n + 2, 0);
            if (op == "shard")
                CHECK(gv_results_copy_shard_device(ctx, 0, scratch.data(), (uint32_t)n, 7));
            else
                CHECK(gv_results_copy_mask_device(ctx, 0, scratch.data(), (uint32_t)((n + 31) / 32)));
        } else {
            std::fprintf(stderr, "schedule %s: unknown operation '%s'\n", path.c_str(), op.c_str());
            std::exit(1);
        }
    }
    gv_destroy(ctx);
}

// ---- allocation failures: every device / pinned allocation of a small frame sequence fails once, in turn ----
// The k-th allocation of the sequence returns hipErrorOutOfMemory (tests/cpp/hip_stub fault injection). Whatever call it lands in
// must come back with an error CODE (GV_E_OOM / GV_E_HIP) — no crash, no leak (ASan), no use of a half-made buffer — and the
// context must still be usable: with the injection off, the same sequence runs through on the same context, and gv_destroy is clean.
#include <hip/hip_runtime.h>
static int frame_sequence(GvCtx* ctx, World& w, const std::vector<float>& depth, bool tolerate)
{
    const GvTransformLayout tl = transform_layout();
    const GvMeshLayout ml = mesh_layout();
    const GvRecordLayout rl = {64, 0, 8, 56, GV_NONE, (uint32_t)sizeof(Mesh), 0};
    auto step = [&](int rc) { return tolerate ? rc : (rc == GV_OK ? GV_OK : (std::fprintf(stderr, "frame_sequence: %d (%s)\n", rc, gv_last_error(ctx)), std::exit(1), rc)); };
    int rc;
    if ((rc = step(gv_transform_bind(ctx, w.xf.data(), sizeof(Transform), (uint32_t)w.xf.size(), &tl, w.e2t.data(), (uint32_t)w.e2t.size()))))
        return rc;
    if ((rc = step(gv_pool_bind(ctx, 0, w.meshes.data(), sizeof(Mesh), (uint32_t)w.meshes.size(), &ml))))
        return rc;
    if ((rc = step(gv_pool_bind(ctx, 1, w.big.data(), sizeof(BigMesh), (uint32_t)w.big.size(), &ml))))
        return rc;
    if ((rc = step(gv_pool_set_record_layout(ctx, 1, &rl))))
        return rc;
    if ((rc = step(gv_hierarchy_rebuild(ctx))))
        return rc;
    if ((rc = step(gv_hiz_build(ctx, depth.data(), 96, 80, GV_MEM_HOST))))
        return rc;
    GvView views[3] = {make_view(-1, 1, 1), make_view(0, 0, 1), make_view(1, 0, 1)};
    for (int frame = 0; frame < 2; frame++) {
        if ((rc = step(gv_mark_dirty(ctx, GV_DIRTY_TRANSFORM, 10, 300))))
            return rc;
        if ((rc = step(gv_sweep(ctx, GV_SWEEP_INCREMENTAL))))
            return rc;
        if ((rc = step(gv_cull_batch_begin(ctx))))
            return rc;
        if ((rc = step(gv_cull(ctx, 0, views, 3))))
            return rc;
        if ((rc = step(gv_cull(ctx, 1, views, 1))))
            return rc;
        if ((rc = step(gv_pool_sort(ctx, 0, 0, 0))))
            return rc;
        if ((rc = step(gv_pool_sort(ctx, 1, 0, 1))))
            return rc;
        GvResult r{};
        for (uint32_t v = 0; v < 3; v++)
            if ((rc = step(gv_pool_results_fetch(ctx, 0, v, 1, &r))))
                return rc;
        if ((rc = step(gv_pool_results_fetch(ctx, 1, 0, 0, &r))))
            return rc;
        const uint32_t* bases = nullptr;
        uint32_t count = 0;
        if ((rc = step(gv_pool_results_instance_bases(ctx, 0, 0, &bases, &count))))
            return rc;
        float world[12 * 4];
        if ((rc = step(gv_get_world(ctx, 0, 4, world))))
            return rc;
    }
    return GV_OK;
}

// the mirror grows by half (entities created) and is put back into spatial order on the device: the path ADVICE r3 asked about
// (a failure after the transform side has been re-ordered must leave the mesh pools marked for rebuild, not pointing at old entries)
static int growth_sequence(GvCtx* ctx, World& small, World& grown, bool tolerate)
{
    const GvTransformLayout tl = transform_layout();
    const GvMeshLayout ml = mesh_layout();
    auto step = [&](int rc) { return tolerate ? rc : (rc == GV_OK ? GV_OK : (std::fprintf(stderr, "growth_sequence: %d (%s)\n", rc, gv_last_error(ctx)), std::exit(1), rc)); };
    GvView v = make_view(-1, 0, 1);
    GvResult r{};
    int rc;
    if ((rc = step(gv_transform_bind(ctx, small.xf.data(), sizeof(Transform), (uint32_t)small.xf.size(), &tl, small.e2t.data(), (uint32_t)small.e2t.size()))))
        return rc;
    if ((rc = step(gv_pool_bind(ctx, 0, small.meshes.data(), sizeof(Mesh), (uint32_t)small.meshes.size(), &ml))))
        return rc;
    if ((rc = step(gv_hierarchy_rebuild(ctx))))
        return rc;
    if ((rc = step(gv_cull(ctx, 0, &v, 1))))
        return rc;
    if ((rc = step(gv_results_fetch(ctx, 0, 1, &r))))
        return rc;
    // the same pools, half as many slots again (the first slots are the same entities): appended, then re-ordered
    if ((rc = step(gv_transform_bind(ctx, grown.xf.data(), sizeof(Transform), (uint32_t)grown.xf.size(), &tl, grown.e2t.data(), (uint32_t)grown.e2t.size()))))
        return rc;
    if ((rc = step(gv_pool_bind(ctx, 0, grown.meshes.data(), sizeof(Mesh), (uint32_t)grown.meshes.size(), &ml))))
        return rc;
    for (int frame = 0; frame < 2; frame++) {
        if ((rc = step(gv_cull(ctx, 0, &v, 1))))
            return rc;
        if ((rc = step(gv_results_fetch(ctx, 0, 1, &r))))
            return rc;
    }
    return GV_OK;
}

// The exchange under allocation failures: a 1-rank communicator (everything on this thread: the countdown is deterministic), lists
// that creep, jump (a short prediction: the rows are widened for the tails) and collapse. Whatever allocation fails, the call says
// GV_E_OOM, nothing leaks (ASan), and the SAME frame can be tried again: every acquired frame holds the whole list.
static long g_exchange_fail = 0, g_exchange_allocations = 0;
static int exchange_sequence(GvCtx* ctx, ExchangeRank& x, bool stop_at_failure, int* completed_frames, bool peers = false)
{
    for (int frame = 0; frame < 10; frame++) {
        x.produce(frame < 5 ? frame : frame + 4, peers ? (uint32_t)GV_EXCHANGE_PEER : (uint32_t)(frame % 3));  // (frames 7.. of the scripted sequence jump)
        const int list_frame = frame < 5 ? frame : frame + 4;
        GvExchangeFrame sent, got;
        bool was_sent = false;
        for (int attempt = 0;; attempt++) {
            // (only the exchange's own allocations are made to fail: the k-th of them over the whole sequence)
            auto guarded = [&](auto&& call) {
                gv_stub_fail_countdown() = g_exchange_fail > g_exchange_allocations ? g_exchange_fail - g_exchange_allocations : 0;
                const long before = gv_stub_allocations();
                const int result = call();
                g_exchange_allocations += gv_stub_allocations() - before;
                gv_stub_fail_countdown() = 0;
                return result;
            };
            int rc = was_sent ? GV_OK : guarded([&] { return gv_exchange_visible(ctx, 0, 0, 0, &sent); });
            was_sent = was_sent || rc == GV_OK;
            if (rc == GV_OK)
                rc = guarded([&] { return gv_exchange_acquire(ctx, sent.frame, &got); });
            if (rc == GV_OK)
                break;
            if (rc != GV_E_OOM && rc != GV_E_HIP) {
                std::fprintf(stderr, "exchange under allocation failures: frame %d: %d (%s)\n", frame, rc, gv_last_error(ctx));
                std::exit(1);
            }
            if (stop_at_failure && attempt == 0)
                ++*completed_frames;  // (counts the failed calls)
            if (attempt > 3) {
                std::fprintf(stderr, "exchange under allocation failures: frame %d does not recover\n", frame);
                std::exit(1);
            }
            // (a frame whose send succeeded and whose acquire failed is acquired again; one whose send failed is sent again)
        }
        const uint32_t* row = (const uint32_t*)got.gathered_device;
        const uint32_t count = list_count(0, list_frame, ExchangeRank::n);
        if (!got.complete || row[0] != count) {
            std::fprintf(stderr, "exchange under allocation failures: frame %d header %u, expected %u\n", frame, row ? row[0] : 0u, count);
            std::exit(1);
        }
        for (uint32_t k = 0; k < count; k++)
            if (row[1 + k] != list_value(0, list_frame, k)) {
                std::fprintf(stderr, "exchange under allocation failures: frame %d entry %u\n", frame, k);
                std::exit(1);
            }
    }
    return GV_OK;
}

static void exchange_allocation_failures(bool peers)
{
    if (!peers && !std::getenv("GV_RCCL_LIBRARY"))
        return;
    g_list_seed = 0;
    long total = 0;
    int failed_calls = 0;
    for (long k = 0;; k++) {
        ExchangeRank x;
        if (!x.create(0, 1))
            std::exit(1);
        GvCtx* ctx = x.ctx;
        if (peers) {
            CHECK(gv_exchange_init_peers(&ctx, 1));  // (a group of one: the same rows, staged and scattered by the peer pattern)
        } else {
            unsigned char id[GV_EXCHANGE_ID_BYTES];
            CHECK(gv_exchange_unique_id(id));
            CHECK(gv_exchange_init(ctx, id, 0, 1));
        }
        g_exchange_fail = k;  // 0: nothing fails (that run counts the exchange's allocations)
        g_exchange_allocations = 0;
        exchange_sequence(ctx, x, true, &failed_calls, peers);
        if (k == 0)
            total = g_exchange_allocations;
        CHECK(gv_exchange_shutdown(ctx));
        gv_destroy(ctx);
        if (k >= total)
            break;
    }
    std::printf("allocation failures in the exchange%s: %ld allocations failed in turn, %d calls reported it, every frame was acquired whole all the same: ok\n",
                peers ? " (peer stores)" : "", total, failed_calls);
}

static void allocation_failures()
{
    {
        World small, grown;
        grown.build(9000, 0);
        small.xf.assign(grown.xf.begin(), grown.xf.begin() + 6000);
        small.meshes.assign(grown.meshes.begin(), grown.meshes.begin() + 6000);
        small.e2t = grown.e2t;
        for (uint32_t e = 0; e < small.e2t.size(); e++)
            if (small.e2t[e] != GV_NONE && small.e2t[e] >= 6000)
                small.e2t[e] = GV_NONE;
        GvConfig config{};
        config.struct_size = sizeof(config);
        GvCtx* probe = nullptr;
        if (gv_create(&config, &probe) != GV_OK)
            std::exit(1);
        const long before = gv_stub_allocations();
        growth_sequence(probe, small, grown, false);
        const long total = gv_stub_allocations() - before;
        GvStats stats{};
        if (gv_stats(probe, &stats) != GV_OK || stats.mirror_reorders == 0) {
            std::fprintf(stderr, "growth_sequence: the mirror was not re-ordered on the device (%llu)\n", (unsigned long long)stats.mirror_reorders);
            std::exit(1);
        }
        gv_destroy(probe);
        int failed_calls = 0;
        for (long k = 1; k <= total; k++) {
            GvCtx* ctx = nullptr;
            if (gv_create(&config, &ctx) != GV_OK)
                std::exit(1);
            gv_stub_fail_countdown() = k;
            const int rc = growth_sequence(ctx, small, grown, true);
            gv_stub_fail_countdown() = 0;
            if (rc != GV_OK) {
                failed_calls++;
                if (rc != GV_E_OOM && rc != GV_E_HIP) {
                    std::fprintf(stderr, "growth: allocation %ld of %ld failing: %d (%s)\n", k, total, rc, gv_last_error(ctx));
                    std::exit(1);
                }
                // carry on from where it stopped, as an engine's next frame would: the grown pools, nothing failing
                const GvTransformLayout tl = transform_layout();
                const GvMeshLayout ml = mesh_layout();
                GvView v = make_view(-1, 0, 1);
                GvResult r{};
                CHECK(gv_transform_bind(ctx, grown.xf.data(), sizeof(Transform), (uint32_t)grown.xf.size(), &tl, grown.e2t.data(), (uint32_t)grown.e2t.size()));
                CHECK(gv_pool_bind(ctx, 0, grown.meshes.data(), sizeof(Mesh), (uint32_t)grown.meshes.size(), &ml));
                CHECK(gv_cull(ctx, 0, &v, 1));
                CHECK(gv_results_fetch(ctx, 0, 1, &r));
                check_permutation(ctx, 0, (uint32_t)grown.meshes.size());
            }
            gv_destroy(ctx);
        }
        std::printf("allocation failures while the mirror grows and is re-ordered: %ld allocations failed in turn, %d calls reported it, every context carried on: ok\n",
                    total, failed_calls);
    }
    World w;
    w.build(5000, 2);
    const std::vector<float> depth(96 * 80, 0.25f);
    // how many allocations the sequence makes on a fresh context
    GvConfig config{};
    config.struct_size = sizeof(config);
    GvCtx* probe = nullptr;
    if (gv_create(&config, &probe) != GV_OK)
        std::exit(1);
    const long before = gv_stub_allocations();
    frame_sequence(probe, w, depth, false);
    const long total = gv_stub_allocations() - before;
    gv_destroy(probe);
    int failed_calls = 0;
    for (long k = 1; k <= total; k++) {
        GvCtx* ctx = nullptr;
        if (gv_create(&config, &ctx) != GV_OK)
            std::exit(1);
        gv_stub_fail_countdown() = k;
        const int rc = frame_sequence(ctx, w, depth, true);
        gv_stub_fail_countdown() = 0;
        if (rc != GV_OK) {
            failed_calls++;
            if (rc != GV_E_OOM && rc != GV_E_HIP) {
                std::fprintf(stderr, "allocation %ld of %ld failing: the call returned %d (%s), not GV_E_OOM / GV_E_HIP\n", k, total, rc, gv_last_error(ctx));
                std::exit(1);
            }
            if (!*gv_last_error(ctx)) {
                std::fprintf(stderr, "allocation %ld: error code %d without a message\n", k, rc);
                std::exit(1);
            }
        }
        // the context is still usable: the same sequence, nothing failing (re-bound, rebuilt)
        const int again = frame_sequence(ctx, w, depth, true);
        if (again != GV_OK) {
            std::fprintf(stderr, "allocation %ld of %ld failing once: the context did not recover: %d (%s)\n", k, total, again, gv_last_error(ctx));
            std::exit(1);
        }
        gv_destroy(ctx);
    }
    std::printf("allocation failures: %ld allocations failed in turn, %d calls reported it, every context recovered: ok\n", total, failed_calls);
    {
        // ... and of a frame of mid-sized pools, whose sorts wait for the first read and go out together (flush_sorts / launch_sort_batch):
        // a view whose sort buffers cannot be had keeps its request, the views in front of it are sorted all the same, the call says GV_E_OOM
        World mid;
        mid.build(40000, 0);
        auto sequence = [&](GvCtx* ctx, bool tolerate) -> int {
            const GvTransformLayout tl = transform_layout();
            const GvMeshLayout ml = mesh_layout();
            auto step = [&](int rc) { return tolerate ? rc : (rc == GV_OK ? GV_OK : (std::fprintf(stderr, "mid-sized frame: %d (%s)\n", rc, gv_last_error(ctx)), std::exit(1), rc)); };
            int rc;
            if ((rc = step(gv_transform_bind(ctx, mid.xf.data(), sizeof(Transform), (uint32_t)mid.xf.size(), &tl, mid.e2t.data(), (uint32_t)mid.e2t.size()))))
                return rc;
            for (uint32_t p = 0; p < 2; p++)
                if ((rc = step(gv_pool_bind(ctx, p, mid.meshes.data(), sizeof(Mesh), (uint32_t)mid.meshes.size() - p * 2000, &ml))))
                    return rc;
            GvView views[2] = {make_view(-1, 0, 1), make_view(0, 0, 1)};
            for (int frame = 0; frame < 2; frame++) {
                for (uint32_t p = 0; p < 2; p++) {
                    if ((rc = step(gv_cull(ctx, p, views, 2))))
                        return rc;
                    for (uint32_t v = 0; v < 2; v++)
                        if ((rc = step(gv_pool_sort(ctx, p, v, (int)(v ^ p)))))
                            return rc;
                }
                GvResult r{};
                for (uint32_t p = 0; p < 2; p++)
                    for (uint32_t v = 0; v < 2; v++)
                        if ((rc = step(gv_pool_results_fetch(ctx, p, v, v == 0, &r))))
                            return rc;
            }
            return GV_OK;
        };
        GvCtx* probe2 = nullptr;
        if (gv_create(&config, &probe2) != GV_OK)
            std::exit(1);
        const long before2 = gv_stub_allocations();
        sequence(probe2, false);
        const long total2 = gv_stub_allocations() - before2;
        gv_destroy(probe2);
        int failed2 = 0;
        for (long k = 1; k <= total2; k++) {
            GvCtx* ctx = nullptr;
            if (gv_create(&config, &ctx) != GV_OK)
                std::exit(1);
            gv_stub_fail_countdown() = k;
            const int rc = sequence(ctx, true);
            gv_stub_fail_countdown() = 0;
            if (rc != GV_OK) {
                failed2++;
                if ((rc != GV_E_OOM && rc != GV_E_HIP) || !*gv_last_error(ctx)) {
                    std::fprintf(stderr, "mid-sized frame, allocation %ld of %ld failing: %d (%s)\n", k, total2, rc, gv_last_error(ctx));
                    std::exit(1);
                }
            }
            if (sequence(ctx, true) != GV_OK) {
                std::fprintf(stderr, "mid-sized frame, allocation %ld of %ld failing once: the context did not recover (%s)\n", k, total2, gv_last_error(ctx));
                std::exit(1);
            }
            gv_destroy(ctx);
        }
        std::printf("allocation failures in a frame of mid-sized pools: %ld allocations failed in turn, %d calls reported it, every context recovered: ok\n", total2, failed2);
    }
}

// The exchange's bounded waits when the work behind them NEVER finishes (the stub's streams / events can be told so): status codes
// inside the limit, every context of the call marked broken, later calls GV_E_STATE, shutdown / destroy still release everything
// (ASan: no leak, nothing freed twice, no use after the abort).
static void exchange_bounded_waits()
{
    if (!std::getenv("GV_RCCL_LIBRARY")) {
        std::printf("exchange, bounded waits: skipped (GV_RCCL_LIBRARY not set)\n");
        return;
    }
    g_list_seed = 0;
    const int ranks = 2;
    std::vector<ExchangeRank> xs(ranks);
    std::vector<GvCtx*> ctxs;
    for (int r = 0; r < ranks; r++) {
        if (!xs[r].create(r, ranks))
            std::exit(1);
        ctxs.push_back(xs[r].ctx);
    }
    GvCtx* ctx = ctxs[0];
    std::vector<uint32_t> views(ranks, 0u);
    std::vector<GvExchangeFrame> sent(ranks), got(ranks);
    auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    // (1) the tails of a frame with short rows never arrive: frame 0 has no history, every row is short
    CHECK(gv_exchange_init_all(ctxs.data(), ranks));
    for (int r = 0; r < ranks; r++) {
        CHECK(gv_exchange_set_timeout(ctxs[r], 40));
        xs[r].produce(0, GV_EXCHANGE_P2P);
    }
    CHECK(gv_exchange_visible_all(ctxs.data(), ranks, views.data(), nullptr, 0, sent.data()));
    gv_stub_never_ready() = 2;
    auto t0 = std::chrono::steady_clock::now();
    EXPECT(gv_exchange_acquire_all(ctxs.data(), ranks, 0, got.data()), GV_E_TIMEOUT);
    if (since(t0) > 5.0 || !strstr(gv_last_error(ctxs[0]), "tails")) {
        std::fprintf(stderr, "bounded waits: the tails' wait took %.1f s / says: %s\n", since(t0), gv_last_error(ctxs[0]));
        std::exit(1);
    }
    gv_stub_never_ready() = 0;
    EXPECT(gv_exchange_visible_all(ctxs.data(), ranks, views.data(), nullptr, 0, sent.data()), GV_E_STATE);  // every context of the call is broken
    EXPECT(gv_exchange_acquire_all(ctxs.data(), ranks, 0, got.data()), GV_E_STATE);
    for (int r = 0; r < ranks; r++) {
        ctx = ctxs[r];
        CHECK(gv_exchange_shutdown(ctx));  // (nothing left to drain: the communicator was aborted on the spot)
    }
    // (2) a frame that was sent and never acquired, on a stream that never drains: shutdown reports it and releases all the same
    CHECK(gv_exchange_init_all(ctxs.data(), ranks));
    for (int r = 0; r < ranks; r++) {
        CHECK(gv_exchange_set_timeout(ctxs[r], 40));
        xs[r].produce(0, GV_EXCHANGE_ALLGATHER);
    }
    CHECK(gv_exchange_visible_all(ctxs.data(), ranks, views.data(), nullptr, 0, sent.data()));
    gv_stub_never_ready() = 1;
    t0 = std::chrono::steady_clock::now();
    EXPECT(gv_exchange_shutdown(ctxs[0]), GV_E_TIMEOUT);
    EXPECT(gv_exchange_visible_all(ctxs.data(), ranks, views.data(), nullptr, 0, sent.data()), GV_E_STATE);  // rank 0 has no communicator any more
    gv_destroy(ctxs[1]);  // (no status to return: comes back inside the limit, everything released)
    if (since(t0) > 5.0) {
        std::fprintf(stderr, "bounded waits: shutdown + destroy behind a stream that never drains took %.1f s\n", since(t0));
        std::exit(1);
    }
    gv_stub_never_ready() = 0;
    // (3) ... and a context is as good as new afterwards
    ctx = ctxs[0];
    CHECK(gv_exchange_init_all(ctxs.data(), 1));
    xs[0].ranks = 1;
    xs[0].produce(0, GV_EXCHANGE_BROADCAST);
    CHECK(gv_exchange_visible_all(ctxs.data(), 1, views.data(), nullptr, 0, sent.data()));
    CHECK(gv_exchange_acquire_all(ctxs.data(), 1, 0, got.data()));
    xs[0].check_acquired(sent[0], got[0], 0);
    if (xs[0].failures) {
        std::fprintf(stderr, "bounded waits: the context did not recover\n");
        std::exit(1);
    }
    CHECK(gv_exchange_shutdown(ctx));
    gv_destroy(ctx);
    std::printf("exchange, bounded waits: tails that never arrive -> GV_E_TIMEOUT on every context of the call; a stream that never drains -> shutdown "
                "GV_E_TIMEOUT, destroy returns; the context recovers: ok\n");
}

int main(int argc, char** argv)
{
    if (argc > 1) {  // schedule files (tests/schedules.py): replayed instead of the fixed exercise
        for (int k = 1; k < argc; k++)
            replay_schedule(argv[k]);
        std::printf("schedules: %d replayed ok\n", argc - 1);
        return 0;
    }
    // spatially ordered mirror (default), pool-slot order, forced block bounds, linear scan; flat and 4-deep
    exercise(0, 40000, 3);
    exercise(GV_CONFIG_KEEP_SLOT_ORDER, 30000, 0);
    exercise(GV_CONFIG_BLOCK_BOUNDS | GV_CONFIG_PROFILE_EVENTS, 30000, 2);
    exercise(GV_CONFIG_BLOCK_BOUNDS, 20000, 0);  // flat + exactly paired: block bounds / emit seeds patched per dirty block (mark_dirty_blocks)
    exercise(GV_CONFIG_LINEAR_SCAN | GV_CONFIG_HIZ_RG16F | GV_CONFIG_KEEP_SLOT_ORDER, 300000, 3);  // (above the device-gather and auto-bounds sizes)
    for (int ranks : {1, 2, 3, 8})
        exchange_in_threads(ranks);
    for (int seed = 1; seed <= 24; seed++)  // lists that jump at random between empty and the whole pool
        exchange_in_threads(2 + seed % 4, seed);
    for (int ranks : {1, 3, 4})
        exchange_in_one_thread(ranks, 0);
    for (int seed = 1; seed <= 6; seed++)
        exchange_in_one_thread(2 + seed % 3, seed);
    for (uint32_t mode : {GV_EXCHANGE_P2P, GV_EXCHANGE_BROADCAST, GV_EXCHANGE_ALLGATHER})
        batched_first_frame(3, mode);
    for (int ranks : {1, 2, 4, 8})
        exchange_by_peers(ranks, 0);
    for (int seed = 1; seed <= 6; seed++)
        exchange_by_peers(2 + seed % 3, seed);
    {  // the host workers as the shim uses them: every task exactly once, every item of a range exactly once
        std::vector<std::atomic<uint32_t>> ran(1000);
        for (uint32_t count : {0u, 1u, 7u, 1000u}) {
            for (auto& r : ran)
                r = 0;
            gv_host_parallel_tasks(count, [](void* user, uint32_t task) { (*static_cast<std::vector<std::atomic<uint32_t>>*>(user))[task]++; }, &ran);
            for (uint32_t k = 0; k < 1000; k++)
                if (ran[k] != (k < count ? 1u : 0u)) {
                    std::fprintf(stderr, "gv_host_parallel_tasks(%u): task %u ran %u times\n", count, k, ran[k].load());
                    std::exit(1);
                }
        }
        std::vector<uint8_t> seen(300000, 0);
        gv_host_parallel_ranges(5, 299990, [](void* user, uint32_t lo, uint32_t hi) {
            auto& v = *static_cast<std::vector<uint8_t>*>(user);
            for (uint32_t i = lo; i < hi; i++)
                v[i]++;
        }, &seen);
        for (uint32_t i = 0; i < seen.size(); i++)
            if (seen[i] != (i >= 5 && i < 299995 ? 1 : 0)) {
                std::fprintf(stderr, "gv_host_parallel_ranges: item %u visited %u times\n", i, seen[i]);
                std::exit(1);
            }
        std::printf("host workers (gv_host_parallel_tasks / _ranges): ok\n");
    }
    allocation_failures();
    exchange_allocation_failures(false);
    exchange_allocation_failures(true);
    exchange_bounded_waits();
    std::printf("host orchestration: ok\n");
    return 0;
}
