// gpu_visibility_system.hpp — the drop-in: an ecsm System that replaces the *prepare* phase of
// MeshRenderSystem (source/system/render/mesh.cpp:331-553 prepareMeshes, :893-903 preDeferredRender) and the
// pyramid build of HizRenderSystem (source/system/render/hiz.cpp:104-174) with calls into libgarden_vis.so.
//
// It subscribes exactly where MeshRenderSystem does (mesh.cpp:35,42-47: "Init", "PreForwardRender" /
// "PreDeferredRender"), reads CommonConstants like mesh.cpp:866-869,899-902, and leaves its results where the
// render phase expects them: MeshRenderComponent::isVisible (mesh.cpp:144-166) and
// UnsortedBuffer::combinedMeshes[0..drawCount) + drawCount/instanceCount (mesh.hpp:207-217), so
// renderUnsorted (mesh.cpp:556-636) can consume them unchanged.
//
// Only C++-ABI surface in the product: everything below it is the extern "C" boundary include/garden_vis.h.
// Errors: gv_* status codes are turned into exceptions here, the way GardenError is used upstream
// (include/garden/error.hpp:32-55).
#pragma once
#include <chrono>
#include <algorithm>
#include <functional>
#include <stdexcept>
#include <type_traits>
#include <string>
#include <vector>

#include "../../../include/garden_vis.h"
#include "garden_host.hpp"
#include "rank_shares.hpp"

namespace garden {

class GardenError : public std::runtime_error {
public:
    explicit GardenError(const std::string& message) : std::runtime_error(message) {}
};

class GpuVisibilitySystem final : public System, public Singleton<GpuVisibilitySystem> {
public:
    struct ShadowPass {  // IShadowMeshRenderSystem::prepareShadowRender outputs (mesh.hpp:166, csm.cpp:260-343)
        f32x4x4 viewProj;
        f32x4 cameraOffset;
        // The shadow system's own number of the pass: what isDrawReady(shadowPass) is asked with. renderShadows (mesh.cpp:809-815)
        // leaves out a pass whose prepareShadowRender says no and goes on counting — list the passes that were prepared, each with
        // its number. -1: the position in the list (no pass was left out).
        int8_t passIndex = -1;
        int8_t index(uint32_t position) const noexcept { return passIndex >= 0 ? passIndex : (int8_t)position; }
    };

private:
    GvCtx* ctx = nullptr;
    // ONE process, N GPUs (the reference's shape: one Manager, source/editor/entry.cpp:135): contexts[r] = rank r's context on
    // device devices[r]; ctx == contexts[0]. One context: everything below runs as before.
    std::vector<GvCtx*> contexts;
    RankShares rankShares;
    uint32_t rankGrid[3] = {1, 1, 1};
    double worldSide = 0.0;
    struct RanksSeen {  // what the shares were dealt from: any change re-deals
        uint64_t hierarchy = ~0ull, reparent = ~0ull;
        uint32_t transformOccupancy = ~0u, transformCount = ~0u;
        std::vector<uint64_t> meshVersion, meshRange;
        std::vector<uint32_t> meshOccupancy, meshCount;
    } ranksSeen;
    std::vector<IMeshRenderSystem*> meshSystems;  // prepareSystems(), mesh.cpp:69-108
    // mesh.hpp:219-223: one UnsortedBuffer per Color/Opaque/OIT/Refracted/TransDepth system; Translucent and UI
    // systems get a SortedBuffer each (counters) and share transSortedMeshes / uiSortedMeshes (records)
    std::vector<UnsortedBuffer*> unsortedBuffers;
    std::vector<SortedBuffer*> sortedBuffers;
    std::vector<SortedMesh> transSortedMeshes, uiSortedMeshes;
    uint32_t transDrawIndex = 0, uiDrawIndex = 0;
    uint32_t unsortedBufferCount = 0, sortedBufferCount = 0;
    std::vector<ShadowPass> shadowPasses;
    std::vector<std::vector<UnsortedBuffer*>> shadowBuffers;    // [unsorted buffer][pass]
    std::vector<std::vector<SortedMesh>> shadowTransMeshes;     // [pass]: the reference re-runs prepareMeshes per
    std::vector<uint32_t> shadowTransDrawIndex;                 // pass and draws at once; here every pass is kept
    std::vector<std::vector<SortedBuffer*>> shadowSortedBuffers; // [pass][bufferIndex of that pass]: the counters of sortedBuffers in a shadow pass
    bool hasAnyRefr = false, hasAnyOIT = false, hasAnyTD = false;  // mesh.hpp:232-234, set by the light pass (mesh.cpp:339,488-490)
    f32x4x4 uiViewProj;                                          // calcUiProjView(), mesh.cpp:851-859
    uint64_t seenHierarchy = ~0ull, seenTransform = ~0ull, seenReparent = 0, seenFlags = 0;
    std::vector<uint64_t> seenMesh;
    bool useHiz = false;

    void check(int rc, const char* what)
    {
        if (rc != GV_OK)
            throw GardenError(std::string(what) + " failed: " + gv_last_error(ctx));
    }

public:
    // host wall time of the prepare phase, accumulated over ticks ("Meshes Prepare" zone of the reference, by step)
    struct TickSeconds {
        double total = 0, cull = 0, sort = 0, fetch = 0, records = 0, share = 0, gather = 0;  // records: filling combinedMeshes from the fetch; share / gather: several ranks
    } tickSeconds;
    struct Stopwatch {
        double& sink;
        std::chrono::steady_clock::time_point start = std::chrono::steady_clock::now();
        explicit Stopwatch(double& into) : sink(into) {}
        ~Stopwatch() { sink += std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count(); }
    };

    bool isEnabled = true;
    // HizRenderSystem::isEnabled (hiz.cpp:107-115): a disabled system clears its pyramid to zero — farthest everywhere, nothing is
    // occluded; here the light pass simply runs without the occlusion query while this is false (the same visible set)
    bool isHizEnabled = true;
    bool isNonTranslucent = false;  // mesh.hpp:275 "Render only non translucent meshes": prepareSystems keeps Color / Opaque / UI systems (mesh.cpp:89-101)
    // true: records arrive as UnsortedMesh / SortedMesh structs (gv_pool_set_record_layout), combinedMeshes is one memcpy
    bool recordStructs = true;
    // true (with recordStructs): the device writes an unsorted buffer's records straight into its combinedMeshes
    // (gv_pool_set_record_target): the fetch leaves the records in the vector itself — no copy loop in the shim
    bool recordTargets = true;
    size_t recordTargetMaxBytes = size_t(256) << 20;  // larger arrays are not made targets: their records are copied after the fetch
    // true (with recordStructs): an unsorted buffer's records are NOT copied anywhere — UnsortedBuffer::meshes() hands the render
    // passes the library's own page-locked result buffer for this frame (valid until the pool's next gv_cull, i.e. through the
    // frame's render phase). The one change on the consumer's side: `unsortedBuffer->meshes()` where the reference reads
    // `unsortedBuffer->combinedMeshes.data()` (mesh.cpp:581,611). The vector is left alone (the engine's own array is never
    // page-locked — round 3 — so filling it costs a second copy: 53 us of a 216 us tick at 100 k entities).
    bool recordSpans = false;
    // true: also produce combinedMeshes records (bakedModel, distanceSq); false: isVisible + counters only
    bool emitRecords = true;
    // true: sortMeshes (mesh.cpp:265-328) runs on the device too: unsorted buffers ascending distanceSq
    // (front to back), so the engine's std::sort over combinedMeshes can be dropped
    bool sortOnDevice = true;
    // true: every frame also leaves the world matrices of all transforms on the device (gv_get_world): the sweep
    // rides on the first pool's cull (GV_SWEEP_WITH_CULL), fused into one pass when that pool is exactly paired
    bool sweepWorldMatrices = false;
    // with sweepWorldMatrices: keep the cache up to date with GV_SWEEP_INCREMENTAL instead (nothing is launched on a
    // frame without transform changes, only the subtrees under moved / re-parented transforms are re-swept otherwise)
    bool sweepIncremental = false;

    // Multi-GPU mode: called after every (mesh system, pass) of a frame has been gathered — frames[r] is rank r's acquired
    // GvExchangeFrame: on every device, every rank's complete list of WORLD mesh slots for that pass (rows valid until the exchange
    // after the next). A GPU-driven renderer enqueues its per-device work here; the host-side buffers are filled afterwards.
    std::function<void(uint32_t meshSystemIndex, int8_t shadowPass, const GvExchangeFrame* frames, uint32_t ranks)> onGathered;

    // blockBounds: GV_CONFIG_BLOCK_BOUNDS — worth it when most of the world is static (same results either way)
    explicit GpuVisibilitySystem(int device = 0, bool profile = false, bool blockBounds = false)
        : GpuVisibilitySystem(std::vector<int>{device}, 0.0, profile, blockBounds)
    {
    }
    // One context per entry of `devices` (the GPUs of the node; the same ordinal may appear more than once — ranks that share a
    // device, as the tests do on a one-GPU box). More than one: the pools are dealt to the ranks by the spatial rule of SURVEY.md
    // §8e (rank_shares.hpp; worldSide = the side of the world cube the cells are cut from), every rank culls its share, the
    // lists are gathered on the devices (gv_exchange_visible_all / _acquire_all: one thread drives all ranks) and the engine's
    // buffers — isVisible of the whole pools, combinedMeshes, the shared sorted arrays — are filled from the ranks' results.
    GpuVisibilitySystem(const std::vector<int>& devices, double worldSide, bool profile = false, bool blockBounds = false) : worldSide(worldSide)
    {
        if (devices.empty() || devices.size() > GV_EXCHANGE_MAX_RANKS)
            throw GardenError("GpuVisibilitySystem: between 1 and GV_EXCHANGE_MAX_RANKS devices");
        for (int device : devices) {
            GvConfig config{};
            config.struct_size = sizeof(GvConfig);
            config.device = device;
            config.hiz_rule = GV_HIZ_RULE_REFERENCE;
            config.flags = (profile ? GV_CONFIG_PROFILE_EVENTS : 0) | (blockBounds ? GV_CONFIG_BLOCK_BOUNDS : 0);
            GvCtx* made = nullptr;
            if (gv_create(&config, &made) != GV_OK) {
                for (auto c : contexts)
                    gv_destroy(c);
                throw GardenError(std::string("GpuVisibilitySystem: ") + gv_last_error(nullptr));
            }
            contexts.push_back(made);
        }
        ctx = contexts[0];
        if (contexts.size() > 1) {  // >= 512 cells per rank, doubling x, y, z in turn (garden_amd/multi.py::cell_grid: 8 ranks -> 16 x 16 x 16)
            for (uint32_t axis = 0; (uint64_t)rankGrid[0] * rankGrid[1] * rankGrid[2] < 512ull * contexts.size(); axis = (axis + 1) % 3)
                rankGrid[axis] *= 2;
            if (!(worldSide > 0.0))
                throw GardenError("GpuVisibilitySystem: several ranks need the side of the world cube");
        }
        setUiSize(1.0f, 1.0f);
        ECSM_SUBSCRIBE_TO_EVENT("Init", GpuVisibilitySystem::init);
    }
    ~GpuVisibilitySystem() override
    {
        // the contexts first: gv_destroy synchronises the stream and un-registers every record target, so no queued publish /
        // sort can still write into a combinedMeshes array (GV_DEBUG_RECORD_TARGET_PAGE_LOCK) and no target outlives its array
        if (contexts.size() > 1)
            for (auto c : contexts)
                (void)gv_exchange_shutdown(c);
        for (auto c : contexts)
            gv_destroy(c);
        ctx = nullptr;
        for (auto b : unsortedBuffers)
            delete b;
        for (auto b : sortedBuffers)
            delete b;
        for (auto& v : shadowBuffers)
            for (auto b : v)
                delete b;
        for (auto& v : shadowSortedBuffers)
            for (auto b : v)
                delete b;
    }

    GvCtx* getContext() const noexcept { return ctx; }
    uint32_t getRankCount() const noexcept { return (uint32_t)contexts.size(); }
    GvCtx* getContext(uint32_t rank) const { return contexts.at(rank); }
    const RankShares& getRankShares() const noexcept { return rankShares; }
    // what preRefrRender / the OIT and TransDepth passes ask (mesh.cpp:917-923): did the light pass prepare any such system?
    bool getHasAnyRefr() const noexcept { return hasAnyRefr; }
    bool getHasAnyOIT() const noexcept { return hasAnyOIT; }
    bool getHasAnyTD() const noexcept { return hasAnyTD; }
    // sortedBuffers as shadow pass `pass` leaves them (Translucent systems only: a UI system takes no bufferIndex there, mesh.cpp:416-419)
    const std::vector<SortedBuffer*>& getShadowSortedBuffers(uint32_t pass) const { return shadowSortedBuffers.at(pass); }
    const std::vector<UnsortedBuffer*>& getUnsortedBuffers() const noexcept { return unsortedBuffers; }
    uint32_t getUnsortedBufferCount() const noexcept { return unsortedBufferCount; }
    const std::vector<SortedBuffer*>& getSortedBuffers() const noexcept { return sortedBuffers; }
    uint32_t getSortedBufferCount() const noexcept { return sortedBufferCount; }
    // [0, getTransDrawCount()) back to front over all Translucent systems (mesh.hpp:222, transDrawIndex :226)
    const std::vector<SortedMesh>& getTransSortedMeshes() const noexcept { return transSortedMeshes; }
    uint32_t getTransDrawCount() const noexcept { return transDrawIndex; }
    const std::vector<SortedMesh>& getUiSortedMeshes() const noexcept { return uiSortedMeshes; }
    uint32_t getUiDrawCount() const noexcept { return uiDrawIndex; }
    const std::vector<UnsortedBuffer*>& getShadowBuffers(uint32_t unsortedBuffer) const { return shadowBuffers.at(unsortedBuffer); }
    const std::vector<SortedMesh>& getShadowTransMeshes(uint32_t pass) const { return shadowTransMeshes.at(pass); }
    uint32_t getShadowTransDrawCount(uint32_t pass) const { return shadowTransDrawIndex.at(pass); }
    void setShadowPasses(std::vector<ShadowPass> passes) { shadowPasses = std::move(passes); }
    // calcUiProjView (mesh.cpp:851-859): calcOrthoProjRevZ over [-w/2,w/2] x [-h/2,h/2], depth [-1,1]
    void setUiSize(float width, float height) noexcept
    {
        const float nearPlane = -1.0f, farPlane = 1.0f;
        memset(uiViewProj.m, 0, sizeof(uiViewProj.m));
        uiViewProj.m[0] = 2.0f / width;
        uiViewProj.m[5] = -2.0f / height;
        uiViewProj.m[10] = -1.0f / (farPlane - nearPlane);
        uiViewProj.m[14] = farPlane / (farPlane - nearPlane);
        uiViewProj.m[15] = 1.0f;
    }
    void setUiViewProj(const f32x4x4& viewProj) noexcept { uiViewProj = viewProj; }
    const f32x4x4& getUiViewProj() const noexcept { return uiViewProj; }

    // HizRenderSystem::downsampleHiz stand-in: hand over this frame's reversed-Z depth (host memory).
    void setHizDepth(const float* depth, uint32_t width, uint32_t height)
    {
        for (uint32_t r = 0; r < contexts.size(); r++)  // the pyramid is screen space: every rank builds the whole of it
            if (gv_hiz_build(contexts[r], depth, width, height, GV_MEM_HOST) != GV_OK)
                throw GardenError(std::string("gv_hiz_build failed: ") + gv_last_error(contexts[r]));
        useHiz = true;
    }

private:
    void init()
    {
        auto manager = Manager::Instance::get();
        // mesh.cpp:42-47 subscribes PreForwardRender / PreDeferredRender when those systems exist
        if (manager->hasEvent("PreForwardRender"))
            ECSM_SUBSCRIBE_TO_EVENT("PreForwardRender", GpuVisibilitySystem::preRender);
        if (manager->hasEvent("PreDeferredRender"))
            ECSM_SUBSCRIBE_TO_EVENT("PreDeferredRender", GpuVisibilitySystem::preRender);
        if (contexts.size() > 1)  // one thread, N ranks: the communicator is made inside one group
            check(gv_exchange_init_all(contexts.data(), (int)contexts.size()), "gv_exchange_init_all");
    }

    static bool isSortedType(MeshRenderType type) noexcept
    {
        return type == MeshRenderType::Translucent || type == MeshRenderType::UI;  // mesh.cpp:358-369,414
    }

    void prepareSystems()  // mesh.cpp:69-108 + the classification pass of prepareMeshes, mesh.cpp:341-395
    {
        meshSystems.clear();
        for (auto& sys : Manager::Instance::get()->getSystems())
            if (auto ms = dynamic_cast<IMeshRenderSystem*>(sys.get())) {
                const auto renderType = ms->getMeshRenderType();
                if (isNonTranslucent && renderType != MeshRenderType::Color && renderType != MeshRenderType::Opaque && renderType != MeshRenderType::UI)
                    continue;  // mesh.cpp:89-101: "Render only non translucent meshes" keeps Color, Opaque and UI systems
                meshSystems.push_back(ms);
            }
        if (meshSystems.size() > GV_MAX_POOLS)
            throw GardenError("GpuVisibilitySystem: more mesh systems than GV_MAX_POOLS");
        unsortedBufferCount = sortedBufferCount = 0;
        for (auto ms : meshSystems)
            (isSortedType(ms->getMeshRenderType()) ? sortedBufferCount : unsortedBufferCount)++;
        while (unsortedBuffers.size() < unsortedBufferCount)
            unsortedBuffers.push_back(new UnsortedBuffer());
        while (sortedBuffers.size() < sortedBufferCount)
            sortedBuffers.push_back(new SortedBuffer());
        shadowBuffers.resize(unsortedBufferCount);
        seenMesh.resize(meshSystems.size(), ~0ull);
    }

    static GvView makeView(const f32x4x4& viewProj, f32x4 cameraPos, f32x4 cameraOffset, int8_t shadowPass,
                           bool hiz, bool emit, bool distance2D = false)
    {
        GvView v{};
        memcpy(v.view_proj, viewProj.m, sizeof(v.view_proj));
        v.camera_position[0] = cameraPos.x; v.camera_position[1] = cameraPos.y; v.camera_position[2] = cameraPos.z;
        v.camera_offset[0] = cameraOffset.x; v.camera_offset[1] = cameraOffset.y; v.camera_offset[2] = cameraOffset.z;
        v.shadow_pass = shadowPass;
        v.use_hiz = hiz ? 1 : 0;
        v.distance_2d = distance2D ? 1 : 0;
        v.emit_records = emit ? 1 : 0;
        return v;
    }

    // The engine's record structs as the library's GvRecordLayout: results then arrive as arrays of UnsortedMesh /
    // SortedMesh and combinedMeshes is one memcpy. A struct the library cannot express (stride not a multiple of 16, or
    // larger than 128 bytes) keeps the three-array fetch.
    template <class Mesh>
    static bool recordLayoutOf(GvRecordLayout& layout, size_t componentSize, uint32_t bufferIndexField, uint32_t bufferIndex)
    {
        static_assert(std::is_trivially_copyable<Mesh>::value, "records are copied bytewise");
        static_assert(sizeof(((Mesh*)nullptr)->componentOffset) == 8 && sizeof(((Mesh*)nullptr)->bakedModel) == 48, "field sizes");
        layout = GvRecordLayout{(uint32_t)sizeof(Mesh), (uint32_t)offsetof(Mesh, componentOffset), (uint32_t)offsetof(Mesh, bakedModel),
                                (uint32_t)offsetof(Mesh, distanceSq), bufferIndexField, (uint32_t)componentSize, bufferIndex};
        return sizeof(Mesh) % 16 == 0 && sizeof(Mesh) <= 128;
    }

    // the fetched records of (pool, view) into meshes[0, count): one copy when the pool has a record layout
    template <class Mesh>
    void copyRecords(Mesh* meshes, const GvResult& r, IMeshRenderSystem* meshSystem, uint32_t pool, uint32_t viewIndex, uint32_t bufferIndex)
    {
        if (r.draw_count && !r.visible_idx) {
            const void* records = nullptr;
            uint32_t count = 0;
            check(gv_pool_results_records(ctx, pool, viewIndex, &records, &count), "gv_pool_results_records");
            memcpy(static_cast<void*>(meshes), records, (size_t)count * sizeof(Mesh));
            return;
        }
        const size_t componentSize = meshSystem->getMeshComponentSize();
        for (uint32_t k = 0; k < r.draw_count; k++) {
            meshes[k].componentOffset = (size_t)r.visible_idx[k] * componentSize;  // mesh.cpp:170 / :247
            memcpy(meshes[k].bakedModel.m, r.baked_model + (size_t)k * 12, 48);     // mesh.cpp:171 / :248
            meshes[k].distanceSq = r.distance_sq[k];                                // mesh.cpp:172 / :249-251
            if constexpr (std::is_same<Mesh, SortedMesh>::value)
                meshes[k].bufferIndex = bufferIndex;                                // mesh.cpp:252
        }
    }

    void fill(UnsortedBuffer* buffer, IMeshRenderSystem* meshSystem, uint32_t pool, uint32_t viewIndex, bool writeBack)
    {
        GvResult r{};
        {
            Stopwatch watch(tickSeconds.fetch);
            check(gv_pool_results_fetch(ctx, pool, viewIndex, writeBack ? 1 : 0, &r), "gv_pool_results_fetch");
        }
        Stopwatch watch(tickSeconds.records);
        buffer->meshSystem = meshSystem;
        buffer->drawCount = r.draw_count;
        buffer->instanceCount = r.instance_count;
        buffer->span = nullptr;
        if (!emitRecords)
            return;
        if (r.draw_count && !r.visible_idx) {
            const void* records = nullptr;
            uint32_t count = 0;
            check(gv_pool_results_records(ctx, pool, viewIndex, &records, &count), "gv_pool_results_records");
            if (records == static_cast<const void*>(buffer->combinedMeshes.data()))
                return;  // written in place by the device (recordTargets)
            if (recordSpans && recordStructs) {
                buffer->span = static_cast<const UnsortedMesh*>(records);  // read where they are (UnsortedBuffer::meshes())
                return;
            }
        }
        if (buffer->combinedMeshes.size() < r.draw_count)
            buffer->combinedMeshes.resize(r.draw_count);  // grown, never shrunk (mesh.cpp:377-395)
        copyRecords(buffer->combinedMeshes.data(), r, meshSystem, pool, viewIndex, 0);
    }

    // prepareSortedMeshes' tail (mesh.cpp:246-261): this system's records go behind the ones already in the shared
    // array, tagged with bufferIndex. Returns the number appended.
    uint32_t append(std::vector<SortedMesh>& combined, uint32_t& drawIndex, MeshBuffer* counters,
                    IMeshRenderSystem* meshSystem, uint32_t pool, uint32_t viewIndex, bool writeBack, uint32_t bufferIndex)
    {
        GvResult r{};
        {
            Stopwatch watch(tickSeconds.fetch);
            check(gv_pool_results_fetch(ctx, pool, viewIndex, writeBack ? 1 : 0, &r), "gv_pool_results_fetch");
        }
        Stopwatch watch(tickSeconds.records);
        if (counters) {
            counters->meshSystem = meshSystem;
            counters->drawCount = r.draw_count;
            counters->instanceCount = r.instance_count;
        }
        if (!emitRecords)
            return 0;
        if (combined.size() < (size_t)drawIndex + r.draw_count)
            combined.resize((size_t)drawIndex + r.draw_count);
        copyRecords(combined.data() + drawIndex, r, meshSystem, pool, viewIndex, bufferIndex);
        drawIndex += r.draw_count;
        return r.draw_count;
    }

    // sortMeshes for a shared array (mesh.cpp:296-326): every system's run arrives back-to-front from gv_sort, so a
    // merge of the runs is all that is left (one run: nothing to do).
    static void mergeRuns(std::vector<SortedMesh>& combined, const std::vector<uint32_t>& runEnds)
    {
        for (size_t i = 1; i < runEnds.size(); i++)
            std::inplace_merge(combined.begin(), combined.begin() + runEnds[i - 1], combined.begin() + runEnds[i]);
    }

    // preForwardRender / preDeferredRender, mesh.cpp:860-903: shadows first, then the main camera.
    void preRender()
    {
        if (!isEnabled)
            return;
        if (contexts.size() > 1) {
            preRenderRanks();
            return;
        }
        auto transformSystem = TransformSystem::Instance::get();
        auto graphicsSystem = GraphicsSystem::Instance::get();
        Stopwatch whole(tickSeconds.total);
        prepareSystems();
        bool sweepRequested = false;

        // Pools may have moved (create() can reallocate): re-bind every frame, as `gv_pool_bind` documents.
        static const GvTransformLayout transformLayout = {
            (uint32_t)offsetof(TransformComponent, entity), (uint32_t)offsetof(TransformComponent, parent),
            (uint32_t)offsetof(TransformComponent, posChildCount), (uint32_t)offsetof(TransformComponent, scaleChildCap),
            (uint32_t)offsetof(TransformComponent, rotation), (uint32_t)offsetof(TransformComponent, selfActive),
            (uint32_t)offsetof(TransformComponent, ancestorsActive),
            (uint32_t)offsetof(TransformComponent, modelWithAncestors)};
        static const GvMeshLayout meshLayout = {
            (uint32_t)offsetof(MeshRenderComponent, entity), (uint32_t)offsetof(MeshRenderComponent, isEnabled),
            (uint32_t)offsetof(MeshRenderComponent, isVisible), (uint32_t)offsetof(MeshRenderComponent, aabb.min),
            (uint32_t)offsetof(MeshRenderComponent, aabb.max)};
        auto& pool = transformSystem->getComponents();
        auto& entityMap = transformSystem->getEntityMap();
        check(gv_transform_bind(ctx, pool.getData(), sizeof(TransformComponent), pool.getOccupancy(), &transformLayout,
                                entityMap.data(), (uint32_t)entityMap.size()), "gv_transform_bind");
        if (seenHierarchy != transformSystem->hierarchyVersion) {
            check(gv_mark_dirty(ctx, GV_DIRTY_HIERARCHY, 0, 0), "gv_mark_dirty");  // entities came or went
            seenHierarchy = transformSystem->hierarchyVersion;
            seenTransform = transformSystem->transformVersion;
        } else {
            if (seenReparent != transformSystem->reparentVersion && transformSystem->reparentLo < transformSystem->reparentHi)
                check(gv_mark_dirty(ctx, GV_DIRTY_HIERARCHY, transformSystem->reparentLo,
                                    transformSystem->reparentHi - transformSystem->reparentLo), "gv_mark_dirty");
            // writers that itemise what they moved (TransformSystem::markMoved): exactly those slots
            for (const auto& moved : transformSystem->movedRanges)
                check(gv_mark_dirty(ctx, GV_DIRTY_TRANSFORM, moved.first, moved.second), "gv_mark_dirty");
            if (seenTransform != transformSystem->transformVersion) {
                check(gv_mark_dirty(ctx, GV_DIRTY_TRANSFORM, 0, pool.getOccupancy()), "gv_mark_dirty");
                seenTransform = transformSystem->transformVersion;
            } else if (seenFlags != transformSystem->flagsVersion && transformSystem->flagsLo < transformSystem->flagsHi) {
                // setActive / re-parenting flipped the active flags of some subtrees: only that slot range moves
                check(gv_mark_dirty(ctx, GV_DIRTY_TRANSFORM, transformSystem->flagsLo,
                                    transformSystem->flagsHi - transformSystem->flagsLo), "gv_mark_dirty");
            }
        }
        seenReparent = transformSystem->reparentVersion;
        seenFlags = transformSystem->flagsVersion;
        transformSystem->clearReparentRange();
        transformSystem->clearFlagsRange();
        transformSystem->clearMovedRanges();

        const auto& cc = graphicsSystem->getCommonConstants();
        const uint32_t passCount = (uint32_t)std::min<size_t>(shadowPasses.size(), GV_MAX_VIEWS - 1);
        transDrawIndex = uiDrawIndex = 0;                  // mesh.cpp:337
        hasAnyRefr = hasAnyOIT = hasAnyTD = false;         // mesh.cpp:339
        shadowTransMeshes.resize(passCount);
        shadowTransDrawIndex.assign(passCount, 0);
        shadowSortedBuffers.resize(passCount);
        std::vector<uint32_t> transRuns, uiRuns;
        std::vector<std::vector<uint32_t>> shadowTransRuns(passCount);

        // Phase 1 — every mesh system's cull (and sort request) is issued before any result is read: results are kept
        // per (pool, view), so the device works through the systems back to back while the host only enqueues; the
        // reference dispatches every system's tasks to its thread pool and waits once, too (mesh.cpp:408-546, :548).
        // viewPass[p][v] = the pass view v of system p's cull stands for (-1 the light pass, s >= 0 shadow pass s): only passes
        // that get through the reference's gate are culled at all (mesh.cpp:426,482).
        std::vector<std::vector<int8_t>> viewPass(meshSystems.size());
        struct Indices {
            uint32_t main = 0, shadow = 0;  // bufferIndex in the light pass / in a shadow pass
        };
        std::vector<Indices> indices(meshSystems.size());
        uint32_t sortedSeen = 0, shadowSortedSeen = 0, unsortedSeen = 0;
        check(gv_cull_batch_begin(ctx), "gv_cull_batch_begin");  // engine-sized pools: one cull / emit / sort / publish launch per tick
        for (uint32_t p = 0; p < meshSystems.size(); p++) {
            auto meshSystem = meshSystems[p];
            const auto renderType = meshSystem->getMeshRenderType();
            const auto& componentPool = meshSystem->getMeshComponentPool();        // mesh.cpp:410
            const uint32_t componentCount = componentPool.getCount();              // mesh.cpp:411
            const uint32_t occupancy = componentPool.getOccupancy();
            const bool sorted = isSortedType(renderType);
            // bufferIndex: in the light pass Translucent and UI systems count (mesh.cpp:419); in a shadow pass a UI system leaves the
            // loop before it takes one (:416-417), so a Translucent system's index there counts Translucent systems only
            if (sorted) {
                indices[p].main = sortedSeen++;
                if (renderType == MeshRenderType::Translucent)
                    indices[p].shadow = shadowSortedSeen++;
            } else {
                indices[p].main = indices[p].shadow = unsortedSeen++;
            }
            // the pool is bound and its changes are taken over whether or not the system is drawn this frame: a system that
            // becomes ready later is culled from the pool as it is then
            check(gv_pool_bind(ctx, p, componentPool.getData(), meshSystem->getMeshComponentSize(), occupancy, &meshLayout), "gv_pool_bind");
            bool inPlace = false;
            {
                GvRecordLayout layout;
                const bool expressible = sorted
                    ? recordLayoutOf<SortedMesh>(layout, meshSystem->getMeshComponentSize(), (uint32_t)offsetof(SortedMesh, bufferIndex), indices[p].main)
                    : recordLayoutOf<UnsortedMesh>(layout, meshSystem->getMeshComponentSize(), GV_NONE, 0);
                check(gv_pool_set_record_layout(ctx, p, emitRecords && recordStructs && expressible ? &layout : nullptr),
                      "gv_pool_set_record_layout");
                inPlace = !sorted && emitRecords && recordStructs && expressible && recordTargets;
            }
            auto versioned = dynamic_cast<VersionedMeshSystem*>(meshSystem);
            if (versioned && versioned->reportsChanges) {
                if (seenMesh[p] != versioned->meshVersion) {
                    check(gv_mark_dirty(ctx, GV_DIRTY_MESH, p << 28, occupancy), "gv_mark_dirty");
                    seenMesh[p] = versioned->meshVersion;
                } else if (versioned->meshLo < versioned->meshHi) {  // created / destroyed / edited components only
                    check(gv_mark_dirty(ctx, GV_DIRTY_MESH, (p << 28) | versioned->meshLo, versioned->meshHi - versioned->meshLo),
                          "gv_mark_dirty");
                }
                versioned->clearMeshRange();
            } else {  // unknown writer: re-mirror the pool every frame (always correct)
                check(gv_mark_dirty(ctx, GV_DIRTY_MESH, p << 28, occupancy), "gv_mark_dirty");
            }

            // The gate of mesh.cpp:426 / :482, pass by pass: `componentCount == 0 || !meshSystem->isDrawReady(shadowPass)` leaves the
            // system's counters at 0 for that pass and touches nothing else — in the light pass isVisible keeps last frame's bytes.
            // The light pass (shadowPass -1: writes isVisible), then the shadow passes (mesh.cpp:809-843). UI: its own ortho
            // frustum, camera at the origin, 2D distance key, no shadow passes (mesh.cpp:416,436-442).
            std::vector<GvView> views;
            auto& passes = viewPass[p];
            const bool lightPass = componentCount != 0 && meshSystem->isDrawReady(-1);
            if (lightPass) {
                if (renderType == MeshRenderType::UI)
                    views.push_back(makeView(uiViewProj, f32x4(), f32x4(), -1, false, emitRecords, true));
                else
                    views.push_back(makeView(cc.viewProj, cc.cameraPos, f32x4(), -1, useHiz && isHizEnabled, emitRecords));
                passes.push_back(-1);
                if (!sorted) {  // mesh.cpp:488-490
                    hasAnyRefr |= renderType == MeshRenderType::Refracted;
                    hasAnyOIT |= renderType == MeshRenderType::OIT;
                    hasAnyTD |= renderType == MeshRenderType::TransDepth;
                }
            }
            if (renderType != MeshRenderType::UI)
                for (uint32_t s = 0; s < passCount; s++)
                    if (componentCount != 0 && meshSystem->isDrawReady(shadowPasses[s].index(s))) {
                        views.push_back(makeView(shadowPasses[s].viewProj, cc.cameraPos, shadowPasses[s].cameraOffset, shadowPasses[s].index(s), false, emitRecords));
                        passes.push_back((int8_t)s);
                    }
            if (!sorted) {
                // An unsorted system's buffers are its own (mesh.hpp:213-217), sized before its tasks run like the
                // reference's scratch (mesh.cpp:377-395; here to the occupancy, which bounds any draw count): the device
                // writes the records where renderUnsorted reads them. Shared sorted arrays keep the append + merge.
                auto& sb = shadowBuffers[indices[p].main];
                while (sb.size() < passCount)
                    sb.push_back(new UnsortedBuffer());
                for (uint32_t v = 0; v < views.size(); v++) {
                    UnsortedBuffer* buffer = passes[v] < 0 ? unsortedBuffers[indices[p].main] : sb[passes[v]];
                    const bool target = inPlace && !recordSpans && occupancy && (size_t)occupancy * sizeof(UnsortedMesh) <= recordTargetMaxBytes;
                    if (target && buffer->combinedMeshes.size() < occupancy) {
                        // growing re-allocates: let the old range go while it is still allocated
                        check(gv_pool_set_record_target(ctx, p, v, nullptr, 0), "gv_pool_set_record_target");
                        buffer->combinedMeshes.resize(occupancy);
                    }
                    if (target && reinterpret_cast<uintptr_t>(buffer->combinedMeshes.data()) % 16 == 0) {
                        check(gv_pool_set_record_target(ctx, p, v, buffer->combinedMeshes.data(),
                                                        buffer->combinedMeshes.size() * sizeof(UnsortedMesh)), "gv_pool_set_record_target");
                    } else {
                        check(gv_pool_set_record_target(ctx, p, v, nullptr, 0), "gv_pool_set_record_target");
                    }
                }
            }
            if (views.empty())
                continue;  // no pass draws this system this frame: nothing is culled, sorted or read for it
            if (sweepWorldMatrices && !sweepRequested) {
                check(gv_sweep(ctx, sweepIncremental ? GV_SWEEP_INCREMENTAL : GV_SWEEP_WITH_CULL), "gv_sweep");
                sweepRequested = true;
            }
            {
                Stopwatch watch(tickSeconds.cull);
                check(gv_cull(ctx, p, views.data(), (uint32_t)views.size()), "gv_cull");
            }
            // sortMeshes, mesh.cpp:270-295: unsorted buffers front to back (UnsortedMesh::operator<, mesh.hpp:196), OIT is
            // not sorted; sorted systems back to front (SortedMesh::operator<, mesh.hpp:204)
            if (emitRecords && sortOnDevice && renderType != MeshRenderType::OIT) {
                Stopwatch watch(tickSeconds.sort);
                for (uint32_t v = 0; v < views.size(); v++)
                    check(gv_pool_sort(ctx, p, v, sorted ? 1 : 0), "gv_pool_sort");
            }
        }
        for (uint32_t s = 0; s < passCount; s++)
            while (shadowSortedBuffers[s].size() < shadowSortedSeen)
                shadowSortedBuffers[s].push_back(new SortedBuffer());
        if (sweepWorldMatrices && !sweepRequested)  // no system is drawn this frame: the cache is kept current all the same
            check(gv_sweep(ctx, sweepIncremental ? GV_SWEEP_INCREMENTAL : GV_SWEEP_VALU), "gv_sweep");

        // Phase 2 — read the results (the first fetch publishes every small pool's views at once). Every buffer of every pass is
        // first put in the state mesh.cpp:420-424 / :476-480 leave it in — its system, both counters 0 — and only the passes
        // that were culled fill theirs.
        auto reset = [](MeshBuffer* buffer, IMeshRenderSystem* meshSystem) {
            buffer->meshSystem = meshSystem;
            buffer->drawCount = 0;
            buffer->instanceCount = 0;
        };
        for (uint32_t p = 0; p < meshSystems.size(); p++) {
            auto meshSystem = meshSystems[p];
            const auto renderType = meshSystem->getMeshRenderType();
            const auto& passes = viewPass[p];
            if (isSortedType(renderType)) {
                const uint32_t bufferIndex = indices[p].main, shadowIndex = indices[p].shadow;
                reset(sortedBuffers[bufferIndex], meshSystem);
                if (renderType == MeshRenderType::Translucent)
                    for (uint32_t s = 0; s < passCount; s++)
                        reset(shadowSortedBuffers[s][shadowIndex], meshSystem);
                for (uint32_t v = 0; v < passes.size(); v++) {
                    if (passes[v] >= 0) {
                        const uint32_t s = (uint32_t)passes[v], first = shadowTransDrawIndex[s];
                        const uint32_t added = append(shadowTransMeshes[s], shadowTransDrawIndex[s], shadowSortedBuffers[s][shadowIndex], meshSystem, p, v,
                                                      false, shadowIndex);
                        if (shadowIndex != bufferIndex)  // (records built on the device carry the light pass's index)
                            for (uint32_t k = 0; k < added; k++)
                                shadowTransMeshes[s][first + k].bufferIndex = shadowIndex;
                        shadowTransRuns[s].push_back(shadowTransDrawIndex[s]);
                    } else if (renderType == MeshRenderType::UI) {
                        append(uiSortedMeshes, uiDrawIndex, sortedBuffers[bufferIndex], meshSystem, p, v, true, bufferIndex);
                        uiRuns.push_back(uiDrawIndex);
                    } else {
                        append(transSortedMeshes, transDrawIndex, sortedBuffers[bufferIndex], meshSystem, p, v, true, bufferIndex);
                        transRuns.push_back(transDrawIndex);
                    }
                }
            } else {
                const uint32_t bufferIndex = indices[p].main;
                auto& sb = shadowBuffers[bufferIndex];
                reset(unsortedBuffers[bufferIndex], meshSystem);
                unsortedBuffers[bufferIndex]->span = nullptr;
                for (uint32_t s = 0; s < passCount; s++) {
                    reset(sb[s], meshSystem);
                    sb[s]->span = nullptr;
                }
                for (uint32_t v = 0; v < passes.size(); v++)
                    if (passes[v] >= 0)
                        fill(sb[passes[v]], meshSystem, p, v, false);
                for (uint32_t v = 0; v < passes.size(); v++)
                    if (passes[v] < 0)
                        fill(unsortedBuffers[bufferIndex], meshSystem, p, v, true);
            }
        }
        if (emitRecords && sortOnDevice) {
            mergeRuns(transSortedMeshes, transRuns);
            mergeRuns(uiSortedMeshes, uiRuns);
            for (uint32_t s = 0; s < passCount; s++)
                mergeRuns(shadowTransMeshes[s], shadowTransRuns[s]);
        }
    }

    // ---- ONE process, N GPUs ----
    void checkRank(uint32_t rank, int rc, const char* what)
    {
        if (rc != GV_OK)
            throw GardenError(std::string(what) + " failed on rank " + std::to_string(rank) + ": " + gv_last_error(contexts[rank]));
    }

    // Deals the pools again when anything structural changed (entities or components came or went, a parent link moved); otherwise
    // copies the transforms that changed into their ranks' pools. Binds every rank's share.
    void syncRanks(TransformSystem* transformSystem)
    {
        const uint32_t ranks = (uint32_t)contexts.size();
        auto& pool = transformSystem->getComponents();
        bool structural = ranksSeen.hierarchy != transformSystem->hierarchyVersion || ranksSeen.reparent != transformSystem->reparentVersion ||
                          ranksSeen.transformOccupancy != pool.getOccupancy() || ranksSeen.transformCount != pool.getCount() ||
                          ranksSeen.meshVersion.size() != meshSystems.size();
        ranksSeen.meshVersion.resize(meshSystems.size(), ~0ull);
        ranksSeen.meshRange.resize(meshSystems.size(), ~0ull);
        ranksSeen.meshOccupancy.resize(meshSystems.size(), ~0u);
        ranksSeen.meshCount.resize(meshSystems.size(), ~0u);
        for (size_t p = 0; p < meshSystems.size(); p++) {
            const auto& meshPool = meshSystems[p]->getMeshComponentPool();
            auto versioned = dynamic_cast<VersionedMeshSystem*>(meshSystems[p]);
            if (versioned && !versioned->reportsChanges)
                versioned = nullptr;
            const uint64_t version = versioned ? versioned->meshVersion : 0, range = versioned ? versioned->rangeVersion : 0;
            structural = structural || !versioned || ranksSeen.meshVersion[p] != version || ranksSeen.meshRange[p] != range ||
                         ranksSeen.meshOccupancy[p] != meshPool.getOccupancy() || ranksSeen.meshCount[p] != meshPool.getCount();
            ranksSeen.meshVersion[p] = version;
            ranksSeen.meshRange[p] = range;
            ranksSeen.meshOccupancy[p] = meshPool.getOccupancy();
            ranksSeen.meshCount[p] = meshPool.getCount();
            if (versioned)
                versioned->clearMeshRange();
        }
        // a slot that was freed and handed to another entity since the deal (same occupancy, same count) is structural too
        if (!structural && seenFlags != transformSystem->flagsVersion)
            for (uint32_t i = transformSystem->flagsLo; i < transformSystem->flagsHi && !structural; i++)
                structural = !rankShares.sameEntity(transformSystem, i);
        ranksSeen.hierarchy = transformSystem->hierarchyVersion;
        ranksSeen.reparent = transformSystem->reparentVersion;
        ranksSeen.transformOccupancy = pool.getOccupancy();
        ranksSeen.transformCount = pool.getCount();
        static const GvTransformLayout transformLayout = {
            (uint32_t)offsetof(TransformComponent, entity), (uint32_t)offsetof(TransformComponent, parent),
            (uint32_t)offsetof(TransformComponent, posChildCount), (uint32_t)offsetof(TransformComponent, scaleChildCap),
            (uint32_t)offsetof(TransformComponent, rotation), (uint32_t)offsetof(TransformComponent, selfActive),
            (uint32_t)offsetof(TransformComponent, ancestorsActive),
            (uint32_t)offsetof(TransformComponent, modelWithAncestors)};
        static const GvMeshLayout meshLayout = {
            (uint32_t)offsetof(MeshRenderComponent, entity), (uint32_t)offsetof(MeshRenderComponent, isEnabled),
            (uint32_t)offsetof(MeshRenderComponent, isVisible), (uint32_t)offsetof(MeshRenderComponent, aabb.min),
            (uint32_t)offsetof(MeshRenderComponent, aabb.max)};
        if (structural) {
            rankShares.deal(transformSystem, meshSystems, ranks, rankGrid, worldSide);
            seenTransform = transformSystem->transformVersion;
            seenFlags = transformSystem->flagsVersion;
        }
        // what moved since the last frame (content only): every slot, or the itemised ones
        std::vector<std::vector<std::pair<uint32_t, uint32_t>>> dirty(ranks);  // per rank: (local slot, count)
        auto touch = [&](uint32_t worldSlot) {
            const uint32_t rank = worldSlot < rankShares.rankOfTransform.size() ? rankShares.rankOfTransform[worldSlot] : GV_NONE;
            if (rank == GV_NONE)
                return;
            rankShares.copyTransform(transformSystem, worldSlot);
            const uint32_t local = rankShares.localOfTransform[worldSlot];
            auto& ranges = dirty[rank];
            if (!ranges.empty() && ranges.back().first + ranges.back().second == local)
                ranges.back().second++;
            else
                ranges.push_back({local, 1u});
        };
        bool everything = false;
        if (!structural) {
            if (seenTransform != transformSystem->transformVersion) {
                everything = true;
                for (uint32_t i = 0; i < pool.getOccupancy(); i++)
                    rankShares.copyTransform(transformSystem, i);
            } else {
                for (const auto& moved : transformSystem->movedRanges)
                    for (uint32_t i = 0; i < moved.second; i++)
                        touch(moved.first + i);
                if (seenFlags != transformSystem->flagsVersion)
                    for (uint32_t i = transformSystem->flagsLo; i < transformSystem->flagsHi; i++)
                        touch(i);
            }
            seenTransform = transformSystem->transformVersion;
            seenFlags = transformSystem->flagsVersion;
        }
        transformSystem->clearReparentRange();
        transformSystem->clearFlagsRange();
        transformSystem->clearMovedRanges();
        for (uint32_t r = 0; r < ranks; r++) {
            auto& share = rankShares.shares[r];
            checkRank(r, gv_transform_bind(contexts[r], share.transforms.data(), sizeof(TransformComponent), (uint32_t)share.transforms.size(), &transformLayout,
                                           share.entityToTransform.data(), (uint32_t)share.entityToTransform.size()), "gv_transform_bind");
            for (uint32_t p = 0; p < meshSystems.size(); p++) {
                auto& mesh = share.meshes[p];
                checkRank(r, gv_pool_bind(contexts[r], p, mesh.components.data(), (uint32_t)mesh.stride, mesh.occupancy(), &meshLayout), "gv_pool_bind");
                checkRank(r, gv_pool_set_record_layout(contexts[r], p, nullptr), "gv_pool_set_record_layout");
                if (structural) {
                    checkRank(r, gv_pool_set_index_map(contexts[r], p, mesh.worldSlot.data(), mesh.occupancy()), "gv_pool_set_index_map");
                    checkRank(r, gv_mark_dirty(contexts[r], GV_DIRTY_MESH, p << 28, mesh.occupancy()), "gv_mark_dirty");
                }
            }
            if (structural) {
                checkRank(r, gv_mark_dirty(contexts[r], GV_DIRTY_HIERARCHY, 0, 0), "gv_mark_dirty");
            } else if (everything) {
                checkRank(r, gv_mark_dirty(contexts[r], GV_DIRTY_TRANSFORM, 0, (uint32_t)share.transforms.size()), "gv_mark_dirty");
            } else {
                for (const auto& range : dirty[r])
                    checkRank(r, gv_mark_dirty(contexts[r], GV_DIRTY_TRANSFORM, range.first, range.second), "gv_mark_dirty");
            }
        }
    }

    // The frame with several ranks. The same classification, gate and buffers as preRender(); per (mesh system, pass): every rank
    // culls (and sorts) its share, the ranks' lists are gathered on the devices, and the engine's buffers are filled from the
    // ranks' results — a rank holds the models of the entities it owns, the host holds all of them:
    //   isVisible      rank r's light-pass bytes, scattered to the engine's pool through the local -> world slot table
    //   records        componentOffset = WORLD slot * componentSize; each rank's run arrives sorted, the runs are merged
    //   counters       summed over the ranks
    void preRenderRanks()
    {
        auto transformSystem = TransformSystem::Instance::get();
        auto graphicsSystem = GraphicsSystem::Instance::get();
        Stopwatch whole(tickSeconds.total);
        if (sweepWorldMatrices)  // (a rank keeps the world matrices of ITS entities: gv_sweep / gv_get_world on getContext(rank), in local slots)
            throw GardenError("GpuVisibilitySystem: sweepWorldMatrices is a one-context option; with several ranks ask each rank's context");
        prepareSystems();
        {
            Stopwatch watch(tickSeconds.share);
            syncRanks(transformSystem);
        }
        const uint32_t ranks = (uint32_t)contexts.size();
        const auto& cc = graphicsSystem->getCommonConstants();
        const uint32_t passCount = (uint32_t)std::min<size_t>(shadowPasses.size(), GV_MAX_VIEWS - 1);
        transDrawIndex = uiDrawIndex = 0;
        hasAnyRefr = hasAnyOIT = hasAnyTD = false;
        shadowTransMeshes.resize(passCount);
        shadowTransDrawIndex.assign(passCount, 0);
        shadowSortedBuffers.resize(passCount);
        std::vector<uint32_t> transRuns, uiRuns;
        std::vector<std::vector<uint32_t>> shadowTransRuns(passCount);
        uint32_t sortedSeen = 0, shadowSortedSeen = 0, unsortedSeen = 0;
        auto reset = [](MeshBuffer* buffer, IMeshRenderSystem* meshSystem) {
            buffer->meshSystem = meshSystem;
            buffer->drawCount = 0;
            buffer->instanceCount = 0;
        };
        std::vector<GvExchangeFrame> frames(ranks);
        std::vector<uint32_t> viewIndices(ranks);
        // Phase 1 — as in preRender(): every system's cull (and sort request) goes to every rank before any result is read, so each
        // device works through the systems back to back (engine-sized pools: one launch per tick, gv_cull_batch_begin) while the host
        // only enqueues. issued[p]: the passes system p was culled for, and its buffers.
        struct Issued {
            std::vector<int8_t> passes;
            uint32_t bufferIndex = 0, shadowIndex = 0;
        };
        std::vector<Issued> issued(meshSystems.size());
        for (uint32_t r = 0; r < ranks; r++)
            checkRank(r, gv_cull_batch_begin(contexts[r]), "gv_cull_batch_begin");
        for (uint32_t p = 0; p < meshSystems.size(); p++) {
            auto meshSystem = meshSystems[p];
            const auto renderType = meshSystem->getMeshRenderType();
            const auto& componentPool = meshSystem->getMeshComponentPool();
            const uint32_t componentCount = componentPool.getCount();
            const bool sorted = isSortedType(renderType);
            uint32_t bufferIndex = 0, shadowIndex = 0;
            if (sorted) {
                bufferIndex = sortedSeen++;
                if (renderType == MeshRenderType::Translucent) {
                    shadowIndex = shadowSortedSeen++;
                    for (uint32_t s = 0; s < passCount; s++) {
                        while (shadowSortedBuffers[s].size() <= shadowIndex)
                            shadowSortedBuffers[s].push_back(new SortedBuffer());
                        reset(shadowSortedBuffers[s][shadowIndex], meshSystem);
                    }
                }
                reset(sortedBuffers[bufferIndex], meshSystem);
            } else {
                bufferIndex = unsortedSeen++;
                auto& sb = shadowBuffers[bufferIndex];
                while (sb.size() < passCount)
                    sb.push_back(new UnsortedBuffer());
                reset(unsortedBuffers[bufferIndex], meshSystem);
                unsortedBuffers[bufferIndex]->span = nullptr;
                for (uint32_t s = 0; s < passCount; s++) {
                    reset(sb[s], meshSystem);
                    sb[s]->span = nullptr;
                }
            }
            // the gate of mesh.cpp:426 / :482, pass by pass (see preRender)
            std::vector<GvView> views;
            std::vector<int8_t> passes;
            if (componentCount != 0 && meshSystem->isDrawReady(-1)) {
                if (renderType == MeshRenderType::UI)
                    views.push_back(makeView(uiViewProj, f32x4(), f32x4(), -1, false, emitRecords, true));
                else
                    views.push_back(makeView(cc.viewProj, cc.cameraPos, f32x4(), -1, useHiz && isHizEnabled, emitRecords));
                passes.push_back(-1);
                if (!sorted) {
                    hasAnyRefr |= renderType == MeshRenderType::Refracted;
                    hasAnyOIT |= renderType == MeshRenderType::OIT;
                    hasAnyTD |= renderType == MeshRenderType::TransDepth;
                }
            }
            if (renderType != MeshRenderType::UI)
                for (uint32_t s = 0; s < passCount; s++)
                    if (componentCount != 0 && meshSystem->isDrawReady(shadowPasses[s].index(s))) {
                        views.push_back(makeView(shadowPasses[s].viewProj, cc.cameraPos, shadowPasses[s].cameraOffset, shadowPasses[s].index(s), false, emitRecords));
                        passes.push_back((int8_t)s);
                    }
            issued[p].bufferIndex = bufferIndex;
            issued[p].shadowIndex = shadowIndex;
            if (views.empty())
                continue;
            issued[p].passes = passes;
            Stopwatch watch(tickSeconds.cull);
            for (uint32_t r = 0; r < ranks; r++) {
                checkRank(r, gv_cull(contexts[r], p, views.data(), (uint32_t)views.size()), "gv_cull");
                if (emitRecords && sortOnDevice && renderType != MeshRenderType::OIT)
                    for (uint32_t v = 0; v < views.size(); v++)
                        checkRank(r, gv_pool_sort(contexts[r], p, v, sorted ? 1 : 0), "gv_pool_sort");
            }
        }
        // Phase 2 — system by system: the gather, then the engine's buffers from the ranks' results (the first reader on a rank
        // launches what that rank recorded).
        for (uint32_t p = 0; p < meshSystems.size(); p++) {
            const auto& passes = issued[p].passes;
            if (passes.empty())
                continue;
            auto meshSystem = meshSystems[p];
            const auto renderType = meshSystem->getMeshRenderType();
            const auto& componentPool = meshSystem->getMeshComponentPool();
            const size_t componentSize = meshSystem->getMeshComponentSize();
            const bool sorted = isSortedType(renderType);
            const uint32_t bufferIndex = issued[p].bufferIndex, shadowIndex = issued[p].shadowIndex;
            // the gather (mesh.cpp:177-183: every worker's records into the shared array): on every device, every rank's list of
            // WORLD slots for the pass — complete in every frame (gv_exchange_acquire_all); the pool is named: other systems have
            // been culled since
            if (emitRecords)
                for (uint32_t v = 0; v < passes.size(); v++) {
                    Stopwatch watch(tickSeconds.gather);
                    std::fill(viewIndices.begin(), viewIndices.end(), v);
                    check(gv_pool_exchange_visible_all(contexts.data(), (int)ranks, p, viewIndices.data(), nullptr, 0, frames.data()), "gv_pool_exchange_visible_all");
                    check(gv_exchange_acquire_all(contexts.data(), (int)ranks, frames[0].frame, frames.data()), "gv_exchange_acquire_all");
                    if (onGathered)
                        onGathered(p, passes[v], frames.data(), ranks);
                }
            // the engine's buffers from the ranks' results
            for (uint32_t v = 0; v < passes.size(); v++) {
                const int8_t pass = passes[v];
                std::vector<uint32_t> runs;
                // every rank's results of the pass (library-owned host memory, valid until the pool's next gv_cull on that rank); the
                // light pass also writes the rank's isVisible bytes into its share, from where they go to the engine's pool
                std::vector<GvResult> results(ranks);
                uint32_t total = 0, instances = 0;
                for (uint32_t r = 0; r < ranks; r++) {
                    {
                        Stopwatch watch(tickSeconds.fetch);
                        checkRank(r, gv_pool_results_fetch(contexts[r], p, v, pass < 0 ? 1 : 0, &results[r]), "gv_pool_results_fetch");
                    }
                    total += results[r].draw_count;
                    instances += results[r].instance_count;
                    if (pass < 0) {  // mesh.cpp:144-166, through the local -> world slot table
                        Stopwatch watch(tickSeconds.records);
                        const auto& share = rankShares.shares[r].meshes[p];
                        uint8_t* world = reinterpret_cast<uint8_t*>(componentPool.getData());
                        for (uint32_t j = 0; j < share.occupancy(); j++)
                            reinterpret_cast<MeshRenderComponent*>(world + (size_t)share.worldSlot[j] * componentSize)->isVisible =
                                reinterpret_cast<const MeshRenderComponent*>(share.components.data() + (size_t)j * componentSize)->isVisible;
                    }
                }
                auto takeRank = [&](uint32_t r, auto* meshes, uint32_t first, uint32_t sortedBufferIndex) {
                    Stopwatch watch(tickSeconds.records);
                    const GvResult& res = results[r];
                    const auto& share = rankShares.shares[r].meshes[p];
                    for (uint32_t k = 0; k < res.draw_count; k++) {
                        auto& m = meshes[first + k];
                        m.componentOffset = (size_t)share.worldSlot[res.visible_idx[k]] * componentSize;  // mesh.cpp:170, in WORLD slots
                        memcpy(m.bakedModel.m, res.baked_model + (size_t)k * 12, 48);                    // mesh.cpp:171
                        m.distanceSq = res.distance_sq[k];                                               // mesh.cpp:172 / :249-251
                        if constexpr (std::is_same<std::remove_reference_t<decltype(m)>, SortedMesh>::value)
                            m.bufferIndex = sortedBufferIndex;                                           // mesh.cpp:252
                    }
                    (void)sortedBufferIndex;
                };
                if (sorted) {
                    const bool ui = renderType == MeshRenderType::UI;
                    auto& combined = pass >= 0 ? shadowTransMeshes[pass] : ui ? uiSortedMeshes : transSortedMeshes;
                    uint32_t& drawIndex = pass >= 0 ? shadowTransDrawIndex[pass] : ui ? uiDrawIndex : transDrawIndex;
                    auto& allRuns = pass >= 0 ? shadowTransRuns[pass] : ui ? uiRuns : transRuns;
                    MeshBuffer* counters = pass >= 0 ? static_cast<MeshBuffer*>(shadowSortedBuffers[pass][shadowIndex]) : sortedBuffers[bufferIndex];
                    if (emitRecords && combined.size() < (size_t)drawIndex + total)
                        combined.resize((size_t)drawIndex + total);
                    for (uint32_t r = 0; r < ranks && emitRecords; r++) {
                        takeRank(r, combined.data(), drawIndex, pass >= 0 ? shadowIndex : bufferIndex);
                        drawIndex += results[r].draw_count;
                        allRuns.push_back(drawIndex);
                    }
                    counters->drawCount = total;
                    counters->instanceCount = instances;
                } else {
                    UnsortedBuffer* buffer = pass >= 0 ? shadowBuffers[bufferIndex][pass] : unsortedBuffers[bufferIndex];
                    if (emitRecords && buffer->combinedMeshes.size() < total)
                        buffer->combinedMeshes.resize(total);  // grown, never shrunk (mesh.cpp:377-395)
                    uint32_t at = 0;
                    for (uint32_t r = 0; r < ranks && emitRecords; r++) {
                        takeRank(r, buffer->combinedMeshes.data(), at, 0);
                        at += results[r].draw_count;
                        runs.push_back(at);
                    }
                    buffer->drawCount = total;
                    buffer->instanceCount = instances;
                    if (emitRecords && sortOnDevice && renderType != MeshRenderType::OIT)  // the ranks' sorted runs -> one sorted list
                        for (size_t i = 1; i < runs.size(); i++)
                            std::inplace_merge(buffer->combinedMeshes.begin(), buffer->combinedMeshes.begin() + runs[i - 1], buffer->combinedMeshes.begin() + runs[i]);
                }
            }
        }
        for (uint32_t s = 0; s < passCount; s++)
            while (shadowSortedBuffers[s].size() < shadowSortedSeen)
                shadowSortedBuffers[s].push_back(new SortedBuffer());
        if (emitRecords && sortOnDevice) {
            mergeRuns(transSortedMeshes, transRuns);
            mergeRuns(uiSortedMeshes, uiRuns);
            for (uint32_t s = 0; s < passCount; s++)
                mergeRuns(shadowTransMeshes[s], shadowTransRuns[s]);
        }
    }
};

}  // namespace garden
