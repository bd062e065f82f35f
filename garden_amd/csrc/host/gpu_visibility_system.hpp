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
    struct RanksSeen {  // what the shares have followed so far (unitemised changes — hierarchyVersion, another set of systems — deal again)
        uint64_t hierarchy = ~0ull, reparent = ~0ull;
        std::vector<IMeshRenderSystem*> meshSystems;
        std::vector<uint64_t> meshVersion;  // VersionedMeshSystem::meshVersion ("the whole pool may have changed": compared slot by slot)
    } ranksSeen;
    RankShares::Changes rankChanges;
    bool exchangeModeChosen = false;
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
        if (rc == GV_OK)
            return;
        // a call over all ranks (gv_exchange_*_all) leaves its text on the context of the rank that failed
        const char* text = gv_last_error(ctx);
        for (auto c : contexts)
            if (text[0] == 0 && gv_last_error(c)[0] != 0)
                text = gv_last_error(c);
        throw GardenError(std::string(what) + " failed: " + text);
    }

public:
    // host wall time of the prepare phase, accumulated over ticks ("Meshes Prepare" zone of the reference, by step)
    struct TickSeconds {
        double total = 0, cull = 0, sort = 0, fetch = 0, records = 0, share = 0, gather = 0;  // records: filling combinedMeshes from the fetch; share / gather: several ranks
    } tickSeconds;
    struct RankCounters {  // several ranks: what keeping the shares current took since the system was made
        uint64_t frames = 0, deals = 0, exchanges = 0, movedTrees = 0, movedTransforms = 0, editedMeshes = 0, copiedTransforms = 0;
    } rankCounters;
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

    // Multi-GPU mode: called ONCE per frame, after the frame's one exchange (gv_exchange_views_all) has been acquired — list i of the
    // frame is (mesh system lists[i].meshSystemIndex, pass lists[i].shadowPass: -1 the light pass); frames[r] is rank r's acquired
    // GvExchangeFrame: on every device, every rank's complete lists of WORLD mesh slots (row layout: include/garden_vis.h,
    // gv_exchange_views). A GPU-driven renderer enqueues its per-device work here ON gv_stream(getContext(r)) — the acquire has ordered
    // that stream behind the rows, and the rows stay valid until the exchange after the next, i.e. through the next frame; a
    // consumer on a stream of its own waits for frames[r].ready_event and must be done before the frame after the next is sent.
    // The host-side buffers are filled afterwards.
    struct GatheredList {
        uint32_t meshSystemIndex;
        int8_t shadowPass;
    };
    std::function<void(const GatheredList* lists, uint32_t listCount, const GvExchangeFrame* frames, uint32_t ranks)> onGathered;
    // Several ranks: time each travel pattern of the exchange (all-gather / send-recv pairs / one broadcast per root) over the first
    // frames and keep the fastest (SURVEY.md §8e argues for the direct patterns on a fully connected node; only a node can tell).
    bool probeExchangeMode = false;
    // Several ranks: how the lists travel between the devices of this one process. Auto: direct stores into every rank's rows
    // (gv_exchange_init_peers — no communicator, nothing predicted, nothing to probe) where every device reaches every other, a
    // communicator (gv_exchange_init_all: RCCL) otherwise. Set before the "Init" event.
    enum class ExchangeTransport { Auto, Peers, Communicator } exchangeTransport = ExchangeTransport::Auto;
    // Several ranks: roots whose position has crossed into a cell of another rank take their trees there (rank_shares.hpp moveTree)
    bool rebinMovedRoots = true;

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
        // the contexts first: gv_destroy synchronises the stream and lets every record target go: no target outlives its array
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
        if (contexts.size() > 1) {
            // one thread, N ranks: peer stores between the devices of this process, or a communicator made inside one group
            int rc = exchangeTransport == ExchangeTransport::Communicator ? GV_E_STATE : gv_exchange_init_peers(contexts.data(), (int)contexts.size());
            if (rc != GV_OK && (exchangeTransport == ExchangeTransport::Peers || rc != GV_E_STATE))
                check(rc, "gv_exchange_init_peers");
            if (rc == GV_OK)
                exchangeMode = GV_EXCHANGE_PEER;
            else
                check(gv_exchange_init_all(contexts.data(), (int)contexts.size()), "gv_exchange_init_all");
        }
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

    // ---- what a frame culls ----
    // One entry per mesh system, in meshSystems order: prepareMeshes' classification (mesh.cpp:341-375, :414), the bufferIndex each
    // system takes (:416-421), the gate pass by pass (:426 / :482) and, behind it, the views that are culled. BOTH frame paths (one
    // context, several ranks) are driven by this table.
    struct SystemPlan {
        IMeshRenderSystem* meshSystem = nullptr;
        MeshRenderType type = MeshRenderType::Opaque;
        bool sorted = false;
        uint32_t bufferIndex = 0;   // in the light pass: Translucent and UI systems count (mesh.cpp:419)
        uint32_t shadowIndex = 0;   // in a shadow pass: a UI system leaves the loop before it takes one (:416-417)
        uint32_t occupancy = 0;
        std::vector<GvView> views;   // the passes that get through the gate, the light pass first
        std::vector<int8_t> passes;  // views[v] stands for: -1 the light pass, s >= 0 shadow pass s (position in shadowPasses)
    };
    std::vector<SystemPlan> plan;
    uint32_t planPassCount = 0;

    void classifyAndGate()
    {
        const auto& cc = GraphicsSystem::Instance::get()->getCommonConstants();
        const uint32_t passCount = planPassCount = (uint32_t)std::min<size_t>(shadowPasses.size(), GV_MAX_VIEWS - 1);
        transDrawIndex = uiDrawIndex = 0;                  // mesh.cpp:337
        hasAnyRefr = hasAnyOIT = hasAnyTD = false;         // mesh.cpp:339
        shadowTransMeshes.resize(passCount);
        shadowTransDrawIndex.assign(passCount, 0);
        shadowSortedBuffers.resize(passCount);
        plan.assign(meshSystems.size(), SystemPlan{});
        uint32_t sortedSeen = 0, shadowSortedSeen = 0, unsortedSeen = 0;
        for (uint32_t p = 0; p < meshSystems.size(); p++) {
            SystemPlan& sp = plan[p];
            auto meshSystem = sp.meshSystem = meshSystems[p];
            const auto renderType = sp.type = meshSystem->getMeshRenderType();
            const auto& componentPool = meshSystem->getMeshComponentPool();        // mesh.cpp:410
            const uint32_t componentCount = componentPool.getCount();              // mesh.cpp:411
            sp.occupancy = componentPool.getOccupancy();
            sp.sorted = isSortedType(renderType);
            // bufferIndex: in the light pass Translucent and UI systems count (mesh.cpp:419); in a shadow pass a UI system leaves the
            // loop before it takes one (:416-417), so a Translucent system's index there counts Translucent systems only
            if (sp.sorted) {
                sp.bufferIndex = sortedSeen++;
                if (renderType == MeshRenderType::Translucent)
                    sp.shadowIndex = shadowSortedSeen++;
            } else {
                sp.bufferIndex = sp.shadowIndex = unsortedSeen++;
                auto& sb = shadowBuffers[sp.bufferIndex];
                while (sb.size() < passCount)
                    sb.push_back(new UnsortedBuffer());
            }
            // The gate of mesh.cpp:426 / :482, pass by pass: `componentCount == 0 || !meshSystem->isDrawReady(shadowPass)` leaves the
            // system's counters at 0 for that pass and touches nothing else — in the light pass isVisible keeps last frame's bytes.
            // The light pass (shadowPass -1: writes isVisible), then the shadow passes (mesh.cpp:809-843). UI: its own ortho
            // frustum, camera at the origin, 2D distance key, no shadow passes (mesh.cpp:416,436-442).
            const bool lightPass = componentCount != 0 && meshSystem->isDrawReady(-1);
            if (lightPass) {
                if (renderType == MeshRenderType::UI)
                    sp.views.push_back(makeView(uiViewProj, f32x4(), f32x4(), -1, false, emitRecords, true));
                else
                    sp.views.push_back(makeView(cc.viewProj, cc.cameraPos, f32x4(), -1, useHiz && isHizEnabled, emitRecords));
                sp.passes.push_back(-1);
                if (!sp.sorted) {  // mesh.cpp:488-490
                    hasAnyRefr |= renderType == MeshRenderType::Refracted;
                    hasAnyOIT |= renderType == MeshRenderType::OIT;
                    hasAnyTD |= renderType == MeshRenderType::TransDepth;
                }
            }
            if (renderType != MeshRenderType::UI)
                for (uint32_t s = 0; s < passCount; s++)
                    if (componentCount != 0 && meshSystem->isDrawReady(shadowPasses[s].index(s))) {
                        sp.views.push_back(makeView(shadowPasses[s].viewProj, cc.cameraPos, shadowPasses[s].cameraOffset, shadowPasses[s].index(s), false, emitRecords));
                        sp.passes.push_back((int8_t)s);
                    }
            // (no view: no pass draws this system this frame — nothing is culled, sorted or read for it: `continue`, mesh.cpp:427 / :483)
        }
        for (uint32_t s = 0; s < passCount; s++)
            while (shadowSortedBuffers[s].size() < shadowSortedSeen)
                shadowSortedBuffers[s].push_back(new SortedBuffer());
        // Every buffer of every pass in the state mesh.cpp:420-424 / :476-480 leave it in — its system, both counters 0; only the
        // passes that are culled fill theirs.
        auto reset = [](MeshBuffer* buffer, IMeshRenderSystem* meshSystem) {
            buffer->meshSystem = meshSystem;
            buffer->drawCount = 0;
            buffer->instanceCount = 0;
        };
        for (const SystemPlan& sp : plan) {
            if (sp.sorted) {
                reset(sortedBuffers[sp.bufferIndex], sp.meshSystem);
                if (sp.type == MeshRenderType::Translucent)
                    for (uint32_t s = 0; s < passCount; s++)
                        reset(shadowSortedBuffers[s][sp.shadowIndex], sp.meshSystem);
            } else {
                reset(unsortedBuffers[sp.bufferIndex], sp.meshSystem);
                unsortedBuffers[sp.bufferIndex]->span = nullptr;
                for (uint32_t s = 0; s < passCount; s++) {
                    reset(shadowBuffers[sp.bufferIndex][s], sp.meshSystem);
                    shadowBuffers[sp.bufferIndex][s]->span = nullptr;
                }
            }
        }
    }

    // the engine's record struct of system p as the library's layout; false: not expressible (the three-array fetch is kept)
    bool recordLayoutOfSystem(const SystemPlan& sp, GvRecordLayout& layout) const
    {
        const size_t componentSize = sp.meshSystem->getMeshComponentSize();
        const bool expressible = sp.sorted ? recordLayoutOf<SortedMesh>(layout, componentSize, (uint32_t)offsetof(SortedMesh, bufferIndex), sp.bufferIndex)
                                           : recordLayoutOf<UnsortedMesh>(layout, componentSize, GV_NONE, 0);
        return emitRecords && recordStructs && expressible;
    }

    static const GvTransformLayout& transformLayout()
    {
        static const GvTransformLayout layout = {
            (uint32_t)offsetof(TransformComponent, entity), (uint32_t)offsetof(TransformComponent, parent),
            (uint32_t)offsetof(TransformComponent, posChildCount), (uint32_t)offsetof(TransformComponent, scaleChildCap),
            (uint32_t)offsetof(TransformComponent, rotation), (uint32_t)offsetof(TransformComponent, selfActive),
            (uint32_t)offsetof(TransformComponent, ancestorsActive),
            (uint32_t)offsetof(TransformComponent, modelWithAncestors)};
        return layout;
    }
    static const GvMeshLayout& meshLayout()
    {
        static const GvMeshLayout layout = {
            (uint32_t)offsetof(MeshRenderComponent, entity), (uint32_t)offsetof(MeshRenderComponent, isEnabled),
            (uint32_t)offsetof(MeshRenderComponent, isVisible), (uint32_t)offsetof(MeshRenderComponent, aabb.min),
            (uint32_t)offsetof(MeshRenderComponent, aabb.max)};
        return layout;
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
        Stopwatch whole(tickSeconds.total);
        prepareSystems();
        bool sweepRequested = false;

        // Pools may have moved (create() can reallocate): re-bind every frame, as `gv_pool_bind` documents.
        auto& pool = transformSystem->getComponents();
        auto& entityMap = transformSystem->getEntityMap();
        check(gv_transform_bind(ctx, pool.getData(), sizeof(TransformComponent), pool.getOccupancy(), &transformLayout(),
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

        classifyAndGate();
        const uint32_t passCount = planPassCount;
        std::vector<uint32_t> transRuns, uiRuns;
        std::vector<std::vector<uint32_t>> shadowTransRuns(passCount);

        // Phase 1 — every mesh system's cull (and sort request) is issued before any result is read: results are kept
        // per (pool, view), so the device works through the systems back to back while the host only enqueues; the
        // reference dispatches every system's tasks to its thread pool and waits once, too (mesh.cpp:408-546, :548).
        check(gv_cull_batch_begin(ctx), "gv_cull_batch_begin");  // engine-sized pools: one cull / emit / sort / publish launch per tick
        for (uint32_t p = 0; p < meshSystems.size(); p++) {
            const SystemPlan& sp = plan[p];
            auto meshSystem = sp.meshSystem;
            const uint32_t occupancy = sp.occupancy;
            // the pool is bound and its changes are taken over whether or not the system is drawn this frame: a system that
            // becomes ready later is culled from the pool as it is then
            check(gv_pool_bind(ctx, p, meshSystem->getMeshComponentPool().getData(), meshSystem->getMeshComponentSize(), occupancy, &meshLayout()), "gv_pool_bind");
            GvRecordLayout layout;
            const bool structs = recordLayoutOfSystem(sp, layout);
            check(gv_pool_set_record_layout(ctx, p, structs ? &layout : nullptr), "gv_pool_set_record_layout");
            const bool inPlace = !sp.sorted && structs && recordTargets;
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
            if (!sp.sorted) {
                // An unsorted system's buffers are its own (mesh.hpp:213-217), sized before its tasks run like the
                // reference's scratch (mesh.cpp:377-395; here to the occupancy, which bounds any draw count): the device
                // writes the records where renderUnsorted reads them. Shared sorted arrays keep the append + merge.
                auto& sb = shadowBuffers[sp.bufferIndex];
                for (uint32_t v = 0; v < sp.views.size(); v++) {
                    UnsortedBuffer* buffer = sp.passes[v] < 0 ? unsortedBuffers[sp.bufferIndex] : sb[sp.passes[v]];
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
            if (sp.views.empty())
                continue;  // no pass draws this system this frame: nothing is culled, sorted or read for it
            if (sweepWorldMatrices && !sweepRequested) {
                check(gv_sweep(ctx, sweepIncremental ? GV_SWEEP_INCREMENTAL : GV_SWEEP_WITH_CULL), "gv_sweep");
                sweepRequested = true;
            }
            {
                Stopwatch watch(tickSeconds.cull);
                check(gv_cull(ctx, p, sp.views.data(), (uint32_t)sp.views.size()), "gv_cull");
            }
            // sortMeshes, mesh.cpp:270-295: unsorted buffers front to back (UnsortedMesh::operator<, mesh.hpp:196), OIT is
            // not sorted; sorted systems back to front (SortedMesh::operator<, mesh.hpp:204)
            if (emitRecords && sortOnDevice && sp.type != MeshRenderType::OIT) {
                Stopwatch watch(tickSeconds.sort);
                for (uint32_t v = 0; v < sp.views.size(); v++)
                    check(gv_pool_sort(ctx, p, v, sp.sorted ? 1 : 0), "gv_pool_sort");
            }
        }
        if (sweepWorldMatrices && !sweepRequested)  // no system is drawn this frame: the cache is kept current all the same
            check(gv_sweep(ctx, sweepIncremental ? GV_SWEEP_INCREMENTAL : GV_SWEEP_VALU), "gv_sweep");

        // Phase 2 — read the results (the first fetch publishes every small pool's views at once).
        for (uint32_t p = 0; p < meshSystems.size(); p++) {
            const SystemPlan& sp = plan[p];
            auto meshSystem = sp.meshSystem;
            const auto& passes = sp.passes;
            if (sp.sorted) {
                const uint32_t bufferIndex = sp.bufferIndex, shadowIndex = sp.shadowIndex;
                for (uint32_t v = 0; v < passes.size(); v++) {
                    if (passes[v] >= 0) {
                        const uint32_t s = (uint32_t)passes[v], first = shadowTransDrawIndex[s];
                        const uint32_t added = append(shadowTransMeshes[s], shadowTransDrawIndex[s], shadowSortedBuffers[s][shadowIndex], meshSystem, p, v,
                                                      false, shadowIndex);
                        if (shadowIndex != bufferIndex)  // (records built on the device carry the light pass's index)
                            for (uint32_t k = 0; k < added; k++)
                                shadowTransMeshes[s][first + k].bufferIndex = shadowIndex;
                        shadowTransRuns[s].push_back(shadowTransDrawIndex[s]);
                    } else if (sp.type == MeshRenderType::UI) {
                        append(uiSortedMeshes, uiDrawIndex, sortedBuffers[bufferIndex], meshSystem, p, v, true, bufferIndex);
                        uiRuns.push_back(uiDrawIndex);
                    } else {
                        append(transSortedMeshes, transDrawIndex, sortedBuffers[bufferIndex], meshSystem, p, v, true, bufferIndex);
                        transRuns.push_back(transDrawIndex);
                    }
                }
            } else {
                auto& sb = shadowBuffers[sp.bufferIndex];
                for (uint32_t v = 0; v < passes.size(); v++)
                    if (passes[v] >= 0)
                        fill(sb[passes[v]], meshSystem, p, v, false);
                for (uint32_t v = 0; v < passes.size(); v++)
                    if (passes[v] < 0)
                        fill(unsortedBuffers[sp.bufferIndex], meshSystem, p, v, true);
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

    // One gathered list of the frame into its place: the ranks' runs as the engine's structs (or built here from the three arrays, through
    // the share's slot table, for a struct the library cannot express), merged; mesh.cpp:252 for a shadow pass's sorted records.
    template <class Mesh, class List>
    void mergeList(Mesh* dst, List& g, const SystemPlan& sp, bool structs, bool ordered, uint32_t ranks)
    {
        const size_t componentSize = sp.meshSystem->getMeshComponentSize();
        std::vector<std::vector<uint8_t>> built;
        std::vector<const Mesh*> typed(ranks);
        for (uint32_t r = 0; r < ranks; r++) {
            if (!structs && g.counts[r]) {  // three arrays, the rank's own slots: through the share's table (mesh.cpp:170-172)
                built.resize(ranks);
                built[r].resize((size_t)g.counts[r] * sizeof(Mesh));
                Mesh* meshes = reinterpret_cast<Mesh*>(built[r].data());
                const auto& worldSlot = rankShares.shares[r].meshes[g.p].worldSlot;
                for (uint32_t k = 0; k < g.counts[r]; k++) {
                    new (&meshes[k]) Mesh();
                    meshes[k].componentOffset = (size_t)worldSlot[g.results[r].visible_idx[k]] * componentSize;
                    memcpy(meshes[k].bakedModel.m, g.results[r].baked_model + (size_t)k * 12, 48);
                    meshes[k].distanceSq = g.results[r].distance_sq[k];
                }
                g.runs[r] = meshes;
            }
            typed[r] = static_cast<const Mesh*>(g.runs[r]);
        }
        mergeRanks(dst, typed.data(), g.counts.data(), ranks, ordered);
        if constexpr (std::is_same<Mesh, SortedMesh>::value) {
            const uint32_t sortedIndex = g.pass >= 0 ? sp.shadowIndex : sp.bufferIndex;
            if (!structs || sortedIndex != sp.bufferIndex)  // mesh.cpp:252 (records built on the device carry the light pass's index)
                for (uint32_t k = 0; k < g.total; k++)
                    dst[k].bufferIndex = sortedIndex;
        }
    }

    // Brings every rank's share up to date with the engine's pools and tells the ranks what changed (rank_shares.hpp). The pools are
    // dealt ONCE; entities and components that come or go, parent links that move, edits and moves are followed slot by slot; a mesh
    // system that cannot say what changed (the reference's have no counters) is compared with the ranks' copies. Dealt again only
    // for changes nobody itemised (hierarchyVersion), another set of mesh systems, or something followEntities cannot follow.
    void syncRanks(TransformSystem* transformSystem)
    {
        const uint32_t ranks = (uint32_t)contexts.size();
        auto& pool = transformSystem->getComponents();
        // dealt: once, and again only for changes nobody itemised (hierarchyVersion) or another set of mesh systems
        bool structural = rankShares.shares.size() != ranks || ranksSeen.hierarchy != transformSystem->hierarchyVersion || ranksSeen.meshSystems != meshSystems;
        ranksSeen.meshVersion.resize(meshSystems.size(), ~0ull);
        rankChanges.reset(ranks, meshSystems.size());
        std::vector<uint32_t> moved;
        bool everything = false;
        try {
        if (!structural) {
            // Mesh components: the slots a system names, or — no counters, or "the whole pool may have changed" — every slot, compared
            // with the ranks' copies in the bytes the cull reads; a slot that holds another entity than the shares know is listed
            std::vector<RankShares::MeshPiece> pieces;
            std::vector<std::vector<uint32_t>> changedHands(meshSystems.size());
            for (size_t p = 0; p < meshSystems.size(); p++) {
                auto versioned = dynamic_cast<VersionedMeshSystem*>(meshSystems[p]);
                const uint32_t occupancy = meshSystems[p]->getMeshComponentPool().getOccupancy();
                uint32_t lo = 0, hi = occupancy;
                if (versioned && versioned->reportsChanges && ranksSeen.meshVersion[p] == versioned->meshVersion) {
                    lo = std::min(versioned->meshLo, versioned->meshHi);
                    hi = std::min(versioned->meshHi, occupancy);
                }
                pieces.push_back(RankShares::MeshPiece{(uint32_t)p, lo, hi, meshSystems[p]});
                for (uint32_t j = (uint32_t)rankShares.meshTables[p].entity.size(); j < occupancy; j++)
                    changedHands[p].push_back(j);  // slots the pool has grown by
            }
            (void)rankShares.syncMeshes(pieces, rankChanges, &changedHands);
            // Entities and components that came or went, parent links that moved: followed slot by slot (rank_shares.hpp followEntities)
            std::vector<std::pair<uint32_t, uint32_t>> transformSlots;  // [first, end)
            if (seenFlags != transformSystem->flagsVersion && transformSystem->flagsLo < transformSystem->flagsHi)  // (entities created / destroyed, setActive)
                transformSlots.push_back({transformSystem->flagsLo, transformSystem->flagsHi});
            const bool relinked = ranksSeen.reparent != transformSystem->reparentVersion && transformSystem->reparentLo < transformSystem->reparentHi;
            if (relinked)
                transformSlots.push_back({transformSystem->reparentLo, transformSystem->reparentHi});
            if (rankShares.rankOfTransform.size() < pool.getOccupancy())
                transformSlots.push_back({(uint32_t)rankShares.rankOfTransform.size(), pool.getOccupancy()});
            bool anything = !transformSlots.empty();
            for (const auto& slots : changedHands)
                anything = anything || !slots.empty();
            if (anything)
                structural = !rankShares.followEntities(transformSystem, meshSystems, ranks, rankGrid, worldSide, std::move(transformSlots), changedHands,
                                                        relinked ? transformSystem->reparentLo : 0u, relinked ? transformSystem->reparentHi : 0u, rankChanges);
        }
        if (!structural) {  // transforms: every slot, or the itemised ones; roots that crossed into another rank's cell take their trees along
            if (seenTransform != transformSystem->transformVersion) {
                everything = true;
                rankShares.syncAllTransforms(transformSystem, rankChanges);
                rankCounters.copiedTransforms += pool.getOccupancy();
            } else {
                for (const auto& range : transformSystem->movedRanges)
                    for (uint32_t i = 0; i < range.second; i++) {
                        rankShares.syncTransform(transformSystem, range.first + i, rankChanges);
                        moved.push_back(range.first + i);
                    }
                rankCounters.copiedTransforms += moved.size();
            }
            if (rebinMovedRoots && (everything || !moved.empty()))
                rankShares.rebin(transformSystem, moved, everything, ranks, rankGrid, worldSide, rankChanges);
        }
        } catch (const std::runtime_error&) {
            structural = true;  // a state the slot-by-slot steps cannot carry over (a link across ranks nobody announced): dealt again, never half-followed
        }
        if (structural) {
            rankShares.deal(transformSystem, meshSystems, ranks, rankGrid, worldSide);
            rankChanges.reset(ranks, meshSystems.size());
            rankCounters.deals++;
        }
        rankCounters.movedTrees += rankChanges.movedTrees;
        rankCounters.movedTransforms += rankChanges.movedTransforms;
        ranksSeen.hierarchy = transformSystem->hierarchyVersion;
        ranksSeen.reparent = transformSystem->reparentVersion;
        ranksSeen.meshSystems = meshSystems;
        for (size_t p = 0; p < meshSystems.size(); p++) {
            if (auto versioned = dynamic_cast<VersionedMeshSystem*>(meshSystems[p])) {
                ranksSeen.meshVersion[p] = versioned->meshVersion;
                versioned->clearMeshRange();
            }
        }
        seenTransform = transformSystem->transformVersion;
        seenFlags = transformSystem->flagsVersion;
        transformSystem->clearReparentRange();
        transformSystem->clearFlagsRange();
        transformSystem->clearMovedRanges();
        // bind every rank's share (vectors may have grown: slots appended by a tree that moved in) and tell the rank what changed
        for (uint32_t r = 0; r < ranks; r++) {
            auto& share = rankShares.shares[r];
            auto& changed = rankChanges.ranks[r];
            checkRank(r, gv_transform_bind(contexts[r], share.transforms.data(), sizeof(TransformComponent), (uint32_t)share.transforms.size(), &transformLayout(),
                                           share.entityToTransform.data(), (uint32_t)share.entityToTransform.size()), "gv_transform_bind");
            for (uint32_t p = 0; p < meshSystems.size(); p++) {
                auto& mesh = share.meshes[p];
                const auto& worldPool = meshSystems[p]->getMeshComponentPool();
                checkRank(r, gv_pool_bind(contexts[r], p, mesh.components.data(), (uint32_t)mesh.stride, mesh.occupancy(), &meshLayout()), "gv_pool_bind");
                // the rank's results in the WORLD's slots: isVisible straight into the engine's pool (mesh.cpp:144-166)
                uint8_t* worldVisible = reinterpret_cast<uint8_t*>(worldPool.getData()) + offsetof(MeshRenderComponent, isVisible);
                rankMapping(r, p, worldVisible, mesh.stride, worldPool.getOccupancy());
                if (structural) {
                    checkRank(r, gv_pool_set_index_map(contexts[r], p, mesh.worldSlot.data(), mesh.occupancy()), "gv_pool_set_index_map");
                    checkRank(r, gv_mark_dirty(contexts[r], GV_DIRTY_MESH, p << 28, mesh.occupancy()), "gv_mark_dirty");
                    continue;
                }
                auto mapRuns = RankShares::Changes::runs(changed.maps[p]);
                if (mapRuns.size() > 4)  // (every call ends in a synchronisation: many scattered entries travel as the one span that holds them)
                    mapRuns = {{mapRuns.front().first, mapRuns.back().first + mapRuns.back().second - mapRuns.front().first}};
                for (const auto& run : mapRuns)
                    checkRank(r, gv_pool_update_index_map(contexts[r], p, run.first, mesh.worldSlot.data() + run.first, run.second), "gv_pool_update_index_map");
                rankCounters.editedMeshes += changed.meshes[p].size();
                for (const auto& run : RankShares::Changes::runs(changed.meshes[p]))
                    checkRank(r, gv_mark_dirty(contexts[r], GV_DIRTY_MESH, (p << 28) | run.first, run.second), "gv_mark_dirty");
            }
            if (structural) {
                checkRank(r, gv_mark_dirty(contexts[r], GV_DIRTY_HIERARCHY, 0, 0), "gv_mark_dirty");
                continue;
            }
            if (changed.allTransforms)
                checkRank(r, gv_mark_dirty(contexts[r], GV_DIRTY_TRANSFORM, 0, (uint32_t)share.transforms.size()), "gv_mark_dirty");
            for (const auto& run : RankShares::Changes::runs(changed.transforms))
                checkRank(r, gv_mark_dirty(contexts[r], GV_DIRTY_TRANSFORM, run.first, run.second), "gv_mark_dirty");
            for (const auto& run : RankShares::Changes::runs(changed.links))  // parent links of the slots a tree moved into
                checkRank(r, gv_mark_dirty(contexts[r], GV_DIRTY_HIERARCHY, run.first, run.second), "gv_mark_dirty");
        }
    }

    // rank r's results of pool p in the engine's own numbering (gv_pool_set_result_mapping); records: decided per frame (the layout)
    std::vector<uint32_t> rankMappingFlags;  // [rank * GV_MAX_POOLS + pool]: GV_RESULTS_MAP_RECORDS wanted
    void rankMapping(uint32_t r, uint32_t p, uint8_t* worldVisible, size_t stride, uint32_t worldOccupancy)
    {
        rankMappingFlags.resize((size_t)contexts.size() * GV_MAX_POOLS, 0u);
        checkRank(r, gv_pool_set_result_mapping(contexts[r], p, GV_RESULTS_MAP_VISIBLE | rankMappingFlags[(size_t)r * GV_MAX_POOLS + p], worldVisible, stride,
                                                worldOccupancy), "gv_pool_set_result_mapping");
    }

    // Several ranks, first frame: the same exchange by each travel pattern — one untimed frame, then five between fences — and the
    // fastest is kept (GvExchangeMode). exchangeModeProbeMs: what each took.
    void chooseExchangeMode(const std::vector<GvExchangeItem>& items, std::vector<GvExchangeFrame>& frames)
    {
        const int ranks = (int)contexts.size();
        auto fence = [&]() {
            for (uint32_t r = 0; r < contexts.size(); r++)
                checkRank(r, gv_wait(contexts[r]), "gv_wait");
        };
        uint32_t best = GV_EXCHANGE_ALLGATHER;
        for (uint32_t mode = GV_EXCHANGE_ALLGATHER; mode <= GV_EXCHANGE_BROADCAST; mode++) {
            for (uint32_t r = 0; r < contexts.size(); r++)
                checkRank(r, gv_exchange_set_mode(contexts[r], mode), "gv_exchange_set_mode");
            double seconds = 0;
            for (int rep = 0; rep < 6; rep++) {
                fence();
                const auto t0 = std::chrono::steady_clock::now();
                check(gv_exchange_views_all(contexts.data(), ranks, items.data(), (uint32_t)items.size(), 0, frames.data()), "gv_exchange_views_all");
                check(gv_exchange_acquire_all(contexts.data(), ranks, frames[0].frame, frames.data()), "gv_exchange_acquire_all");
                fence();
                if (rep > 0)
                    seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
            exchangeModeProbeMs[mode] = seconds / 5 * 1e3;
            if (exchangeModeProbeMs[mode] < exchangeModeProbeMs[best])
                best = mode;
        }
        for (uint32_t r = 0; r < contexts.size(); r++)
            checkRank(r, gv_exchange_set_mode(contexts[r], best), "gv_exchange_set_mode");
        exchangeMode = best;
    }

public:
    double exchangeModeProbeMs[3] = {0, 0, 0};  // probeExchangeMode: milliseconds per exchange by GvExchangeMode (0: not probed)
    uint32_t exchangeMode = GV_EXCHANGE_ALLGATHER;  // GV_EXCHANGE_PEER: the ranks store into each other's rows (exchangeTransport)

    // (public for tests/cpp/rank_shares_test.cpp: the merge is checked on the CPU against std::sort)
    // The ranks' runs of one list -> dst[0, total): each rank's records arrive in sortMeshes order (gv_pool_sort), so one pass that
    // always takes the smallest head (operator< of the record: ascending distanceSq for unsorted buffers, descending for sorted
    // ones, mesh.hpp:196,204) leaves the whole list in that order; not ordered (OIT, or the engine sorts itself): the runs back to back.
    // Long lists are merged on the library's worker threads: every worker takes a piece [a, b) of the OUTPUT, finds where that piece
    // begins and ends in every run (splitRuns) and merges its sub-runs — the pieces are independent.
    static uint32_t orderKey(float distanceSq, bool descending) noexcept  // monotone in the records' order
    {
        uint32_t u;
        memcpy(&u, &distanceSq, 4);
        u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
        return descending ? ~u : u;
    }
    // cuts[r]: how many records of run r lie in front of output position `position` (sum of cuts == position; records with equal keys
    // are taken run by run, so that two positions always give nested cuts)
    template <class Mesh>
    static void splitRuns(const Mesh* const* runs, const uint32_t* counts, uint32_t ranks, uint64_t position, uint32_t* cuts)
    {
        constexpr bool descending = std::is_same<Mesh, SortedMesh>::value;
        auto inFront = [&](uint32_t key, bool orEqual, uint32_t* out) {  // per run: records whose key is < key (<= key)
            uint64_t total = 0;
            for (uint32_t r = 0; r < ranks; r++) {
                uint32_t lo = 0, hi = counts[r];
                while (lo < hi) {
                    const uint32_t mid = lo + (hi - lo) / 2;
                    const uint32_t k = orderKey(runs[r][mid].distanceSq, descending);
                    if (k < key || (orEqual && k == key))
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                out[r] = lo;
                total += lo;
            }
            return total;
        };
        uint32_t scratch[GV_EXCHANGE_MAX_RANKS];
        uint32_t lo = 0, hi = 0xFFFFFFFFu;  // the smallest key with at least `position` records at or in front of it
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (inFront(mid, true, scratch) >= position)
                hi = mid;
            else
                lo = mid + 1;
        }
        uint32_t upTo[GV_EXCHANGE_MAX_RANKS];
        inFront(lo, true, upTo);
        uint64_t left = position - std::min<uint64_t>(position, inFront(lo, false, cuts));
        for (uint32_t r = 0; r < ranks; r++) {  // the records with exactly that key: run by run
            const uint32_t take = (uint32_t)std::min<uint64_t>(left, upTo[r] - cuts[r]);
            cuts[r] += take;
            left -= take;
        }
    }
    template <class Mesh>
    static void mergePiece(Mesh* dst, const Mesh* const* runs, const uint32_t* from, const uint32_t* to, uint32_t ranks)
    {
        uint32_t at[GV_EXCHANGE_MAX_RANKS], live[GV_EXCHANGE_MAX_RANKS], n = 0;
        for (uint32_t r = 0; r < ranks; r++) {
            at[r] = from[r];
            if (from[r] < to[r])
                live[n++] = r;
        }
        while (n > 1) {
            uint32_t best = 0;
            for (uint32_t k = 1; k < n; k++)
                if (runs[live[k]][at[live[k]]] < runs[live[best]][at[live[best]]])
                    best = k;
            const uint32_t r = live[best];
            memcpy(static_cast<void*>(dst++), static_cast<const void*>(runs[r] + at[r]), sizeof(Mesh));
            if (++at[r] == to[r])
                live[best] = live[--n];
        }
        if (n == 1)
            memcpy(static_cast<void*>(dst), static_cast<const void*>(runs[live[0]] + at[live[0]]), (size_t)(to[live[0]] - at[live[0]]) * sizeof(Mesh));
    }
    template <class Mesh>
    static void mergeRanks(Mesh* dst, const Mesh* const* runs, const uint32_t* counts, uint32_t ranks, bool ordered)
    {
        uint64_t total = 0;
        for (uint32_t r = 0; r < ranks; r++)
            total += counts[r];
        if (!ordered) {
            for (uint32_t r = 0; r < ranks; r++) {
                memcpy(static_cast<void*>(dst), static_cast<const void*>(runs[r]), (size_t)counts[r] * sizeof(Mesh));
                dst += counts[r];
            }
            return;
        }
        struct Job {
            Mesh* dst;
            const Mesh* const* runs;
            const uint32_t* counts;
            uint32_t ranks;
            uint64_t total;
        } job{dst, runs, counts, ranks, total};
        gv_host_parallel_ranges(0, (uint32_t)total, [](void* user, uint32_t a, uint32_t b) {
            const Job& j = *static_cast<const Job*>(user);
            uint32_t from[GV_EXCHANGE_MAX_RANKS] = {}, to[GV_EXCHANGE_MAX_RANKS];
            if (a != 0)
                splitRuns(j.runs, j.counts, j.ranks, a, from);
            if (b == j.total)
                std::copy(j.counts, j.counts + j.ranks, to);
            else
                splitRuns(j.runs, j.counts, j.ranks, b, to);
            mergePiece(j.dst + a, j.runs, from, to, j.ranks);
        }, &job);
    }

private:
    // The frame with several ranks. The same classification, gate and buffers as preRender() (classifyAndGate); every rank culls (and
    // sorts) its share of every system, ALL the frame's lists are gathered on the devices by ONE exchange — the reference dispatches
    // every system's tasks and waits once (mesh.cpp:408-546, :548) — and the engine's buffers are filled from the ranks' results — a
    // rank holds the models of the entities it owns, the host holds all of them:
    //   isVisible      rank r's light-pass bytes go straight to the engine's pool through the share's slot -> world slot table
    //   records        arrive as the engine's structs with componentOffset in WORLD slots; each rank's run is sorted, the runs are merged
    //   counters       summed over the ranks
    void preRenderRanks()
    {
        auto transformSystem = TransformSystem::Instance::get();
        Stopwatch whole(tickSeconds.total);
        if (sweepWorldMatrices)  // (a rank keeps the world matrices of ITS entities: gv_sweep / gv_get_world on getContext(rank), in local slots)
            throw GardenError("GpuVisibilitySystem: sweepWorldMatrices is a one-context option; with several ranks ask each rank's context");
        prepareSystems();
        classifyAndGate();
        const uint32_t ranks = (uint32_t)contexts.size();
        // records as the engine's structs, in world slots, for the systems whose struct the library can express
        std::vector<GvRecordLayout> layouts(meshSystems.size());
        std::vector<uint8_t> structs(meshSystems.size(), 0);
        rankMappingFlags.assign((size_t)ranks * GV_MAX_POOLS, 0u);
        for (uint32_t p = 0; p < meshSystems.size(); p++) {
            structs[p] = recordLayoutOfSystem(plan[p], layouts[p]) ? 1 : 0;
            for (uint32_t r = 0; r < ranks; r++)
                rankMappingFlags[(size_t)r * GV_MAX_POOLS + p] = structs[p] ? GV_RESULTS_MAP_RECORDS : 0u;
        }
        {
            Stopwatch watch(tickSeconds.share);
            syncRanks(transformSystem);
        }
        rankCounters.frames++;
        const uint32_t passCount = planPassCount;
        std::vector<uint32_t> transRuns, uiRuns;
        std::vector<std::vector<uint32_t>> shadowTransRuns(passCount);
        std::vector<GvExchangeFrame> frames(ranks);
        // Phase 1 — as in preRender(): every system's cull (and sort request) goes to every rank before any result is read, so each
        // device works through the systems back to back (engine-sized pools: one launch per tick, gv_cull_batch_begin) while the host
        // only enqueues.
        for (uint32_t r = 0; r < ranks; r++)
            checkRank(r, gv_cull_batch_begin(contexts[r]), "gv_cull_batch_begin");
        try {
            for (uint32_t p = 0; p < meshSystems.size(); p++) {
                const SystemPlan& sp = plan[p];
                for (uint32_t r = 0; r < ranks; r++)
                    checkRank(r, gv_pool_set_record_layout(contexts[r], p, structs[p] ? &layouts[p] : nullptr), "gv_pool_set_record_layout");
                if (sp.views.empty())
                    continue;
                Stopwatch watch(tickSeconds.cull);
                for (uint32_t r = 0; r < ranks; r++) {
                    checkRank(r, gv_cull(contexts[r], p, sp.views.data(), (uint32_t)sp.views.size()), "gv_cull");
                    if (emitRecords && sortOnDevice && sp.type != MeshRenderType::OIT)
                        for (uint32_t v = 0; v < sp.views.size(); v++)
                            checkRank(r, gv_pool_sort(contexts[r], p, v, sp.sorted ? 1 : 0), "gv_pool_sort");
                }
            }
        } catch (...) {
            for (auto c : contexts)  // no rank is left recording: the next frame starts from a clean batch
                (void)gv_cull_batch_end(c);
            throw;
        }
        // Phase 2 — the gather (mesh.cpp:177-183: every worker's records into the shared array): ONE exchange carries every list of the
        // frame; on every device, every rank's lists of WORLD slots, complete (gv_exchange_acquire_all).
        if (emitRecords) {
            std::vector<GvExchangeItem> items;
            std::vector<GatheredList> lists;
            for (uint32_t p = 0; p < meshSystems.size(); p++)
                for (uint32_t v = 0; v < plan[p].passes.size(); v++) {
                    items.push_back(GvExchangeItem{p, v, 0u});
                    lists.push_back(GatheredList{p, plan[p].passes[v]});
                }
            if (!items.empty()) {
                {
                    Stopwatch watch(tickSeconds.gather);
                    if (probeExchangeMode && !exchangeModeChosen && exchangeMode != GV_EXCHANGE_PEER) {
                        chooseExchangeMode(items, frames);
                        exchangeModeChosen = true;
                    }
                    check(gv_exchange_views_all(contexts.data(), (int)ranks, items.data(), (uint32_t)items.size(), 0, frames.data()), "gv_exchange_views_all");
                    check(gv_exchange_acquire_all(contexts.data(), (int)ranks, frames[0].frame, frames.data()), "gv_exchange_acquire_all");
                    rankCounters.exchanges++;
                }
                if (onGathered)  // (the consumer's own time: not the gather's)
                    onGathered(lists.data(), (uint32_t)lists.size(), frames.data(), ranks);
            }
        }
        // Phase 3 — the engine's buffers from the ranks' results (library-owned host memory, valid until the pool's next gv_cull on that
        // rank); the light pass's fetch also writes the rank's isVisible bytes into the engine's pool. (a) every list is fetched from
        // every rank: counters, totals, each list's place in its array; (b) the arrays are sized; (c) the ranks' runs are merged into
        // place — the lists are independent: short ones side by side on the library's worker threads, long ones one after the other,
        // each in pieces on those threads.
        struct Gathered {
            uint32_t p, v, total = 0, first = 0;  // first: where the list starts in a shared sorted array
            int8_t pass;
            std::vector<GvResult> results;
            std::vector<const void*> runs;
            std::vector<uint32_t> counts;
            void* destination = nullptr;
        };
        std::vector<Gathered> lists;
        for (uint32_t p = 0; p < meshSystems.size(); p++) {
            const SystemPlan& sp = plan[p];
            for (uint32_t v = 0; v < sp.passes.size(); v++) {
                Gathered g;
                g.p = p, g.v = v, g.pass = sp.passes[v];
                g.results.resize(ranks), g.runs.assign(ranks, nullptr), g.counts.assign(ranks, 0u);
                uint32_t instances = 0;
                for (uint32_t r = 0; r < ranks; r++) {
                    Stopwatch watch(tickSeconds.fetch);
                    checkRank(r, gv_pool_results_fetch(contexts[r], p, v, g.pass < 0 ? 1 : 0, &g.results[r]), "gv_pool_results_fetch");
                    g.total += g.results[r].draw_count;
                    instances += g.results[r].instance_count;
                    g.counts[r] = emitRecords ? g.results[r].draw_count : 0u;
                    if (emitRecords && structs[p]) {
                        uint32_t n = 0;
                        checkRank(r, gv_pool_results_records(contexts[r], p, v, &g.runs[r], &n), "gv_pool_results_records");
                    }
                }
                MeshBuffer* counters;
                if (sp.sorted) {
                    const bool ui = sp.type == MeshRenderType::UI;
                    uint32_t& drawIndex = g.pass >= 0 ? shadowTransDrawIndex[g.pass] : ui ? uiDrawIndex : transDrawIndex;
                    auto& allRuns = g.pass >= 0 ? shadowTransRuns[g.pass] : ui ? uiRuns : transRuns;
                    counters = g.pass >= 0 ? static_cast<MeshBuffer*>(shadowSortedBuffers[g.pass][sp.shadowIndex]) : sortedBuffers[sp.bufferIndex];
                    if (emitRecords) {  // prepareSortedMeshes' tail (mesh.cpp:246-261): behind the records already in the shared array
                        g.first = drawIndex;
                        drawIndex += g.total;
                        allRuns.push_back(drawIndex);
                    }
                } else {
                    counters = g.pass >= 0 ? shadowBuffers[sp.bufferIndex][g.pass] : unsortedBuffers[sp.bufferIndex];
                }
                counters->drawCount = g.total;
                counters->instanceCount = instances;
                if (emitRecords && g.total)
                    lists.push_back(std::move(g));
            }
        }
        Stopwatch watch(tickSeconds.records);
        auto grown = [](auto& array, size_t records) {
            if (array.size() < records)
                array.resize(records);  // grown, never shrunk (mesh.cpp:377-395)
        };
        grown(transSortedMeshes, transDrawIndex);
        grown(uiSortedMeshes, uiDrawIndex);
        for (uint32_t s = 0; s < passCount; s++)
            grown(shadowTransMeshes[s], shadowTransDrawIndex[s]);
        for (Gathered& g : lists) {
            const SystemPlan& sp = plan[g.p];
            if (sp.sorted) {
                auto& combined = g.pass >= 0 ? shadowTransMeshes[g.pass] : sp.type == MeshRenderType::UI ? uiSortedMeshes : transSortedMeshes;
                g.destination = combined.data() + g.first;
            } else {
                UnsortedBuffer* buffer = g.pass >= 0 ? shadowBuffers[sp.bufferIndex][g.pass] : unsortedBuffers[sp.bufferIndex];
                grown(buffer->combinedMeshes, g.total);
                g.destination = buffer->combinedMeshes.data();
            }
        }
        struct Merge {
            GpuVisibilitySystem* self;
            std::vector<Gathered>* lists;
            const std::vector<uint8_t>* structs;
            uint32_t ranks;
            std::vector<uint32_t> order;  // which lists this call merges
        } merge{this, &lists, &structs, ranks, {}};
        auto mergeOne = [](void* user, uint32_t task) {
            Merge& m = *static_cast<Merge*>(user);
            Gathered& g = (*m.lists)[m.order[task]];
            const SystemPlan& sp = m.self->plan[g.p];
            const bool ordered = m.self->sortOnDevice && sp.type != MeshRenderType::OIT;
            if (sp.sorted)
                m.self->mergeList(static_cast<SortedMesh*>(g.destination), g, sp, (*m.structs)[g.p] != 0, ordered, m.ranks);
            else
                m.self->mergeList(static_cast<UnsortedMesh*>(g.destination), g, sp, (*m.structs)[g.p] != 0, ordered, m.ranks);
        };
        constexpr uint32_t kLongList = 1u << 17;  // (from here on mergeRanks itself spreads over the worker threads)
        for (uint32_t k = 0; k < lists.size(); k++)
            if (lists[k].total < kLongList)
                merge.order.push_back(k);
        gv_host_parallel_tasks((uint32_t)merge.order.size(), mergeOne, &merge);
        merge.order.clear();
        for (uint32_t k = 0; k < lists.size(); k++)
            if (lists[k].total >= kLongList)
                merge.order.push_back(k);
        for (uint32_t task = 0; task < merge.order.size(); task++)
            mergeOne(&merge, task);
        if (emitRecords && sortOnDevice) {
            mergeRuns(transSortedMeshes, transRuns);
            mergeRuns(uiSortedMeshes, uiRuns);
            for (uint32_t s = 0; s < passCount; s++)
                mergeRuns(shadowTransMeshes[s], shadowTransRuns[s]);
        }
    }
};

}  // namespace garden
