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
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/garden_vis.h"
#include "garden_host.hpp"

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
    };

private:
    GvCtx* ctx = nullptr;
    std::vector<IMeshRenderSystem*> meshSystems;  // prepareSystems(), mesh.cpp:69-108
    std::vector<UnsortedBuffer*> unsortedBuffers;
    std::vector<ShadowPass> shadowPasses;
    std::vector<std::vector<UnsortedBuffer*>> shadowBuffers;  // [pool][pass]
    uint64_t seenHierarchy = ~0ull, seenTransform = ~0ull, seenReparent = 0;
    std::vector<uint64_t> seenMesh;
    bool useHiz = false;

    void check(int rc, const char* what)
    {
        if (rc != GV_OK)
            throw GardenError(std::string(what) + " failed: " + gv_last_error(ctx));
    }

public:
    bool isEnabled = true;
    // true: also produce combinedMeshes records (bakedModel, distanceSq); false: isVisible + counters only
    bool emitRecords = true;
    // true: sortMeshes (mesh.cpp:265-328) runs on the device too: unsorted buffers ascending distanceSq
    // (front to back), so the engine's std::sort over combinedMeshes can be dropped
    bool sortOnDevice = true;

    explicit GpuVisibilitySystem(int device = 0, bool profile = false)
    {
        GvConfig config{};
        config.struct_size = sizeof(GvConfig);
        config.device = device;
        config.hiz_rule = GV_HIZ_RULE_REFERENCE;
        config.flags = profile ? GV_CONFIG_PROFILE_EVENTS : 0;
        if (gv_create(&config, &ctx) != GV_OK)
            throw GardenError(std::string("GpuVisibilitySystem: ") + gv_last_error(nullptr));
        ECSM_SUBSCRIBE_TO_EVENT("Init", GpuVisibilitySystem::init);
    }
    ~GpuVisibilitySystem() override
    {
        for (auto b : unsortedBuffers)
            delete b;
        for (auto& v : shadowBuffers)
            for (auto b : v)
                delete b;
        gv_destroy(ctx);
    }

    GvCtx* getContext() const noexcept { return ctx; }
    const std::vector<UnsortedBuffer*>& getUnsortedBuffers() const noexcept { return unsortedBuffers; }
    const std::vector<UnsortedBuffer*>& getShadowBuffers(uint32_t pool) const { return shadowBuffers.at(pool); }
    void setShadowPasses(std::vector<ShadowPass> passes) { shadowPasses = std::move(passes); }

    // HizRenderSystem::downsampleHiz stand-in: hand over this frame's reversed-Z depth (host memory).
    void setHizDepth(const float* depth, uint32_t width, uint32_t height)
    {
        check(gv_hiz_build(ctx, depth, width, height, GV_MEM_HOST), "gv_hiz_build");
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
    }

    void prepareSystems()  // mesh.cpp:69-108
    {
        meshSystems.clear();
        for (auto& sys : Manager::Instance::get()->getSystems())
            if (auto ms = dynamic_cast<IMeshRenderSystem*>(sys.get()))
                meshSystems.push_back(ms);
        if (meshSystems.size() > GV_MAX_POOLS)
            throw GardenError("GpuVisibilitySystem: more mesh systems than GV_MAX_POOLS");
        while (unsortedBuffers.size() < meshSystems.size())
            unsortedBuffers.push_back(new UnsortedBuffer());
        shadowBuffers.resize(meshSystems.size());
        seenMesh.resize(meshSystems.size(), ~0ull);
    }

    static GvView makeView(const f32x4x4& viewProj, f32x4 cameraPos, f32x4 cameraOffset, int8_t shadowPass,
                           bool hiz, bool emit)
    {
        GvView v{};
        memcpy(v.view_proj, viewProj.m, sizeof(v.view_proj));
        v.camera_position[0] = cameraPos.x; v.camera_position[1] = cameraPos.y; v.camera_position[2] = cameraPos.z;
        v.camera_offset[0] = cameraOffset.x; v.camera_offset[1] = cameraOffset.y; v.camera_offset[2] = cameraOffset.z;
        v.shadow_pass = shadowPass;
        v.use_hiz = hiz ? 1 : 0;
        v.emit_records = emit ? 1 : 0;
        return v;
    }

    void fill(UnsortedBuffer* buffer, IMeshRenderSystem* meshSystem, uint32_t viewIndex, bool writeBack)
    {
        GvResult r{};
        check(gv_results_fetch(ctx, viewIndex, writeBack ? 1 : 0, &r), "gv_results_fetch");
        buffer->meshSystem = meshSystem;
        buffer->drawCount = r.draw_count;
        buffer->instanceCount = r.instance_count;
        if (!emitRecords)
            return;
        if (buffer->combinedMeshes.size() < r.draw_count)
            buffer->combinedMeshes.resize(r.draw_count);  // grown, never shrunk (mesh.cpp:377-395)
        const size_t componentSize = meshSystem->getMeshComponentSize();
        auto meshes = buffer->combinedMeshes.data();
        for (uint32_t k = 0; k < r.draw_count; k++) {
            meshes[k].componentOffset = (size_t)r.visible_idx[k] * componentSize;  // mesh.cpp:170
            memcpy(meshes[k].bakedModel.m, r.baked_model + (size_t)k * 12, 48);     // mesh.cpp:171
            meshes[k].distanceSq = r.distance_sq[k];                                // mesh.cpp:172
        }
    }

    // preForwardRender / preDeferredRender, mesh.cpp:860-903: shadows first, then the main camera.
    void preRender()
    {
        if (!isEnabled)
            return;
        auto manager = Manager::Instance::get();
        auto transformSystem = TransformSystem::Instance::get();
        auto graphicsSystem = GraphicsSystem::Instance::get();
        prepareSystems();

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
            if (seenTransform != transformSystem->transformVersion) {
                check(gv_mark_dirty(ctx, GV_DIRTY_TRANSFORM, 0, pool.getOccupancy()), "gv_mark_dirty");
                seenTransform = transformSystem->transformVersion;
            }
        }
        seenReparent = transformSystem->reparentVersion;
        transformSystem->clearReparentRange();

        const auto& cc = graphicsSystem->getCommonConstants();
        for (uint32_t p = 0; p < meshSystems.size(); p++) {
            auto meshSystem = meshSystems[p];
            check(gv_pool_bind(ctx, p, meshSystem->getMeshComponentData(), meshSystem->getMeshComponentSize(),
                               meshSystem->getMeshComponentOccupancy(), &meshLayout), "gv_pool_bind");
            if (auto versioned = dynamic_cast<OpaqueMeshSystem*>(meshSystem)) {
                if (seenMesh[p] != versioned->meshVersion) {
                    check(gv_mark_dirty(ctx, GV_DIRTY_MESH, p << 28, meshSystem->getMeshComponentOccupancy()), "gv_mark_dirty");
                    seenMesh[p] = versioned->meshVersion;
                }
            }
            // view 0 = main camera (shadowPass -1: writes isVisible), views 1.. = shadow passes (mesh.cpp:809-843)
            std::vector<GvView> views;
            views.push_back(makeView(cc.viewProj, cc.cameraPos, f32x4(), -1, useHiz, emitRecords));
            for (size_t s = 0; s < shadowPasses.size() && views.size() < GV_MAX_VIEWS; s++)
                views.push_back(makeView(shadowPasses[s].viewProj, cc.cameraPos, shadowPasses[s].cameraOffset,
                                         (int8_t)s, false, emitRecords));
            check(gv_cull(ctx, p, views.data(), (uint32_t)views.size()), "gv_cull");
            if (emitRecords && sortOnDevice)
                for (uint32_t v = 0; v < views.size(); v++)
                    check(gv_sort(ctx, v, 0), "gv_sort");  // opaque/unsorted: operator< (render/mesh.hpp:196)
            auto& sb = shadowBuffers[p];
            while (sb.size() + 1 < views.size())
                sb.push_back(new UnsortedBuffer());
            for (uint32_t v = 1; v < views.size(); v++)
                fill(sb[v - 1], meshSystem, v, false);
            fill(unsortedBuffers[p], meshSystem, 0, true);
        }
        (void)manager;
    }
};

}  // namespace garden
