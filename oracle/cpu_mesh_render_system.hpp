// cpu_mesh_render_system.hpp — the reference's CPU prepare phase as an ecsm System, driven by the scalar oracle
// (gv_oracle.c). TEST INFRASTRUCTURE ONLY: this is BASELINE.json configs[0] ("10k entities, flat hierarchy,
// frustum-only cull on the reference CPU path, headless ecsm tick, no Vulkan") and the comparator for the GPU
// system in tests/cpp/headless_tick.cpp. Mirrors MeshRenderSystem::preDeferredRender -> prepareMeshes
// (source/system/render/mesh.cpp:893-903, :331-553) with the ThreadPool::addItems range split.
#pragma once
#include <algorithm>
#include <vector>

#include "../garden_amd/csrc/host/garden_host.hpp"
#include "gv_oracle.h"

namespace garden {

class CpuMeshRenderSystem final : public System, public Singleton<CpuMeshRenderSystem> {
public:
    struct ShadowPass {
        f32x4x4 viewProj;
        f32x4 cameraOffset;
        int8_t passIndex = -1;  // the shadow system's own number of the pass (mesh.cpp:809-815); -1: the position in the list
        int8_t index(uint32_t position) const noexcept { return passIndex >= 0 ? passIndex : (int8_t)position; }
    };

private:
    std::vector<IMeshRenderSystem*> meshSystems;
    std::vector<UnsortedBuffer*> unsortedBuffers;
    std::vector<SortedBuffer*> sortedBuffers;
    std::vector<SortedMesh> transSortedMeshes, uiSortedMeshes;
    uint32_t transDrawIndex = 0, uiDrawIndex = 0;
    uint32_t unsortedBufferCount = 0, sortedBufferCount = 0;
    std::vector<ShadowPass> shadowPasses;
    std::vector<std::vector<UnsortedBuffer*>> shadowBuffers;  // [unsorted buffer][pass]
    std::vector<std::vector<SortedMesh>> shadowTransMeshes;   // [pass]
    std::vector<uint32_t> shadowTransDrawIndex;
    std::vector<std::vector<SortedBuffer*>> shadowSortedBuffers;  // [pass][bufferIndex of that pass]
    bool hasAnyRefr = false, hasAnyOIT = false, hasAnyTD = false;  // mesh.hpp:232-234
    std::vector<uint32_t> idx;
    std::vector<float> baked, dist;
    // the build-defined per-AABB occlusion query (SURVEY.md 8a-7'; the reference has none): when a depth image has been handed
    // over, the light pass of the non-UI systems runs it after the frustum test — the same rule the drop-in applies (GvView.use_hiz)
    GvoHiz hiz{};
    std::vector<float> hizDepth, hizMips;
    bool useHiz = false;

public:
    // HizRenderSystem::downsampleHiz stand-in (hiz.cpp:104-167, hiz.frag:23-63): this frame's reversed-Z depth
    void setHizDepth(const float* depth, uint32_t width, uint32_t height)
    {
        hizDepth.assign(depth, depth + (size_t)width * height);
        const uint64_t pairs = gvo_hiz_layout(width, height, &hiz);
        hizMips.assign((size_t)pairs * 2 + 2, 0.0f);
        hiz.depth = hizDepth.data();
        hiz.mips = hizMips.data();
        gvo_hiz_build(&hiz, hizMips.data(), GVO_HIZ_RULE_REFERENCE);
        useHiz = true;
    }
    bool isEnabled = true;
    bool isNonTranslucent = false;  // mesh.hpp:275
    bool sortMeshesEnabled = true;  // mesh.cpp:548-551: prepareMeshes ends with sortMeshes()
    bool useAvx2 = false;    // 8-wide AVX2+FMA SoA path instead of the scalar AoS loop (bit-identical)
    uint32_t threads = 1;    // asyncPreparing (mesh.cpp:399): >1 fans out like ThreadPool::addItems
    f32x4x4 uiViewProj;      // calcUiProjView(), mesh.cpp:851-859 (set by the driver)

    CpuMeshRenderSystem() { ECSM_SUBSCRIBE_TO_EVENT("Init", CpuMeshRenderSystem::init); }
    ~CpuMeshRenderSystem() override
    {
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
    bool getHasAnyRefr() const noexcept { return hasAnyRefr; }
    bool getHasAnyOIT() const noexcept { return hasAnyOIT; }
    bool getHasAnyTD() const noexcept { return hasAnyTD; }
    const std::vector<SortedBuffer*>& getShadowSortedBuffers(uint32_t pass) const { return shadowSortedBuffers.at(pass); }
    const std::vector<UnsortedBuffer*>& getUnsortedBuffers() const noexcept { return unsortedBuffers; }
    uint32_t getUnsortedBufferCount() const noexcept { return unsortedBufferCount; }
    const std::vector<SortedBuffer*>& getSortedBuffers() const noexcept { return sortedBuffers; }
    uint32_t getSortedBufferCount() const noexcept { return sortedBufferCount; }
    const std::vector<SortedMesh>& getTransSortedMeshes() const noexcept { return transSortedMeshes; }
    uint32_t getTransDrawCount() const noexcept { return transDrawIndex; }
    const std::vector<SortedMesh>& getUiSortedMeshes() const noexcept { return uiSortedMeshes; }
    uint32_t getUiDrawCount() const noexcept { return uiDrawIndex; }
    const std::vector<UnsortedBuffer*>& getShadowBuffers(uint32_t unsortedBuffer) const { return shadowBuffers.at(unsortedBuffer); }
    const std::vector<SortedMesh>& getShadowTransMeshes(uint32_t pass) const { return shadowTransMeshes.at(pass); }
    uint32_t getShadowTransDrawCount(uint32_t pass) const { return shadowTransDrawIndex.at(pass); }
    void setShadowPasses(std::vector<ShadowPass> passes) { shadowPasses = std::move(passes); }

private:
    void init()
    {
        if (Manager::Instance::get()->hasEvent("PreDeferredRender"))
            ECSM_SUBSCRIBE_TO_EVENT("PreDeferredRender", CpuMeshRenderSystem::preDeferredRender);
    }

    static GvoView makeView(const f32x4x4& viewProj, f32x4 cameraPos, f32x4 cameraOffset, int8_t shadowPass, bool distance2D)
    {
        GvoView view{};
        memcpy(view.view_proj, viewProj.m, sizeof(view.view_proj));
        view.camera_position[0] = cameraPos.x; view.camera_position[1] = cameraPos.y; view.camera_position[2] = cameraPos.z;
        view.camera_offset[0] = cameraOffset.x; view.camera_offset[1] = cameraOffset.y; view.camera_offset[2] = cameraOffset.z;
        view.shadow_pass = shadowPass;
        view.distance_2d = distance2D ? 1 : 0;
        return view;
    }

    // one prepareUnsortedMeshes / prepareSortedMeshes dispatch (mesh.cpp:111-262) through the oracle: the scalar loop over
    // the AoS pools, or (useAvx2) the 8-wide AVX2+FMA loop over an SoA copy of them, rebuilt per dispatch — the reference
    // build targets -march=haswell (cmake/compile-options.cmake:34-36); both give the same bits
    GvoCullOut run(const GvoMeshPool& mp, const GvoTransformPool& tp, GvoView view)
    {
        // occlusion queries: the camera's light pass only (UI and shadow views have no depth image of their own)
        const GvoHiz* pyramid = useHiz && view.shadow_pass < 0 && !view.distance_2d ? &hiz : nullptr;
        view.use_hiz = pyramid ? 1 : 0;
        const size_t n = mp.occupancy ? mp.occupancy : 1;
        idx.resize(n); baked.resize(n * 12); dist.resize(n);
        GvoCullOut out{idx.data(), baked.data(), dist.data(), 0, 0};
        if (useAvx2) {
            GvoSoa* soa = gvo_soa_build(&mp, &tp);
            gvo_prepare_meshes_avx2(soa, &mp, &view, pyramid, threads, &out);
            gvo_soa_free(soa);
            // the AVX2 path reports records in slot order per thread range too: same contract as the scalar one
        } else {
            gvo_prepare_meshes(&mp, &tp, &view, pyramid, threads, &out);
        }
        return out;
    }
    // the tail of prepareUnsortedMeshes (mesh.cpp:168-183): the range's records behind the ones already there, both counters added
    void fill(UnsortedBuffer* buffer, const GvoCullOut& out, size_t stride)
    {
        const uint32_t drawOffset = buffer->drawCount.fetch_add(out.draw_count);
        buffer->instanceCount.fetch_add(out.instance_count);
        if (buffer->combinedMeshes.size() < (size_t)drawOffset + out.draw_count)
            buffer->combinedMeshes.resize((size_t)drawOffset + out.draw_count);
        for (uint32_t k = 0; k < out.draw_count; k++) {
            auto& m = buffer->combinedMeshes[drawOffset + k];
            m.componentOffset = (size_t)idx[k] * stride;                  // mesh.cpp:170
            memcpy(m.bakedModel.m, baked.data() + (size_t)k * 12, 48);    // mesh.cpp:171
            m.distanceSq = dist[k];                                       // mesh.cpp:172
        }
    }
    // ... of prepareSortedMeshes (mesh.cpp:246-261): into the array all systems of the kind share, tagged with bufferIndex
    void append(std::vector<SortedMesh>& combined, uint32_t& drawIndex, SortedBuffer* buffer, const GvoCullOut& out, size_t stride, uint32_t bufferIndex)
    {
        buffer->drawCount.fetch_add(out.draw_count);
        buffer->instanceCount.fetch_add(out.instance_count);
        if (combined.size() < (size_t)drawIndex + out.draw_count)
            combined.resize((size_t)drawIndex + out.draw_count);
        for (uint32_t k = 0; k < out.draw_count; k++) {
            auto& m = combined[drawIndex + k];
            m.componentOffset = (size_t)idx[k] * stride;
            memcpy(m.bakedModel.m, baked.data() + (size_t)k * 12, 48);
            m.distanceSq = dist[k];
            m.bufferIndex = bufferIndex;                                  // mesh.cpp:252
        }
        drawIndex += out.draw_count;
    }

    // MeshRenderSystem::sortMeshes, mesh.cpp:265-328
    void sortMeshes()
    {
        for (uint32_t i = 0; i < unsortedBufferCount; i++) {
            auto unsortedBuffer = unsortedBuffers[i];
            if (unsortedBuffer->meshSystem->getMeshRenderType() == MeshRenderType::OIT || unsortedBuffer->drawCount.load() == 0)
                continue;  // :273-277 "No need to sort OIT meshes at all."
            auto& meshes = unsortedBuffer->combinedMeshes;
            std::sort(meshes.begin(), meshes.begin() + unsortedBuffer->drawCount.load());
        }
        if (transDrawIndex > 0)
            std::sort(transSortedMeshes.begin(), transSortedMeshes.begin() + transDrawIndex);
        if (uiDrawIndex > 0)
            std::sort(uiSortedMeshes.begin(), uiSortedMeshes.begin() + uiDrawIndex);
    }

    // MeshRenderSystem::prepareMeshes, mesh.cpp:331-553, statement by statement (the thread pool's range split is inside
    // gvo_prepare_meshes; editor counters and debug asserts have no counterpart here). uiViewProj stands for uiFrustum.
    void prepareMeshes(const f32x4x4& viewProj, const f32x4x4* uiViewProj, f32x4 cameraOffset, int8_t shadowPass, const GvoTransformPool& tp)
    {
        uint32_t transMeshMaxCount = 0, uiMeshMaxCount = 0;                                     // :336
        transDrawIndex = 0; uiDrawIndex = 0;                                                    // :337
        unsortedBufferCount = sortedBufferCount = 0;                                            // :338
        hasAnyRefr = hasAnyOIT = hasAnyTD = false;                                              // :339
        for (auto meshSystem : meshSystems) {                                                   // :341-375
            auto renderType = meshSystem->getMeshRenderType();
            if (renderType == MeshRenderType::Translucent) {
                transMeshMaxCount += meshSystem->getMeshComponentPool().getCount();
                sortedBufferCount++;
            } else if (renderType == MeshRenderType::UI) {
                if (shadowPass < 0) {
                    uiMeshMaxCount += meshSystem->getMeshComponentPool().getCount();
                    sortedBufferCount++;
                }
            } else {
                unsortedBufferCount++;
            }
        }
        while (unsortedBuffers.size() < unsortedBufferCount)                                    // :377-384
            unsortedBuffers.push_back(new UnsortedBuffer());
        while (sortedBuffers.size() < sortedBufferCount)                                        // :385-391
            sortedBuffers.push_back(new SortedBuffer());
        if (transSortedMeshes.size() < transMeshMaxCount)                                       // :393-396
            transSortedMeshes.resize(transMeshMaxCount);
        if (uiSortedMeshes.size() < uiMeshMaxCount)
            uiSortedMeshes.resize(uiMeshMaxCount);
        const auto& cc = GraphicsSystem::Instance::get()->getCommonConstants();                 // :401
        const f32x4 cameraPosition = cc.cameraPos;                                              // :402
        uint32_t unsortedBufferIndex = 0, sortedBufferIndex = 0;                                // :403
        for (auto meshSystem : meshSystems) {                                                   // :408
            const auto& componentPool = meshSystem->getMeshComponentPool();
            auto componentCount = componentPool.getCount();
            auto renderType = meshSystem->getMeshRenderType();
            GvoMeshPool mp{};
            mp.base = reinterpret_cast<uint8_t*>(componentPool.getData());
            mp.stride = meshSystem->getMeshComponentSize();
            mp.occupancy = componentPool.getOccupancy();
            mp.off_entity = offsetof(MeshRenderComponent, entity);
            mp.off_is_enabled = offsetof(MeshRenderComponent, isEnabled);
            mp.off_is_visible = offsetof(MeshRenderComponent, isVisible);
            mp.off_aabb_min = offsetof(MeshRenderComponent, aabb.min);
            mp.off_aabb_max = offsetof(MeshRenderComponent, aabb.max);
            if (renderType == MeshRenderType::Translucent || renderType == MeshRenderType::UI) {
                if (renderType == MeshRenderType::UI && shadowPass >= 0)                        // :416-417
                    continue;
                auto bufferIndex = sortedBufferIndex++;                                         // :419
                auto sortedBuffer = sortedBuffers[bufferIndex];
                sortedBuffer->meshSystem = meshSystem;                                          // :421
                sortedBuffer->drawCount.store(0);
                sortedBuffer->instanceCount.store(0);
                if (componentCount == 0 || !meshSystem->isDrawReady(shadowPass))                // :426
                    continue;
                if (renderType == MeshRenderType::Translucent) {                                // :432-435: the view frustum, the camera, 3-D key
                    auto out = run(mp, tp, makeView(viewProj, cameraPosition, cameraOffset, shadowPass, false));
                    append(transSortedMeshes, transDrawIndex, sortedBuffer, out, mp.stride, bufferIndex);
                } else {                                                                        // :438-441: the UI frustum, camera at the origin, 2-D key
                    auto out = run(mp, tp, makeView(*uiViewProj, f32x4(), cameraOffset, shadowPass, true));
                    append(uiSortedMeshes, uiDrawIndex, sortedBuffer, out, mp.stride, bufferIndex);
                }
            } else {
                auto unsortedBuffer = unsortedBuffers[unsortedBufferIndex++];                   // :475
                unsortedBuffer->meshSystem = meshSystem;
                unsortedBuffer->drawCount.store(0);
                unsortedBuffer->instanceCount.store(0);
                if (componentCount == 0 || !meshSystem->isDrawReady(shadowPass))                // :482
                    continue;
                if (unsortedBuffer->combinedMeshes.size() < componentCount)                     // :485-486
                    unsortedBuffer->combinedMeshes.resize(componentCount);
                hasAnyRefr |= renderType == MeshRenderType::Refracted;                          // :488-490
                hasAnyOIT |= renderType == MeshRenderType::OIT;
                hasAnyTD |= renderType == MeshRenderType::TransDepth;
                auto out = run(mp, tp, makeView(viewProj, cameraPosition, cameraOffset, shadowPass, false));
                fill(unsortedBuffer, out, mp.stride);
            }
        }
        if (!meshSystems.empty() && sortMeshesEnabled)                                          // :546-552
            sortMeshes();
    }

    void preDeferredRender()  // mesh.cpp:893-903: prepareSystems(); renderShadows(); prepareMeshes(light pass)
    {
        if (!isEnabled)
            return;
        meshSystems.clear();  // prepareSystems, mesh.cpp:69-108
        for (auto& sys : Manager::Instance::get()->getSystems())
            if (auto meshSystem = dynamic_cast<IMeshRenderSystem*>(sys.get())) {
                if (isNonTranslucent) {  // :89-101
                    auto renderType = meshSystem->getMeshRenderType();
                    if (renderType == MeshRenderType::Color || renderType == MeshRenderType::Opaque || renderType == MeshRenderType::UI)
                        meshSystems.push_back(meshSystem);
                } else {
                    meshSystems.push_back(meshSystem);  // :103-107
                }
            }
        auto transformSystem = TransformSystem::Instance::get();
        const auto& cc = GraphicsSystem::Instance::get()->getCommonConstants();
        auto& tpool = transformSystem->getComponents();
        auto& emap = transformSystem->getEntityMap();
        GvoTransformPool tp{};
        tp.base = reinterpret_cast<const uint8_t*>(tpool.getData());
        tp.stride = sizeof(TransformComponent);
        tp.occupancy = tpool.getOccupancy();
        tp.off_entity = offsetof(TransformComponent, entity);
        tp.off_parent = offsetof(TransformComponent, parent);
        tp.off_position = offsetof(TransformComponent, posChildCount);
        tp.off_scale = offsetof(TransformComponent, scaleChildCap);
        tp.off_rotation = offsetof(TransformComponent, rotation);
        tp.off_self_active = offsetof(TransformComponent, selfActive);
        tp.off_ancestors_active = offsetof(TransformComponent, ancestorsActive);
        tp.off_model_with_ancestors = offsetof(TransformComponent, modelWithAncestors);
        tp.entity_to_transform = emap.data();
        tp.entity_capacity = (uint32_t)emap.size();

        // renderShadows, mesh.cpp:795-847: one prepareMeshes per shadow pass, drawn at once there — KEPT here, pass by pass,
        // so that the drop-in (which prepares all passes of a frame together) can be compared with every one of them
        const uint32_t passCount = (uint32_t)shadowPasses.size();
        shadowTransMeshes.resize(passCount);
        shadowTransDrawIndex.assign(passCount, 0);
        shadowSortedBuffers.resize(passCount);
        for (uint32_t s = 0; s < passCount; s++) {
            // (mesh.cpp:812-815: a pass whose prepareShadowRender said no is not in the list; the others keep their passIndex)
            prepareMeshes(shadowPasses[s].viewProj, nullptr, shadowPasses[s].cameraOffset, shadowPasses[s].index(s), tp);
            if (shadowBuffers.size() < unsortedBufferCount)
                shadowBuffers.resize(unsortedBufferCount);
            for (uint32_t b = 0; b < unsortedBufferCount; b++) {
                auto& sb = shadowBuffers[b];
                while (sb.size() < passCount)
                    sb.push_back(new UnsortedBuffer());
                sb[s]->meshSystem = unsortedBuffers[b]->meshSystem;
                sb[s]->drawCount = unsortedBuffers[b]->drawCount.load();
                sb[s]->instanceCount = unsortedBuffers[b]->instanceCount.load();
                sb[s]->combinedMeshes.assign(unsortedBuffers[b]->combinedMeshes.begin(), unsortedBuffers[b]->combinedMeshes.begin() + sb[s]->drawCount);
            }
            while (shadowSortedBuffers[s].size() < sortedBufferCount)
                shadowSortedBuffers[s].push_back(new SortedBuffer());
            for (uint32_t b = 0; b < sortedBufferCount; b++) {
                shadowSortedBuffers[s][b]->meshSystem = sortedBuffers[b]->meshSystem;
                shadowSortedBuffers[s][b]->drawCount = sortedBuffers[b]->drawCount.load();
                shadowSortedBuffers[s][b]->instanceCount = sortedBuffers[b]->instanceCount.load();
            }
            shadowTransMeshes[s].assign(transSortedMeshes.begin(), transSortedMeshes.begin() + transDrawIndex);
            shadowTransDrawIndex[s] = transDrawIndex;
        }
        prepareMeshes(cc.viewProj, &uiViewProj, f32x4(), -1, tp);  // :899-902
        if (shadowBuffers.size() < unsortedBufferCount)
            shadowBuffers.resize(unsortedBufferCount);
    }
};

}  // namespace garden
