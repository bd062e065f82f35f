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
    std::vector<uint32_t> idx;
    std::vector<float> baked, dist;

public:
    bool isEnabled = true;
    bool sortMeshes = true;  // mesh.cpp:548-551: prepareMeshes ends with sortMeshes()
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
    }
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
    GvoCullOut run(const GvoMeshPool& mp, const GvoTransformPool& tp, const GvoView& view)
    {
        const size_t n = mp.occupancy ? mp.occupancy : 1;
        idx.resize(n); baked.resize(n * 12); dist.resize(n);
        GvoCullOut out{idx.data(), baked.data(), dist.data(), 0, 0};
        if (useAvx2) {
            GvoSoa* soa = gvo_soa_build(&mp, &tp);
            gvo_prepare_meshes_avx2(soa, &mp, &view, nullptr, threads, &out);
            gvo_soa_free(soa);
            // the AVX2 path reports records in slot order per thread range too: same contract as the scalar one
        } else {
            gvo_prepare_meshes(&mp, &tp, &view, nullptr, threads, &out);
        }
        return out;
    }
    void fill(UnsortedBuffer* buffer, IMeshRenderSystem* ms, const GvoCullOut& out, size_t stride)
    {
        buffer->meshSystem = ms;
        buffer->drawCount = out.draw_count;
        buffer->instanceCount = out.instance_count;
        if (buffer->combinedMeshes.size() < out.draw_count)
            buffer->combinedMeshes.resize(out.draw_count);
        for (uint32_t k = 0; k < out.draw_count; k++) {
            auto& m = buffer->combinedMeshes[k];
            m.componentOffset = (size_t)idx[k] * stride;
            memcpy(m.bakedModel.m, baked.data() + (size_t)k * 12, 48);
            m.distanceSq = dist[k];
        }
        if (sortMeshes && ms->getMeshRenderType() != MeshRenderType::OIT)  // mesh.cpp:273-277
            std::sort(buffer->combinedMeshes.begin(), buffer->combinedMeshes.begin() + out.draw_count);
    }
    void append(std::vector<SortedMesh>& combined, uint32_t& drawIndex, const GvoCullOut& out, size_t stride, uint32_t bufferIndex)
    {
        if (combined.size() < (size_t)drawIndex + out.draw_count)
            combined.resize((size_t)drawIndex + out.draw_count);
        for (uint32_t k = 0; k < out.draw_count; k++) {
            auto& m = combined[drawIndex + k];
            m.componentOffset = (size_t)idx[k] * stride;
            memcpy(m.bakedModel.m, baked.data() + (size_t)k * 12, 48);
            m.distanceSq = dist[k];
            m.bufferIndex = bufferIndex;
        }
        drawIndex += out.draw_count;
    }

    void preDeferredRender()
    {
        if (!isEnabled)
            return;
        meshSystems.clear();
        for (auto& sys : Manager::Instance::get()->getSystems())
            if (auto ms = dynamic_cast<IMeshRenderSystem*>(sys.get()))
                meshSystems.push_back(ms);
        unsortedBufferCount = sortedBufferCount = 0;
        for (auto ms : meshSystems) {
            const auto type = ms->getMeshRenderType();
            ((type == MeshRenderType::Translucent || type == MeshRenderType::UI) ? sortedBufferCount : unsortedBufferCount)++;
        }
        while (unsortedBuffers.size() < unsortedBufferCount)
            unsortedBuffers.push_back(new UnsortedBuffer());
        while (sortedBuffers.size() < sortedBufferCount)
            sortedBuffers.push_back(new SortedBuffer());
        shadowBuffers.resize(unsortedBufferCount);
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

        const uint32_t passCount = (uint32_t)shadowPasses.size();
        transDrawIndex = uiDrawIndex = 0;
        shadowTransMeshes.resize(passCount);
        shadowTransDrawIndex.assign(passCount, 0);
        uint32_t unsortedBufferIndex = 0, sortedBufferIndex = 0;
        for (size_t p = 0; p < meshSystems.size(); p++) {
            auto ms = meshSystems[p];
            const auto type = ms->getMeshRenderType();
            GvoMeshPool mp{};
            mp.base = ms->getMeshComponentData();
            mp.stride = ms->getMeshComponentSize();
            mp.occupancy = ms->getMeshComponentOccupancy();
            mp.off_entity = offsetof(MeshRenderComponent, entity);
            mp.off_is_enabled = offsetof(MeshRenderComponent, isEnabled);
            mp.off_is_visible = offsetof(MeshRenderComponent, isVisible);
            mp.off_aabb_min = offsetof(MeshRenderComponent, aabb.min);
            mp.off_aabb_max = offsetof(MeshRenderComponent, aabb.max);
            // renderShadows() first (mesh.cpp:795-847), then the main camera (mesh.cpp:899-902)
            if (type == MeshRenderType::UI) {  // mesh.cpp:416,436-442: main pass only, UI frustum, camera at the origin
                const uint32_t bufferIndex = sortedBufferIndex++;
                auto out = run(mp, tp, makeView(uiViewProj, f32x4(), f32x4(), -1, true));
                sortedBuffers[bufferIndex]->meshSystem = ms;
                sortedBuffers[bufferIndex]->drawCount = out.draw_count;
                sortedBuffers[bufferIndex]->instanceCount = out.instance_count;
                append(uiSortedMeshes, uiDrawIndex, out, mp.stride, bufferIndex);
            } else if (type == MeshRenderType::Translucent) {
                const uint32_t bufferIndex = sortedBufferIndex++;
                for (uint32_t s = 0; s < passCount; s++) {
                    auto out = run(mp, tp, makeView(shadowPasses[s].viewProj, cc.cameraPos, shadowPasses[s].cameraOffset, (int8_t)s, false));
                    append(shadowTransMeshes[s], shadowTransDrawIndex[s], out, mp.stride, bufferIndex);
                }
                auto out = run(mp, tp, makeView(cc.viewProj, cc.cameraPos, f32x4(), -1, false));
                sortedBuffers[bufferIndex]->meshSystem = ms;
                sortedBuffers[bufferIndex]->drawCount = out.draw_count;
                sortedBuffers[bufferIndex]->instanceCount = out.instance_count;
                append(transSortedMeshes, transDrawIndex, out, mp.stride, bufferIndex);
            } else {
                const uint32_t bufferIndex = unsortedBufferIndex++;
                auto& sb = shadowBuffers[bufferIndex];
                while (sb.size() < passCount)
                    sb.push_back(new UnsortedBuffer());
                for (uint32_t s = 0; s < passCount; s++) {
                    auto out = run(mp, tp, makeView(shadowPasses[s].viewProj, cc.cameraPos, shadowPasses[s].cameraOffset, (int8_t)s, false));
                    fill(sb[s], ms, out, mp.stride);
                }
                auto out = run(mp, tp, makeView(cc.viewProj, cc.cameraPos, f32x4(), -1, false));
                fill(unsortedBuffers[bufferIndex], ms, out, mp.stride);
            }
        }
        if (sortMeshes) {  // mesh.cpp:296-326
            std::sort(transSortedMeshes.begin(), transSortedMeshes.begin() + transDrawIndex);
            std::sort(uiSortedMeshes.begin(), uiSortedMeshes.begin() + uiDrawIndex);
            for (uint32_t s = 0; s < passCount; s++)
                std::sort(shadowTransMeshes[s].begin(), shadowTransMeshes[s].begin() + shadowTransDrawIndex[s]);
        }
    }
};

}  // namespace garden
