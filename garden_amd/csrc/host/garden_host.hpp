// garden_host.hpp — host-side mirror of the Garden types on the visibility path, with the reference's byte
// layouts and member names, so a system written here reads like one written inside the engine.
//   TransformComponent / TransformSystem    include/garden/system/transform.hpp:31-61,74-110, source/system/transform.cpp:75-195
//   MeshRenderComponent / IMeshRenderSystem  include/garden/system/render/mesh.hpp:45-55,60-147
//   UnsortedMesh / MeshBuffer                include/garden/system/render/mesh.hpp:191-218
//   CommonConstants (viewProj, cameraPos)    include/garden/graphics/constants.hpp:26-56
//   event chain Update -> Render -> PreDeferredRender   source/system/graphics.cpp:312,409; render/deferred.cpp:441-446
// The math types are plain PODs (cfnptr/math is absent); all arithmetic of the path lives behind the C-ABI.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "ecsm_lite.hpp"

namespace garden {
using namespace ecsm;

struct f32x4 {
    float x = 0, y = 0, z = 0, w = 0;
    f32x4() = default;
    f32x4(float x_, float y_, float z_, float w_ = 0) : x(x_), y(y_), z(z_), w(w_) {}
};
struct quat {
    float x = 0, y = 0, z = 0, w = 1;
    quat() = default;
    quat(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
};
struct Aabb {
    f32x4 min = f32x4(-0.5f, -0.5f, -0.5f), max = f32x4(0.5f, 0.5f, 0.5f);  // Aabb::one
};
struct f32x4x4 {
    float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};  // column-major c0..c3
};
struct float4x3 {
    float m[12] = {};  // c0.xyz c1.xyz c2.xyz c3.xyz
};

// transform.hpp:31-61 — 80 bytes in release builds.
struct alignas(16) TransformComponent final : public Component {
    ID<Entity> parent = {};
    uint64_t uid = 0;
    f32x4 posChildCount = f32x4(0, 0, 0, 0);
    f32x4 scaleChildCap = f32x4(1, 1, 1, 0);
    quat rotation;
    ID<Entity>* childs = nullptr;
    volatile bool selfActive = true;
    volatile bool ancestorsActive = true;
    volatile bool modelWithAncestors = true;

    uint32_t childCount() const noexcept
    {
        uint32_t v;
        memcpy(&v, &posChildCount.w, 4);
        return v;
    }
    void setChildCount(uint32_t v) noexcept { memcpy(&posChildCount.w, &v, 4); }
    uint32_t childCapacity() const noexcept
    {
        uint32_t v;
        memcpy(&v, &scaleChildCap.w, 4);
        return v;
    }
    void setChildCapacity(uint32_t v) noexcept { memcpy(&scaleChildCap.w, &v, 4); }
    void setPosition(float x, float y, float z) noexcept { posChildCount.x = x; posChildCount.y = y; posChildCount.z = z; }
    void setScale(float x, float y, float z) noexcept { scaleChildCap.x = x; scaleChildCap.y = y; scaleChildCap.z = z; }
    void setRotation(quat r) noexcept { rotation = r; }
    bool isActive() const noexcept { return selfActive && ancestorsActive; }  // transform.hpp:110
    ID<Entity> getParent() const noexcept { return parent; }
};
static_assert(sizeof(TransformComponent) == 80, "TransformComponent must keep the reference's 80-byte layout");

class TransformSystem final : public ComponentSystem<TransformComponent>, public Singleton<TransformSystem> {
public:
    uint64_t hierarchyVersion = 0, transformVersion = 0;  // bumped by the mutators below; consumers re-mirror
    uint64_t reparentVersion = 0;  // setParent on existing entities: only links of [reparentLo, reparentHi) changed
    uint32_t reparentLo = UINT32_MAX, reparentHi = 0;
    void clearReparentRange() noexcept { reparentLo = UINT32_MAX; reparentHi = 0; }
    // setActive / setParent flip selfActive / ancestorsActive of a subtree: the slots they touched, so consumers can
    // re-mirror [flagsLo, flagsHi) instead of the pool (flagsVersion bumps with them; transformVersion is left for
    // writers that do not say what they changed)
    uint64_t flagsVersion = 0;
    uint32_t flagsLo = UINT32_MAX, flagsHi = 0;
    void clearFlagsRange() noexcept { flagsLo = UINT32_MAX; flagsHi = 0; }

    // hierarchyVersion is left for structural changes nobody itemised (consumers rebuild their mirror); creating and
    // destroying entities is itemised: the slots involved are recorded like any other change
    View<TransformComponent> add(ID<Entity> entity)
    {
        auto view = addTo(entity);
        touchFlags((uint32_t)(*view - components.getData()));
        return view;
    }
    void disposeComponents() override  // the slots are wiped (entity = null) here, at the end of the frame
    {
        for (auto id : components.getGarbage())
            touchFlags(*id - 1);
        ComponentSystem<TransformComponent>::disposeComponents();
    }
    // transform.cpp:129-195: unlink from the old parent's childs[] (order kept), append to the new one, take
    // ancestorsActive from the new parent — for this entity only: the reference does not push it down the subtree
    void setParent(ID<Entity> entity, ID<Entity> newParent)
    {
        auto view = tryGetOf(entity);
        if (!view || view->parent == newParent)
            return;
        for (auto p = newParent; p; p = tryGetOf(p)->parent)  // cycle check, transform.cpp:137-143
            if (p == entity)
                throw std::runtime_error("setParent: cycle");
        if (view->parent) {
            auto old = tryGetOf(view->parent);
            uint32_t n = old->childCount();
            for (uint32_t i = 0; i < n; i++)
                if (old->childs[i] == entity) {
                    for (uint32_t j = i + 1; j < n; j++)
                        old->childs[j - 1] = old->childs[j];
                    old->setChildCount(n - 1);
                    break;
                }
        }
        view->parent = newParent;
        bool active = true;
        if (newParent) {
            auto np = tryGetOf(newParent);
            uint32_t n = np->childCount(), cap = np->childCapacity();
            if (n == cap) {
                cap = cap ? cap * 2 : 1;
                np->childs = static_cast<ID<Entity>*>(std::realloc(np->childs, sizeof(ID<Entity>) * cap));
                np->setChildCapacity(cap);
            }
            np->childs[n] = entity;
            np->setChildCount(n + 1);
            active = np->isActive();
        }
        view->ancestorsActive = active;
        const uint32_t slot = (uint32_t)(*view - components.getData());
        touchFlags(slot);
        reparentLo = std::min(reparentLo, slot);
        reparentHi = std::max(reparentHi, slot + 1);
        reparentVersion++;
    }
    // TransformComponent::setActive, transform.cpp:75-127, walk for walk — including its stack, which is `static thread_local`
    // (transform.cpp:27) and is NOT emptied by the early return of an activation under inactive ancestors (:86-87): the entity stays
    // on it, and the NEXT setActive of any entity — after its own subtree — also walks the subtree of that left-over entity, in ITS
    // direction (an activation marks the left-over's children ancestorsActive = true although the left-over's own ancestors are
    // inactive; a deactivation marks them false). Activating: children become ancestorsActive = true, and the walk does NOT descend
    // through a child that is itself inactive (:94-95 `continue`): the bytes below it are left as they are — which is what makes the
    // stored byte history-dependent. Deactivating: every descendant, whatever its own state (:110-124). The cull reads the stored
    // bytes (isActive = selfActive && ancestorsActive, transform.hpp:110): whatever history leaves there is what both systems see.
    void setActive(ID<Entity> entity, bool isActive)
    {
        static thread_local std::vector<ID<Entity>> entityStack;  // transform.cpp:27
        auto view = tryGetOf(entity);
        if (!view || view->selfActive == isActive)
            return;
        view->selfActive = isActive;
        touchFlags((uint32_t)(*view - components.getData()));
        entityStack.push_back(entity);
        if (isActive) {
            if (!view->ancestorsActive)
                return;  // (the entity stays on the stack, as in the reference)
            while (!entityStack.empty()) {
                auto v = tryGetOf(entityStack.back());
                entityStack.pop_back();
                if (!v || !v->selfActive)  // (!v: a left-over entity that has been destroyed since — the reference would read freed memory)
                    continue;
                for (uint32_t i = 0, n = v->childCount(); i < n; i++)
                    if (auto child = tryGetOf(v->childs[i])) {
                        child->ancestorsActive = true;
                        touchFlags((uint32_t)(*child - components.getData()));
                        entityStack.push_back(v->childs[i]);
                    }
            }
        } else {
            while (!entityStack.empty()) {
                auto v = tryGetOf(entityStack.back());
                entityStack.pop_back();
                if (!v)
                    continue;
                for (uint32_t i = 0, n = v->childCount(); i < n; i++)
                    if (auto child = tryGetOf(v->childs[i])) {
                        child->ancestorsActive = false;
                        touchFlags((uint32_t)(*child - components.getData()));
                        entityStack.push_back(v->childs[i]);
                    }
            }
        }
    }
    void markTransformsChanged() noexcept { transformVersion++; }
    // Itemised form for writers that know what they moved (an animation / physics system walking its own list): the
    // slots whose position / rotation / scale changed since the consumers last looked. Consumers re-mirror exactly these
    // (GV_DIRTY_TRANSFORM per range) and, with a kept world-matrix cache, re-sweep only the subtrees under them.
    std::vector<std::pair<uint32_t, uint32_t>> movedRanges;  // (first slot, count)
    void markMoved(ID<Entity> entity)
    {
        if (*entity >= entityToComponent.size() || entityToComponent[*entity] == none)
            return;
        const uint32_t slot = entityToComponent[*entity];
        if (!movedRanges.empty() && movedRanges.back().first + movedRanges.back().second == slot)
            movedRanges.back().second++;
        else
            movedRanges.push_back({slot, 1u});
    }
    void clearMovedRanges() noexcept { movedRanges.clear(); }
    // TransformComponent::destroy (transform.cpp:29-73): unlink from the parent's childs[] keeping their order,
    // orphan the children (parent = null, ancestorsActive = true, not pushed further down), free childs[]
    void removeOf(ID<Entity> entity) override
    {
        auto view = tryGetOf(entity);
        if (view) {
            if (view->parent) {
                if (auto parentView = tryGetOf(view->parent)) {
                    uint32_t n = parentView->childCount();
                    for (uint32_t i = 0; i < n; i++)
                        if (parentView->childs[i] == entity) {
                            for (uint32_t j = i + 1; j < n; j++)
                                parentView->childs[j - 1] = parentView->childs[j];
                            parentView->setChildCount(n - 1);
                            break;
                        }
                }
            }
            for (uint32_t i = 0, n = view->childCount(); i < n; i++)
                if (auto child = tryGetOf(view->childs[i])) {
                    child->parent = {};
                    child->ancestorsActive = true;
                    const uint32_t childSlot = (uint32_t)(*child - components.getData());
                    touchFlags(childSlot);
                    reparentLo = std::min(reparentLo, childSlot);
                    reparentHi = std::max(reparentHi, childSlot + 1);
                    reparentVersion++;
                }
            std::free(view->childs);
            view->childs = nullptr;
            view->setChildCount(0);
            view->setChildCapacity(0);
            touchFlags((uint32_t)(*view - components.getData()));
        }
        ComponentSystem<TransformComponent>::removeOf(entity);
    }
    ~TransformSystem() override
    {
        auto data = components.getData();
        for (uint32_t i = 0; i < components.getOccupancy(); i++)
            std::free(data[i].childs);
    }

private:
    void touchFlags(uint32_t slot) noexcept
    {
        flagsLo = std::min(flagsLo, slot);
        flagsHi = std::max(flagsHi, slot + 1);
        flagsVersion++;
    }
};

// render/mesh.hpp:45-55 — 48 bytes; derived components are larger, hence getMeshComponentSize().
struct alignas(16) MeshRenderComponent : public Component {
protected:
    uint32_t reserved0 = 0;
    uint32_t reserved1 = 0;
    uint16_t reserved2 = 0;

public:
    volatile bool isEnabled = true;
    volatile bool isVisible = false;
    Aabb aabb;
};
static_assert(sizeof(MeshRenderComponent) == 48, "MeshRenderComponent must keep the reference's 48-byte layout");

enum class MeshRenderType : uint8_t { Color, Opaque, Translucent, OIT, Refracted, TransDepth, UI, Count };

// render/mesh.hpp:60-147 — the members the prepare phase consumes, with the reference's names, signatures and access: isDrawReady is
// protected and reached through `friend class MeshRenderSystem` (mesh.hpp:69,118); the two classes that stand where
// MeshRenderSystem stands here are its friends instead.
class GpuVisibilitySystem;
class CpuMeshRenderSystem;
class IMeshRenderSystem {
public:
    using MeshRenderPool = LinearPool<MeshRenderComponent, false>;
    virtual ~IMeshRenderSystem() = default;

protected:
    // "Is mesh system ready for rendering. (All resources loaded, etc.)" shadowPass: shadow pass index, light pass = -1
    // (mesh.hpp:64-69; pass-dependent in InstanceRenderSystem::isDrawReady, instance.cpp:61-…, UiLabelSystem, label.cpp:262-269)
    virtual bool isDrawReady(int8_t shadowPass) = 0;

public:
    virtual MeshRenderType getMeshRenderType() const = 0;         // mesh.hpp:123
    virtual MeshRenderPool& getMeshComponentPool() const = 0;     // mesh.hpp:127: walked with getMeshComponentSize() as byte stride
    virtual size_t getMeshComponentSize() const = 0;              // mesh.hpp:131

    friend class GpuVisibilitySystem;
    friend class CpuMeshRenderSystem;
};

// One mesh system per component type, as in the engine (ModelRenderSystem, SpriteRenderSystem, ... each own a pool
// of their MeshRenderComponent-derived struct and report a MeshRenderType, render/mesh.hpp:60-147).
class VersionedMeshSystem {
public:
    // false: the system behaves like the engine's own mesh systems (sprite.cpp, 9-slice, label.cpp, instance.cpp: no counter at
    // all) — consumers must find out themselves what changed (headless_tick --unversioned)
    bool reportsChanges = true;
    uint64_t meshVersion = 0;  // bumped by markMeshesChanged(): consumers re-mirror the whole pool
    void markMeshesChanged() noexcept { meshVersion++; }
    // itemised changes (components created / destroyed / edited through touchMesh): only [meshLo, meshHi) moves
    uint64_t rangeVersion = 0;
    uint32_t meshLo = UINT32_MAX, meshHi = 0;
    void touchMesh(uint32_t slot) noexcept
    {
        meshLo = std::min(meshLo, slot);
        meshHi = std::max(meshHi, slot + 1);
        rangeVersion++;
    }
    void clearMeshRange() noexcept { meshLo = UINT32_MAX; meshHi = 0; }
};
// stand-in for "all resources loaded": what a system's isDrawReady answers for the light pass and for each shadow pass
class ReadinessSwitch {
public:
    bool readyMain = true;
    uint32_t readyShadowMask = ~0u;  // bit s = shadow pass s
    bool drawReady(int8_t shadowPass) const noexcept { return shadowPass < 0 ? readyMain : ((readyShadowMask >> shadowPass) & 1u) != 0; }
};
template <class C, MeshRenderType TYPE>
class MeshSystemOf : public ComponentSystem<C, false>, public IMeshRenderSystem, public VersionedMeshSystem, public ReadinessSwitch {
    static_assert(std::is_base_of<MeshRenderComponent, C>::value, "mesh components derive from MeshRenderComponent");

public:
    View<C> add(ID<Entity> entity)
    {
        auto view = this->addTo(entity);
        touchMesh((uint32_t)(*view - this->components.getData()));
        return view;
    }
    void removeOf(ID<Entity> entity) override
    {
        if (auto view = this->tryGetOf(entity))
            touchMesh((uint32_t)(*view - this->components.getData()));
        ComponentSystem<C, false>::removeOf(entity);
    }
    void disposeComponents() override
    {
        for (auto id : this->components.getGarbage())
            touchMesh(*id - 1);
        ComponentSystem<C, false>::disposeComponents();
    }
    MeshRenderType getMeshRenderType() const override { return TYPE; }
    // the engine's own spelling (instance.hpp:108, sprite.hpp:192, 9-slice.hpp:120): the pool of the derived component type, seen
    // as a pool of the base type — data, occupancy and count do not depend on the element type
    // (the engine's own cast; a translation unit that calls this is built with -fno-strict-aliasing, like tests/cpp/Makefile)
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wstrict-aliasing"
    MeshRenderPool& getMeshComponentPool() const override { return *((MeshRenderPool*)&this->components); }
#pragma GCC diagnostic pop
    size_t getMeshComponentSize() const override { return sizeof(C); }


protected:
    bool isDrawReady(int8_t shadowPass) override { return drawReady(shadowPass); }
};
using OpaqueMeshSystem = MeshSystemOf<MeshRenderComponent, MeshRenderType::Opaque>;

// Derived components with the reference's shape: base header + system-specific payload, so pools are walked with a
// byte stride larger than 48 (sprite.hpp:29-43).
struct alignas(16) TranslucentMeshComponent final : public MeshRenderComponent {
    float colorFactor[4] = {1, 1, 1, 1};
};
struct alignas(16) OitMeshComponent final : public MeshRenderComponent {
    float colorFactor[4] = {1, 1, 1, 0.5f};
    uint64_t descriptorSet = 0;
};
struct alignas(16) UiMeshComponent final : public MeshRenderComponent {
    float uvSize[2] = {1, 1}, uvOffset[2] = {0, 0};
};
struct alignas(16) RefractedMeshComponent final : public MeshRenderComponent {
    float indexOfRefraction[4] = {1.33f, 0, 0, 0};
};
struct alignas(16) TransDepthMeshComponent final : public MeshRenderComponent {
    float colorFactor[4] = {1, 1, 1, 0.25f};
    float depthBias[4] = {0, 0, 0, 0};
};
using RefractedMeshSystem = MeshSystemOf<RefractedMeshComponent, MeshRenderType::Refracted>;
using TransDepthMeshSystem = MeshSystemOf<TransDepthMeshComponent, MeshRenderType::TransDepth>;
using TranslucentMeshSystem = MeshSystemOf<TranslucentMeshComponent, MeshRenderType::Translucent>;
using OitMeshSystem = MeshSystemOf<OitMeshComponent, MeshRenderType::OIT>;
using UiMeshSystem = MeshSystemOf<UiMeshComponent, MeshRenderType::UI>;

// graphics/constants.hpp:26-56 (fields the path reads) + render/mesh.hpp:166 shadow-pass inputs.
struct CommonConstants {
    f32x4x4 viewProj;
    f32x4 cameraPos;
};

// render/mesh.hpp:191-217
struct UnsortedMesh final {
    size_t componentOffset = 0;
    float4x3 bakedModel;
    float distanceSq = 0.0f;
    bool operator<(const UnsortedMesh& m) const noexcept { return distanceSq < m.distanceSq; }
};
struct MeshBuffer {
    IMeshRenderSystem* meshSystem = nullptr;
    std::atomic<uint32_t> drawCount{0};
    alignas(64) std::atomic<uint32_t> instanceCount{0};
};
struct UnsortedBuffer final : public MeshBuffer {
    std::vector<UnsortedMesh> combinedMeshes;
    // What the render passes read — `const auto meshes = unsortedBuffer->combinedMeshes.data()` in the reference
    // (mesh.cpp:581,611). NULL: the vector itself; otherwise records [0, drawCount) that live elsewhere for this frame: the
    // drop-in's own page-locked result buffer (GpuVisibilitySystem::recordSpans — no copy into the vector at all).
    const UnsortedMesh* span = nullptr;
    const UnsortedMesh* meshes() const noexcept { return span ? span : combinedMeshes.data(); }
};
// render/mesh.hpp:198-205,218: translucent / UI meshes of ALL such systems share one array per kind
// (transSortedMeshes / uiSortedMeshes, mesh.hpp:222-223), drawn back to front; bufferIndex names the system.
struct SortedMesh final {
    size_t componentOffset = 0;
    float4x3 bakedModel;
    float distanceSq = 0.0f;
    uint32_t bufferIndex = 0;
    bool operator<(const SortedMesh& m) const noexcept { return distanceSq > m.distanceSq; }
};
struct SortedBuffer final : public MeshBuffer {};

// Headless stand-ins for GraphicsSystem ("Update": prepareCommonConstants -> runEvent("Render"),
// graphics.cpp:312,409) and DeferredRenderSystem ("Render" -> "PreDeferredRender"/"DeferredRender",
// deferred.cpp:441-489). No Vulkan.
class GraphicsSystem final : public System, public Singleton<GraphicsSystem> {
    CommonConstants commonConstants;

public:
    GraphicsSystem()
    {
        auto manager = Manager::Instance::get();
        manager->registerEvent("Render");
        ECSM_SUBSCRIBE_TO_EVENT("Update", GraphicsSystem::update);
    }
    const CommonConstants& getCommonConstants() const noexcept { return commonConstants; }
    void setCamera(const f32x4x4& viewProj, f32x4 cameraPos) noexcept
    {
        commonConstants.viewProj = viewProj;
        commonConstants.cameraPos = cameraPos;
    }

private:
    void update() { Manager::Instance::get()->runEvent("Render"); }
};

class DeferredRenderSystem final : public System, public Singleton<DeferredRenderSystem> {
public:
    DeferredRenderSystem()
    {
        auto manager = Manager::Instance::get();
        manager->registerEvent("PreDeferredRender");
        manager->registerEvent("DeferredRender");
        manager->registerEvent("PreHdrRender");
        ECSM_SUBSCRIBE_TO_EVENT("Render", DeferredRenderSystem::render);
    }

private:
    void render()
    {
        auto manager = Manager::Instance::get();
        manager->runEvent("PreDeferredRender");
        manager->runEvent("DeferredRender");
        manager->runEvent("PreHdrRender");
    }
};

}  // namespace garden
