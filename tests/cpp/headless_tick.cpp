// headless_tick.cpp — TEST driver: a headless ecsm-style frame loop (Manager::update() firing
// Input -> Update -> Render -> PreDeferredRender, no Vulkan) with either the CPU reference-path system
// (oracle-backed, BASELINE.json configs[0]) or the GPU drop-in (GpuVisibilitySystem over libgarden_vis.so),
// and, in `both` mode, a bit-for-bit comparison of what each leaves behind for the render phase.
//
//   headless_tick --mode cpu|gpu|both [--ranks R] [--gate never|shadow|reverse|empty] [--non-translucent] [--hiz] [--entities N] [--ticks T] [--threads K] [--hier] [--mutate] [--mixed] [--toggle] [--bounds] [--churn R] [--avx2] [--animate K] [--itemised] [--world] [--csm] [--soa-records] [--copy-records] [--span-records] [--seed S]
// --mixed spreads the meshes over Opaque, OIT, two Translucent and one UI system and adds two shadow passes, so the
// unsorted/sorted classification of prepareMeshes (mesh.cpp:341-546) and sortMeshes (mesh.cpp:265-328) are compared too.
// --gate (with --mixed): the per-system gate of mesh.cpp:426 / :482 — `componentCount == 0 || !isDrawReady(shadowPass)`:
//   never: the OIT, the Refracted and the second Translucent system are never ready; shadow: the Opaque and the first Translucent
//   system are ready for the light pass and shadow pass 0, not for shadow pass 1 (InstanceRenderSystem::isDrawReady answers per
//   pass, instance.cpp:61-…); reverse: the TransDepth system is ready for the shadow passes only; empty: the OIT and the second
//   Translucent system have lost all their components (occupancy > 0, count 0). Besides the comparison of the two systems, each
//   system's results are checked against the reference TEXT (gateHolds below): a system that is not drawn in a pass has its
//   counters at 0 for that pass, contributes no record, and — light pass — its isVisible bytes are exactly what they were before
//   the tick (every tick starts from a pattern no cull would leave behind).
// --non-translucent: MeshRenderSystem::isNonTranslucent (mesh.hpp:275): prepareSystems keeps Color / Opaque / UI systems only
// (mesh.cpp:89-101) — the others get no buffer and their isVisible bytes are never touched.
// --ranks R: the GPU drop-in's own multi-GPU mode — ONE process, ONE thread, R contexts (all on device 0 here: with R > 1 the rows
// travel through the test transport named by GV_RCCL_LIBRARY): the pools are dealt to the ranks, every rank culls its share, the
// lists are gathered on the devices (every rank's rows are read back and compared: all ranks hold the same rows, their union is the
// set of WORLD slots the pass's buffer holds) and the engine's buffers are filled from the ranks' results — compared with the CPU
// system's like any other run.
// --same-frame (with --mode both): the two systems run in the SAME Manager::update() — the CPU system, a hook that takes its snapshot,
// then the GPU drop-in — instead of a frame each. Entities destroyed since the last frame are then still in their pools (components are
// wiped at the END of a frame, docs/ECS/Entities.md:52-54) while Manager::tryGet no longer finds them: the state an engine really
// presents on the frame after a destroy, which a CPU frame in front of the GPU frame would have disposed of.
// Prints one JSON line; exit code 0 = ok, 1 = mismatch/failure.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>

#include <hip/hip_runtime.h>

#include "../../garden_amd/csrc/host/gpu_visibility_system.hpp"
#include "../../garden_amd/csrc/host/csm_lite.hpp"
#include "../../oracle/cpu_mesh_render_system.hpp"

using namespace garden;

static_assert(offsetof(TransformComponent, parent) == 4 && offsetof(TransformComponent, uid) == 8 &&
              offsetof(TransformComponent, posChildCount) == 16 && offsetof(TransformComponent, scaleChildCap) == 32 &&
              offsetof(TransformComponent, rotation) == 48 && offsetof(TransformComponent, childs) == 64 &&
              offsetof(TransformComponent, selfActive) == 72 && offsetof(TransformComponent, modelWithAncestors) == 74,
              "transform.hpp:31-61 layout");
static_assert(offsetof(MeshRenderComponent, isEnabled) == 14 && offsetof(MeshRenderComponent, isVisible) == 15 &&
              offsetof(MeshRenderComponent, aabb) == 16, "render/mesh.hpp:45-55 layout");

struct Rng {  // PCG32
    uint64_t state = 0x853c49e6748fea9bull, inc = 0xda3e39cb94b95bdbull;
    uint32_t next()
    {
        uint64_t old = state;
        state = old * 6364136223846793005ull + (inc | 1);
        uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
        return (xs >> rot) | (xs << ((-rot) & 31));
    }
    float uniform(float lo, float hi) { return lo + (hi - lo) * (float)(next() >> 8) * (1.0f / 16777216.0f); }
};

// a second Translucent system with its own component type (one pool per component type, as in the engine)
struct alignas(16) GlassMeshComponent final : public MeshRenderComponent {
    float indexOfRefraction = 1.5f;
    uint32_t materialId = 0;
};
using GlassMeshSystem = MeshSystemOf<GlassMeshComponent, MeshRenderType::Translucent>;

// --same-frame: runs between the CPU system and the GPU drop-in (systems are called in the order they were created)
class MidFrameHook final : public System {
public:
    std::function<void()> between;
    MidFrameHook() { ECSM_SUBSCRIBE_TO_EVENT("Init", MidFrameHook::init); }

private:
    void init() { ECSM_SUBSCRIBE_TO_EVENT("PreDeferredRender", MidFrameHook::run); }
    void run()
    {
        if (between)
            between();
    }
};

// What a prepare phase leaves behind for the render phase. Record arrays are compared as sets (the reference's order
// is fetch_add arrival order before sortMeshes and unspecified among equal keys after it): canonical order here is
// (bufferIndex, componentOffset).
struct Record {
    uint32_t bufferIndex;
    size_t componentOffset;
    float bakedModel[12];
    float distanceSq;
    bool operator<(const Record& r) const noexcept
    {
        return bufferIndex != r.bufferIndex ? bufferIndex < r.bufferIndex : componentOffset < r.componentOffset;
    }
};
struct Snapshot {
    bool hasAnyRefr = false, hasAnyOIT = false, hasAnyTD = false;  // mesh.hpp:232-234
    std::vector<std::vector<uint8_t>> isVisible;  // [mesh system][slot]
    std::vector<std::vector<Record>> lists;       // unsorted buffers, their shadow buffers, trans, ui, shadow trans
    std::vector<uint32_t> counters;               // drawCount / instanceCount of every MeshBuffer, in the same walk
    bool ordered = true;                          // every list obeys its operator< (mesh.hpp:196,204)
    std::string disorder;
};

static MeshRenderComponent* componentAt(IMeshRenderSystem* ms, uint32_t slot)  // mesh.cpp:120,139: base + i * componentSize
{
    return reinterpret_cast<MeshRenderComponent*>(reinterpret_cast<uint8_t*>(ms->getMeshComponentPool().getData()) + (size_t)slot * ms->getMeshComponentSize());
}

static std::vector<IMeshRenderSystem*> allMeshSystems(Manager& manager)
{
    std::vector<IMeshRenderSystem*> out;
    for (auto& sys : manager.getSystems())
        if (auto ms = dynamic_cast<IMeshRenderSystem*>(sys.get()))
            out.push_back(ms);
    return out;
}

template <class M>
static void addList(Snapshot& s, const M* meshes, uint32_t count, bool mustBeOrdered, const char* name);
template <class M>
static void addList(Snapshot& s, const std::vector<M>& meshes, uint32_t count, bool mustBeOrdered, const char* name)
{
    addList(s, meshes.data(), count, mustBeOrdered, name);
}
template <class M>
static void addList(Snapshot& s, const M* meshes, uint32_t count, bool mustBeOrdered, const char* name)
{
    std::vector<Record> list(count);
    for (uint32_t k = 0; k < count; k++) {
        list[k].componentOffset = meshes[k].componentOffset;
        memcpy(list[k].bakedModel, meshes[k].bakedModel.m, 48);
        list[k].distanceSq = meshes[k].distanceSq;
        if constexpr (std::is_same<M, SortedMesh>::value)
            list[k].bufferIndex = meshes[k].bufferIndex;
        else
            list[k].bufferIndex = 0;
        if (mustBeOrdered && k > 0 && meshes[k] < meshes[k - 1] && s.ordered) {
            s.ordered = false;
            s.disorder = name;
        }
    }
    std::sort(list.begin(), list.end());
    s.lists.push_back(std::move(list));
}

template <class SystemT>
static Snapshot snapshot(Manager& manager, const SystemT* system, uint32_t passCount)
{
    Snapshot s;
    s.hasAnyRefr = system->getHasAnyRefr();
    s.hasAnyOIT = system->getHasAnyOIT();
    s.hasAnyTD = system->getHasAnyTD();
    for (auto ms : allMeshSystems(manager)) {
        const auto& pool = ms->getMeshComponentPool();
        std::vector<uint8_t> vis(pool.getOccupancy());
        for (uint32_t i = 0; i < vis.size(); i++)
            vis[i] = componentAt(ms, i)->isVisible;
        s.isVisible.push_back(std::move(vis));
    }
    for (uint32_t b = 0; b < system->getUnsortedBufferCount(); b++) {
        auto buffer = system->getUnsortedBuffers()[b];
        const bool sorted = buffer->meshSystem->getMeshRenderType() != MeshRenderType::OIT;  // mesh.cpp:273-277
        addList(s, buffer->meshes(), buffer->drawCount, sorted, "unsorted buffer");  // what the render passes read (mesh.cpp:581)
        s.counters.push_back(buffer->drawCount);
        s.counters.push_back(buffer->instanceCount);
        for (uint32_t pass = 0; pass < passCount; pass++) {
            auto shadow = system->getShadowBuffers(b)[pass];
            addList(s, shadow->meshes(), shadow->drawCount, sorted, "shadow unsorted buffer");
            s.counters.push_back(shadow->drawCount);
            s.counters.push_back(shadow->instanceCount);
        }
    }
    for (uint32_t b = 0; b < system->getSortedBufferCount(); b++) {
        s.counters.push_back(system->getSortedBuffers()[b]->drawCount);
        s.counters.push_back(system->getSortedBuffers()[b]->instanceCount);
    }
    for (uint32_t pass = 0; pass < passCount; pass++)  // sortedBuffers as each shadow pass leaves them (Translucent systems only)
        for (auto buffer : system->getShadowSortedBuffers(pass)) {
            s.counters.push_back(buffer->drawCount);
            s.counters.push_back(buffer->instanceCount);
        }
    addList(s, system->getTransSortedMeshes(), system->getTransDrawCount(), true, "transSortedMeshes");
    addList(s, system->getUiSortedMeshes(), system->getUiDrawCount(), true, "uiSortedMeshes");
    for (uint32_t pass = 0; pass < passCount; pass++)
        addList(s, system->getShadowTransMeshes(pass), system->getShadowTransDrawCount(pass), true, "shadow transSortedMeshes");
    return s;
}

static bool same(const Snapshot& a, const Snapshot& b, std::string& why)
{
    if (a.hasAnyRefr != b.hasAnyRefr || a.hasAnyOIT != b.hasAnyOIT || a.hasAnyTD != b.hasAnyTD) { why = "hasAnyRefr / hasAnyOIT / hasAnyTD differ"; return false; }
    if (a.isVisible != b.isVisible) { why = "isVisible differs"; return false; }
    if (a.counters != b.counters) { why = "counters differ"; return false; }
    if (a.lists.size() != b.lists.size()) { why = "buffer count differs"; return false; }
    for (size_t l = 0; l < a.lists.size(); l++) {
        if (a.lists[l].size() != b.lists[l].size()) { why = "list length differs"; return false; }
        for (size_t k = 0; k < a.lists[l].size(); k++) {
            const Record &x = a.lists[l][k], &y = b.lists[l][k];
            if (x.bufferIndex != y.bufferIndex) { why = "bufferIndex differs"; return false; }
            if (x.componentOffset != y.componentOffset) { why = "componentOffset differs"; return false; }
            if (memcmp(x.bakedModel, y.bakedModel, 48) != 0) { why = "bakedModel differs"; return false; }
            if (memcmp(&x.distanceSq, &y.distanceSq, 4) != 0) { why = "distanceSq differs"; return false; }
        }
    }
    return true;
}

// Every tick starts from isVisible bytes no cull would leave behind: a system that is drawn in the light pass rewrites every one
// of its slots (each exit of mesh.cpp:140-166 stores the byte), a system that is not must leave them exactly like this.
static bool g_nonTranslucent = false;  // --non-translucent: prepareSystems keeps Color / Opaque / UI systems only (mesh.cpp:89-101)
static bool patternAt(size_t system, uint32_t slot) { return ((slot * 7u + (uint32_t)system * 3u + 3u) % 5u) == 0; }
static void poisonVisible(Manager& manager)
{
    size_t k = 0;
    for (auto ms : allMeshSystems(manager)) {
        for (uint32_t i = 0; i < ms->getMeshComponentPool().getOccupancy(); i++)
            componentAt(ms, i)->isVisible = patternAt(k, i);
        k++;
    }
}

// The gate of prepareMeshes, checked against the reference TEXT (not against the other system): for every mesh system and pass with
// `componentCount == 0 || !isDrawReady(shadowPass)` (mesh.cpp:426,482) the buffer of that pass names the system and has both counters
// at 0 (:419-424, :475-480), no record of the shared sorted arrays carries its bufferIndex, and — light pass — every isVisible byte
// is the one the tick started from. hasAnyRefr / hasAnyOIT / hasAnyTD are what :339,488-490 compute. Empty string: holds.
static std::vector<int8_t> g_passIndex;  // by position in the list of prepared shadow passes: the pass's own number (--skip-pass)
template <class SystemT>
static std::string gateHolds(Manager& manager, const SystemT* system, uint32_t passCount)
{
    uint32_t unsortedIndex = 0, sortedIndex = 0, shadowSortedIndex = 0;
    bool anyRefr = false, anyOit = false, anyTd = false;
    size_t k = 0;
    for (auto ms : allMeshSystems(manager)) {
        const auto type = ms->getMeshRenderType();
        const auto ready = dynamic_cast<const ReadinessSwitch*>(ms);
        const uint32_t count = ms->getMeshComponentPool().getCount(), occupancy = ms->getMeshComponentPool().getOccupancy();
        const bool sorted = type == MeshRenderType::Translucent || type == MeshRenderType::UI;
        const bool kept = !g_nonTranslucent || type == MeshRenderType::Color || type == MeshRenderType::Opaque || type == MeshRenderType::UI;
        const bool light = kept && count != 0 && (!ready || ready->drawReady(-1));
        const std::string name = "mesh system " + std::to_string(k);
        if (!light)
            for (uint32_t i = 0; i < occupancy; i++)
                if ((bool)componentAt(ms, i)->isVisible != patternAt(k, i))
                    return name + " is not drawn in the light pass, yet isVisible of slot " + std::to_string(i) + " was written";
        if (!kept) {  // not among meshSystems at all (mesh.cpp:89-101): no buffer, no index
            k++;
            continue;
        }
        if (sorted) {
            const uint32_t index = sortedIndex++;
            const uint32_t shadowIndex = type == MeshRenderType::Translucent ? shadowSortedIndex++ : 0u;
            const auto buffer = system->getSortedBuffers()[index];
            if (buffer->meshSystem != ms)
                return name + ": sortedBuffers[bufferIndex] does not name it";
            if (!light) {
                if (buffer->drawCount != 0 || buffer->instanceCount != 0)
                    return name + " is not drawn in the light pass, yet its sorted buffer counts draws";
                const auto& list = type == MeshRenderType::UI ? system->getUiSortedMeshes() : system->getTransSortedMeshes();
                const uint32_t n = type == MeshRenderType::UI ? system->getUiDrawCount() : system->getTransDrawCount();
                for (uint32_t r = 0; r < n; r++)
                    if (list[r].bufferIndex == index)
                        return name + " is not drawn in the light pass, yet the shared sorted array holds a record of it";
            }
            if (type == MeshRenderType::Translucent)
                for (uint32_t s = 0; s < passCount; s++) {
                    const auto shadow = system->getShadowSortedBuffers(s).at(shadowIndex);
                    if (shadow->meshSystem != ms)
                        return name + ": a shadow pass's sortedBuffers[bufferIndex] does not name it";
                    if (count != 0 && (!ready || ready->drawReady(g_passIndex.at(s))))
                        continue;
                    if (shadow->drawCount != 0 || shadow->instanceCount != 0)
                        return name + " is not drawn in shadow pass " + std::to_string(s) + ", yet its sorted buffer counts draws";
                    for (uint32_t r = 0; r < system->getShadowTransDrawCount(s); r++)
                        if (system->getShadowTransMeshes(s)[r].bufferIndex == shadowIndex)
                            return name + " is not drawn in shadow pass " + std::to_string(s) + ", yet that pass's sorted array holds a record of it";
                }
        } else {
            const uint32_t index = unsortedIndex++;
            const auto buffer = system->getUnsortedBuffers()[index];
            if (buffer->meshSystem != ms)
                return name + ": unsortedBuffers[index] does not name it";
            if (!light && (buffer->drawCount != 0 || buffer->instanceCount != 0))
                return name + " is not drawn in the light pass, yet its unsorted buffer counts draws";
            if (light) {
                anyRefr = anyRefr || type == MeshRenderType::Refracted;
                anyOit = anyOit || type == MeshRenderType::OIT;
                anyTd = anyTd || type == MeshRenderType::TransDepth;
            }
            for (uint32_t s = 0; s < passCount; s++) {
                if (count != 0 && (!ready || ready->drawReady(g_passIndex.at(s))))
                    continue;
                const auto shadow = system->getShadowBuffers(index).at(s);
                if (shadow->drawCount != 0 || shadow->instanceCount != 0)
                    return name + " is not drawn in shadow pass " + std::to_string(s) + ", yet its unsorted buffer counts draws";
            }
        }
        k++;
    }
    if (system->getHasAnyRefr() != anyRefr || system->getHasAnyOIT() != anyOit || system->getHasAnyTD() != anyTd)
        return "hasAnyRefr / hasAnyOIT / hasAnyTD are not what mesh.cpp:339,488-490 compute";
    return "";
}

// What prepareSortedMeshes / prepareUnsortedMeshes / sortMeshes leave behind, checked against the reference TEXT (not against the other
// system), for one system's buffers after a tick:
//  (i)   mesh.cpp:252,419-421  every record of transSortedMeshes / uiSortedMeshes carries a bufferIndex that names a sortedBuffers entry
//        whose meshSystem OWNS componentOffset: a whole number of components inside that system's pool, on a live, enabled component
//        that the light pass has just marked visible (:166,170)
//  (ii)  mesh.cpp:416-419      a shadow pass prepares no UI system (`continue` before the index is taken): its sorted array holds records
//        of Translucent systems only, and their bufferIndex counts Translucent systems only — it indexes that pass's sortedBuffers
//  (iii) mesh.cpp:270-295, mesh.hpp:196,204   unsorted buffers ascending distanceSq (OIT: not sorted, any order), shared sorted arrays descending
//  (iv)  mesh.cpp:255-259      sum of drawCount over the buffers that share an array == that array's draw index
// Empty string: holds.
template <class SystemT>
static std::string sortedArraysHold(Manager& manager, const SystemT* system, uint32_t passCount)
{
    (void)manager;
    auto owns = [](const MeshBuffer* buffer, size_t componentOffset, bool mustBeVisible) -> const char* {
        if (!buffer || !buffer->meshSystem)
            return "the buffer names no mesh system";
        const size_t size = buffer->meshSystem->getMeshComponentSize();
        const auto& pool = buffer->meshSystem->getMeshComponentPool();
        if (componentOffset % size != 0 || componentOffset / size >= pool.getOccupancy())
            return "componentOffset is not a component of the system the bufferIndex names";
        const auto* c = reinterpret_cast<const MeshRenderComponent*>(reinterpret_cast<const uint8_t*>(pool.getData()) + componentOffset);
        if (!*c->entity || !c->isEnabled)
            return "componentOffset names a free or disabled component";
        if (mustBeVisible && !c->isVisible)
            return "a record of the light pass names a component whose isVisible is false";
        return nullptr;
    };
    // (i) + (iii) + (iv), the light pass's shared arrays
    for (int ui = 0; ui < 2; ui++) {
        const auto& list = ui ? system->getUiSortedMeshes() : system->getTransSortedMeshes();
        const uint32_t n = ui ? system->getUiDrawCount() : system->getTransDrawCount();
        const char* name = ui ? "uiSortedMeshes" : "transSortedMeshes";
        uint64_t sum = 0;
        for (uint32_t b = 0; b < system->getSortedBufferCount(); b++) {
            const auto buffer = system->getSortedBuffers()[b];
            if ((buffer->meshSystem->getMeshRenderType() == MeshRenderType::UI) == (ui != 0))
                sum += buffer->drawCount;
        }
        if (sum != n)
            return std::string(name) + ": the buffers' drawCounts sum to " + std::to_string(sum) + ", the array's draw index is " + std::to_string(n) + " (mesh.cpp:255-259)";
        std::vector<uint32_t> perBuffer(system->getSortedBufferCount(), 0);
        for (uint32_t k = 0; k < n; k++) {
            const SortedMesh& m = list[k];
            if (m.bufferIndex >= system->getSortedBufferCount())
                return std::string(name) + ": a record's bufferIndex is past sortedBuffers (mesh.cpp:252)";
            const auto buffer = system->getSortedBuffers()[m.bufferIndex];
            if ((buffer->meshSystem->getMeshRenderType() == MeshRenderType::UI) != (ui != 0))
                return std::string(name) + ": a record's bufferIndex names a system of the other kind (mesh.cpp:414-421)";
            if (const char* why = owns(buffer, m.componentOffset, true))
                return std::string(name) + ": record " + std::to_string(k) + ": " + why + " (mesh.cpp:170,252)";
            perBuffer[m.bufferIndex]++;
            if (k > 0 && list[k - 1].distanceSq < m.distanceSq)
                return std::string(name) + ": not in descending distanceSq order at record " + std::to_string(k) + " (mesh.hpp:204, mesh.cpp:296-326)";
        }
        for (uint32_t b = 0; b < system->getSortedBufferCount(); b++) {
            const auto buffer = system->getSortedBuffers()[b];
            if ((buffer->meshSystem->getMeshRenderType() == MeshRenderType::UI) == (ui != 0) && perBuffer[b] != buffer->drawCount)
                return std::string(name) + ": sortedBuffers[" + std::to_string(b) + "] counts " + std::to_string((uint32_t)buffer->drawCount) + " draws, the array holds " +
                       std::to_string(perBuffer[b]) + " records with its index (mesh.cpp:255-259)";
        }
    }
    // (ii) + (iii) + (iv), every shadow pass's sorted array
    for (uint32_t s = 0; s < passCount; s++) {
        const auto& buffers = system->getShadowSortedBuffers(s);
        const auto& list = system->getShadowTransMeshes(s);
        const uint32_t n = system->getShadowTransDrawCount(s);
        uint64_t sum = 0;
        for (auto buffer : buffers) {
            if (buffer->meshSystem->getMeshRenderType() != MeshRenderType::Translucent)
                return "shadow pass " + std::to_string(s) + ": its sortedBuffers hold a system that is not Translucent (a UI system takes no index there, mesh.cpp:416-419)";
            sum += buffer->drawCount;
        }
        if (sum != n)
            return "shadow pass " + std::to_string(s) + ": the buffers' drawCounts sum to " + std::to_string(sum) + ", the array's draw index is " + std::to_string(n);
        for (uint32_t k = 0; k < n; k++) {
            const SortedMesh& m = list[k];
            if (m.bufferIndex >= buffers.size())
                return "shadow pass " + std::to_string(s) + ": a record's bufferIndex counts more than the Translucent systems (mesh.cpp:416-419)";
            if (const char* why = owns(buffers[m.bufferIndex], m.componentOffset, false))
                return "shadow pass " + std::to_string(s) + ": record " + std::to_string(k) + ": " + why + " (mesh.cpp:170,252)";
            if (k > 0 && list[k - 1].distanceSq < m.distanceSq)
                return "shadow pass " + std::to_string(s) + ": its sorted array is not in descending distanceSq order (mesh.hpp:204)";
        }
    }
    // (i) + (iii) for the unsorted buffers: records on their own system's components, ascending (OIT: any order)
    for (uint32_t b = 0; b < system->getUnsortedBufferCount(); b++) {
        for (int pass = -1; pass < (int)passCount; pass++) {
            const UnsortedBuffer* buffer = pass < 0 ? system->getUnsortedBuffers()[b] : system->getShadowBuffers(b)[pass];
            const bool sorted = buffer->meshSystem->getMeshRenderType() != MeshRenderType::OIT;  // mesh.cpp:273-277
            const UnsortedMesh* meshes = buffer->meshes();
            for (uint32_t k = 0; k < buffer->drawCount; k++) {
                if (const char* why = owns(buffer, meshes[k].componentOffset, pass < 0))
                    return "unsorted buffer " + std::to_string(b) + " pass " + std::to_string(pass) + ": record " + std::to_string(k) + ": " + why + " (mesh.cpp:170)";
                if (sorted && k > 0 && meshes[k].distanceSq < meshes[k - 1].distanceSq)
                    return "unsorted buffer " + std::to_string(b) + " pass " + std::to_string(pass) + ": not in ascending distanceSq order (mesh.hpp:196, mesh.cpp:270-295)";
            }
        }
    }
    return "";
}

int main(int argc, char** argv)
{
    std::string mode = "cpu";
    uint32_t entities = 10000, ticks = 20, threads = 1;
    bool hier = false, mutate = false, mixed = false, toggle = false, bounds = false, avx2 = false;
    std::string gate;        // --gate never|shadow|reverse|empty (see the head of this file)
    uint32_t ranks = 1;      // --ranks R: the drop-in's multi-GPU mode, R contexts driven by this one thread
    bool sameFrame = false;      // --same-frame: both systems in one Manager::update() (see the head of this file)
    bool communicator = false;   // --communicator (with --ranks): the ranks' lists travel through a communicator (gv_exchange_init_all: RCCL, or the tests' transport named
                                 // by GV_RCCL_LIBRARY) instead of the drop-in's default for the devices of one process, peer stores (gv_exchange_init_peers)
    bool probeExchange = false;  // --probe-exchange (with --ranks): the drop-in times the three travel patterns on its first frame and keeps the fastest
    bool noRebin = false;        // --no-rebin (with --ranks): roots that cross cells stay on their rank (the balance decays; results are the same)
    bool unversioned = false;  // --unversioned: the mesh systems carry no change counters (like every mesh system of the reference)
    bool hiz = false;        // --hiz: a depth image with walls is handed to both systems: the light pass of the non-UI systems runs the
                             // per-AABB occlusion query behind the frustum test (with --ranks: the pyramid is built on every rank)
    bool csmPasses = false;  // --csm: three cascades from calcLightViewProj (csm_lite.hpp) as the shadow passes
    int skipPass = -1;       // --skip-pass K: the shadow system's prepareShadowRender says no for pass K (renderShadows, mesh.cpp:812-813:
                             // `continue`): the pass is not prepared, the others keep their numbers — isDrawReady is asked with THOSE
    float animateStep = 0.25f;  // --animate-step S: how far a moved entity goes per tick (large steps: roots cross cells, with --ranks their trees change ranks)
    uint32_t animate = 0;  // --animate K: before every tick, every K-th entity moves (a dynamic scene: the mirror follows every frame)
    bool world = false;     // --world: the GPU system keeps the world-matrix cache (incremental sweep); every compared tick
                            // checks gv_get_world of every transform slot against the oracle's chain walk, bit for bit
    bool itemised = false;  // --itemised: --animate reports the moved entities one by one (TransformSystem::markMoved)
    bool copyRecords = false;  // --copy-records: records arrive in the library's buffer and are copied into combinedMeshes (no record target)
    bool spanRecords = false;  // --span-records: the render passes read the library's page-locked buffer (UnsortedBuffer::meshes()), no copy
    bool soaRecords = false;  // --soa-records: the GPU system fetches three arrays and fills combinedMeshes itself (no record layout)
    uint32_t churn = 0;  // --churn R: R extra rounds that destroy and create entities (itemised: no mirror rebuild asked for)
    uint64_t seed = 0;   // --seed S: another world and another sequence of mutations (tools/tick_soak.sh)
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--mode" && i + 1 < argc) mode = argv[++i];
        else if (a == "--entities" && i + 1 < argc) entities = (uint32_t)atoi(argv[++i]);
        else if (a == "--ticks" && i + 1 < argc) ticks = (uint32_t)atoi(argv[++i]);
        else if (a == "--threads" && i + 1 < argc) threads = (uint32_t)atoi(argv[++i]);
        else if (a == "--hier") hier = true;
        else if (a == "--mutate") mutate = true;
        else if (a == "--mixed") mixed = true;
        else if (a == "--churn" && i + 1 < argc) churn = (uint32_t)atoi(argv[++i]);
        else if (a == "--seed" && i + 1 < argc) seed = strtoull(argv[++i], nullptr, 10);
        else if (a == "--animate" && i + 1 < argc) animate = (uint32_t)atoi(argv[++i]);
        else if (a == "--animate-step" && i + 1 < argc) animateStep = (float)atof(argv[++i]);
        else if (a == "--probe-exchange") probeExchange = true;
        else if (a == "--communicator") communicator = true;
        else if (a == "--same-frame") sameFrame = true;
        else if (a == "--no-rebin") noRebin = true;
        else if (a == "--csm") csmPasses = true;
        else if (a == "--skip-pass" && i + 1 < argc) skipPass = atoi(argv[++i]);
        else if (a == "--gate" && i + 1 < argc) gate = argv[++i];
        else if (a == "--non-translucent") g_nonTranslucent = true;
        else if (a == "--hiz") hiz = true;
        else if (a == "--unversioned") unversioned = true;
        else if (a == "--ranks" && i + 1 < argc) ranks = (uint32_t)atoi(argv[++i]);
        else if (a == "--world") world = true;
        else if (a == "--itemised") itemised = true;
        else if (a == "--soa-records") soaRecords = true;
        else if (a == "--copy-records") copyRecords = true;
        else if (a == "--span-records") spanRecords = true;
        else if (a == "--avx2") avx2 = true;  // CPU system: AVX2+FMA SoA path (bit-identical to the scalar loop)
        else if (a == "--bounds") bounds = true;  // GV_CONFIG_BLOCK_BOUNDS in the GPU system
        else if (a == "--toggle") toggle = mutate = true;  // second round: only setActive / setParent (ranged re-mirror)
    }
    try {
        Manager manager;
        auto transformSystem = manager.createSystem<TransformSystem>();
        manager.registerComponents<TransformComponent>(transformSystem);
        auto graphicsSystem = manager.createSystem<GraphicsSystem>();
        manager.createSystem<DeferredRenderSystem>();
        auto meshSystem = manager.createSystem<OpaqueMeshSystem>();
        manager.registerComponents<MeshRenderComponent>(meshSystem);
        OitMeshSystem* oitSystem = nullptr;
        RefractedMeshSystem* refrSystem = nullptr;
        TransDepthMeshSystem* tdSystem = nullptr;
        TranslucentMeshSystem* transSystem = nullptr;
        GlassMeshSystem* glassSystem = nullptr;
        UiMeshSystem* uiSystem = nullptr;
        if (mixed) {  // creation order = order in meshSystems = bufferIndex order (mesh.cpp:69-108)
            transSystem = manager.createSystem<TranslucentMeshSystem>();
            manager.registerComponents<TranslucentMeshComponent>(transSystem);
            oitSystem = manager.createSystem<OitMeshSystem>();
            manager.registerComponents<OitMeshComponent>(oitSystem);
            uiSystem = manager.createSystem<UiMeshSystem>();
            manager.registerComponents<UiMeshComponent>(uiSystem);
            glassSystem = manager.createSystem<GlassMeshSystem>();
            manager.registerComponents<GlassMeshComponent>(glassSystem);
            refrSystem = manager.createSystem<RefractedMeshSystem>();
            manager.registerComponents<RefractedMeshComponent>(refrSystem);
            tdSystem = manager.createSystem<TransDepthMeshSystem>();
            manager.registerComponents<TransDepthMeshComponent>(tdSystem);
            // the gate of mesh.cpp:426 / :482 (see the head of this file)
            if (gate == "never")
                oitSystem->readyMain = refrSystem->readyMain = glassSystem->readyMain = false,
                oitSystem->readyShadowMask = refrSystem->readyShadowMask = glassSystem->readyShadowMask = 0;
            else if (gate == "shadow")
                meshSystem->readyShadowMask = transSystem->readyShadowMask = ~2u;
            else if (gate == "reverse")
                tdSystem->readyMain = false;
        }
        if (!gate.empty() && !mixed) {
            printf("{\"ok\": false, \"why\": \"--gate needs --mixed\"}\n");
            return 1;
        }
        CpuMeshRenderSystem* cpu = nullptr;
        GpuVisibilitySystem* gpu = nullptr;
        if (mode == "cpu" || mode == "both") {
            cpu = manager.createSystem<CpuMeshRenderSystem>();
            cpu->threads = threads;
            cpu->useAvx2 = avx2;
        }
        MidFrameHook* hook = mode == "both" ? manager.createSystem<MidFrameHook>() : nullptr;
        // what the devices held after the gather of each (mesh system, pass) of the last tick: the union of the ranks' rows
        std::map<std::pair<uint32_t, int>, std::vector<uint32_t>> gathered;
        std::string gatherProblem;
        uint64_t exchangesSeen = 0;
        if (mode == "gpu" || mode == "both") {
            if (ranks > 1)
                gpu = manager.createSystem<GpuVisibilitySystem>(std::vector<int>(ranks, 0), (double)(100.0f * std::cbrt((float)entities)), false, bounds);
            else
                gpu = manager.createSystem<GpuVisibilitySystem>(0, false, bounds);
        }
        // (a timing run — --mode gpu with GV_TICK_BREAKDOWN — does not read every rank's rows back every frame: that check, ~10 ms per
        // frame at 10^6 entities, is --mode both's, which the tests run)
        if (gpu && ranks > 1 && (mode == "both" || !getenv("GV_TICK_BREAKDOWN")))
            gpu->onGathered = [&](const GpuVisibilitySystem::GatheredList* lists, uint32_t listCount, const GvExchangeFrame* frames, uint32_t n) {
                // ONE frame of the exchange carries every list of the tick: row q = [listCount + total, c_0 .. c_{listCount-1}, list 0, list 1 ...]
                std::vector<std::vector<uint32_t>> rows(n);
                for (uint32_t r = 0; r < n; r++) {
                    const GvExchangeFrame& f = frames[r];
                    if (!f.complete || !f.gathered_device || f.world_size != n || f.items != listCount || !f.item_counts) {
                        gatherProblem = "a gathered frame is not complete";
                        return;
                    }
                    rows[r].resize((size_t)n * f.row_words);
                    if (hipStreamSynchronize((hipStream_t)gv_stream(gpu->getContext(r))) != hipSuccess ||  // (acquire ordered the stream behind the rows)
                        hipMemcpy(rows[r].data(), f.gathered_device, rows[r].size() * 4, hipMemcpyDeviceToHost) != hipSuccess) {
                        gatherProblem = "reading the gathered rows back failed";
                        return;
                    }
                }
                std::vector<std::vector<uint32_t>> all(listCount);
                for (uint32_t q = 0; q < n; q++) {
                    const uint32_t* mine = rows[0].data() + (size_t)q * frames[0].row_words;
                    for (uint32_t r = 1; r < n; r++) {
                        const uint32_t* theirs = rows[r].data() + (size_t)q * frames[r].row_words;
                        if (theirs[0] != mine[0] || memcmp(theirs + 1, mine + 1, (size_t)mine[0] * 4) != 0)
                            gatherProblem = "ranks hold different rows after the gather";
                    }
                    if (mine[0] != frames[0].counts[q])
                        gatherProblem = "a row's header is not the count the frame reports";
                    uint32_t at = 1 + listCount, sum = 0;
                    for (uint32_t i = 0; i < listCount; i++) {
                        const uint32_t c = mine[1 + i];
                        if (c != frames[0].item_counts[(size_t)q * listCount + i])
                            gatherProblem = "a row's count table is not what the frame reports";
                        if (at + c > 1 + mine[0]) {
                            gatherProblem = "a row's count table runs past its header";
                            return;
                        }
                        all[i].insert(all[i].end(), mine + at, mine + at + c);
                        at += c;
                        sum += c;
                    }
                    if (mine[0] != listCount + sum)
                        gatherProblem = "a row's header is not the size of its table and lists";
                }
                for (uint32_t i = 0; i < listCount; i++) {
                    std::sort(all[i].begin(), all[i].end());
                    gathered[{lists[i].meshSystemIndex, (int)lists[i].shadowPass}] = std::move(all[i]);
                }
                exchangesSeen++;
            };
        if (cpu) cpu->isNonTranslucent = g_nonTranslucent;
        if (gpu) {
            gpu->isNonTranslucent = g_nonTranslucent;
            gpu->recordStructs = !soaRecords;
            gpu->recordTargets = !copyRecords;
            gpu->recordSpans = spanRecords;
            gpu->probeExchangeMode = probeExchange;
            if (communicator)
                gpu->exchangeTransport = GpuVisibilitySystem::ExchangeTransport::Communicator;
            gpu->rebinMovedRoots = !noRebin;
        }
        if (gpu && world)
            gpu->sweepWorldMatrices = gpu->sweepIncremental = true;
        if (unversioned)
            for (auto ms : allMeshSystems(manager))
                if (auto versioned = dynamic_cast<VersionedMeshSystem*>(ms))
                    versioned->reportsChanges = false;
        manager.initialize();

        // scene: SURVEY.md §8d distribution (cube side 100 * N^(1/3), scale [0.5,2], half-extent [0.25,1])
        Rng rng;
        rng.state += seed * 0x9E3779B97F4A7C15ull;
        const float side = 100.0f * std::cbrt((float)entities);
        std::vector<ID<Entity>> ents;
        auto spawn = [&](uint32_t i) {
            auto e = manager.createEntity();
            ents.push_back(e);
            auto t = transformSystem->add(e);
            t->setPosition(rng.uniform(-0.5f * side, 0.5f * side), rng.uniform(-0.5f * side, 0.5f * side),
                           rng.uniform(-0.5f * side, 0.5f * side));
            t->setScale(rng.uniform(0.5f, 2.0f), rng.uniform(0.5f, 2.0f), rng.uniform(0.5f, 2.0f));
            float q[4] = {rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-1, 1)};
            const float inv = 1.0f / std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + 1e-12f);
            t->setRotation(quat(q[0] * inv, q[1] * inv, q[2] * inv, q[3] * inv));
            t->uid = i + 1;
            MeshRenderComponent* m = nullptr;
            const uint32_t kind = mixed ? i % 8 : 0;
            if (kind == 2) m = *refrSystem->add(e);
            else if (kind == 3) m = *tdSystem->add(e);
            else if (kind == 4) m = *oitSystem->add(e);
            else if (kind == 5) m = *transSystem->add(e);
            else if (kind == 6) m = *glassSystem->add(e);
            else if (kind == 7) {
                m = *uiSystem->add(e);  // UI space: the ortho box of calcUiProjView (mesh.cpp:851-859), some outside
                t->setPosition(rng.uniform(-1200, 1200), rng.uniform(-600, 600), rng.uniform(-1.5f, 1.5f));
            } else m = *meshSystem->add(e);
            const float hx = rng.uniform(0.25f, 1.0f), hy = rng.uniform(0.25f, 1.0f), hz = rng.uniform(0.25f, 1.0f);
            m->aabb.min = f32x4(-hx, -hy, -hz);
            m->aabb.max = f32x4(hx, hy, hz);
            const uint32_t r = rng.next() % 100;
            if (r == 0) m->isEnabled = false;
            if (r == 1) m->aabb.max = m->aabb.min;
            return e;
        };
        for (uint32_t i = 0; i < entities; i++)
            spawn(i);
        if (hier)  // every entity beyond the first tenth gets a parent among earlier entities: depth ~4
            for (uint32_t i = entities / 10; i < entities; i++) {
                if (mixed && i % 8 == 7)
                    continue;  // UI entities stay in UI space
                auto t = transformSystem->tryGetOf(ents[i]);
                t->setPosition(rng.uniform(-40, 40), rng.uniform(-40, 40), rng.uniform(-40, 40));
                transformSystem->setParent(ents[i], ents[rng.next() % (i / 4 + 1)]);
            }
        for (uint32_t i = 0; i < entities; i += 97)
            transformSystem->setActive(ents[i], false);
        if (gate == "empty") {  // two systems lose every component: occupancy stays, getCount() == 0 once the frame has disposed of them
            for (uint32_t i = 0; i < entities; i++)
                if (i % 8 == 4) oitSystem->removeOf(ents[i]);
                else if (i % 8 == 6) glassSystem->removeOf(ents[i]);
        }

        // camera: looks down +z from the origin, FOV 90, 16:9, near 0.01, infinite reversed-Z (camera.hpp:111-121)
        f32x4x4 viewProj;
        memset(viewProj.m, 0, sizeof(viewProj.m));
        viewProj.m[0] = 9.0f / 16.0f; viewProj.m[5] = -1.0f; viewProj.m[11] = 1.0f; viewProj.m[14] = 0.01f;
        graphicsSystem->setCamera(viewProj, f32x4(0, 0, 0));

        // --mixed: a 2000 x 1000 UI canvas and two orthographic shadow passes (csm.cpp:260-343 shapes: reversed-Z ortho
        // boxes around the camera, cameraOffset = light-space shift of the cascade centre)
        uint32_t passCount = 0;
        if (mixed) {
            if (gpu) gpu->setUiSize(2000.0f, 1000.0f);
            f32x4x4 ui;
            memset(ui.m, 0, sizeof(ui.m));
            ui.m[0] = 2.0f / 2000.0f; ui.m[5] = -2.0f / 1000.0f; ui.m[10] = -0.5f; ui.m[14] = 0.5f; ui.m[15] = 1.0f;
            if (cpu) cpu->uiViewProj = ui;
            std::vector<GpuVisibilitySystem::ShadowPass> gpuPasses;
            std::vector<CpuMeshRenderSystem::ShadowPass> cpuPasses;
            for (int c = 0; c < 2; c++) {
                if (c == skipPass)
                    continue;
                const float size = side * (c == 0 ? 0.25f : 0.6f), nearPlane = -side, farPlane = side;
                f32x4x4 vp;
                memset(vp.m, 0, sizeof(vp.m));
                vp.m[0] = 2.0f / size; vp.m[5] = -2.0f / size; vp.m[10] = -1.0f / (farPlane - nearPlane);
                vp.m[14] = farPlane / (farPlane - nearPlane); vp.m[15] = 1.0f;
                const f32x4 offset(3.0f * (float)(c + 1), -7.0f, 11.0f);
                gpuPasses.push_back({vp, offset, (int8_t)c});
                cpuPasses.push_back({vp, offset, (int8_t)c});
                g_passIndex.push_back((int8_t)c);
            }
            if (gpu) gpu->setShadowPasses(gpuPasses);
            if (cpu) cpu->setShadowPasses(cpuPasses);
            passCount = (uint32_t)gpuPasses.size();
        }
        if (csmPasses) {
            g_passIndex.clear();
            // CsmRenderSystem::prepareShadowRender (csm.cpp:308-325) for the camera above (identity view, FOV 90, 16:9,
            // near 0.01): three cascades over the nearest eighth of the world, light from above and behind
            const float splits[2] = {0.1f, 0.35f}, distance = 0.125f * side, zCoeff = 10.0f;
            const uint32_t cascades = 3, mapSize = 2048;
            const float l = std::sqrt(0.3f * 0.3f + 0.8f * 0.8f + 0.5f * 0.5f);
            const f32x4 lightDir(0.3f / l, -0.8f / l, 0.5f / l);
            std::vector<GpuVisibilitySystem::ShadowPass> gpuPasses;
            std::vector<CpuMeshRenderSystem::ShadowPass> cpuPasses;
            float nearPlane = 0.01f;
            for (uint32_t c = 0; c < cascades; c++) {
                const auto cs = csm::cascade(c, cascades, splits, distance, f32x4x4(), lightDir, 1.5707963f, 16.0f / 9.0f, 0.01f, zCoeff, mapSize);
                if ((int)c != skipPass) {
                    gpuPasses.push_back({cs.viewProj, cs.cameraOffset, (int8_t)c});
                    cpuPasses.push_back({cs.viewProj, cs.cameraOffset, (int8_t)c});
                    g_passIndex.push_back((int8_t)c);
                }
                // property of calcLightViewProj: the slice's corners lie in the light's box (up to the texel snap)
                const float farPlane = c + 1 < cascades ? distance * splits[c] : distance;
                const float eps = 4.0f / (float)mapSize + 1e-3f;
                for (int k = 0; k < 8; k++) {
                    const float z = (k & 4) ? farPlane : nearPlane;
                    const float x = ((k & 1) ? 1.0f : -1.0f) * z * (16.0f / 9.0f), y = ((k & 2) ? 1.0f : -1.0f) * z;  // tan(45 deg) = 1
                    const float* m = cs.viewProj.m;
                    const float cx = m[0] * x + m[4] * y + m[8] * z + m[12], cy = m[1] * x + m[5] * y + m[9] * z + m[13];
                    const float cz = m[2] * x + m[6] * y + m[10] * z + m[14], cw = m[3] * x + m[7] * y + m[11] * z + m[15];
                    if (!(std::fabs(cx) <= (1 + eps) * cw && std::fabs(cy) <= (1 + eps) * cw && cz >= -eps * cw && cz <= (1 + eps) * cw && cw > 0)) {
                        printf("{\"ok\": false, \"why\": \"cascade %u: slice corner %d outside the light box (%g %g %g %g)\"}\n", c, k, cx, cy, cz, cw);
                        return 1;
                    }
                }
                nearPlane = farPlane;
            }
            if (gpu) gpu->setShadowPasses(gpuPasses);
            if (cpu) cpu->setShadowPasses(cpuPasses);
            passCount = (uint32_t)gpuPasses.size();
        }

        if (hiz) {  // reversed-Z depth (1 = near, 0 = far): an empty background with a few walls at different depths, 480 x 272
            const uint32_t dw = 480, dh = 272;
            std::vector<float> depth((size_t)dw * dh, 0.0f);
            Rng walls;
            for (int k = 0; k < 24; k++) {
                const uint32_t x0 = walls.next() % dw, y0 = walls.next() % dh, w = 20 + walls.next() % 120, h = 16 + walls.next() % 90;
                const float d = walls.uniform(0.00002f, 0.002f);  // near / distance: walls 5 .. 500 units away (near plane 0.01)
                for (uint32_t y = y0; y < std::min(dh, y0 + h); y++)
                    for (uint32_t x = x0; x < std::min(dw, x0 + w); x++)
                        depth[(size_t)y * dw + x] = std::max(depth[(size_t)y * dw + x], d);
            }
            if (cpu) cpu->setHizDepth(depth.data(), dw, dh);
            if (gpu) gpu->setHizDepth(depth.data(), dw, dh);
        }
        uint32_t animateTick = 0;
        auto run = [&](bool useCpu, bool useGpu, uint32_t n, bool moving = true) {
            if (cpu) cpu->isEnabled = useCpu;
            if (gpu) gpu->isEnabled = useGpu;
            auto t0 = std::chrono::steady_clock::now();
            for (uint32_t i = 0; i < n; i++) {
                if (animate && moving) {  // timed with the tick: the engine's own systems would be doing this
                    for (uint32_t k = animateTick % animate; k < (uint32_t)ents.size(); k += animate)
                        if (auto t = transformSystem->tryGetOf(ents[k])) {
                            t->posChildCount.x += animateStep;
                            if (std::fabs(t->posChildCount.x) > 0.5f * side)  // (stay inside the world cube: wrap around)
                                t->posChildCount.x -= std::copysign(side, t->posChildCount.x);
                            if (itemised)
                                transformSystem->markMoved(ents[k]);
                        }
                    if (!itemised)
                        transformSystem->markTransformsChanged();
                    animateTick++;
                }
                manager.update();
            }
            return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        };
        std::vector<bool> gone(entities, false);  // destroyed: its handle in ents[] is stale (the id may be handed out again)
        auto doMutate = [&]() {
            if (toggle) {  // nothing but active-flag and parent-link changes: the systems' ranged update paths
                for (uint32_t i = 5; i < entities; i += 53)
                    transformSystem->setActive(ents[i], (i & 1) != 0);
                for (uint32_t i = 0; i < entities; i += 97)
                    transformSystem->setActive(ents[i], true);
                if (hier)
                    for (uint32_t i = entities / 2; i < entities; i += 31)
                        transformSystem->setParent(ents[i], ents[i / 8]);
                return;
            }
            for (uint32_t i = 3; i < entities; i += 11) {
                auto t = transformSystem->tryGetOf(ents[i]);
                if (t) t->setPosition(rng.uniform(-0.5f * side, 0.5f * side), rng.uniform(-50, 50), rng.uniform(0, 0.5f * side));
            }
            transformSystem->markTransformsChanged();
            for (uint32_t i = 5; i < entities; i += 53)
                transformSystem->setActive(ents[i], (i & 1) != 0);
            if (hier)
                for (uint32_t i = entities / 2; i < entities; i += 31)
                    transformSystem->setParent(ents[i], ents[i / 8]);
            for (uint32_t i = 7; i < entities; i += 101) {
                manager.destroy(ents[i]);
                gone[i] = true;  // (a later churn round must not take the stale handle for a live parent: with the id recycled it
                                 //  could be the very entity being parented — "setParent: cycle", tools/tick_soak.sh seed 10)
            }
            meshSystem->markMeshesChanged();
            if (mixed) {
                transSystem->markMeshesChanged(); oitSystem->markMeshesChanged();
                uiSystem->markMeshesChanged(); glassSystem->markMeshesChanged();
                refrSystem->markMeshesChanged(); tdSystem->markMeshesChanged();
            }
            // destroying entities needs no rebuild request: TransformSystem / the mesh systems itemise the slots
            graphicsSystem->setCamera(viewProj, f32x4(12.5f, -3.0f, 40.0f));
        };

        bool ok = true;
        std::string why;
        double seconds = 0;
        uint32_t drawCount = 0, sortedDrawCount = 0;
        auto doChurn = [&]() {  // ~1 % destroyed, ~2 % created (some under existing parents), a few flags flipped
            const uint32_t count = (uint32_t)ents.size();
            for (uint32_t k = 0; k < count / 100 + 1; k++) {
                const uint32_t i = rng.next() % count;
                if (!gone[i]) {
                    manager.destroy(ents[i]);
                    gone[i] = true;
                }
            }
            for (uint32_t k = 0; k < count / 50 + 1; k++) {
                const uint32_t i = (uint32_t)ents.size();
                auto e = spawn(i);
                gone.push_back(false);
                if (hier && (k & 1)) {
                    const uint32_t parent = rng.next() % count;
                    if (!gone[parent] && transformSystem->tryGetOf(ents[parent]))
                        transformSystem->setParent(e, ents[parent]);
                }
            }
            for (uint32_t k = 0; k < 20; k++) {
                const uint32_t i = rng.next() % (uint32_t)ents.size();
                if (!gone[i])
                    transformSystem->setActive(ents[i], (k & 1) != 0);
            }
        };
        if (gate == "empty")
            run(false, false, 1);  // (one frame with neither system: the removed components are disposed of at its end)
        int rounds = (mutate ? 2 : 1) + (int)churn;
        for (int round = 0; round < rounds && ok; round++) {
            if (round == 1 && mutate)
                doMutate();
            else if (round >= 1)
                doChurn();
            if (round >= 1 && (!gate.empty() || g_nonTranslucent))
                run(false, false, 1);  // (components destroyed above are disposed of at the end of a frame — T(): isVisible = false — also in
                                       //  pools no system draws: that frame is kept out of the two ticks whose isVisible bytes are compared)
            if (mode == "both") {
                // --animate: every tick is compared (the scene moves, the CPU system ticks, the GPU system ticks on the
                // same state through its dirty-range path); otherwise one CPU tick against `ticks` GPU ticks
                const uint32_t compared = animate ? ticks : 1;
                Snapshot a, b;
                for (uint32_t c = 0; c < compared && ok; c++) {
                    std::string cpuGate, gpuGate;
                    poisonVisible(manager);
                    if (sameFrame) {  // one frame: the CPU system, its snapshot, the pattern again, the GPU drop-in
                        hook->between = [&]() {
                            a = snapshot(manager, cpu, passCount);
                            cpuGate = gateHolds(manager, cpu, passCount);
                            if (cpuGate.empty())
                                cpuGate = sortedArraysHold(manager, cpu, passCount);
                            poisonVisible(manager);
                        };
                        seconds += run(true, true, 1);
                        hook->between = nullptr;
                    } else {
                        run(true, false, 1);
                        a = snapshot(manager, cpu, passCount);
                        cpuGate = gateHolds(manager, cpu, passCount);
                        if (cpuGate.empty())
                            cpuGate = sortedArraysHold(manager, cpu, passCount);
                        poisonVisible(manager);
                        seconds += run(false, true, 1, false);
                    }
                    gpuGate = gateHolds(manager, gpu, passCount);  // (after ONE tick from the pattern: a later tick
                    if (gpuGate.empty())                           //  would find a non-drawn system's bytes unchanged anyway)
                        gpuGate = sortedArraysHold(manager, gpu, passCount);
                    if (!animate && ticks > 1)
                        seconds += run(false, true, ticks - 1, false);
                    b = snapshot(manager, gpu, passCount);
                    if (ranks > 1 && ok) {
                        // the gather against what the render phase will read: for every unsorted mesh system and pass, the union of
                        // the rows every device holds == the WORLD slots of the records in that pass's buffer
                        if (!gatherProblem.empty()) {
                            ok = false;
                            why = gatherProblem;
                        }
                        auto systems = allMeshSystems(manager);
                        for (uint32_t bi = 0; bi < gpu->getUnsortedBufferCount() && ok; bi++) {
                            auto buffer = gpu->getUnsortedBuffers()[bi];
                            const uint32_t p = (uint32_t)(std::find(systems.begin(), systems.end(), buffer->meshSystem) - systems.begin());
                            for (int pass = -1; pass < (int)passCount && ok; pass++) {
                                const UnsortedBuffer* pb = pass < 0 ? buffer : gpu->getShadowBuffers(bi)[pass];
                                std::vector<uint32_t> slots(pb->drawCount);
                                for (uint32_t k = 0; k < pb->drawCount; k++)
                                    slots[k] = (uint32_t)(pb->meshes()[k].componentOffset / buffer->meshSystem->getMeshComponentSize());
                                std::sort(slots.begin(), slots.end());
                                auto it = gathered.find({p, pass});
                                if (it == gathered.end() ? !slots.empty() : it->second != slots) {
                                    ok = false;
                                    why = "the rows gathered on the devices are not the world slots of the pass's records (mesh system " + std::to_string(p) +
                                          ", pass " + std::to_string(pass) + ")";
                                }
                            }
                        }
                        gathered.clear();
                    }
                    // first each system against the reference text, then the two against each other
                    if (!cpuGate.empty()) {
                        ok = false;
                        why = "CPU system vs the text of mesh.cpp:187-328,408-490: " + cpuGate;
                    } else if (!gpuGate.empty()) {
                        ok = false;
                        why = "GPU system vs the text of mesh.cpp:187-328,408-490: " + gpuGate;
                    } else {
                        ok = same(a, b, why);
                    }
                    if (ok && world) {  // the kept cache == TransformComponent::calcModel() of every slot (transform.hpp:197-214)
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
                        std::vector<float> want((size_t)tp.occupancy * 12), got((size_t)tp.occupancy * 12);
                        gvo_world_matrices(&tp, 0, tp.occupancy, want.data());
                        if (gv_get_world(gpu->getContext(), 0, tp.occupancy, got.data()) != GV_OK) {
                            ok = false;
                            why = std::string("gv_get_world: ") + gv_last_error(gpu->getContext());
                        } else if (memcmp(want.data(), got.data(), want.size() * 4) != 0) {
                            ok = false;
                            why = "world-matrix cache differs from the oracle's chain walk";
                        }
                    }
                }
                if (ok && !a.ordered) {
                    ok = false;
                    why = "CPU system: " + a.disorder + " not in sortMeshes order";
                }
                if (ok && !b.ordered) {  // gv_sort (+ run merge) == sortMeshes order
                    ok = false;
                    why = "GPU system: " + b.disorder + " not in sortMeshes order";
                }
                drawCount = gpu->getUnsortedBuffers()[0]->drawCount;
                sortedDrawCount = gpu->getTransDrawCount() + gpu->getUiDrawCount();
            } else {
                if (round == 0) {  // untimed first tick: the GPU system builds its device mirror, the CPU system its scratch
                    poisonVisible(manager);
                    run(mode == "cpu", mode == "gpu", 1);
                    if (gpu) gpu->tickSeconds = {};
                    // the gate of mesh.cpp:426 / :482 against the reference text, for the one system that ran
                    std::string held = cpu ? gateHolds(manager, cpu, passCount) : gateHolds(manager, gpu, passCount);
                    if (held.empty())
                        held = cpu ? sortedArraysHold(manager, cpu, passCount) : sortedArraysHold(manager, gpu, passCount);
                    if (!held.empty()) {
                        ok = false;
                        why = std::string(cpu ? "CPU" : "GPU") + " system vs the text of mesh.cpp:187-328,408-490: " + held;
                        break;
                    }
                }
                seconds += run(mode == "cpu", mode == "gpu", ticks);
                drawCount = (cpu ? cpu->getUnsortedBuffers()[0] : gpu->getUnsortedBuffers()[0])->drawCount;
                sortedDrawCount = cpu ? cpu->getTransDrawCount() + cpu->getUiDrawCount() : gpu->getTransDrawCount() + gpu->getUiDrawCount();
            }
        }
        uint32_t visibleFlags = 0;
        for (uint32_t i = 0; i < meshSystem->getComponents().getOccupancy(); i++)
            visibleFlags += meshSystem->getComponents().getData()[i].isVisible ? 1 : 0;
        std::string shadowCounts = "[";
        for (uint32_t pass = 0; pass < passCount; pass++) {
            const auto& sb = cpu ? cpu->getShadowBuffers(0) : gpu->getShadowBuffers(0);
            shadowCounts += std::string(pass ? ", " : "") + std::to_string(pass < sb.size() ? (uint32_t)sb[pass]->drawCount : 0u);
        }
        shadowCounts += "]";
        if (gpu && ranks > 1) {
            std::string shares = "[";
            for (uint32_t r = 0; r < ranks; r++)
                shares += std::string(r ? ", " : "") + std::to_string(gpu->getRankShares().shares[r].transforms.size());
            fprintf(stderr, "ranks %u: transforms per rank %s]\n", ranks, shares.c_str());
        }
        std::string rankJson;
        if (gpu && ranks > 1) {
            const auto& c = gpu->rankCounters;
            char text[512];
            snprintf(text, sizeof(text), "\"ranks\": %u, \"rank_frames\": %llu, \"exchanges\": %llu, \"exchanges_seen\": %llu, \"deals\": %llu, \"moved_trees\": %llu, "
                     "\"moved_transforms\": %llu, \"edited_meshes\": %llu, \"exchange_mode\": %u, \"exchange_mode_probe_ms\": [%.4f, %.4f, %.4f], ", ranks,
                     (unsigned long long)c.frames, (unsigned long long)c.exchanges, (unsigned long long)exchangesSeen, (unsigned long long)c.deals,
                     (unsigned long long)c.movedTrees, (unsigned long long)c.movedTransforms, (unsigned long long)c.editedMeshes, gpu->exchangeMode,
                     gpu->exchangeModeProbeMs[0], gpu->exchangeModeProbeMs[1], gpu->exchangeModeProbeMs[2]);
            rankJson = text;
        }
        printf("{\"mode\": \"%s\", \"entities\": %u, \"ticks\": %u, \"threads\": %u, \"hier\": %s, \"shadow_draw_counts\": %s, \"draw_count\": %u, "
               "\"sorted_draw_count\": %u, \"is_visible_set\": %u, \"culls_per_s\": %.1f, %s\"ok\": %s, \"why\": \"%s\"}\n",
               mode.c_str(), entities, ticks * rounds, threads, hier ? "true" : "false", shadowCounts.c_str(), drawCount, sortedDrawCount, visibleFlags,
               (double)entities * ticks * rounds / seconds, rankJson.c_str(), ok ? "true" : "false", why.c_str());
        if (gpu && getenv("GV_TICK_BREAKDOWN")) {
            const auto& t = gpu->tickSeconds;
            const double n = (double)ticks * rounds * 1e-6;  // -> microseconds per tick
            fprintf(stderr, "gpu prepare us/tick: total %.1f = cull %.1f + sort %.1f + fetch %.1f + records %.1f + shares %.1f + gather %.1f + other %.1f\n", t.total / n,
                    t.cull / n, t.sort / n, t.fetch / n, t.records / n, t.share / n, t.gather / n,
                    (t.total - t.cull - t.sort - t.fetch - t.records - t.share - t.gather) / n);
            if (ranks > 1) {
                const auto& c = gpu->rankCounters;
                GvStats stats{};
                gv_stats(gpu->getContext(0), &stats);
                fprintf(stderr, "ranks: %llu frames, exchanges per frame %.2f (gv_stats of rank 0: %llu sent, %llu with a second exchange), deals %llu, trees moved %llu "
                        "(%llu transforms), mesh slots edited %llu, transforms copied %llu\n", (unsigned long long)c.frames,
                        c.frames ? (double)c.exchanges / (double)c.frames : 0.0, (unsigned long long)stats.exchanges, (unsigned long long)stats.exchange_tail_rounds,
                        (unsigned long long)c.deals, (unsigned long long)c.movedTrees, (unsigned long long)c.movedTransforms, (unsigned long long)c.editedMeshes,
                        (unsigned long long)c.copiedTransforms);
            }
        }
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        printf("{\"ok\": false, \"why\": \"exception: %s\"}\n", e.what());
        return 1;
    }
}
