// headless_tick.cpp — TEST driver: a headless ecsm-style frame loop (Manager::update() firing
// Input -> Update -> Render -> PreDeferredRender, no Vulkan) with either the CPU reference-path system
// (oracle-backed, BASELINE.json configs[0]) or the GPU drop-in (GpuVisibilitySystem over libgarden_vis.so),
// and, in `both` mode, a bit-for-bit comparison of what each leaves behind for the render phase.
//
//   headless_tick --mode cpu|gpu|both [--entities N] [--ticks T] [--threads K] [--hier] [--mutate]
// Prints one JSON line; exit code 0 = ok, 1 = mismatch/failure.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../garden_amd/csrc/host/gpu_visibility_system.hpp"
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

struct Snapshot {
    std::vector<uint8_t> isVisible;
    std::vector<UnsortedMesh> meshes;
    uint32_t drawCount = 0, instanceCount = 0;
};

static bool isSortedByDistance(const UnsortedBuffer* buffer)
{
    for (uint32_t k = 1; k < buffer->drawCount; k++)
        if (buffer->combinedMeshes[k] < buffer->combinedMeshes[k - 1])  // operator< render/mesh.hpp:196
            return false;
    return true;
}

static Snapshot snapshot(OpaqueMeshSystem* meshSystem, const UnsortedBuffer* buffer)
{
    Snapshot s;
    auto data = meshSystem->getComponents().getData();
    for (uint32_t i = 0; i < meshSystem->getComponents().getOccupancy(); i++)
        s.isVisible.push_back(data[i].isVisible);
    s.drawCount = buffer->drawCount;
    s.instanceCount = buffer->instanceCount;
    s.meshes.assign(buffer->combinedMeshes.begin(), buffer->combinedMeshes.begin() + s.drawCount);
    // the reference's order is fetch_add arrival order (mesh.cpp:177): compare as a set, keyed by componentOffset
    std::sort(s.meshes.begin(), s.meshes.end(),
              [](const UnsortedMesh& a, const UnsortedMesh& b) { return a.componentOffset < b.componentOffset; });
    return s;
}

static bool same(const Snapshot& a, const Snapshot& b, std::string& why)
{
    if (a.isVisible != b.isVisible) { why = "isVisible differs"; return false; }
    if (a.drawCount != b.drawCount || a.instanceCount != b.instanceCount) { why = "counters differ"; return false; }
    for (uint32_t k = 0; k < a.drawCount; k++) {
        if (a.meshes[k].componentOffset != b.meshes[k].componentOffset) { why = "componentOffset differs"; return false; }
        if (memcmp(a.meshes[k].bakedModel.m, b.meshes[k].bakedModel.m, 48) != 0) { why = "bakedModel differs"; return false; }
        if (memcmp(&a.meshes[k].distanceSq, &b.meshes[k].distanceSq, 4) != 0) { why = "distanceSq differs"; return false; }
    }
    return true;
}

int main(int argc, char** argv)
{
    std::string mode = "cpu";
    uint32_t entities = 10000, ticks = 20, threads = 1;
    bool hier = false, mutate = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--mode" && i + 1 < argc) mode = argv[++i];
        else if (a == "--entities" && i + 1 < argc) entities = (uint32_t)atoi(argv[++i]);
        else if (a == "--ticks" && i + 1 < argc) ticks = (uint32_t)atoi(argv[++i]);
        else if (a == "--threads" && i + 1 < argc) threads = (uint32_t)atoi(argv[++i]);
        else if (a == "--hier") hier = true;
        else if (a == "--mutate") mutate = true;
    }
    try {
        Manager manager;
        auto transformSystem = manager.createSystem<TransformSystem>();
        manager.registerComponents<TransformComponent>(transformSystem);
        auto graphicsSystem = manager.createSystem<GraphicsSystem>();
        manager.createSystem<DeferredRenderSystem>();
        auto meshSystem = manager.createSystem<OpaqueMeshSystem>();
        manager.registerComponents<MeshRenderComponent>(meshSystem);
        CpuMeshRenderSystem* cpu = nullptr;
        GpuVisibilitySystem* gpu = nullptr;
        if (mode == "cpu" || mode == "both") {
            cpu = manager.createSystem<CpuMeshRenderSystem>();
            cpu->threads = threads;
        }
        if (mode == "gpu" || mode == "both")
            gpu = manager.createSystem<GpuVisibilitySystem>(0, false);
        manager.initialize();

        // scene: SURVEY.md §8d distribution (cube side 100 * N^(1/3), scale [0.5,2], half-extent [0.25,1])
        Rng rng;
        const float side = 100.0f * std::cbrt((float)entities);
        std::vector<ID<Entity>> ents;
        for (uint32_t i = 0; i < entities; i++) {
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
            auto m = meshSystem->add(e);
            const float hx = rng.uniform(0.25f, 1.0f), hy = rng.uniform(0.25f, 1.0f), hz = rng.uniform(0.25f, 1.0f);
            m->aabb.min = f32x4(-hx, -hy, -hz);
            m->aabb.max = f32x4(hx, hy, hz);
            const uint32_t r = rng.next() % 100;
            if (r == 0) m->isEnabled = false;
            if (r == 1) m->aabb.max = m->aabb.min;
        }
        if (hier)  // every entity beyond the first tenth gets a parent among earlier entities: depth ~4
            for (uint32_t i = entities / 10; i < entities; i++) {
                auto t = transformSystem->tryGetOf(ents[i]);
                t->setPosition(rng.uniform(-40, 40), rng.uniform(-40, 40), rng.uniform(-40, 40));
                transformSystem->setParent(ents[i], ents[rng.next() % (i / 4 + 1)]);
            }
        for (uint32_t i = 0; i < entities; i += 97)
            transformSystem->setActive(ents[i], false);

        // camera: looks down +z from the origin, FOV 90, 16:9, near 0.01, infinite reversed-Z (camera.hpp:111-121)
        f32x4x4 viewProj;
        memset(viewProj.m, 0, sizeof(viewProj.m));
        viewProj.m[0] = 9.0f / 16.0f; viewProj.m[5] = -1.0f; viewProj.m[11] = 1.0f; viewProj.m[14] = 0.01f;
        graphicsSystem->setCamera(viewProj, f32x4(0, 0, 0));

        auto run = [&](bool useCpu, bool useGpu, uint32_t n) {
            if (cpu) cpu->isEnabled = useCpu;
            if (gpu) gpu->isEnabled = useGpu;
            auto t0 = std::chrono::steady_clock::now();
            for (uint32_t i = 0; i < n; i++)
                manager.update();
            return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        };
        auto doMutate = [&]() {
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
            }
            meshSystem->markMeshesChanged();
            transformSystem->hierarchyVersion++;  // entities were destroyed: full rebuild
            graphicsSystem->setCamera(viewProj, f32x4(12.5f, -3.0f, 40.0f));
        };

        bool ok = true;
        std::string why;
        double seconds = 0;
        uint32_t drawCount = 0;
        int rounds = mutate ? 2 : 1;
        for (int round = 0; round < rounds && ok; round++) {
            if (round == 1)
                doMutate();
            if (mode == "both") {
                run(true, false, 1);
                Snapshot a = snapshot(meshSystem, cpu->getUnsortedBuffers()[0]);
                for (uint32_t i = 0; i < meshSystem->getComponents().getOccupancy(); i++)
                    meshSystem->getComponents().getData()[i].isVisible = false;
                seconds += run(false, true, ticks);
                const bool sorted = isSortedByDistance(gpu->getUnsortedBuffers()[0]);  // gv_sort == sortMeshes order
                Snapshot b = snapshot(meshSystem, gpu->getUnsortedBuffers()[0]);
                ok = same(a, b, why);
                if (ok && !sorted) {
                    ok = false;
                    why = "combinedMeshes not ascending by distanceSq after gv_sort";
                }
                drawCount = b.drawCount;
            } else {
                seconds += run(mode == "cpu", mode == "gpu", ticks);
                drawCount = (cpu ? cpu->getUnsortedBuffers()[0] : gpu->getUnsortedBuffers()[0])->drawCount;
            }
        }
        uint32_t visibleFlags = 0;
        for (uint32_t i = 0; i < meshSystem->getComponents().getOccupancy(); i++)
            visibleFlags += meshSystem->getComponents().getData()[i].isVisible ? 1 : 0;
        printf("{\"mode\": \"%s\", \"entities\": %u, \"ticks\": %u, \"threads\": %u, \"hier\": %s, \"draw_count\": %u, "
               "\"is_visible_set\": %u, \"culls_per_s\": %.1f, \"ok\": %s, \"why\": \"%s\"}\n",
               mode.c_str(), entities, ticks * rounds, threads, hier ? "true" : "false", drawCount, visibleFlags,
               (double)entities * ticks * rounds / seconds, ok ? "true" : "false", why.c_str());
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        printf("{\"ok\": false, \"why\": \"exception: %s\"}\n", e.what());
        return 1;
    }
}
