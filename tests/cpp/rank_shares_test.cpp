// rank_shares_test.cpp — TEST: garden_amd/csrc/host/rank_shares.hpp (what each rank of the drop-in's multi-GPU mode holds of the
// engine's pools) on the CPU, under AddressSanitizer + UndefinedBehaviorSanitizer: worlds with hierarchies, free slots, meshes
// without a transform, destroyed entities, re-parenting; ranks 1 .. 8. Checked after every deal:
//   * every live transform lives on exactly one rank, on the rank gv_cell_owner gives its ROOT's position; a parent is on its
//     child's rank and the local parent id names it; local entity ids are slot + 1; the copied bytes are the world's
//   * every slot of every mesh pool lives on exactly one rank (free slots and meshes without a transform included), its local entity
//     id resolves to the transform the world's entity resolves to (or to nothing), the world slot table is a permutation
//   * copyTransform refreshes a moved transform in place
// ... and frame to frame, without another deal (round 6): edited mesh components are found and copied (syncMeshes, with and
// without the system saying which), a slot that changed hands is reported as structural, every transform is re-copied on the worker
// threads (syncAllTransforms), and roots that crossed into another rank's cell take their trees — transforms and meshes — to that
// rank (rebin / moveTree: holes are left and reused); the same invariants hold after every step, and what the steps report
// (Changes) names exactly the local slots whose bytes differ from before.
// Links libgarden_vis.so for gv_cell_owner (host-only; no device is touched).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../garden_amd/csrc/host/gpu_visibility_system.hpp"

using namespace garden;

struct alignas(16) WideMeshComponent final : public MeshRenderComponent {
    float payload[8] = {};
};
using WideMeshSystem = MeshSystemOf<WideMeshComponent, MeshRenderType::Translucent>;

static int failures = 0;
#define EXPECT(cond, ...)                          \
    do {                                           \
        if (!(cond)) {                             \
            std::fprintf(stderr, __VA_ARGS__);     \
            std::fprintf(stderr, "\n");            \
            failures++;                            \
        }                                          \
    } while (0)

static void check(const TransformSystem* ts, const std::vector<IMeshRenderSystem*>& meshSystems, const RankShares& shares, uint32_t ranks,
                  const uint32_t grid[3], double side, const char* what, bool ownersMatter = true)
{
    auto& pool = const_cast<TransformSystem*>(ts)->getComponents();
    const auto& emap = ts->getEntityMap();
    const uint32_t occupancy = pool.getOccupancy();
    const TransformComponent* world = pool.getData();
    std::vector<uint32_t> owners(occupancy ? occupancy : 1);
    if (occupancy)
        EXPECT(gv_cell_owner(grid, side, ranks, reinterpret_cast<const float*>(&world[0].posChildCount), sizeof(TransformComponent), occupancy, owners.data()) == GV_OK,
               "%s: gv_cell_owner", what);
    std::vector<uint32_t> seen(occupancy, 0);
    for (uint32_t r = 0; r < ranks; r++) {
        const auto& share = shares.shares[r];
        EXPECT(share.transforms.size() == share.transformWorldSlot.size() && share.entityToTransform.size() == share.transforms.size() + 1, "%s: rank %u sizes", what, r);
        uint32_t holes = 0;
        for (uint32_t k = 0; k < share.transforms.size(); k++) {
            const uint32_t w = share.transformWorldSlot[k];
            const TransformComponent& local = share.transforms[k];
            if (w == GV_NONE) {  // a hole left by a tree that moved away
                EXPECT(*local.entity == 0 && *local.parent == 0, "%s: rank %u hole %u still holds an entity", what, r, k);
                holes++;
                continue;
            }
            seen[w]++;
            EXPECT(shares.rankOfTransform[w] == r && shares.localOfTransform[w] == k, "%s: the tables do not lead to rank %u slot %u", what, r, k);
            EXPECT(*local.entity == k + 1 && share.entityToTransform[k + 1] == k, "%s: rank %u local entity id of slot %u", what, r, k);
            EXPECT(std::memcmp(&local.posChildCount, &world[w].posChildCount, 12) == 0 && std::memcmp(&local.rotation, &world[w].rotation, 16) == 0 &&
                       local.selfActive == world[w].selfActive && local.ancestorsActive == world[w].ancestorsActive, "%s: rank %u slot %u is not the world's transform", what, r, k);
            uint32_t root = w;  // the world's root of w
            for (uint32_t guard = 0; *world[root].parent && guard < occupancy; guard++)
                root = emap[*world[root].parent];
            if (ownersMatter)
                EXPECT(owners[root] == r, "%s: transform %u lives on rank %u, its root's cell belongs to rank %u", what, w, r, owners[root]);
            if (*world[w].parent) {
                const uint32_t parentSlot = emap[*world[w].parent];
                EXPECT(shares.rankOfTransform[parentSlot] == r && *local.parent == shares.localOfTransform[parentSlot] + 1, "%s: parent link of transform %u", what, w);
            } else {
                EXPECT(*local.parent == 0, "%s: a root with a parent (world slot %u entity %u, rank %u local %u, local parent id %u)", what, w, *world[w].entity, r, k, *local.parent);
            }
        }
        EXPECT(holes == share.freeTransforms.size(), "%s: rank %u has %u holes and %zu free slots", what, r, holes, share.freeTransforms.size());
    }
    for (uint32_t i = 0; i < occupancy; i++)
        EXPECT(seen[i] == (*world[i].entity ? 1u : 0u), "%s: transform slot %u is held %u times", what, i, seen[i]);
    // the meshes that wait for a transform: each knows its place in the list, and the list holds nothing else
    size_t flagged = 0;
    for (const auto& table : shares.meshTables)
        for (uint32_t t : table.transform)
            flagged += t == RankShares::kLoose ? 1 : 0;
    EXPECT(flagged == shares.looseMeshes.size(), "%s: %zu meshes are marked as waiting for a transform, the list holds %zu", what, flagged, shares.looseMeshes.size());
    for (size_t k = 0; k < shares.looseMeshes.size(); k++) {
        const uint32_t ref = shares.looseMeshes[k], p = ref >> 28, j = ref & 0x0FFFFFFFu;
        EXPECT(p < shares.meshTables.size() && j < shares.meshTables[p].transform.size() && shares.meshTables[p].transform[j] == RankShares::kLoose &&
                   shares.meshTables[p].next[j] == k, "%s: entry %zu of the waiting list", what, k);
    }
    for (size_t p = 0; p < meshSystems.size(); p++) {
        const auto& meshPool = meshSystems[p]->getMeshComponentPool();
        const size_t stride = meshSystems[p]->getMeshComponentSize();
        const uint8_t* data = reinterpret_cast<const uint8_t*>(meshPool.getData());
        std::vector<uint32_t> held(meshPool.getOccupancy(), 0);
        for (uint32_t r = 0; r < ranks; r++) {
            const auto& mesh = shares.shares[r].meshes[p];
            EXPECT(mesh.stride == stride && mesh.components.size() == (size_t)mesh.occupancy() * stride, "%s: pool %zu rank %u sizes", what, p, r);
            uint32_t holes = 0;
            for (uint32_t j = 0; j < mesh.occupancy(); j++) {
                const uint32_t w = mesh.worldSlot[j];
                const auto* local = reinterpret_cast<const MeshRenderComponent*>(mesh.components.data() + (size_t)j * stride);
                if (w == GV_NONE) {
                    EXPECT(*local->entity == 0, "%s: pool %zu rank %u hole %u still holds an entity", what, p, r, j);
                    holes++;
                    continue;
                }
                held[w]++;
                EXPECT(shares.meshTables[p].rank[w] == r && shares.meshTables[p].local[w] == j, "%s: pool %zu: the tables do not lead to rank %u slot %u", what, p, r, j);
                const auto* global = reinterpret_cast<const MeshRenderComponent*>(data + (size_t)w * stride);
                EXPECT(std::memcmp(&local->aabb, &global->aabb, sizeof(Aabb)) == 0 && local->isEnabled == global->isEnabled, "%s: pool %zu slot %u is not the world's component", what, p, w);
                const uint32_t entity = *global->entity;
                const uint32_t transformSlot = entity && entity < emap.size() ? emap[entity] : GV_NONE;
                const uint32_t localEntity = *local->entity;
                if (!entity) {
                    EXPECT(localEntity == 0, "%s: a free mesh slot with an entity", what);
                } else if (transformSlot == GV_NONE) {
                    EXPECT(localEntity == RankShares::kNoTransformEntity && localEntity >= shares.shares[r].entityToTransform.size(),
                           "%s: a mesh without a transform resolves to one on its rank", what);
                } else {
                    EXPECT(shares.rankOfTransform[transformSlot] == r && shares.shares[r].entityToTransform[localEntity] == shares.localOfTransform[transformSlot],
                           "%s: pool %zu slot %u does not resolve to its entity's transform on rank %u (entity %u, transform slot %u on rank %u local %u; local entity %u; linked to %u which holds entity %u / known as %u)", what, p, w, r,
                           entity, transformSlot, shares.rankOfTransform[transformSlot], shares.localOfTransform[transformSlot], localEntity, shares.meshTables[p].transform[w],
                           shares.meshTables[p].transform[w] < occupancy ? *world[shares.meshTables[p].transform[w]].entity : 0u,
                           shares.meshTables[p].transform[w] < occupancy ? shares.entityOfTransform[shares.meshTables[p].transform[w]] : 0u);
                }
            }
            EXPECT(holes == mesh.freeSlots.size(), "%s: pool %zu rank %u has %u holes and %zu free slots", what, p, r, holes, mesh.freeSlots.size());
        }
        for (uint32_t j = 0; j < meshPool.getOccupancy(); j++)
            EXPECT(held[j] == 1, "%s: pool %zu slot %u is held %u times", what, p, j, held[j]);
    }
}

// GpuVisibilitySystem::mergeRanks: the ranks' sorted runs of one list -> one sorted list, long lists in pieces on the worker threads
// (splitRuns finds where an output position falls in every run). Random runs with many equal keys, negative keys, empty runs: the
// output is a permutation of the input in the records' order (mesh.hpp:196,204), piece boundaries included.
template <class Mesh>
static void checkMerge(std::mt19937& rng, uint32_t ranks, uint32_t perRank, uint32_t distinctKeys, const char* what)
{
    std::vector<std::vector<Mesh>> runs(ranks);
    std::vector<const Mesh*> ptrs(ranks);
    std::vector<uint32_t> counts(ranks);
    size_t total = 0;
    for (uint32_t r = 0; r < ranks; r++) {
        const uint32_t n = (r == 1 && ranks > 2) ? 0u : perRank / 2 + rng() % (perRank + 1);
        runs[r].resize(n);
        for (uint32_t k = 0; k < n; k++) {
            runs[r][k].componentOffset = ((size_t)r << 32) | k;
            runs[r][k].distanceSq = (float)(int)(rng() % distinctKeys) * 0.25f - (distinctKeys > 8 ? 3.0f : 0.0f);
        }
        std::stable_sort(runs[r].begin(), runs[r].end());  // (the record's own operator<)
        ptrs[r] = runs[r].data();
        counts[r] = n;
        total += n;
    }
    std::vector<Mesh> merged(total);
    GpuVisibilitySystem::mergeRanks(merged.data(), ptrs.data(), counts.data(), ranks, true);
    std::vector<uint8_t> seen(total, 0);
    std::vector<size_t> base(ranks, 0);
    for (uint32_t r = 1; r < ranks; r++)
        base[r] = base[r - 1] + counts[r - 1];
    for (size_t k = 0; k < total; k++) {
        EXPECT(k == 0 || !(merged[k] < merged[k - 1]), "%s: record %zu is out of order", what, k);
        const uint32_t r = (uint32_t)(merged[k].componentOffset >> 32), j = (uint32_t)merged[k].componentOffset;
        EXPECT(r < ranks && j < counts[r] && !seen[base[r] + j], "%s: record %zu is not one of the input's, or appears twice", what, k);
        if (r < ranks && j < counts[r])
            seen[base[r] + j] = 1;
    }
    // (every input record appears: `seen` has `total` distinct hits)
    size_t hits = 0;
    for (uint8_t v : seen)
        hits += v;
    EXPECT(hits == total, "%s: %zu of %zu records arrived", what, hits, total);
}

int main()
{
    {
        std::mt19937 mergeRng(99u);
        for (uint32_t ranks : {1u, 2u, 4u, 8u}) {
            checkMerge<UnsortedMesh>(mergeRng, ranks, 3000, 1u << 20, "short ascending runs");
            checkMerge<SortedMesh>(mergeRng, ranks, 3000, 5, "short descending runs, five distinct keys");
            checkMerge<UnsortedMesh>(mergeRng, ranks, 300000 / ranks, 7, "long ascending runs, seven distinct keys (pieces on the workers)");
            checkMerge<SortedMesh>(mergeRng, ranks, 300000 / ranks, 1u << 22, "long descending runs (pieces on the workers)");
            checkMerge<SortedMesh>(mergeRng, ranks, 300000 / ranks, 1, "long runs of ONE key");
        }
    }
    std::mt19937 rng(20261004u);
    auto uniform = [&](float lo, float hi) { return lo + (hi - lo) * (float)(rng() >> 8) * (1.0f / 16777216.0f); };
    for (uint32_t ranks : {1u, 2u, 3u, 8u, 4u}) {
        Manager manager;
        auto transformSystem = manager.createSystem<TransformSystem>();
        manager.registerComponents<TransformComponent>(transformSystem);
        auto opaque = manager.createSystem<OpaqueMeshSystem>();
        manager.registerComponents<MeshRenderComponent>(opaque);
        auto wide = manager.createSystem<WideMeshSystem>();
        manager.registerComponents<WideMeshComponent>(wide);
        manager.initialize();
        const uint32_t n = ranks == 4 ? 210000 : 6000;  // (the last world is long enough for the passes to spread over the worker threads)
        const double side = 100.0 * std::cbrt((double)n);
        uint32_t grid[3] = {1, 1, 1};
        for (uint32_t axis = 0; (uint64_t)grid[0] * grid[1] * grid[2] < 512ull * ranks; axis = (axis + 1) % 3)
            grid[axis] *= 2;
        std::vector<ID<Entity>> ents;
        for (uint32_t i = 0; i < n; i++) {
            auto e = manager.createEntity();
            ents.push_back(e);
            MeshRenderComponent* m = (i % 3 == 0) ? static_cast<MeshRenderComponent*>(*wide->add(e)) : *opaque->add(e);
            m->aabb.min = f32x4(-1, -1, -1);
            m->aabb.max = f32x4(uniform(0.1f, 2.0f), 1, 1);
            if (i % 41 == 0)
                continue;  // a mesh whose entity has no transform (mesh.cpp:149-153)
            auto t = transformSystem->add(e);
            t->setPosition(uniform(-0.5f * (float)side, 0.5f * (float)side), uniform(-0.5f * (float)side, 0.5f * (float)side), uniform(-0.5f * (float)side, 0.5f * (float)side));
            t->setScale(1, 1, 1);
            t->setRotation(quat(0, 0, 0, 1));
        }
        for (uint32_t i = n / 10; i < n; i++)  // depth ~4 trees
            if (i % 41 != 0) {
                uint32_t parent = rng() % (i / 4 + 1);
                if (parent % 41 == 0)
                    parent++;
                if (parent != i && parent % 41 != 0 && transformSystem->tryGetOf(ents[parent]))
                    transformSystem->setParent(ents[i], ents[parent]);
            }
        std::vector<IMeshRenderSystem*> meshSystems{opaque, wide};
        RankShares shares;
        shares.deal(transformSystem, meshSystems, ranks, grid, side);
        check(transformSystem, meshSystems, shares, ranks, grid, side, "first deal");
        // entities go (free slots once the frame has disposed of them), subtrees are re-parented, a few move: deal again
        for (uint32_t i = 7; i < n; i += 13)
            if (i % 41 != 0)
                manager.destroy(ents[i]);
        manager.update();
        for (uint32_t i = n / 2; i < n; i += 29)
            if (i % 41 != 0 && i % 13 != 7 && (i / 8) % 41 != 0 && (i / 8) % 13 != 7 && transformSystem->tryGetOf(ents[i]) && transformSystem->tryGetOf(ents[i / 8]))
                transformSystem->setParent(ents[i], ents[i / 8]);
        shares.deal(transformSystem, meshSystems, ranks, grid, side);
        check(transformSystem, meshSystems, shares, ranks, grid, side, "after destruction and re-parenting");
        // a transform moves (inside its cell or not: ownership is re-examined by the next deal only): its rank's copy follows
        for (uint32_t i = 1; i < n; i += 97)
            if (auto t = transformSystem->tryGetOf(ents[i])) {
                t->setPosition(uniform(-10, 10), uniform(-10, 10), uniform(-10, 10));
                const uint32_t slot = (uint32_t)(*t - transformSystem->getComponents().getData());
                shares.copyTransform(transformSystem, slot);
                const uint32_t rank = shares.rankOfTransform[slot];
                EXPECT(rank != GV_NONE && std::memcmp(&shares.shares[rank].transforms[shares.localOfTransform[slot]].posChildCount, &t->posChildCount, 12) == 0,
                       "copyTransform: slot %u", slot);
            }
        // ---- frame to frame, no further deal ----
        RankShares::Changes changes;
        changes.reset(ranks, meshSystems.size());
        // (1) mesh components edited behind the systems' backs: found by comparing, copied, reported once each
        uint32_t edited = 0;
        for (uint32_t j = 3; j < opaque->getComponents().getOccupancy(); j += 17) {
            auto& c = opaque->getComponents().getData()[j];
            if (!*c.entity)
                continue;
            c.aabb.max = f32x4(uniform(2.5f, 3.0f), 2, 2);
            c.isEnabled = (j & 1) != 0;
            edited++;
        }
        EXPECT(shares.syncMeshes(0, opaque, 0, opaque->getComponents().getOccupancy(), changes), "syncMeshes: an edit is not structural");
        EXPECT(shares.syncMeshes(1, wide, 0, wide->getComponents().getOccupancy(), changes), "syncMeshes: an untouched pool is not structural");
        uint32_t reported = 0;
        for (uint32_t r = 0; r < ranks; r++) {
            reported += (uint32_t)changes.ranks[r].meshes[0].size();
            EXPECT(changes.ranks[r].meshes[1].empty(), "syncMeshes: an untouched pool reports edits");
        }
        EXPECT(reported == edited, "syncMeshes: %u slots edited, %u reported", edited, reported);
        check(transformSystem, meshSystems, shares, ranks, grid, side, "after mesh edits", false);
        changes.reset(ranks, meshSystems.size());
        EXPECT(shares.syncMeshes(0, opaque, 0, opaque->getComponents().getOccupancy(), changes), "syncMeshes twice");
        for (uint32_t r = 0; r < ranks; r++)
            EXPECT(changes.ranks[r].meshes[0].empty(), "syncMeshes: nothing changed, yet rank %u is told of %zu slots", r, changes.ranks[r].meshes[0].size());
        // (2) every transform re-copied (a writer that does not say what it moved), then the roots that left their rank's cells move
        for (uint32_t i = 2; i < n; i += 5)
            if (auto t = transformSystem->tryGetOf(ents[i]))
                if (!*t->parent)
                    t->setPosition(uniform(-0.5f * (float)side, 0.5f * (float)side), uniform(-0.5f * (float)side, 0.5f * (float)side), uniform(-0.5f * (float)side, 0.5f * (float)side));
        shares.syncAllTransforms(transformSystem, changes);
        check(transformSystem, meshSystems, shares, ranks, grid, side, "after syncAllTransforms", false);
        shares.rebin(transformSystem, {}, true, ranks, grid, side, changes);
        EXPECT(ranks == 1 ? changes.movedTrees == 0 : changes.movedTrees > 0, "rebin: %u trees moved with %u ranks", changes.movedTrees, ranks);
        check(transformSystem, meshSystems, shares, ranks, grid, side, "after rebin of every root");
        // (3) itemised: a few roots jump, only those are looked at; they come back, their old holes are reused
        for (int round = 0; round < 3; round++) {
            changes.reset(ranks, meshSystems.size());
            std::vector<uint32_t> moved;
            for (uint32_t i = 4 + (uint32_t)round; i < n; i += 11)
                if (auto t = transformSystem->tryGetOf(ents[i]))
                    if (!*t->parent) {
                        t->setPosition(uniform(-0.5f * (float)side, 0.5f * (float)side), uniform(-0.5f * (float)side, 0.5f * (float)side), uniform(-0.5f * (float)side, 0.5f * (float)side));
                        const uint32_t slot = (uint32_t)(*t - transformSystem->getComponents().getData());
                        shares.syncTransform(transformSystem, slot, changes);
                        moved.push_back(slot);
                    }
            shares.rebin(transformSystem, moved, false, ranks, grid, side, changes);
            // what a rank is told covers every local slot a tree left or entered
            for (uint32_t r = 0; r < ranks; r++) {
                auto& told = changes.ranks[r];
                std::vector<uint32_t> slots = told.transforms;
                for (const auto& run : RankShares::Changes::runs(slots))
                    EXPECT(run.first + run.second <= shares.shares[r].transforms.size(), "rebin: a run past the share of rank %u", r);
                for (size_t p = 0; p < meshSystems.size(); p++)
                    for (uint32_t l : told.maps[p])
                        EXPECT(l < shares.shares[r].meshes[p].occupancy(), "rebin: an index-map entry past the share of rank %u", r);
            }
            check(transformSystem, meshSystems, shares, ranks, grid, side, "after an itemised rebin", false);
        }
        // every moved root is where its position says (the others were not re-examined: ownership is a matter of balance)
        // (4) entities and components that come and go, parent links that move, pools that grow: followed, never dealt again
        auto follow = [&](const char* what) {
            changes.reset(ranks, meshSystems.size());
            std::vector<std::vector<uint32_t>> changedHands(meshSystems.size());
            std::vector<RankShares::MeshPiece> pieces;
            for (uint32_t p = 0; p < meshSystems.size(); p++) {
                const uint32_t occupancy = meshSystems[p]->getMeshComponentPool().getOccupancy();
                pieces.push_back(RankShares::MeshPiece{p, 0u, occupancy, meshSystems[p]});
                for (uint32_t j = (uint32_t)shares.meshTables[p].entity.size(); j < occupancy; j++)
                    changedHands[p].push_back(j);
            }
            (void)shares.syncMeshes(pieces, changes, &changedHands);
            std::vector<std::pair<uint32_t, uint32_t>> slots;
            if (transformSystem->flagsLo < transformSystem->flagsHi)
                slots.push_back({transformSystem->flagsLo, transformSystem->flagsHi});
            if (transformSystem->reparentLo < transformSystem->reparentHi)
                slots.push_back({transformSystem->reparentLo, transformSystem->reparentHi});
            if (shares.rankOfTransform.size() < transformSystem->getComponents().getOccupancy())
                slots.push_back({(uint32_t)shares.rankOfTransform.size(), transformSystem->getComponents().getOccupancy()});
            const bool followed = shares.followEntities(transformSystem, meshSystems, ranks, grid, side, slots, changedHands, transformSystem->reparentLo,
                                                        transformSystem->reparentHi, changes);
            EXPECT(followed, "%s: followEntities gave up", what);
            transformSystem->clearFlagsRange();
            transformSystem->clearReparentRange();
            for (auto sys : {static_cast<VersionedMeshSystem*>(opaque), static_cast<VersionedMeshSystem*>(wide)})
                sys->clearMeshRange();
            check(transformSystem, meshSystems, shares, ranks, grid, side, what, false);
            // what the ranks are told stays inside their shares
            for (uint32_t r = 0; r < ranks; r++) {
                for (uint32_t l : changes.ranks[r].transforms)
                    EXPECT(l < shares.shares[r].transforms.size(), "%s: a transform mark past the share of rank %u", what, r);
                for (size_t p = 0; p < meshSystems.size(); p++)
                    for (uint32_t l : changes.ranks[r].meshes[p])
                        EXPECT(l < shares.shares[r].meshes[p].occupancy(), "%s: a mesh mark past the share of rank %u", what, r);
            }
        };
        transformSystem->clearFlagsRange();
        transformSystem->clearReparentRange();
        std::vector<bool> dead(ents.size(), false);
        for (uint32_t i = 7; i < n; i += 13)  // (destroyed further up: their handles are stale, the ids may have been handed out again)
            dead[i] = i % 41 != 0;
        for (int round = 0; round < 6; round++) {
            // entities go (with their subtrees' links: children become roots) ...
            for (uint32_t k = 0; k < n / 60 + 3; k++) {
                const uint32_t i = rng() % (uint32_t)ents.size();
                if (!dead[i]) {
                    manager.destroy(ents[i]);
                    dead[i] = true;
                }
            }
            manager.update();  // (the components are wiped at the end of the frame)
            // ... entities come: roots, children of old entities, children of entities that came in this very round, some without a
            // transform, some without a mesh; the pools grow past what was dealt
            const uint32_t before = (uint32_t)ents.size();
            for (uint32_t k = 0; k < n / 40 + 5; k++) {
                auto e = manager.createEntity();
                ents.push_back(e);
                dead.push_back(false);
                if (k % 7 != 3) {
                    MeshRenderComponent* m = (k % 3 == 0) ? static_cast<MeshRenderComponent*>(*wide->add(e)) : *opaque->add(e);
                    m->aabb.max = f32x4(uniform(0.1f, 2.0f), 1, 1);
                }
                if (k % 11 == 5)
                    continue;  // a mesh whose entity has no transform
                auto t = transformSystem->add(e);
                t->setPosition(uniform(-0.5f * (float)side, 0.5f * (float)side), uniform(-0.5f * (float)side, 0.5f * (float)side), uniform(-0.5f * (float)side, 0.5f * (float)side));
                if (k % 2) {
                    const uint32_t parent = (k % 4 == 1 && ents.size() - before > 2) ? before + rng() % (uint32_t)(ents.size() - before - 1) : rng() % before;
                    if (!dead[parent] && transformSystem->tryGetOf(ents[parent]) && !(ents[parent] == e))
                        transformSystem->setParent(e, ents[parent]);
                }
            }
            // ... subtrees are handed to parents that live elsewhere, some become roots again, flags flip
            for (uint32_t k = 0; k < n / 50 + 3; k++) {
                const uint32_t i = rng() % (uint32_t)ents.size(), q = rng() % (uint32_t)ents.size();
                if (dead[i] || dead[q] || i == q || !transformSystem->tryGetOf(ents[i]) || !transformSystem->tryGetOf(ents[q]))
                    continue;
                bool cycle = false;
                for (auto up = ents[q]; up; up = transformSystem->tryGetOf(up)->parent)
                    cycle = cycle || up == ents[i];
                if (!cycle)
                    transformSystem->setParent(ents[i], k % 5 == 0 ? ID<Entity>() : ents[q]);
                transformSystem->setActive(ents[q], (k & 1) != 0);
            }
            // ... a transform is taken from an entity that keeps its mesh, another entity gets one at last
            for (uint32_t k = 0; k < 5; k++) {
                const uint32_t i = rng() % (uint32_t)ents.size();
                if (dead[i])
                    continue;
                if (auto t = transformSystem->tryGetOf(ents[i])) {
                    if (t->childCount() == 0 && round % 2 == 0)
                        transformSystem->removeOf(ents[i]);
                } else {
                    transformSystem->add(ents[i])->setPosition(uniform(-100, 100), uniform(-100, 100), uniform(-100, 100));
                }
            }
            manager.update();
            follow("entities came and went");
        }
        // ... and after all that a deal gives shares that satisfy the same invariants (the tables were kept consistent: nothing depends on it)
        shares.deal(transformSystem, meshSystems, ranks, grid, side);
        check(transformSystem, meshSystems, shares, ranks, grid, side, "dealt again at the end");
    }
    std::printf("{\"ok\": %s, \"failures\": %d}\n", failures ? "false" : "true", failures);
    return failures ? 1 : 0;
}
