// rank_shares.hpp — ONE process, N GPUs: what each rank (GPU) holds of the engine's component pools, and how that follows the engine
// from frame to frame without being dealt again.
//
// The reference is one process with one Manager (source/editor/entry.cpp:135) whose workers index the pools IN PLACE
// (source/system/render/mesh.cpp:119-120,139: componentData + i * componentSize; ranges from ThreadPool::addItems,
// source/thread-pool.cpp:180-194). A GPU cannot: it needs its part of the pools in its own memory. Ranks split the WORLD
// (SURVEY.md §8e): the world cube is cut into many more cells than ranks, dealt in rotating rounds (gv_cell_owner: the rule
// gv_scene_extract_rank and garden_amd/multi.py::partition_world use); a ROOT transform's position decides, its descendants follow —
// no parent chain is cut, so a rank computes the world's matrices bit for bit (TransformComponent::calcModel, transform.hpp:197-214).
//
// A share is a set of pools in the engine's own layouts (TransformComponent / MeshRenderComponent-derived, same byte strides), with
// entity ids renumbered per rank (local transform slot i <-> entity i + 1) and, per pool, the local slot -> WORLD slot table that
// gv_pool_set_index_map takes: exchanged lists, records and isVisible bytes carry the engine's own slots. Every slot of every mesh
// pool lives on exactly one rank — free slots and meshes without a transform included: the light pass writes isVisible of ALL of
// them (mesh.cpp:140-153).
//
// Frame to frame (round 6): deal() runs once — and again only for changes nobody itemised (hierarchyVersion, another set of mesh
// systems). Everything else is carried over slot by slot — the tables below say where every world slot lives:
//   * a transform that moved / changed its flags: copied to its rank (copyTransform);
//   * a mesh component that was edited — or, for a mesh system that cannot say what changed (every one of the reference's:
//     sprite.cpp, 9-slice, label.cpp, instance.cpp carry no counter), whichever slots turn out to differ from the rank's copy in the
//     bytes the cull reads (entity, isEnabled, aabb: syncMeshes compares them on the worker threads): copied to its rank;
//   * a ROOT whose position crossed into a cell of another rank (SURVEY.md §8e "re-bin only roots whose position crosses a cell";
//     physics writes positions every tick, source/system/physics.cpp:1033-1034): its tree's transforms and their meshes move from
//     one share to the other (moveTree) — holes are left behind and reused, nothing else is touched;
//   * entities and components that came or went, parent links that moved (followEntities): a transform that went leaves a hole in its
//     share, one that came takes a hole (or a new slot) on its parent's rank — a root: on the rank its position falls to —, a mesh
//     component follows its entity's transform, a subtree whose new parent lives on another rank moves there (moveTree).
// What changed where is recorded per rank (Changes) in the units gv_mark_dirty / gv_pool_update_index_map take.
#pragma once
#include <algorithm>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <vector>

#include "../../../include/garden_vis.h"
#include "garden_host.hpp"

namespace garden {

struct RankShares {
    // a live mesh whose entity has no transform (mesh.cpp:149-153): an entity id beyond every rank's entity map
    static constexpr uint32_t kNoTransformEntity = 0x7FFFFFFFu;

    struct MeshShare {
        std::vector<uint8_t> components;   // local pool, the engine's stride
        std::vector<uint32_t> worldSlot;   // local slot -> slot of the engine's pool (GV_NONE: a hole left by a tree that moved away)
        std::vector<uint32_t> freeSlots;   // the holes
        size_t stride = 0;
        uint32_t occupancy() const noexcept { return (uint32_t)worldSlot.size(); }
        MeshRenderComponent* at(uint32_t local) noexcept { return reinterpret_cast<MeshRenderComponent*>(components.data() + (size_t)local * stride); }
    };
    struct Share {
        std::vector<TransformComponent> transforms;  // local transform pool (childs = NULL: the cull path never reads it)
        std::vector<uint32_t> transformWorldSlot;    // local -> world transform slot (GV_NONE: a hole)
        std::vector<uint32_t> entityToTransform;     // local entity id -> local transform slot: [0] = none, [k + 1] = k
        std::vector<uint32_t> freeTransforms;
        std::vector<MeshShare> meshes;               // one per mesh system, in meshSystems order
        uint32_t liveTransforms() const noexcept { return (uint32_t)(transforms.size() - freeTransforms.size()); }
    };
    std::vector<Share> shares;
    std::vector<uint32_t> rankOfTransform, localOfTransform;  // world transform slot -> rank, local slot (GV_NONE: a free slot)
    std::vector<uint32_t> entityOfTransform;                  // world transform slot -> the entity it held when the pools were dealt
    struct MeshTable {                                        // per world mesh slot of one pool
        std::vector<uint32_t> rank, local, entity;            // where it lives; the entity it holds as far as the shares know
        std::vector<uint32_t> transform, next;                // the world transform slot it is linked to (GV_NONE: none; kLoose: a live
    };                                                        // entity without a transform), the next ref of that transform's list
    std::vector<MeshTable> meshTables;
    // world transform slot -> the mesh components of its entity, as a list through MeshTable::next: ref = (pool << 28) | mesh slot
    std::vector<uint32_t> firstMesh;
    std::vector<uint32_t> looseMeshes;  // refs of live meshes whose entity has no transform (mesh.cpp:149-153): picked up when one arrives
    static constexpr uint32_t kLoose = 0xFFFFFFFEu;
    static uint32_t refOf(uint32_t pool, uint32_t slot) noexcept { return (pool << 28) | slot; }

    void linkMesh(uint32_t transformSlot, uint32_t pool, uint32_t slot)
    {
        MeshTable& table = meshTables[pool];
        table.transform[slot] = transformSlot;
        table.next[slot] = firstMesh[transformSlot];
        firstMesh[transformSlot] = refOf(pool, slot);
    }
    void unlinkMesh(uint32_t pool, uint32_t slot)
    {
        MeshTable& table = meshTables[pool];
        const uint32_t t = table.transform[slot];
        if (t == kLoose) {  // (a loose mesh keeps its place in looseMeshes where a linked one keeps its list's next ref)
            const uint32_t at = table.next[slot], last = looseMeshes.back();
            looseMeshes[at] = last;
            meshTables[last >> 28].next[last & 0x0FFFFFFFu] = at;
            looseMeshes.pop_back();
        } else if (t != GV_NONE) {
            uint32_t* at = &firstMesh[t];
            while (*at != GV_NONE && *at != refOf(pool, slot))
                at = &meshTables[*at >> 28].next[*at & 0x0FFFFFFFu];
            if (*at != GV_NONE)
                *at = table.next[slot];
        }
        table.transform[slot] = GV_NONE;
        table.next[slot] = GV_NONE;
    }
    void looseMesh(uint32_t pool, uint32_t slot)
    {
        meshTables[pool].transform[slot] = kLoose;
        meshTables[pool].next[slot] = (uint32_t)looseMeshes.size();
        looseMeshes.push_back(refOf(pool, slot));
    }
    // share slots: holes are reused before the pools grow
    static uint32_t allocTransform(Share& share, uint32_t worldSlot)
    {
        uint32_t local;
        if (!share.freeTransforms.empty()) {
            local = share.freeTransforms.back();
            share.freeTransforms.pop_back();
        } else {
            local = (uint32_t)share.transforms.size();
            share.transforms.emplace_back();
            share.transformWorldSlot.push_back(GV_NONE);
            share.entityToTransform.push_back(local);  // entity local + 1
        }
        share.transformWorldSlot[local] = worldSlot;
        return local;
    }
    static void freeTransform(Share& share, uint32_t local)
    {
        share.transforms[local] = TransformComponent();  // a free slot (entity = null)
        share.transformWorldSlot[local] = GV_NONE;
        share.freeTransforms.push_back(local);
    }
    static uint32_t allocMesh(MeshShare& mesh, uint32_t worldSlot)
    {
        uint32_t local;
        if (!mesh.freeSlots.empty()) {
            local = mesh.freeSlots.back();
            mesh.freeSlots.pop_back();
        } else {
            local = mesh.occupancy();
            mesh.worldSlot.push_back(GV_NONE);
            mesh.components.resize(mesh.components.size() + mesh.stride);
        }
        mesh.worldSlot[local] = worldSlot;
        return local;
    }
    static void freeMesh(MeshShare& mesh, uint32_t local)
    {
        mesh.at(local)->entity = ID<Entity>();  // a free slot on the rank it left (mesh.cpp:142)
        mesh.worldSlot[local] = GV_NONE;
        mesh.freeSlots.push_back(local);
    }

    // What a frame's synchronisation changed on each rank, in LOCAL slots: what gv_mark_dirty (GV_DIRTY_TRANSFORM, ranged
    // GV_DIRTY_HIERARCHY, GV_DIRTY_MESH) and gv_pool_update_index_map are told.
    struct Changes {
        struct PerRank {
            std::vector<uint32_t> transforms, links;
            std::vector<std::vector<uint32_t>> meshes, maps;  // [pool]
            bool allTransforms = false;
        };
        std::vector<PerRank> ranks;
        uint32_t movedTrees = 0, movedTransforms = 0;
        void reset(uint32_t rankCount, size_t pools)
        {
            ranks.assign(rankCount, PerRank{});
            for (auto& r : ranks) {
                r.meshes.assign(pools, {});
                r.maps.assign(pools, {});
            }
            movedTrees = movedTransforms = 0;
        }
        // sorted, without duplicates, as (first, count) runs
        static std::vector<std::pair<uint32_t, uint32_t>> runs(std::vector<uint32_t>& slots)
        {
            std::sort(slots.begin(), slots.end());
            slots.erase(std::unique(slots.begin(), slots.end()), slots.end());
            std::vector<std::pair<uint32_t, uint32_t>> out;
            for (uint32_t s : slots)
                if (!out.empty() && out.back().first + out.back().second == s)
                    out.back().second++;
                else
                    out.push_back({s, 1u});
            return out;
        }
    };

    // The entity whose transform lives in `slot` as Manager::tryGet<TransformComponent>(entity) sees it (mesh.cpp:149) — 0: none. A
    // component that was destroyed in this frame still sits in the pool until the frame's dispose() (docs/ECS/Entities.md:52-54), but
    // the entity no longer resolves to it: for the cull it is gone already.
    static uint32_t heldBy(const TransformSystem* ts, uint32_t slot) noexcept
    {
        const uint32_t entity = *worldTransform(ts, slot)->entity;
        const auto& emap = ts->getEntityMap();
        return entity && entity < emap.size() && emap[entity] == slot ? entity : 0u;
    }

    static const TransformComponent* worldTransform(const TransformSystem* ts, uint32_t slot) noexcept
    {
        return const_cast<TransformSystem*>(ts)->getComponents().getData() + slot;
    }

    // one world transform into its rank's pool: the same bytes, ids renumbered
    void copyTransform(const TransformSystem* ts, uint32_t worldSlot)
    {
        if (!tryCopyTransform(ts, worldSlot))
            throw std::runtime_error("RankShares: a parent without a transform, or on another rank than its child");
    }
    bool tryCopyTransform(const TransformSystem* ts, uint32_t worldSlot) noexcept
    {
        const uint32_t rank = rankOfTransform[worldSlot];
        if (rank == GV_NONE)
            return true;
        const uint32_t local = localOfTransform[worldSlot];
        const auto& emap = ts->getEntityMap();
        TransformComponent& dst = shares[rank].transforms[local];
        std::memcpy(static_cast<void*>(&dst), worldTransform(ts, worldSlot), sizeof(TransformComponent));
        dst.childs = nullptr;
        dst.entity = ID<Entity>(local + 1);
        const uint32_t parent = *dst.parent;
        if (parent) {
            const uint32_t parentSlot = parent < emap.size() && emap[parent] < rankOfTransform.size() ? emap[parent] : GV_NONE;
            if (parentSlot == GV_NONE) {
                dst.parent = ID<Entity>();  // a link that leads nowhere (a parent destroyed in this frame: its children are orphaned at once,
                return true;                //  transform.cpp:53-63 — only a transform that is itself on its way out still names it)
            }
            if (rankOfTransform[parentSlot] != rank)
                return false;
            dst.parent = ID<Entity>(localOfTransform[parentSlot] + 1);
        }
        return true;
    }

    // Deals the engine's pools to `ranks` shares. grid / side: the cell grid over the world cube [-side/2, side/2]^3.
    void deal(const TransformSystem* ts, const std::vector<IMeshRenderSystem*>& meshSystems, uint32_t ranks, const uint32_t grid[3], double side)
    {
        auto& pool = const_cast<TransformSystem*>(ts)->getComponents();
        const auto& emap = ts->getEntityMap();
        const uint32_t occupancy = pool.getOccupancy();
        const TransformComponent* world = pool.getData();
        // the owner of every position, then of every transform = the owner of its ROOT's position
        std::vector<uint32_t> ownerByPosition(occupancy ? occupancy : 1);
        if (occupancy && gv_cell_owner(grid, side, ranks, reinterpret_cast<const float*>(&world[0].posChildCount), (uint32_t)sizeof(TransformComponent), occupancy,
                                       ownerByPosition.data()) != GV_OK)
            throw std::runtime_error("RankShares: gv_cell_owner failed");
        std::vector<uint32_t> rootOf(occupancy, GV_NONE);
        std::vector<uint32_t> chain;
        for (uint32_t i = 0; i < occupancy; i++) {
            if (!heldBy(ts, i) || rootOf[i] != GV_NONE)
                continue;
            chain.clear();
            uint32_t s = i;
            while (rootOf[s] == GV_NONE) {
                chain.push_back(s);
                const uint32_t parent = *world[s].parent;
                const uint32_t up = parent && parent < emap.size() && emap[parent] < occupancy ? emap[parent] : GV_NONE;
                if (up == GV_NONE || !heldBy(ts, up) || chain.size() > occupancy) {
                    rootOf[s] = s;  // a root (or a broken link: treated as one)
                    break;
                }
                s = up;
            }
            const uint32_t root = rootOf[s];
            for (uint32_t c : chain)
                rootOf[c] = root;
        }
        shares.assign(ranks, Share{});
        rankOfTransform.assign(occupancy, GV_NONE);
        localOfTransform.assign(occupancy, GV_NONE);
        entityOfTransform.assign(occupancy, 0u);
        for (uint32_t i = 0; i < occupancy; i++)
            entityOfTransform[i] = heldBy(ts, i);
        for (uint32_t i = 0; i < occupancy; i++) {
            if (!entityOfTransform[i])
                continue;
            const uint32_t rank = ownerByPosition[rootOf[i]];
            rankOfTransform[i] = rank;
            localOfTransform[i] = (uint32_t)shares[rank].transformWorldSlot.size();
            shares[rank].transformWorldSlot.push_back(i);
        }
        for (uint32_t r = 0; r < ranks; r++) {
            Share& share = shares[r];
            const uint32_t n = (uint32_t)share.transformWorldSlot.size();
            share.transforms.resize(n);
            share.entityToTransform.assign((size_t)n + 1, GV_NONE);  // id 0 = null, ids 1..n
            for (uint32_t k = 0; k < n; k++)
                share.entityToTransform[k + 1] = k;
            share.meshes.assign(meshSystems.size(), MeshShare{});
        }
        for (uint32_t i = 0; i < occupancy; i++)
            copyTransform(ts, i);
        meshTables.assign(meshSystems.size(), MeshTable{});
        firstMesh.assign(occupancy, GV_NONE);
        looseMeshes.clear();
        for (size_t p = 0; p < meshSystems.size(); p++) {
            const auto& meshPool = meshSystems[p]->getMeshComponentPool();
            const size_t stride = meshSystems[p]->getMeshComponentSize();
            const uint8_t* data = reinterpret_cast<const uint8_t*>(meshPool.getData());
            const uint32_t meshOccupancy = meshPool.getOccupancy();
            MeshTable& table = meshTables[p];
            table.rank.assign(meshOccupancy, 0u);
            table.local.assign(meshOccupancy, 0u);
            table.entity.assign(meshOccupancy, 0u);
            table.transform.assign(meshOccupancy, GV_NONE);
            table.next.assign(meshOccupancy, GV_NONE);
            for (uint32_t r = 0; r < ranks; r++)
                shares[r].meshes[p].stride = stride;
            for (uint32_t j = 0; j < meshOccupancy; j++) {
                const auto* component = reinterpret_cast<const MeshRenderComponent*>(data + (size_t)j * stride);
                const uint32_t entity = *component->entity;
                uint32_t transformSlot = entity && entity < emap.size() && emap[entity] < occupancy ? emap[entity] : GV_NONE;
                if (transformSlot != GV_NONE && rankOfTransform[transformSlot] == GV_NONE)
                    transformSlot = GV_NONE;  // (a slot its entity no longer resolves to lives nowhere: the mesh has no transform, as in placeMesh)
                const uint32_t rank = transformSlot != GV_NONE ? rankOfTransform[transformSlot] : j % ranks;
                MeshShare& share = shares[rank].meshes[p];
                const size_t at = share.components.size();
                share.components.resize(at + stride);
                std::memcpy(share.components.data() + at, component, stride);
                auto* local = reinterpret_cast<MeshRenderComponent*>(share.components.data() + at);
                if (!entity)
                    local->entity = ID<Entity>();
                else if (transformSlot != GV_NONE)
                    local->entity = ID<Entity>(localOfTransform[transformSlot] + 1);
                else
                    local->entity = ID<Entity>(kNoTransformEntity);  // alive, no transform (mesh.cpp:149-153)
                table.rank[j] = rank;
                table.local[j] = share.occupancy();
                table.entity[j] = entity;
                share.worldSlot.push_back(j);
                if (transformSlot != GV_NONE) {
                    linkMesh(transformSlot, (uint32_t)p, j);  // (what follows a tree from share to share)
                } else if (entity) {
                    looseMesh((uint32_t)p, j);
                }
            }
        }
    }

    // ---- frame to frame ----

    // Mesh slots against the ranks' copies, in the bytes the cull reads — entity (mesh.cpp:142,149), isEnabled (:142), aabb
    // (:140-141,158): what differs is copied and recorded. pieces: per pool, the slots [lo, hi) to look at; ALL pieces are walked in
    // ONE pass over the library's worker threads (a frame of seven systems of 10^5 components each is one pass over 7 * 10^5 slots,
    // not seven short ones on the calling thread). Slots that hold another entity than the shares know (a component came or went)
    // are left alone and listed in changedEntity[pool] (NULL: false is returned at the first one instead): followEntities takes them.
    struct MeshPiece {
        uint32_t pool, lo, hi;
        IMeshRenderSystem* meshSystem;
    };
    bool syncMeshes(const std::vector<MeshPiece>& pieces, Changes& changes, std::vector<std::vector<uint32_t>>* changedEntity = nullptr)
    {
        struct Job {
            RankShares* self;
            std::vector<MeshPiece> pieces;
            std::vector<uint32_t> start;  // running slot counts: piece k covers [start[k], start[k + 1]) of the pass
            Changes* changes;
            std::vector<std::vector<uint32_t>>* changedEntity;
            std::mutex merge;
            bool structural = false;
        } job;
        job.self = this;
        job.changes = &changes;
        job.changedEntity = changedEntity;
        job.start.push_back(0u);
        for (MeshPiece piece : pieces) {
            piece.hi = std::min<uint32_t>(piece.hi, (uint32_t)meshTables[piece.pool].entity.size());
            if (piece.lo >= piece.hi)
                continue;
            job.pieces.push_back(piece);
            job.start.push_back(job.start.back() + (piece.hi - piece.lo));
        }
        if (job.pieces.empty())
            return true;
        gv_host_parallel_ranges(0, job.start.back(), [](void* user, uint32_t a, uint32_t b) {
            Job& j = *static_cast<Job*>(user);
            std::vector<uint32_t> edited;  // (pool << 28 | rank << 24 ...) would not fit: triples, flattened
            std::vector<uint32_t> handsChanged;  // pairs (pool, slot)
            bool structural = false;
            constexpr size_t kAabb = offsetof(MeshRenderComponent, aabb);
            size_t k = (size_t)(std::upper_bound(j.start.begin(), j.start.end(), a) - j.start.begin()) - 1;
            for (uint32_t at = a; at < b && (!structural || j.changedEntity); k++) {
                const MeshPiece& piece = j.pieces[k];
                const uint32_t end = std::min(b, j.start[k + 1]);
                const MeshTable& table = j.self->meshTables[piece.pool];
                const uint8_t* world = reinterpret_cast<const uint8_t*>(piece.meshSystem->getMeshComponentPool().getData());
                const size_t stride = piece.meshSystem->getMeshComponentSize();
                for (uint32_t s = piece.lo + (at - j.start[k]), e = piece.lo + (end - j.start[k]); s < e; s++) {
                    const auto* w = reinterpret_cast<const MeshRenderComponent*>(world + (size_t)s * stride);
                    if (*w->entity != table.entity[s]) {
                        structural = true;
                        if (!j.changedEntity)
                            break;
                        handsChanged.insert(handsChanged.end(), {piece.pool, s});
                        continue;
                    }
                    MeshRenderComponent* c = j.self->shares[table.rank[s]].meshes[piece.pool].at(table.local[s]);
                    if (c->isEnabled == w->isEnabled && std::memcmp(&c->aabb, &w->aabb, sizeof(Aabb)) == 0)
                        continue;
                    c->isEnabled = w->isEnabled;
                    std::memcpy(reinterpret_cast<uint8_t*>(c) + kAabb, reinterpret_cast<const uint8_t*>(w) + kAabb, sizeof(Aabb));
                    edited.insert(edited.end(), {piece.pool, table.rank[s], table.local[s]});
                }
                at = end;
            }
            if (!structural && edited.empty())
                return;
            std::lock_guard<std::mutex> lock(j.merge);
            j.structural = j.structural || structural;
            for (size_t e = 0; e + 2 < edited.size(); e += 3)
                j.changes->ranks[edited[e + 1]].meshes[edited[e]].push_back(edited[e + 2]);
            for (size_t e = 0; e + 1 < handsChanged.size(); e += 2)
                (*j.changedEntity)[handsChanged[e]].push_back(handsChanged[e + 1]);
        }, &job);
        return !job.structural;
    }
    bool syncMeshes(uint32_t p, IMeshRenderSystem* meshSystem, uint32_t lo, uint32_t hi, Changes& changes)
    {
        return syncMeshes(std::vector<MeshPiece>{MeshPiece{p, lo, hi, meshSystem}}, changes);
    }

    // One transform that moved or changed its flags: its bytes to its rank.
    void syncTransform(const TransformSystem* ts, uint32_t worldSlot, Changes& changes)
    {
        const uint32_t rank = worldSlot < rankOfTransform.size() ? rankOfTransform[worldSlot] : GV_NONE;
        if (rank == GV_NONE)
            return;
        copyTransform(ts, worldSlot);
        changes.ranks[rank].transforms.push_back(localOfTransform[worldSlot]);
    }
    // Every transform (a writer that does not say what it moved), on the worker threads.
    void syncAllTransforms(const TransformSystem* ts, Changes& changes)
    {
        struct Job {
            RankShares* self;
            const TransformSystem* ts;
            std::atomic<bool> broken;
        } job{this, ts, {false}};
        gv_host_parallel_ranges(0, (uint32_t)rankOfTransform.size(), [](void* user, uint32_t a, uint32_t b) {
            Job& j = *static_cast<Job*>(user);
            for (uint32_t s = a; s < b; s++)
                if (!j.self->tryCopyTransform(j.ts, s))  // (no exception may leave a worker thread)
                    j.broken = true;
        }, &job);
        if (job.broken)
            throw std::runtime_error("RankShares: a parent without a transform, or on another rank than its child");
        for (auto& r : changes.ranks)
            r.allTransforms = true;
    }

    // ROOTS among `slots` (world transform slots that moved; empty + all == true: every root) whose position now lies in a cell of
    // another rank: their trees change shares. SURVEY.md §8e: ownership is a matter of balance, not of correctness.
    void rebin(const TransformSystem* ts, const std::vector<uint32_t>& slots, bool all, uint32_t ranks, const uint32_t grid[3], double side, Changes& changes)
    {
        auto& pool = const_cast<TransformSystem*>(ts)->getComponents();
        const TransformComponent* world = pool.getData();
        const uint32_t occupancy = (uint32_t)rankOfTransform.size();
        std::vector<std::pair<uint32_t, uint32_t>> moves;  // (root slot, new rank)
        if (all) {
            std::vector<uint32_t> owner(occupancy ? occupancy : 1);
            if (occupancy && gv_cell_owner(grid, side, ranks, reinterpret_cast<const float*>(&world[0].posChildCount), (uint32_t)sizeof(TransformComponent), occupancy,
                                           owner.data()) != GV_OK)
                throw std::runtime_error("RankShares: gv_cell_owner failed");
            for (uint32_t s = 0; s < occupancy; s++)
                if (rankOfTransform[s] != GV_NONE && !*world[s].parent && owner[s] != rankOfTransform[s])
                    moves.push_back({s, owner[s]});
        } else {
            std::vector<uint32_t> roots;
            std::vector<float> positions;  // (one call for all of them: gv_cell_owner deals the cells anew every time it is asked)
            for (uint32_t s : slots) {
                if (s >= occupancy || rankOfTransform[s] == GV_NONE || *world[s].parent)
                    continue;
                roots.push_back(s);
                positions.insert(positions.end(), {world[s].posChildCount.x, world[s].posChildCount.y, world[s].posChildCount.z});
            }
            std::vector<uint32_t> owner(roots.size() ? roots.size() : 1);
            if (!roots.empty() && gv_cell_owner(grid, side, ranks, positions.data(), 12, (uint32_t)roots.size(), owner.data()) != GV_OK)
                throw std::runtime_error("RankShares: gv_cell_owner failed");
            for (size_t k = 0; k < roots.size(); k++)
                if (owner[k] != rankOfTransform[roots[k]])
                    moves.push_back({roots[k], owner[k]});
        }
        for (const auto& m : moves)
            if (rankOfTransform[m.first] != m.second)  // (a slot listed twice has moved already)
                moveTree(ts, m.first, m.second, changes);
    }

    // Mesh slot (pool, slot) to rank `to` (nothing happens when it lives there already), its local entity id set to `localEntity`.
    void moveMesh(uint32_t p, uint32_t slot, uint32_t to, uint32_t localEntity, Changes& changes)
    {
        MeshTable& table = meshTables[p];
        const uint32_t from = table.rank[slot];
        if (from != to) {
            MeshShare& ms = shares[from].meshes[p];
            MeshShare& md = shares[to].meshes[p];
            const uint32_t was = table.local[slot], now = allocMesh(md, slot);
            std::memcpy(md.components.data() + (size_t)now * md.stride, ms.components.data() + (size_t)was * ms.stride, md.stride);
            freeMesh(ms, was);
            table.rank[slot] = to;
            table.local[slot] = now;
            changes.ranks[from].meshes[p].push_back(was);
            changes.ranks[from].maps[p].push_back(was);
            changes.ranks[to].maps[p].push_back(now);
        }
        shares[to].meshes[p].at(table.local[slot])->entity = ID<Entity>(localEntity);
        changes.ranks[to].meshes[p].push_back(table.local[slot]);
    }

    // Entities and components that came or went, parent links that moved — followed slot by slot instead of dealing the pools again.
    //   transformRanges world transform slots [first, end) that may hold another entity than the shares know, or whose flags / parent
    //                   link may have changed (the engine's flags / re-parent ranges — hulls: most slots in them are as they were —,
    //                   slots the pool has grown by); they are classified in ONE pass on the library's worker threads: went / came /
    //                   changed / as it was (the last kind, nearly all of a hull, costs a compare and nothing else)
    //   meshSlots[p]    the same for mesh pool p (what syncMeshes found, slots the pool has grown by)
    // A transform that went leaves a hole on its rank (its entity's remaining meshes become "no transform", mesh.cpp:149-153); one
    // that came goes to its parent's rank — a root: where gv_cell_owner puts its position —; a subtree whose new parent lives on
    // another rank moves there (moveTree); a mesh component follows its entity's transform. false: something that cannot be followed
    // (a parent without a place, a pool that shrank): the caller deals again.
    // [linkLo, linkHi): the slots whose parent link may have moved (the engine's re-parent range): the ranks re-validate those links.
    bool followEntities(const TransformSystem* ts, const std::vector<IMeshRenderSystem*>& meshSystems, uint32_t ranks, const uint32_t grid[3], double side,
                        std::vector<std::pair<uint32_t, uint32_t>> transformRanges, const std::vector<std::vector<uint32_t>>& meshSlots, uint32_t linkLo,
                        uint32_t linkHi, Changes& changes)
    {
        auto& pool = const_cast<TransformSystem*>(ts)->getComponents();
        const auto& emap = ts->getEntityMap();
        const TransformComponent* world = pool.getData();
        const uint32_t occupancy = pool.getOccupancy();
        if (occupancy < rankOfTransform.size() || meshSystems.size() != meshTables.size() || shares.size() != ranks)
            return false;
        rankOfTransform.resize(occupancy, GV_NONE);
        localOfTransform.resize(occupancy, GV_NONE);
        entityOfTransform.resize(occupancy, 0u);
        firstMesh.resize(occupancy, GV_NONE);
        auto slotOf = [&](uint32_t entity) { return entity && entity < emap.size() && emap[entity] < occupancy ? emap[entity] : GV_NONE; };
        // 0. what happened to every slot of the ranges: one pass over the worker threads
        struct Classify {
            RankShares* self;
            const TransformSystem* ts;
            std::vector<std::pair<uint32_t, uint32_t>> ranges;
            std::vector<uint32_t> start;
            std::mutex merge;
            std::vector<uint32_t> went, came, changed;
        } classify;
        classify.self = this;
        classify.ts = ts;
        std::sort(transformRanges.begin(), transformRanges.end());
        classify.start.push_back(0u);
        for (auto range : transformRanges) {
            range.second = std::min(range.second, occupancy);
            if (!classify.ranges.empty() && range.first < classify.ranges.back().second)
                range.first = classify.ranges.back().second;  // (the engine's hulls overlap)
            if (range.first >= range.second)
                continue;
            classify.ranges.push_back(range);
            classify.start.push_back(classify.start.back() + (range.second - range.first));
        }
        if (!classify.ranges.empty())
            gv_host_parallel_ranges(0, classify.start.back(), [](void* user, uint32_t a, uint32_t b) {
                Classify& c = *static_cast<Classify*>(user);
                const RankShares& self = *c.self;
                const auto& emap = c.ts->getEntityMap();
                std::vector<uint32_t> went, came, changed;
                size_t k = (size_t)(std::upper_bound(c.start.begin(), c.start.end(), a) - c.start.begin()) - 1;
                for (uint32_t at = a; at < b; k++) {
                    const uint32_t end = std::min(b, c.start[k + 1]);
                    for (uint32_t s = c.ranges[k].first + (at - c.start[k]), e = c.ranges[k].first + (end - c.start[k]); s < e; s++) {
                        const uint32_t now = heldBy(c.ts, s), was = self.entityOfTransform[s];
                        if (was && now != was)
                            went.push_back(s);
                        if (now && now != was) {
                            came.push_back(s);
                        } else if (now) {  // the same entity: is the rank's copy still what the engine holds?
                            const TransformComponent* w = worldTransform(c.ts, s);
                            const uint32_t rank = self.rankOfTransform[s];
                            const TransformComponent& mine = self.shares[rank].transforms[self.localOfTransform[s]];
                            const uint32_t parent = *w->parent;
                            const uint32_t up = parent && parent < emap.size() && emap[parent] < self.rankOfTransform.size() ? emap[parent] : GV_NONE;
                            const uint32_t parentHere = up == GV_NONE ? 0u : self.rankOfTransform[up] == rank ? self.localOfTransform[up] + 1 : GV_NONE;
                            if (parentHere != *mine.parent || std::memcmp(&mine.posChildCount, &w->posChildCount, 48) != 0 || mine.selfActive != w->selfActive ||
                                mine.ancestorsActive != w->ancestorsActive || mine.modelWithAncestors != w->modelWithAncestors)
                                changed.push_back(s);
                        }
                    }
                    at = end;
                }
                if (went.empty() && came.empty() && changed.empty())
                    return;
                std::lock_guard<std::mutex> lock(c.merge);
                c.went.insert(c.went.end(), went.begin(), went.end());
                c.came.insert(c.came.end(), came.begin(), came.end());
                c.changed.insert(c.changed.end(), changed.begin(), changed.end());
            }, &classify);
        std::sort(classify.went.begin(), classify.went.end());
        std::sort(classify.came.begin(), classify.came.end());
        std::sort(classify.changed.begin(), classify.changed.end());
        std::vector<uint32_t>& arrivals = classify.came;
        std::vector<uint32_t>& kept = classify.changed;
        // 1. transforms that went (or whose slot changed hands)
        for (uint32_t s : classify.went) {
            {
                for (uint32_t ref = firstMesh[s]; ref != GV_NONE;) {  // its entity's meshes: whatever is still alive has no transform now
                    const uint32_t p = ref >> 28, j = ref & 0x0FFFFFFFu;
                    MeshTable& table = meshTables[p];
                    ref = table.next[j];
                    table.transform[j] = table.next[j] = GV_NONE;
                    const auto* component = reinterpret_cast<const MeshRenderComponent*>(
                        reinterpret_cast<const uint8_t*>(meshSystems[p]->getMeshComponentPool().getData()) + (size_t)j * meshSystems[p]->getMeshComponentSize());
                    if (j < meshSystems[p]->getMeshComponentPool().getOccupancy() && *component->entity == table.entity[j] && table.entity[j]) {
                        shares[table.rank[j]].meshes[p].at(table.local[j])->entity = ID<Entity>(kNoTransformEntity);
                        changes.ranks[table.rank[j]].meshes[p].push_back(table.local[j]);
                        looseMesh(p, j);
                    }  // (else: the component went or changed hands too — step 4 takes it from there)
                }
                firstMesh[s] = GV_NONE;
                freeTransform(shares[rankOfTransform[s]], localOfTransform[s]);
                changes.ranks[rankOfTransform[s]].transforms.push_back(localOfTransform[s]);
                rankOfTransform[s] = localOfTransform[s] = GV_NONE;
                entityOfTransform[s] = 0u;
            }
        }
        // 2. transforms that came: roots where their position falls (one gv_cell_owner call), children on their parent's rank
        {
            std::vector<float> positions;
            std::vector<uint32_t> roots;
            for (uint32_t s : arrivals)
                if (slotOf(*world[s].parent) == GV_NONE) {
                    roots.push_back(s);
                    positions.insert(positions.end(), {world[s].posChildCount.x, world[s].posChildCount.y, world[s].posChildCount.z});
                }
            std::vector<uint32_t> owner(roots.size() ? roots.size() : 1);
            if (!roots.empty() && gv_cell_owner(grid, side, ranks, positions.data(), 12, (uint32_t)roots.size(), owner.data()) != GV_OK)
                throw std::runtime_error("RankShares: gv_cell_owner failed");
            auto place = [&](uint32_t s, uint32_t rank) {
                rankOfTransform[s] = rank;
                localOfTransform[s] = allocTransform(shares[rank], s);
                entityOfTransform[s] = heldBy(ts, s);
            };
            for (size_t k = 0; k < roots.size(); k++)
                place(roots[k], owner[k]);
            std::vector<uint32_t> chain;
            for (uint32_t s : arrivals) {  // a child whose parent arrived in this frame too waits for it
                chain.clear();
                for (uint32_t c = s; rankOfTransform[c] == GV_NONE; c = slotOf(*world[c].parent)) {
                    chain.push_back(c);
                    const uint32_t up = slotOf(*world[c].parent);
                    if (up == GV_NONE || (rankOfTransform[up] == GV_NONE && !heldBy(ts, up)) || chain.size() > occupancy)
                        return false;  // (a parent without a place, or a cycle)
                    if (rankOfTransform[up] == GV_NONE && !std::binary_search(arrivals.begin(), arrivals.end(), up))
                        return false;
                }
                for (size_t k = chain.size(); k-- > 0;)
                    place(chain[k], rankOfTransform[slotOf(*world[chain[k]].parent)]);
            }
            for (uint32_t s : arrivals) {
                if (!tryCopyTransform(ts, s))
                    return false;
                changes.ranks[rankOfTransform[s]].transforms.push_back(localOfTransform[s]);
                if (*world[s].parent)
                    changes.ranks[rankOfTransform[s]].links.push_back(localOfTransform[s]);
            }
        }
        // 3. transforms that stayed: a parent link that now leads to another rank takes the subtree there; the others are refreshed in place
        for (uint32_t s : kept) {
            if (rankOfTransform[s] == GV_NONE)
                return false;
            const uint32_t up = slotOf(*world[s].parent);
            if (up != GV_NONE && rankOfTransform[up] == GV_NONE)
                return false;
            if (up != GV_NONE && rankOfTransform[up] != rankOfTransform[s]) {
                moveTree(ts, s, rankOfTransform[up], changes);
                continue;
            }
            if (!tryCopyTransform(ts, s))
                return false;
            changes.ranks[rankOfTransform[s]].transforms.push_back(localOfTransform[s]);
            if (s >= linkLo && s < linkHi)
                changes.ranks[rankOfTransform[s]].links.push_back(localOfTransform[s]);
        }
        // 4. mesh components that came, went or changed hands: each follows its entity's transform
        for (size_t p = 0; p < meshSystems.size(); p++) {
            const auto& meshPool = meshSystems[p]->getMeshComponentPool();
            const size_t stride = meshSystems[p]->getMeshComponentSize();
            const uint8_t* data = reinterpret_cast<const uint8_t*>(meshPool.getData());
            const uint32_t meshOccupancy = meshPool.getOccupancy();
            MeshTable& table = meshTables[p];
            if (meshOccupancy < table.entity.size())
                return false;
            table.rank.resize(meshOccupancy, GV_NONE);
            table.local.resize(meshOccupancy, GV_NONE);
            table.entity.resize(meshOccupancy, 0u);
            table.transform.resize(meshOccupancy, GV_NONE);
            table.next.resize(meshOccupancy, GV_NONE);
            for (uint32_t j : meshSlots[p]) {
                if (j >= meshOccupancy)
                    continue;
                const auto* component = reinterpret_cast<const MeshRenderComponent*>(data + (size_t)j * stride);
                placeMesh((uint32_t)p, j, component, slotOf(*component->entity), ranks, changes);
            }
        }
        // 5. a transform that came for an entity whose meshes were waiting without one
        if (!looseMeshes.empty() && !arrivals.empty()) {
            const std::vector<uint32_t> waiting = looseMeshes;
            for (uint32_t ref : waiting) {
                const uint32_t p = ref >> 28, j = ref & 0x0FFFFFFFu;
                const uint32_t t = slotOf(meshTables[p].entity[j]);
                if (t == GV_NONE || rankOfTransform[t] == GV_NONE)
                    continue;
                const auto* component = reinterpret_cast<const MeshRenderComponent*>(
                    reinterpret_cast<const uint8_t*>(meshSystems[p]->getMeshComponentPool().getData()) + (size_t)j * meshSystems[p]->getMeshComponentSize());
                placeMesh(p, j, component, t, ranks, changes);
            }
        }
        return true;
    }

    // Mesh slot (p, j) as the engine holds it now: on the rank of its entity's transform `t` (GV_NONE: it stays where it is — a slot the
    // pool has grown by goes to rank j % ranks), the engine's bytes, the local entity id, the list of its transform.
    void placeMesh(uint32_t p, uint32_t j, const MeshRenderComponent* component, uint32_t t, uint32_t ranks, Changes& changes)
    {
        MeshTable& table = meshTables[p];
        unlinkMesh(p, j);
        const uint32_t entity = *component->entity;
        if (t != GV_NONE && rankOfTransform[t] == GV_NONE)
            t = GV_NONE;
        const bool fresh = table.rank[j] == GV_NONE;
        const uint32_t want = t != GV_NONE ? rankOfTransform[t] : fresh ? j % ranks : table.rank[j];
        if (fresh) {
            table.rank[j] = want;
            table.local[j] = allocMesh(shares[want].meshes[p], j);
            changes.ranks[want].maps[p].push_back(table.local[j]);
        }
        const uint32_t localEntity = !entity ? 0u : t != GV_NONE ? localOfTransform[t] + 1 : kNoTransformEntity;
        moveMesh(p, j, want, localEntity, changes);
        MeshShare& share = shares[want].meshes[p];
        std::memcpy(share.components.data() + (size_t)table.local[j] * share.stride, component, share.stride);
        share.at(table.local[j])->entity = ID<Entity>(localEntity);
        table.entity[j] = entity;
        if (t != GV_NONE) {
            linkMesh(t, p, j);
        } else if (entity) {
            looseMesh(p, j);
        }
    }

    // The tree under world transform slot `root` from its rank to `to`: transforms parents first, each with its meshes.
    void moveTree(const TransformSystem* ts, uint32_t root, uint32_t to, Changes& changes)
    {
        const auto& emap = ts->getEntityMap();
        const uint32_t from = rankOfTransform[root];
        std::vector<uint32_t> tree{root};
        for (size_t k = 0; k < tree.size(); k++) {  // (breadth first: a parent is always in front of its children)
            const TransformComponent* t = worldTransform(ts, tree[k]);
            for (uint32_t c = 0, n = t->childCount(); c < n; c++) {
                const uint32_t child = *t->childs[c];
                const uint32_t slot = child && child < emap.size() ? emap[child] : GV_NONE;
                if (slot != GV_NONE && rankOfTransform[slot] == from)
                    tree.push_back(slot);
            }
            if (tree.size() > rankOfTransform.size())
                throw std::runtime_error("RankShares: a cycle in the transform hierarchy");
        }
        Share& src = shares[from];
        Share& dst = shares[to];
        for (uint32_t s : tree) {
            const uint32_t was = localOfTransform[s];
            freeTransform(src, was);
            changes.ranks[from].transforms.push_back(was);
            const uint32_t now = allocTransform(dst, s);
            rankOfTransform[s] = to;
            localOfTransform[s] = now;
            copyTransform(ts, s);
            changes.ranks[to].transforms.push_back(now);
            if (*dst.transforms[now].parent)
                changes.ranks[to].links.push_back(now);
            for (uint32_t ref = firstMesh[s]; ref != GV_NONE; ref = meshTables[ref >> 28].next[ref & 0x0FFFFFFFu])
                moveMesh(ref >> 28, ref & 0x0FFFFFFFu, to, now + 1, changes);
        }
        changes.movedTrees++;
        changes.movedTransforms += (uint32_t)tree.size();
    }
};

}  // namespace garden
