// rank_shares.hpp — ONE process, N GPUs: what each rank (GPU) holds of the engine's component pools.
//
// The reference is one process with one Manager (source/editor/entry.cpp:135) whose workers split every pool into index ranges
// (ThreadPool::addItems, source/thread-pool.cpp:180-194). Ranks split the WORLD instead: the world cube is cut
// into many more cells than ranks, dealt in rotating rounds (gv_cell_owner: the rule gv_scene_extract_rank and
// garden_amd/multi.py::partition_world use); a ROOT transform's position decides, its descendants follow — no parent chain is
// cut, so a rank computes the world's matrices bit for bit (TransformComponent::calcModel, transform.hpp:197-214).
//
// A share is a set of pools in the engine's own layouts (TransformComponent / MeshRenderComponent-derived, same byte strides), with
// entity ids renumbered per rank (local transform slot i <-> entity i + 1) and, per pool, the local slot -> WORLD slot table that
// gv_pool_set_index_map takes: exchanged lists carry the engine's own slots. Every slot of every mesh pool lives on exactly one
// rank — free slots and meshes without a transform included: the light pass writes isVisible of ALL of them (mesh.cpp:140-153).
#pragma once
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../../../include/garden_vis.h"
#include "garden_host.hpp"

namespace garden {

struct RankShares {
    struct MeshShare {
        std::vector<uint8_t> components;  // local pool, the engine's stride
        std::vector<uint32_t> worldSlot;  // local slot -> slot of the engine's pool
        size_t stride = 0;
        uint32_t occupancy() const noexcept { return (uint32_t)worldSlot.size(); }
    };
    struct Share {
        std::vector<TransformComponent> transforms;  // local transform pool (childs = NULL: the cull path never reads it)
        std::vector<uint32_t> transformWorldSlot;    // local -> world transform slot
        std::vector<uint32_t> entityToTransform;     // local entity id -> local transform slot (GV_NONE: none)
        std::vector<MeshShare> meshes;               // one per mesh system, in meshSystems order
    };
    std::vector<Share> shares;
    std::vector<uint32_t> rankOfTransform, localOfTransform;  // world transform slot -> rank, local slot (GV_NONE: a free slot)
    std::vector<uint32_t> entityOfTransform;                  // world transform slot -> the entity it held when the pools were dealt

    // true: slot still holds the entity it was dealt with (a slot that was freed and handed to another entity since must be dealt again)
    bool sameEntity(const TransformSystem* ts, uint32_t worldSlot) const noexcept
    {
        return worldSlot < entityOfTransform.size() && entityOfTransform[worldSlot] == *worldTransform(ts, worldSlot)->entity;
    }

    static const TransformComponent* worldTransform(const TransformSystem* ts, uint32_t slot) noexcept
    {
        return const_cast<TransformSystem*>(ts)->getComponents().getData() + slot;
    }

    // one world transform into its rank's pool: the same bytes, ids renumbered
    void copyTransform(const TransformSystem* ts, uint32_t worldSlot)
    {
        const uint32_t rank = rankOfTransform[worldSlot];
        if (rank == GV_NONE)
            return;
        const uint32_t local = localOfTransform[worldSlot];
        const auto& emap = ts->getEntityMap();
        TransformComponent& dst = shares[rank].transforms[local];
        std::memcpy(static_cast<void*>(&dst), worldTransform(ts, worldSlot), sizeof(TransformComponent));
        dst.childs = nullptr;
        dst.entity = ID<Entity>(local + 1);
        const uint32_t parent = *dst.parent;
        if (parent) {
            const uint32_t parentSlot = parent < emap.size() ? emap[parent] : GV_NONE;
            if (parentSlot == GV_NONE || rankOfTransform[parentSlot] != rank)
                throw std::runtime_error("RankShares: a parent without a transform, or on another rank than its child");
            dst.parent = ID<Entity>(localOfTransform[parentSlot] + 1);
        }
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
            if (!*world[i].entity || rootOf[i] != GV_NONE)
                continue;
            chain.clear();
            uint32_t s = i;
            while (rootOf[s] == GV_NONE) {
                chain.push_back(s);
                const uint32_t parent = *world[s].parent;
                const uint32_t up = parent && parent < emap.size() ? emap[parent] : GV_NONE;
                if (up == GV_NONE || chain.size() > occupancy) {
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
            entityOfTransform[i] = *world[i].entity;
        for (uint32_t i = 0; i < occupancy; i++) {
            if (!*world[i].entity)
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
            share.entityToTransform.assign((size_t)n + 2, GV_NONE);  // id 0 = null, ids 1..n, id n + 1 = "an entity without a transform"
            for (uint32_t k = 0; k < n; k++)
                share.entityToTransform[k + 1] = k;
            share.meshes.assign(meshSystems.size(), MeshShare{});
        }
        for (uint32_t i = 0; i < occupancy; i++)
            copyTransform(ts, i);
        for (size_t p = 0; p < meshSystems.size(); p++) {
            const auto& meshPool = meshSystems[p]->getMeshComponentPool();
            const size_t stride = meshSystems[p]->getMeshComponentSize();
            const uint8_t* data = reinterpret_cast<const uint8_t*>(meshPool.getData());
            const uint32_t meshOccupancy = meshPool.getOccupancy();
            for (uint32_t r = 0; r < ranks; r++)
                shares[r].meshes[p].stride = stride;
            for (uint32_t j = 0; j < meshOccupancy; j++) {
                const auto* component = reinterpret_cast<const MeshRenderComponent*>(data + (size_t)j * stride);
                const uint32_t entity = *component->entity;
                const uint32_t transformSlot = entity && entity < emap.size() ? emap[entity] : GV_NONE;
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
                    local->entity = ID<Entity>((uint32_t)shares[rank].transforms.size() + 1);  // alive, no transform (mesh.cpp:149-153)
                share.worldSlot.push_back(j);
            }
        }
    }
};

}  // namespace garden
