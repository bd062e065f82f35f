// ecsm_lite.hpp — the slice of the ecsm API (cfnptr/ecsm, an EMPTY submodule in the reference checkout:
// .gitmodules:1-3) that Garden's visibility path touches, so the GPU system can be written and tested as the
// ecsm System it would be in the engine. API surface inferred from the reference's docs and call sites:
//   docs/ECS/Systems.md:23-27,86-129,139-153   System, createSystem, ordered/unordered events, subscribe
//   docs/ECS/Components.md:21-24,42-46,116-172  Component, LinearPool getData/getOccupancy/getCount, Singleton
//   docs/ECS/Entities.md:38-83                  create/destroy entities, deferred destroy until dispose()
//   source/system/render/mesh.cpp:35,42-63      ECSM_SUBSCRIBE_TO_EVENT(name, Class::method)
// Not a reimplementation of ecsm: no Ref<>, no serialization, no component-type reflection.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <typeindex>
#include <unordered_map>
#include <vector>

namespace ecsm {

struct Entity;

// 1-based index into a LinearPool; 0 = null (docs/ECS/Components.md:42-46).
template <class T>
struct ID {
    uint32_t index = 0;
    ID() = default;
    explicit ID(uint32_t i) : index(i) {}
    explicit operator bool() const noexcept { return index != 0; }
    uint32_t operator*() const noexcept { return index; }
    bool operator==(ID o) const noexcept { return index == o.index; }
    bool operator!=(ID o) const noexcept { return index != o.index; }
};

template <class T>
struct View {
    T* ptr = nullptr;
    T* operator->() const noexcept { return ptr; }
    T* operator*() const noexcept { return ptr; }
    explicit operator bool() const noexcept { return ptr != nullptr; }
};

struct Component {
    ID<Entity> entity = {};
    ID<Entity> getEntity() const noexcept { return entity; }
};

// Contiguous item pool with holes: getData() is the base pointer, getOccupancy() the high-water slot count
// including holes, getCount() the live count; destroyed items are recycled on dispose() so pointers stay
// valid within a frame; create() may reallocate (docs/ECS/Components.md:137-170, Entities.md:40-59).
template <class T, bool DestroyItems = true>
class LinearPool {
    T* items = nullptr;
    uint32_t occupancy = 0, capacity = 0;
    std::vector<ID<T>> freeItems, garbageItems;

public:
    LinearPool() = default;
    LinearPool(const LinearPool&) = delete;
    ~LinearPool() { std::free(items); }
    ID<T> create()
    {
        if (!freeItems.empty()) {
            auto id = freeItems.back();
            freeItems.pop_back();
            new (&items[*id - 1]) T();
            return id;
        }
        if (occupancy == capacity) {
            const uint32_t cap = capacity ? capacity * 2 : 64;
            void* mem = nullptr;
            if (posix_memalign(&mem, 64, sizeof(T) * (size_t)cap) != 0)
                throw std::bad_alloc();
            if (items)
                std::memcpy(mem, items, sizeof(T) * (size_t)occupancy);  // components here are trivially relocatable
            std::free(items);
            items = static_cast<T*>(mem);
            capacity = cap;
        }
        new (&items[occupancy]) T();
        return ID<T>(++occupancy);
    }
    void destroy(ID<T> id)
    {
        if (id)
            garbageItems.push_back(id);
    }
    void dispose()
    {
        for (auto id : garbageItems) {
            items[*id - 1] = T();  // entity == null marks a free slot (mesh.cpp:142)
            freeItems.push_back(id);
        }
        garbageItems.clear();
    }
    View<T> get(ID<T> id) const noexcept { return View<T>{&items[*id - 1]}; }
    T* getData() const noexcept { return items; }
    uint32_t getOccupancy() const noexcept { return occupancy; }
    uint32_t getCount() const noexcept { return occupancy - (uint32_t)freeItems.size(); }
    const std::vector<ID<T>>& getGarbage() const noexcept { return garbageItems; }  // destroyed, not yet disposed
};

class Manager;

class System {
public:
    virtual ~System() = default;
};

// Base of systems that own one component type's pool and the entity -> component map behind
// Manager::tryGet<C>(entity) (docs/ECS/Components.md:21-24).
class ComponentSystemBase : public System {
public:
    virtual void removeOf(ID<Entity> entity) = 0;
    virtual void disposeComponents() = 0;
};

template <class C, bool DestroyItems = true>
class ComponentSystem : public ComponentSystemBase {
protected:
    LinearPool<C, DestroyItems> components;
    std::vector<uint32_t> entityToComponent;  // entity index -> pool slot (0-based) or UINT32_MAX

public:
    static constexpr uint32_t none = UINT32_MAX;
    LinearPool<C, DestroyItems>& getComponents() noexcept { return components; }
    const std::vector<uint32_t>& getEntityMap() const noexcept { return entityToComponent; }
    View<C> addTo(ID<Entity> entity)
    {
        auto id = components.create();
        auto view = components.get(id);
        view->entity = entity;
        if (entityToComponent.size() <= *entity)
            entityToComponent.resize((size_t)*entity * 2 + 64, none);
        entityToComponent[*entity] = *id - 1;
        return view;
    }
    View<C> tryGetOf(ID<Entity> entity) const noexcept
    {
        if (*entity >= entityToComponent.size() || entityToComponent[*entity] == none)
            return {};
        return View<C>{components.getData() + entityToComponent[*entity]};
    }
    void removeOf(ID<Entity> entity) override
    {
        if (*entity < entityToComponent.size() && entityToComponent[*entity] != none) {
            components.destroy(ID<C>(entityToComponent[*entity] + 1));
            entityToComponent[*entity] = none;
        }
    }
    void disposeComponents() override { components.dispose(); }
};

template <class T>
class Singleton {
protected:
    // the base pointer is stored while the derived object is still under construction; the downcast happens at
    // access time, when the object is a complete T
    inline static Singleton* singletonInstance = nullptr;

public:
    struct Instance {
        static T* get()
        {
            if (!singletonInstance)
                throw std::runtime_error("singleton not created");
            return static_cast<T*>(singletonInstance);
        }
        static T* tryGet() noexcept { return static_cast<T*>(singletonInstance); }
    };
    Singleton() { singletonInstance = this; }
    ~Singleton() { singletonInstance = nullptr; }
};

// Entities, systems and events. Ordered events run each Manager::update() in registration order
// (Input -> Update -> Output: source/system/loop.cpp:58-59); unordered events run on runEvent(name)
// (docs/ECS/Systems.md:86-116).
class Manager final : public Singleton<Manager> {
    using Callback = std::function<void()>;
    struct Event {
        std::vector<Callback> subscribers;
        bool ordered = false;
    };
    std::vector<std::unique_ptr<System>> systems;
    std::unordered_map<std::type_index, System*> systemByType;
    std::unordered_map<std::type_index, ComponentSystemBase*> componentSystems;
    std::map<std::string, Event> events;
    std::vector<std::string> orderedEvents;
    uint32_t entityOccupancy = 0;
    std::vector<uint32_t> freeEntities, garbageEntities;
    bool initialized = false;

public:
    Manager()
    {
        for (auto name : {"PreInit", "Init", "PostInit"})
            registerEvent(name);
        for (auto name : {"Input", "Update", "Output"})
            registerEventAfter(name);
    }
    void registerEvent(const std::string& name) { events.emplace(name, Event{}); }
    void registerEventAfter(const std::string& name)
    {
        events[name].ordered = true;
        orderedEvents.push_back(name);
    }
    bool hasEvent(const std::string& name) const { return events.count(name) != 0; }
    void subscribeToEvent(const std::string& name, Callback fn)
    {
        auto it = events.find(name);
        if (it == events.end())
            throw std::runtime_error("event is not registered: " + name);
        it->second.subscribers.push_back(std::move(fn));
    }
    void runEvent(const std::string& name)
    {
        auto it = events.find(name);
        if (it == events.end())
            throw std::runtime_error("event is not registered: " + name);
        for (auto& fn : it->second.subscribers)
            fn();
    }
    template <class T, class... Args>
    T* createSystem(Args&&... args)
    {
        auto sys = std::unique_ptr<T>(new T(std::forward<Args>(args)...));
        T* raw = sys.get();
        systemByType[std::type_index(typeid(T))] = raw;
        systems.push_back(std::move(sys));
        return raw;
    }
    template <class C, class S>
    void registerComponents(S* system)
    {
        componentSystems[std::type_index(typeid(C))] = system;
    }
    template <class T>
    T* get() const
    {
        auto it = systemByType.find(std::type_index(typeid(T)));
        if (it == systemByType.end())
            throw std::runtime_error("system is not created");
        return static_cast<T*>(it->second);
    }
    template <class T>
    T* tryGet() const noexcept
    {
        auto it = systemByType.find(std::type_index(typeid(T)));
        return it == systemByType.end() ? nullptr : static_cast<T*>(it->second);
    }
    const std::vector<std::unique_ptr<System>>& getSystems() const noexcept { return systems; }

    ID<Entity> createEntity()
    {
        if (!freeEntities.empty()) {
            auto e = freeEntities.back();
            freeEntities.pop_back();
            return ID<Entity>(e);
        }
        return ID<Entity>(++entityOccupancy);
    }
    void destroy(ID<Entity> entity)
    {
        for (auto& cs : componentSystems)
            cs.second->removeOf(entity);
        garbageEntities.push_back(*entity);
    }
    uint32_t getEntityOccupancy() const noexcept { return entityOccupancy; }

    void initialize()
    {
        runEvent("PreInit");
        runEvent("Init");
        runEvent("PostInit");
        initialized = true;
    }
    void update()
    {
        for (auto& name : orderedEvents)
            runEvent(name);
        disposeGarbage();
    }
    void disposeGarbage()
    {
        for (auto& cs : componentSystems)
            cs.second->disposeComponents();
        for (auto e : garbageEntities)
            freeEntities.push_back(e);
        garbageEntities.clear();
    }
};

#define ECSM_SUBSCRIBE_TO_EVENT(name, func) \
    ::ecsm::Manager::Instance::get()->subscribeToEvent(name, std::bind(&func, this))

}  // namespace ecsm
