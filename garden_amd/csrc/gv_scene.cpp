// gv_scene.cpp — scene ingest: a Garden scene file goes straight into column (SoA) pools that the
// visibility pass binds with gv_*_bind_columns; the 80-byte TransformComponent / 48+-byte MeshRenderComponent AoS is
// never materialised (10^8 entities: 4.5 GB of columns instead of 13 GB of components).
//
// Restates, for the fields the path reads, what the reference does when it loads a scene:
//   ResourceSystem::loadScene                 source/system/resource.cpp:2421-2510   entities[] -> components[] -> ".type"
//   TransformSystem::deserialize              source/system/transform.cpp:517-560    uid / position / rotation / scale /
//                                                                                    isActive / parent
//   TransformSystem::postDeserialize          source/system/transform.cpp:561-583    parent uids -> setParent, in file order
//   TransformComponent::setParent             source/system/transform.cpp:129-195    ancestorsActive taken from the parent
//                                                                                    AT THAT MOMENT, for this entity only
//   <Mesh>RenderSystem::deserialize           e.g. source/system/render/sprite.cpp:206-207   "aabb", "isEnabled"
//   JsonDeserializer::read(f32x4, n) / quat / Aabb / bool / string   source/json-serialize.cpp:873-898,768-781,851-863
// Numbers follow nlohmann's typing as the reference sees it: only literals with a fraction or an exponent are
// "number_float" and accepted for float fields (an integer literal leaves the default in place); text -> double
// (strtod) -> float.
//
// Host code only (no HIP): GvScene can be parsed, inspected and destroyed without a device; gv_scene_bind needs a context.
#include <cerrno>
#include <algorithm>
#include <charconv>
#include <new>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/garden_vis.h"

namespace {

struct MeshColumnsOwned {
    std::vector<uint32_t> entity;
    std::vector<uint8_t> is_enabled, is_visible;
    std::vector<float> aabb_min, aabb_max;  // 3 per mesh
    std::string type;
    bool mapped = false;
};

}  // namespace

struct GvScene {
    // transform columns, slot = order of appearance
    std::vector<uint32_t> entity, parent;
    std::vector<uint64_t> uid;
    std::vector<float> position, scale, rotation;  // 3, 3, 4 per transform
    std::vector<uint8_t> self_active, ancestors_active, model_with_ancestors;
    std::vector<uint32_t> entity_to_transform;  // [entity id] -> slot or GV_NONE
    MeshColumnsOwned pools[GV_MAX_POOLS];
    GvSceneInfo info{};
    // a tile cut out of a larger scene (gv_scene_extract_tile): local slot -> slot in the scene it was cut from
    std::vector<uint32_t> transform_global, mesh_global[GV_MAX_POOLS];
    bool is_tile = false;
};

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;

struct Parser {
    const char* p;
    const char* end;
    std::string error;

    bool fail(const char* fmt, ...)
    {
        if (error.empty()) {
            char buf[256];
            va_list ap;
            va_start(ap, fmt);
            vsnprintf(buf, sizeof(buf), fmt, ap);
            va_end(ap);
            error = buf;
        }
        return false;
    }
    void ws()
    {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r'))
            p++;
    }
    bool eat(char c)
    {
        ws();
        if (p < end && *p == c) {
            p++;
            return true;
        }
        return false;
    }
    char peek()
    {
        ws();
        return p < end ? *p : '\0';
    }
    bool at_number()
    {
        const char c = peek();
        return c == '-' || (c >= '0' && c <= '9') || (allow_nan && c == 'n' && (size_t)(end - p) >= 3 && memcmp(p, "nan", 3) == 0);
    }
    bool string(std::string* out)
    {
        ws();
        if (p >= end || *p != '"')
            return fail("expected a string at byte %zu", offset());
        p++;
        if (out)
            out->clear();
        while (p < end && *p != '"') {
            char c = *p++;
            if (c == '\\') {
                if (p >= end)
                    break;
                const char e = *p++;
                switch (e) {
                case 'b': c = '\b'; break;
                case 'f': c = '\f'; break;
                case 'n': c = '\n'; break;
                case 'r': c = '\r'; break;
                case 't': c = '\t'; break;
                case 'u': {
                    if (end - p < 4)
                        return fail("truncated \\u escape");
                    unsigned cp = 0;
                    for (int k = 0; k < 4; k++) {
                        const char h = *p++;
                        cp = cp * 16 + (unsigned)(h >= '0' && h <= '9' ? h - '0' : (h | 32) >= 'a' && (h | 32) <= 'f' ? (h | 32) - 'a' + 10 : 0);
                    }
                    if (out) {  // UTF-8 (surrogate pairs are not combined: names this loader reads are ASCII)
                        if (cp < 0x80) {
                            out->push_back((char)cp);
                        } else if (cp < 0x800) {
                            out->push_back((char)(0xC0 | (cp >> 6)));
                            out->push_back((char)(0x80 | (cp & 0x3F)));
                        } else {
                            out->push_back((char)(0xE0 | (cp >> 12)));
                            out->push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
                            out->push_back((char)(0x80 | (cp & 0x3F)));
                        }
                    }
                    continue;
                }
                default: c = e; break;  // \" \\ \/
                }
            }
            if (out)
                out->push_back(c);
        }
        if (p >= end)
            return fail("unterminated string");
        p++;
        return true;
    }
    // JSON number; is_float = the literal has a fraction or an exponent (nlohmann's number_float)
    bool number(double* value, bool* is_float)
    {
        ws();
        const char* s = p;
        if (allow_nan && (size_t)(end - p) >= 3 && memcmp(p, "nan", 3) == 0) {  // only the BSON transcoder writes this
            p += 3;
            if (value)
                *value = NAN;
            if (is_float)
                *is_float = true;
            return true;
        }
        if (p < end && (*p == '-' || *p == '+'))
            p++;
        bool digits = false, flt = false;
        while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '-' || *p == '+')) {
            if (*p >= '0' && *p <= '9')
                digits = true;
            if (*p == '.' || *p == 'e' || *p == 'E')
                flt = true;
            p++;
        }
        if (!digits)
            return fail("expected a value at byte %zu", (size_t)(s - begin));
        if (value) {  // correctly rounded text -> double, as strtod / nlohmann's lexer give it
            const char* first = *s == '+' ? s + 1 : s;
            const auto res = std::from_chars(first, p, *value);
            if (res.ec == std::errc::result_out_of_range)
                *value = (*first == '-') ? -HUGE_VAL : HUGE_VAL;
            else if (res.ec != std::errc() || res.ptr != p)
                return fail("malformed number at byte %zu", (size_t)(s - begin));
        }
        if (is_float)
            *is_float = flt;
        return true;
    }
    bool literal(const char* word)
    {
        const size_t n = strlen(word);
        if ((size_t)(end - p) >= n && memcmp(p, word, n) == 0) {
            p += n;
            return true;
        }
        return fail("unexpected token at byte %zu", offset());
    }
    bool skip_value()
    {
        const char c = peek();
        if (c == '{') {
            p++;
            if (eat('}'))
                return true;
            do {
                if (!string(nullptr) || !eat(':'))
                    return fail("malformed object at byte %zu", offset());
                if (!skip_value())
                    return false;
            } while (eat(','));
            return eat('}') || fail("expected '}' at byte %zu", offset());
        }
        if (c == '[') {
            p++;
            if (eat(']'))
                return true;
            do {
                if (!skip_value())
                    return false;
            } while (eat(','));
            return eat(']') || fail("expected ']' at byte %zu", offset());
        }
        if (c == '"')
            return string(nullptr);
        if (c == 't')
            return literal("true");
        if (c == 'f')
            return literal("false");
        if (c == 'n' && allow_nan && (size_t)(end - p) >= 3 && memcmp(p, "nan", 3) == 0)
            return number(nullptr, nullptr);
        if (c == 'n')
            return literal("null");
        return number(nullptr, nullptr);
    }
    const char* begin = nullptr;
    bool allow_nan = false;
    size_t offset() const { return (size_t)(p - begin); }
};

// JsonDeserializer::read(name, f32x4&, components) (json-serialize.cpp:873-898): a float literal splats, an object
// sets the components it has as float literals; anything else leaves `v` alone.
bool read_vector(Parser& ps, float* v, int components)
{
    const char c = ps.peek();
    if (c == '{') {
        ps.p++;
        if (ps.eat('}'))
            return true;
        std::string key;
        do {
            if (!ps.string(&key) || !ps.eat(':'))
                return ps.fail("malformed vector object at byte %zu", ps.offset());
            int k = -1;
            if (key.size() == 1)
                k = key[0] == 'x' ? 0 : key[0] == 'y' ? 1 : key[0] == 'z' ? 2 : key[0] == 'w' ? 3 : -1;
            if (k >= 0 && k < components && ps.at_number()) {
                double d;
                bool flt;
                if (!ps.number(&d, &flt))
                    return false;
                if (flt)
                    v[k] = (float)d;
            } else if (!ps.skip_value()) {
                return false;
            }
        } while (ps.eat(','));
        return ps.eat('}') || ps.fail("expected '}' at byte %zu", ps.offset());
    }
    if (ps.at_number()) {
        double d;
        bool flt;
        if (!ps.number(&d, &flt))
            return false;
        if (flt)
            for (int k = 0; k < components; k++)
                v[k] = (float)d;
        return true;
    }
    return ps.skip_value();
}

// modp_b64 URL alphabet (A-Z a-z 0-9 - _), 11 characters = 8 bytes with the padding character cut off
// (transform.cpp:473-475,520-523): the stored uint64, little-endian.
bool decode_uid(const std::string& s, uint64_t* uid)
{
    if (s.size() != 11)
        return false;  // transform.cpp:520: size + 1 == modp_b64_encode_data_len(8)
    uint8_t bytes[9] = {};
    uint32_t acc = 0;
    int bits = 0, n = 0;
    for (char ch : s) {
        int v;
        if (ch >= 'A' && ch <= 'Z') v = ch - 'A';
        else if (ch >= 'a' && ch <= 'z') v = ch - 'a' + 26;
        else if (ch >= '0' && ch <= '9') v = ch - '0' + 52;
        else if (ch == '-') v = 62;
        else if (ch == '_') v = 63;
        else return false;
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            if (n < 8)
                bytes[n++] = (uint8_t)(acc >> bits);
        }
    }
    if (n != 8)
        return false;
    memcpy(uid, bytes, 8);
    return true;
}

struct PendingParent {
    uint32_t slot;
    uint64_t parent_uid;
};

struct Loader {
    Parser ps;
    GvScene* sc;
    std::unordered_map<std::string, uint32_t> pool_of_type;
    std::unordered_map<uint64_t, uint32_t> entity_of_uid;  // deserializedEntities (transform.cpp:525)
    std::vector<PendingParent> parents;                    // deserializedParents  (transform.cpp:553)
    bool add_root = false;                                 // loadScene's addRootEntity
    uint32_t root_entity = 0;

    // finds ".type" in the component object that starts at ps.p (which is left unchanged)
    bool component_type(std::string* type)
    {
        const char* start = ps.p;
        type->clear();
        bool found = false, first_key = true;
        if (!ps.eat('{'))
            return ps.fail("component is not an object at byte %zu", ps.offset());
        if (!ps.eat('}')) {
            std::string key;
            do {
                if (!ps.string(&key) || !ps.eat(':'))
                    return ps.fail("malformed component at byte %zu", ps.offset());
                if (key == ".type" && ps.peek() == '"') {
                    if (!ps.string(type))
                        return false;
                    found = true;
                    if (first_key) {  // the reference's writer sorts keys: ".type" leads and nothing else needs scanning
                        ps.p = start;
                        return true;
                    }
                } else if (!ps.skip_value()) {
                    return false;
                }
                first_key = false;
            } while (ps.eat(','));
            if (!ps.eat('}'))
                return ps.fail("expected '}' at byte %zu", ps.offset());
        }
        ps.p = start;
        if (!found)
            type->clear();
        return true;
    }

    bool transform(uint32_t entity)  // TransformSystem::deserialize, transform.cpp:517-560
    {
        const uint32_t slot = (uint32_t)sc->entity.size();
        float pos[3] = {0, 0, 0}, scl[3] = {1, 1, 1}, rot[4] = {0, 0, 0, 1};
        bool self_active = true, have_uid = false, have_parent = false;
        uint64_t uid = 0, parent_uid = 0;
        std::string key, text;
        ps.eat('{');
        if (!ps.eat('}')) {
            do {
                if (!ps.string(&key) || !ps.eat(':'))
                    return ps.fail("malformed Transform at byte %zu", ps.offset());
                const char c = ps.peek();
                if (key == "uid" && c == '"') {
                    if (!ps.string(&text))
                        return false;
                    have_uid = decode_uid(text, &uid);
                } else if (key == "parent" && c == '"') {
                    if (!ps.string(&text))
                        return false;
                    have_parent = decode_uid(text, &parent_uid);
                } else if (key == "position") {
                    if (!read_vector(ps, pos, 3))
                        return false;
                } else if (key == "scale") {
                    if (!read_vector(ps, scl, 3))
                        return false;
                } else if (key == "rotation" && c == '{') {  // quat: objects only (json-serialize.cpp:772)
                    if (!read_vector(ps, rot, 4))
                        return false;
                } else if (key == "isActive" && (c == 't' || c == 'f')) {
                    self_active = c == 't';
                    if (!ps.literal(c == 't' ? "true" : "false"))
                        return false;
                } else if (!ps.skip_value()) {
                    return false;
                }
            } while (ps.eat(','));
            if (!ps.eat('}'))
                return ps.fail("expected '}' at byte %zu", ps.offset());
        }
        sc->entity.push_back(entity);
        sc->parent.push_back(0);
        sc->uid.push_back(have_uid ? uid : 0);
        sc->position.insert(sc->position.end(), pos, pos + 3);
        sc->scale.insert(sc->scale.end(), scl, scl + 3);
        sc->rotation.insert(sc->rotation.end(), rot, rot + 4);
        sc->self_active.push_back(self_active ? 1 : 0);
        sc->ancestors_active.push_back(1);
        sc->model_with_ancestors.push_back(1);
        if (sc->entity_to_transform.size() <= entity)
            sc->entity_to_transform.resize((size_t)entity + 1, kNone);
        sc->entity_to_transform[entity] = slot;
        if (have_uid && !entity_of_uid.emplace(uid, entity).second)
            sc->info.duplicate_uids++;  // "Deserialized entity with already existing UID": the first one keeps it
        if (have_parent) {
            if (have_uid && parent_uid == uid)
                sc->info.self_parents++;  // "Deserialized entity with the same parent UID": no link
            else
                parents.push_back({slot, parent_uid});
        }
        return true;
    }

    bool aabb(float* mn, float* mx)  // JsonDeserializer::read(name, Aabb&), json-serialize.cpp:851-863
    {
        // both start from the component's current min — the reference's quirk (line 856) — and only a valid pair is taken
        float lo[3] = {mn[0], mn[1], mn[2]}, hi[3] = {mn[0], mn[1], mn[2]};
        if (ps.peek() != '{')
            return ps.skip_value();
        ps.p++;
        std::string key;
        if (!ps.eat('}')) {
            do {
                if (!ps.string(&key) || !ps.eat(':'))
                    return ps.fail("malformed aabb at byte %zu", ps.offset());
                if (key == "min") {
                    if (!read_vector(ps, lo, 3))
                        return false;
                } else if (key == "max") {
                    if (!read_vector(ps, hi, 3))
                        return false;
                } else if (!ps.skip_value()) {
                    return false;
                }
            } while (ps.eat(','));
            if (!ps.eat('}'))
                return ps.fail("expected '}' at byte %zu", ps.offset());
        }
        // Aabb::trySet (cfnptr/math, absent): build-defined as "min <= max on every axis, else unchanged"
        if (lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]) {
            memcpy(mn, lo, 12);
            memcpy(mx, hi, 12);
        }
        return true;
    }

    bool mesh(uint32_t entity, MeshColumnsOwned& pool)  // e.g. SpriteRenderSystem::deserialize, sprite.cpp:206-207
    {
        float mn[3] = {-0.5f, -0.5f, -0.5f}, mx[3] = {0.5f, 0.5f, 0.5f};  // Aabb::one (mesh.hpp:54)
        bool enabled = true;
        std::string key;
        ps.eat('{');
        if (!ps.eat('}')) {
            do {
                if (!ps.string(&key) || !ps.eat(':'))
                    return ps.fail("malformed mesh component at byte %zu", ps.offset());
                const char c = ps.peek();
                if (key == "aabb") {
                    if (!aabb(mn, mx))
                        return false;
                } else if (key == "isEnabled" && (c == 't' || c == 'f')) {
                    enabled = c == 't';
                    if (!ps.literal(c == 't' ? "true" : "false"))
                        return false;
                } else if (!ps.skip_value()) {
                    return false;
                }
            } while (ps.eat(','));
            if (!ps.eat('}'))
                return ps.fail("expected '}' at byte %zu", ps.offset());
        }
        pool.entity.push_back(entity);
        pool.is_enabled.push_back(enabled ? 1 : 0);
        pool.is_visible.push_back(0);
        pool.aabb_min.insert(pool.aabb_min.end(), mn, mn + 3);
        pool.aabb_max.insert(pool.aabb_max.end(), mx, mx + 3);
        return true;
    }

    bool entity_object(uint32_t* next_entity)  // resource.cpp:2428-2506
    {
        if (!ps.eat('{'))
            return ps.fail("entity is not an object at byte %zu", ps.offset());
        bool had_components = false;
        if (!ps.eat('}')) {
            std::string key, type;
            do {
                if (!ps.string(&key) || !ps.eat(':'))
                    return ps.fail("malformed entity at byte %zu", ps.offset());
                if (key != "components" || ps.peek() != '[' || had_components) {
                    if (!ps.skip_value())
                        return false;
                    continue;
                }
                had_components = true;
                ps.p++;
                if (ps.eat(']')) {
                    sc->info.skipped_entities++;  // "Missing scene entity components": no entity is created
                    continue;
                }
                const uint32_t entity = (*next_entity)++;  // manager->createEntity()
                sc->info.entity_count++;
                uint32_t seen_pools = 0;
                bool seen_transform = false;
                do {
                    if (!component_type(&type))
                        return false;
                    if (type == "Transform") {
                        if (seen_transform)
                            return ps.fail("entity %u has two Transform components", entity);
                        seen_transform = true;
                        if (!transform(entity))
                            return false;
                        continue;
                    }
                    auto it = pool_of_type.find(type);
                    if (it != pool_of_type.end()) {
                        if (seen_pools & (1u << it->second))
                            return ps.fail("entity %u has two %s components", entity, type.c_str());
                        seen_pools |= 1u << it->second;
                        if (!mesh(entity, sc->pools[it->second]))
                            return false;
                        continue;
                    }
                    sc->info.other_components++;  // a component this pass does not read
                    if (!ps.skip_value())
                        return false;
                } while (ps.eat(','));
                if (!ps.eat(']'))
                    return ps.fail("expected ']' at byte %zu", ps.offset());
                if (root_entity && seen_transform) {  // resource.cpp:2497-2502: transformView->setParent(rootEntity)
                    const uint32_t slot = sc->entity_to_transform[entity];
                    sc->parent[slot] = root_entity;
                    sc->ancestors_active[slot] = 1;  // the root is active (default flags)
                }
            } while (ps.eat(','));
            if (!ps.eat('}'))
                return ps.fail("expected '}' at byte %zu", ps.offset());
        }
        return true;
    }

    bool run()
    {
        if (!ps.eat('{'))
            return ps.fail("scene is not a JSON object");
        uint32_t next_entity = 1;
        if (add_root) {  // loadScene(path, addRootEntity = true), resource.cpp:2398-2407: a default transform on top
            root_entity = next_entity++;
            sc->info.entity_count++;
            const float pos[3] = {0, 0, 0}, scl[3] = {1, 1, 1}, rot[4] = {0, 0, 0, 1};
            sc->entity.push_back(root_entity);
            sc->parent.push_back(0);
            sc->uid.push_back(0);
            sc->position.insert(sc->position.end(), pos, pos + 3);
            sc->scale.insert(sc->scale.end(), scl, scl + 3);
            sc->rotation.insert(sc->rotation.end(), rot, rot + 4);
            sc->self_active.push_back(1);
            sc->ancestors_active.push_back(1);
            sc->model_with_ancestors.push_back(1);
            sc->entity_to_transform.resize((size_t)root_entity + 1, kNone);
            sc->entity_to_transform[root_entity] = 0;
        }
        if (!ps.eat('}')) {
            std::string key;
            do {
                if (!ps.string(&key) || !ps.eat(':'))
                    return ps.fail("malformed scene at byte %zu", ps.offset());
                if (key == "entities" && ps.peek() == '[') {
                    ps.p++;
                    if (!ps.eat(']')) {
                        do {
                            if (!entity_object(&next_entity))
                                return false;
                        } while (ps.eat(','));
                        if (!ps.eat(']'))
                            return ps.fail("expected ']' at byte %zu", ps.offset());
                    }
                } else if (!ps.skip_value()) {
                    return false;
                }
            } while (ps.eat(','));
            if (!ps.eat('}'))
                return ps.fail("expected '}' at byte %zu", ps.offset());
        }
        ps.ws();
        if (ps.p != ps.end)
            return ps.fail("trailing bytes after the scene object at byte %zu", ps.offset());
        if (sc->entity_to_transform.size() < next_entity)
            sc->entity_to_transform.resize(next_entity, kNone);
        // postDeserialize (transform.cpp:561-583): links in file order; setParent (transform.cpp:129-195) takes
        // ancestorsActive from the parent's flags as they are at that moment and does not touch descendants
        for (const PendingParent& pp : parents) {
            auto it = entity_of_uid.find(pp.parent_uid);
            if (it == entity_of_uid.end()) {
                sc->info.unresolved_parents++;  // "Deserialized entity parent does not exist"
                continue;
            }
            const uint32_t parent_entity = it->second;
            const uint32_t ps_slot = sc->entity_to_transform[parent_entity];
            if (parent_entity == sc->entity[pp.slot])
                continue;  // GARDEN_ASSERT(parent != entity): a duplicate uid can point an entity at itself
            sc->parent[pp.slot] = parent_entity;
            sc->ancestors_active[pp.slot] = (sc->self_active[ps_slot] && sc->ancestors_active[ps_slot]) ? 1 : 0;
        }
        sc->info.transform_count = (uint32_t)sc->entity.size();
        for (uint32_t k = 0; k < GV_MAX_POOLS; k++)
            sc->info.mesh_count[k] = (uint32_t)sc->pools[k].entity.size();
        return true;
    }
};


// ---- BSON (packed builds): ResourceSystem::loadScene reads the scene through JsonDeserializer::load(vector<uint8>)
// = nlohmann::json::from_bson (resource.cpp:2359-2377, json-serialize.cpp:338-344); tools ship it as
// json::to_bson(text) with "debugName" keys erased (json2bson.cpp:41-66). from_bson maps 0x01 double -> number_float,
// 0x10 / 0x12 int32 / int64 -> number_integer, 0x11 -> number_unsigned, 0x08 bool, 0x0A null, 0x02 string,
// 0x03 document, 0x04 array (keys ignored, element order kept), 0x05 binary — the same value categories the text
// parser distinguishes, so the document is re-written as JSON text for the one loader above: doubles with 17
// significant digits and always a fraction or exponent (the float / integer distinction IS the typing rule), NaN as
// the private token `nan`, infinities as an overflowing literal. Everything else nlohmann rejects is rejected here.
struct BsonReader {
    const uint8_t* p;
    const uint8_t* end;
    std::string* out;
    std::string error;

    bool fail(const char* what)
    {
        if (error.empty())
            error = std::string("BSON: ") + what;
        return false;
    }
    bool need(size_t n) { return (size_t)(end - p) >= n || fail("truncated document"); }
    template <typename T>
    bool get(T* v)
    {
        if (!need(sizeof(T)))
            return false;
        memcpy(v, p, sizeof(T));  // BSON is little-endian, as is every host this library builds for
        p += sizeof(T);
        return true;
    }
    bool cstring(std::string* s)
    {
        const void* z = memchr(p, 0, (size_t)(end - p));
        if (!z)
            return fail("unterminated key");
        s->assign(reinterpret_cast<const char*>(p), reinterpret_cast<const char*>(z));
        p = static_cast<const uint8_t*>(z) + 1;
        return true;
    }
    void quoted(const char* s, size_t n)
    {
        out->push_back('"');
        for (size_t k = 0; k < n; k++) {
            const unsigned char c = (unsigned char)s[k];
            if (c == '"' || c == '\\') {
                out->push_back('\\');
                out->push_back((char)c);
            } else if (c < 0x20) {
                char buf[8];
                snprintf(buf, sizeof(buf), "\\u%04x", c);
                out->append(buf);
            } else {
                out->push_back((char)c);
            }
        }
        out->push_back('"');
    }
    bool document(bool as_array, int depth)
    {
        if (depth > 256)
            return fail("nesting deeper than 256 levels");
        int32_t size;
        const uint8_t* start = p;
        if (!get(&size))
            return false;
        if (size < 5 || (size_t)size > (size_t)(end - start))
            return fail("document size out of range");
        const uint8_t* stop = start + size;
        out->push_back(as_array ? '[' : '{');
        bool first = true;
        std::string key;
        for (;;) {
            uint8_t type;
            if (p >= stop || !get(&type))
                return fail("document without terminator");
            if (type == 0)
                break;
            if (!cstring(&key))
                return false;
            if (!first)
                out->push_back(',');
            first = false;
            if (!as_array) {
                quoted(key.data(), key.size());
                out->push_back(':');
            }
            char buf[40];
            switch (type) {
            case 0x01: {
                double d;
                if (!get(&d))
                    return false;
                if (d != d) {
                    out->append("nan");
                } else if (d == HUGE_VAL || d == -HUGE_VAL) {
                    out->append(d < 0 ? "-1e999" : "1e999");
                } else {
                    snprintf(buf, sizeof(buf), "%.17g", d);
                    out->append(buf);
                    if (!strpbrk(buf, ".eE"))
                        out->append(".0");
                }
                break;
            }
            case 0x02: {
                int32_t len;
                if (!get(&len) || len < 1 || !need((size_t)len))
                    return fail("bad string length");
                quoted(reinterpret_cast<const char*>(p), (size_t)len - 1);
                p += len;
                break;
            }
            case 0x03:
            case 0x04:
                if (!document(type == 0x04, depth + 1))
                    return false;
                break;
            case 0x05: {  // binary: no reader of this loader takes one; keep the document well-formed
                int32_t len;
                if (!get(&len) || len < 0 || !need((size_t)len + 1))
                    return fail("bad binary length");
                p += (size_t)len + 1;
                out->append("null");
                break;
            }
            case 0x08: {
                uint8_t b;
                if (!get(&b))
                    return false;
                out->append(b ? "true" : "false");
                break;
            }
            case 0x0A:
                out->append("null");
                break;
            case 0x10: {
                int32_t v;
                if (!get(&v))
                    return false;
                snprintf(buf, sizeof(buf), "%d", v);
                out->append(buf);
                break;
            }
            case 0x12: {
                int64_t v;
                if (!get(&v))
                    return false;
                snprintf(buf, sizeof(buf), "%lld", (long long)v);
                out->append(buf);
                break;
            }
            case 0x11: {
                uint64_t v;
                if (!get(&v))
                    return false;
                snprintf(buf, sizeof(buf), "%llu", (unsigned long long)v);
                out->append(buf);
                break;
            }
            default:
                return fail("unsupported element type");  // nlohmann: parse_error.114
            }
        }
        if (p != stop)
            return fail("document size does not match its content");
        out->push_back(as_array ? ']' : '}');
        return true;
    }
};

void set_error(char* error, size_t capacity, const std::string& text)
{
    if (error && capacity) {
        snprintf(error, capacity, "%s", text.c_str());
    }
}

GvColumn column(const void* data, uint32_t stride) { return GvColumn{data, stride}; }

}  // namespace

static int parse_scene_text(const char* text, size_t length, bool allow_nan, const GvScenePool* pools, uint32_t pool_count,
                     uint32_t flags, GvScene** out_scene, char* error, size_t error_capacity)
{
    if (!text || !out_scene || (pool_count && !pools)) {
        set_error(error, error_capacity, "gv_scene_parse: NULL argument");
        return GV_E_ARG;
    }
    *out_scene = nullptr;
    GvScene* sc = new GvScene();
    Loader ld;
    ld.sc = sc;
    ld.ps.p = ld.ps.begin = text;
    ld.ps.end = text + length;
    ld.ps.allow_nan = allow_nan;
    ld.add_root = (flags & GV_SCENE_ADD_ROOT_ENTITY) != 0;
    for (uint32_t k = 0; k < pool_count; k++) {
        if (!pools[k].component_type || pools[k].pool_id >= GV_MAX_POOLS || sc->pools[pools[k].pool_id].mapped ||
            strcmp(pools[k].component_type, "Transform") == 0) {
            set_error(error, error_capacity, "gv_scene_parse: bad pool mapping");
            delete sc;
            return GV_E_ARG;
        }
        sc->pools[pools[k].pool_id].mapped = true;
        sc->pools[pools[k].pool_id].type = pools[k].component_type;
        ld.pool_of_type[pools[k].component_type] = pools[k].pool_id;
    }
    if (!ld.run()) {
        set_error(error, error_capacity, ld.ps.error);
        delete sc;
        return GV_E_ARG;
    }
    *out_scene = sc;
    return GV_OK;
}


extern "C" {

int gv_scene_parse_json(const char* text, size_t length, const GvScenePool* pools, uint32_t pool_count, uint32_t flags,
                        GvScene** out_scene, char* error, size_t error_capacity)
{
    return parse_scene_text(text, length, false, pools, pool_count, flags, out_scene, error, error_capacity);
}

int gv_scene_parse_bson(const void* data, size_t length, const GvScenePool* pools, uint32_t pool_count, uint32_t flags,
                        GvScene** out_scene, char* error, size_t error_capacity)
{
    if (!data || !out_scene || (pool_count && !pools)) {
        set_error(error, error_capacity, "gv_scene_parse_bson: NULL argument");
        return GV_E_ARG;
    }
    *out_scene = nullptr;
    std::string text;
    text.reserve(length * 2);
    BsonReader reader{static_cast<const uint8_t*>(data), static_cast<const uint8_t*>(data) + length, &text, {}};
    if (!reader.document(false, 0)) {
        set_error(error, error_capacity, reader.error);
        return GV_E_ARG;
    }
    return parse_scene_text(text.data(), text.size(), true, pools, pool_count, flags, out_scene, error, error_capacity);
}

void gv_scene_destroy(GvScene* scene) { delete scene; }

int gv_scene_info(const GvScene* scene, GvSceneInfo* out)
{
    if (!scene || !out)
        return GV_E_ARG;
    *out = scene->info;
    return GV_OK;
}

int gv_scene_transform_columns(const GvScene* scene, GvTransformColumns* columns, uint32_t* occupancy,
                               const uint32_t** entity_to_transform, uint32_t* entity_capacity, const uint64_t** uids)
{
    if (!scene || !columns)
        return GV_E_ARG;
    columns->entity = column(scene->entity.data(), 4);
    columns->parent = column(scene->parent.data(), 4);
    columns->position = column(scene->position.data(), 12);
    columns->scale = column(scene->scale.data(), 12);
    columns->rotation = column(scene->rotation.data(), 16);
    columns->self_active = column(scene->self_active.data(), 1);
    columns->ancestors_active = column(scene->ancestors_active.data(), 1);
    columns->model_with_ancestors = column(scene->model_with_ancestors.data(), 1);
    if (occupancy)
        *occupancy = (uint32_t)scene->entity.size();
    if (entity_to_transform)
        *entity_to_transform = scene->entity_to_transform.data();
    if (entity_capacity)
        *entity_capacity = (uint32_t)scene->entity_to_transform.size();
    if (uids)
        *uids = scene->uid.data();
    return GV_OK;
}

int gv_scene_mesh_columns(GvScene* scene, uint32_t pool_id, GvMeshColumns* columns, uint32_t* occupancy)
{
    if (!scene || !columns || pool_id >= GV_MAX_POOLS)
        return GV_E_ARG;
    MeshColumnsOwned& p = scene->pools[pool_id];
    columns->entity = column(p.entity.data(), 4);
    columns->is_enabled = column(p.is_enabled.data(), 1);
    columns->aabb_min = column(p.aabb_min.data(), 12);
    columns->aabb_max = column(p.aabb_max.data(), 12);
    columns->is_visible = p.is_visible.data();
    columns->is_visible_stride = 1;
    if (occupancy)
        *occupancy = (uint32_t)p.entity.size();
    return GV_OK;
}

int gv_scene_bind(GvCtx* ctx, GvScene* scene)
{
    if (!ctx || !scene)
        return GV_E_ARG;
    GvTransformColumns tc;
    uint32_t n = 0, cap = 0;
    const uint32_t* e2t = nullptr;
    gv_scene_transform_columns(scene, &tc, &n, &e2t, &cap, nullptr);
    int rc = gv_transform_bind_columns(ctx, &tc, n, e2t, cap);
    if (rc != GV_OK)
        return rc;
    rc = gv_mark_dirty(ctx, GV_DIRTY_HIERARCHY, 0, 0);  // a new scene: full mirror build
    if (rc != GV_OK)
        return rc;
    for (uint32_t k = 0; k < GV_MAX_POOLS; k++) {
        if (!scene->pools[k].mapped)
            continue;
        GvMeshColumns mc;
        uint32_t count = 0;
        gv_scene_mesh_columns(scene, k, &mc, &count);
        rc = gv_pool_bind_columns(ctx, k, &mc, count);
        if (rc != GV_OK)
            return rc;
        rc = gv_mark_dirty(ctx, GV_DIRTY_MESH, k << 28, count);
        if (rc != GV_OK)
            return rc;
        // a tile's pool slots are not a contiguous range of the world's: the exchange carries the world's mesh slots
        rc = gv_pool_set_index_map(ctx, k, scene->is_tile ? scene->mesh_global[k].data() : nullptr,
                                   scene->is_tile ? (uint32_t)scene->mesh_global[k].size() : 0u);
        if (rc != GV_OK)
            return rc;
    }
    return GV_OK;
}

// One spatial tile of a scene as a scene of its own (entities shard by spatial tile; the nearest reference
// analogue is the contiguous range split of ThreadPool::addItems, source/thread-pool.cpp:173-200 — here the cut is by
// space so that a tile is culled as a unit). Same rule as garden_amd/multi.py::partition_world, which the tests compare
// it with: every ROOT transform goes to the tile its position falls in (grid cells of a cube of edge `side` centred on
// the origin), every descendant follows its root (a parent chain is never cut), a mesh follows its entity's transform;
// free slots and meshes without a transform go to tile 0; inside a tile slots keep their order, entity ids are
// renumbered from 1 (live transforms in slot order, then mesh entities without a transform in ascending old id), parents
// are remapped (a parent without a transform stays an id nothing maps to).
}  // extern "C"

// the cell a position falls in (the rule of extract_owned / multi.py::tile_of_positions)
static uint32_t cell_of(const uint32_t grid[3], double side, const float position[3])
{
    uint32_t t = 0, mul = 1;
    for (int a = 0; a < 3; a++) {
        const double cell = ((double)position[a] / side + 0.5) * (double)grid[a];
        // truncation, as numpy's astype(int64) — whose result for NaN and for values outside the int64 range is INT64_MIN
        // (x86 cvttsd2si), i.e. cell 0 after the clamp; decided in double here: the cast itself would be undefined behaviour
        long long c = (cell > -9.2e18 && cell < 9.2e18) ? (long long)cell : 0;
        c = c < 0 ? 0 : (c > (long long)grid[a] - 1 ? (long long)grid[a] - 1 : c);
        t += (uint32_t)c * mul;
        mul *= grid[a];
    }
    return t;
}

// owner[cell]: which tile each cell of the grid belongs to (identity: one tile per cell; cells dealt to ranks: see
// gv_scene_extract_rank); `tile` is the one to cut out
// spread_strays > 0: free transform slots and meshes without a transform go to tile slot % spread_strays instead of tile 0
static int extract_owned(const GvScene* scene, const uint32_t grid[3], double side, const std::vector<uint32_t>& owner, uint32_t tile,
                         uint32_t spread_strays, GvScene** out_tile)
{
    *out_tile = nullptr;
    GvScene* out = nullptr;
    try {  // (the vectors and the map below allocate: nothing may escape into C / ctypes callers)
    const uint32_t nt = (uint32_t)scene->entity.size();
    const auto& e2t = scene->entity_to_transform;
    auto slot_of = [&](uint32_t entity) -> uint32_t {
        return (entity == 0 || entity >= e2t.size()) ? kNone : e2t[entity];
    };
    // root ancestor of every transform slot (chains are short; a cycle — the loader cannot produce one — ends at 64 steps)
    std::vector<uint32_t> xf_tile(nt, 0);
    for (uint32_t s = 0; s < nt; s++) {
        if (scene->entity[s] == 0) {
            xf_tile[s] = spread_strays ? s % spread_strays : 0u;  // free slot
            continue;
        }
        uint32_t root = s;
        for (int hop = 0; hop < 64; hop++) {
            const uint32_t ps = slot_of(scene->parent[root]);
            if (ps == kNone || ps >= nt || scene->entity[ps] == 0)
                break;
            root = ps;
        }
        xf_tile[s] = owner[cell_of(grid, side, &scene->position[(size_t)root * 3])];
    }
    out = new (std::nothrow) GvScene();
    if (!out)
        return GV_E_OOM;
    out->is_tile = true;
    // transforms of the tile, ascending slot; new ids for live ones
    std::vector<uint32_t> new_id(e2t.size() + 1, 0);
    uint32_t k = 0;
    for (uint32_t s = 0; s < nt; s++)
        if (xf_tile[s] == tile) {
            out->transform_global.push_back(s);
            if (scene->entity[s] != 0 && scene->entity[s] < new_id.size())
                new_id[scene->entity[s]] = ++k;
        }
    // meshes of the tile per pool; entities without a transform in this tile get ids behind the transforms'
    std::vector<uint32_t> stray;
    for (uint32_t pid = 0; pid < GV_MAX_POOLS; pid++) {
        const MeshColumnsOwned& src = scene->pools[pid];
        out->pools[pid].type = src.type;
        out->pools[pid].mapped = src.mapped;
        for (uint32_t i = 0; i < src.entity.size(); i++) {
            const uint32_t ts = slot_of(src.entity[i]);
            const uint32_t mt = (ts != kNone && ts < nt) ? xf_tile[ts] : (spread_strays ? i % spread_strays : 0u);
            if (mt != tile)
                continue;
            out->mesh_global[pid].push_back(i);
            const uint32_t e = src.entity[i];
            if (e != 0 && (e >= new_id.size() || new_id[e] == 0))
                stray.push_back(e);
        }
    }
    std::sort(stray.begin(), stray.end());
    stray.erase(std::unique(stray.begin(), stray.end()), stray.end());
    std::unordered_map<uint32_t, uint32_t> stray_id;  // (ids beyond the entity map included)
    for (size_t q = 0; q < stray.size(); q++) {
        if (stray[q] < new_id.size())
            new_id[stray[q]] = k + 1 + (uint32_t)q;
        stray_id[stray[q]] = k + 1 + (uint32_t)q;
    }
    auto mapped_id = [&](uint32_t e) -> uint32_t {
        if (e == 0)
            return 0;
        if (e < new_id.size())
            return new_id[e];
        auto it = stray_id.find(e);
        return it == stray_id.end() ? 0u : it->second;
    };
    bool dangling = false;
    for (uint32_t s : out->transform_global)
        if (scene->parent[s] != 0 && mapped_id(scene->parent[s]) == 0)
            dangling = true;
    const uint32_t cap = k + 1 + (uint32_t)stray.size() + (dangling ? 1u : 0u);
    out->entity_to_transform.assign(cap, kNone);
    for (uint32_t local = 0; local < out->transform_global.size(); local++) {
        const uint32_t s = out->transform_global[local];
        const uint32_t id = mapped_id(scene->entity[s]);
        out->entity.push_back(id);
        uint32_t parent = mapped_id(scene->parent[s]);
        if (scene->parent[s] != 0 && parent == 0)
            parent = cap - 1;  // an entity id with no transform: the chain ends there, as in the whole scene
        out->parent.push_back(parent);
        out->uid.push_back(scene->uid[s]);
        for (int c = 0; c < 3; c++) {
            out->position.push_back(scene->position[(size_t)s * 3 + c]);
            out->scale.push_back(scene->scale[(size_t)s * 3 + c]);
        }
        for (int c = 0; c < 4; c++)
            out->rotation.push_back(scene->rotation[(size_t)s * 4 + c]);
        out->self_active.push_back(scene->self_active[s]);
        out->ancestors_active.push_back(scene->ancestors_active[s]);
        out->model_with_ancestors.push_back(scene->model_with_ancestors[s]);
        if (id != 0)
            out->entity_to_transform[id] = local;
    }
    for (uint32_t pid = 0; pid < GV_MAX_POOLS; pid++) {
        const MeshColumnsOwned& src = scene->pools[pid];
        MeshColumnsOwned& dst = out->pools[pid];
        for (uint32_t i : out->mesh_global[pid]) {
            dst.entity.push_back(mapped_id(src.entity[i]));
            dst.is_enabled.push_back(src.is_enabled[i]);
            dst.is_visible.push_back(0);
            for (int c = 0; c < 3; c++) {
                dst.aabb_min.push_back(src.aabb_min[(size_t)i * 3 + c]);
                dst.aabb_max.push_back(src.aabb_max[(size_t)i * 3 + c]);
            }
        }
        out->info.mesh_count[pid] = (uint32_t)dst.entity.size();
    }
    out->info.entity_count = cap - 1;
    out->info.transform_count = (uint32_t)out->entity.size();
    *out_tile = out;
    return GV_OK;
    } catch (...) {  // std::bad_alloc
        delete out;
        return GV_E_OOM;
    }
}

extern "C" {

int gv_scene_extract_tile(const GvScene* scene, const uint32_t grid[3], double side, uint32_t tile, GvScene** out_tile)
{
    if (!scene || !grid || !out_tile || !(side > 0.0) || grid[0] == 0 || grid[1] == 0 || grid[2] == 0 ||
        (uint64_t)grid[0] * grid[1] * grid[2] > 4096u || tile >= grid[0] * grid[1] * grid[2])
        return GV_E_ARG;
    try {
        std::vector<uint32_t> owner((size_t)grid[0] * grid[1] * grid[2]);
        for (uint32_t c = 0; c < owner.size(); c++)
            owner[c] = c;
        return extract_owned(scene, grid, side, owner, tile, 0, out_tile);
    } catch (...) {
        return GV_E_OOM;
    }
}

// owner[cell] of the dealing rule below, for every cell of the grid (linear id x + y * gx + z * gx * gy)
static void deal_cells(const uint32_t grid[3], uint32_t world_size, std::vector<uint32_t>& owner)
{
    const uint32_t cells = grid[0] * grid[1] * grid[2];
    std::vector<std::pair<uint64_t, uint32_t>> order(cells);
    for (uint32_t c = 0; c < cells; c++) {
        const uint32_t x = c % grid[0], y = (c / grid[0]) % grid[1], z = c / (grid[0] * grid[1]);
        uint64_t code = 0;
        for (uint32_t b = 0; b < 12; b++)
            code |= (uint64_t)((x >> b) & 1u) << (3 * b) | (uint64_t)((y >> b) & 1u) << (3 * b + 1) | (uint64_t)((z >> b) & 1u) << (3 * b + 2);
        order[c] = {code, c};
    }
    std::sort(order.begin(), order.end());
    owner.assign(cells, 0);
    for (uint32_t k = 0; k < cells; k++) {  // one cell per rank per round; the rotation changes from round to round
        const uint32_t turn = (uint32_t)(((uint64_t)(k / world_size) * 2654435761ull) & 0xFFFFFFFFull) >> 16;
        owner[order[k].second] = (uint32_t)(((uint64_t)k + turn) % world_size);
    }
}

// Cells in Morton (Z-curve) order of their (x, y, z) coordinates, dealt in rounds of world_size with a rotation that changes
// from round to round: cell k of that order belongs to rank (k + h(k / world_size)) % world_size. The same table as
// garden_amd/multi.py::cell_owners (tests compare them; the docstring there says why not plain k % world_size).
int gv_scene_extract_rank(const GvScene* scene, const uint32_t grid[3], double side, uint32_t rank, uint32_t world_size, GvScene** out_tile)
{
    if (!scene || !grid || !out_tile || !(side > 0.0) || grid[0] == 0 || grid[1] == 0 || grid[2] == 0 ||
        (uint64_t)grid[0] * grid[1] * grid[2] > 32768u || world_size == 0 || rank >= world_size)
        return GV_E_ARG;
    try {
        std::vector<uint32_t> owner;
        deal_cells(grid, world_size, owner);
        return extract_owned(scene, grid, side, owner, rank, world_size, out_tile);
    } catch (...) {
        return GV_E_OOM;
    }
}

int gv_cell_owner(const uint32_t grid[3], double side, uint32_t world_size, const float* positions, uint32_t stride, uint32_t count,
                  uint32_t* owners)
{
    if (!grid || !(side > 0.0) || grid[0] == 0 || grid[1] == 0 || grid[2] == 0 || (uint64_t)grid[0] * grid[1] * grid[2] > 32768u ||
        world_size == 0 || (count && (!positions || !owners || stride < 12)))
        return GV_E_ARG;
    try {
        std::vector<uint32_t> owner;
        deal_cells(grid, world_size, owner);
        for (uint32_t i = 0; i < count; i++)
            owners[i] = owner[cell_of(grid, side, reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(positions) + (size_t)i * stride))];
        return GV_OK;
    } catch (...) {
        return GV_E_OOM;
    }
}

int gv_scene_tile_maps(const GvScene* tile, uint32_t pool_id, const uint32_t** transform_global, uint32_t* transform_count,
                       const uint32_t** mesh_global, uint32_t* mesh_count)
{
    if (!tile || !tile->is_tile || pool_id >= GV_MAX_POOLS)
        return GV_E_ARG;
    if (transform_global)
        *transform_global = tile->transform_global.data();
    if (transform_count)
        *transform_count = (uint32_t)tile->transform_global.size();
    if (mesh_global)
        *mesh_global = tile->mesh_global[pool_id].data();
    if (mesh_count)
        *mesh_count = (uint32_t)tile->mesh_global[pool_id].size();
    return GV_OK;
}

}  // extern "C"
