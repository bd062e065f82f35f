// TEST: the host-only scene ingest (garden_amd/csrc/gv_scene.cpp) under AddressSanitizer + UBSan — built by
// tests/test_scene_ingest.py on the CPU tier. argv: seed files (*.json parsed as text, anything else as BSON). Every
// prefix of every seed and a few thousand random byte mutations are parsed: an error or a scene, never a fault.
#include "../../include/garden_vis.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

// gv_scene_bind's callees live in the device half of the library; never reached here
extern "C" {
int gv_transform_bind_columns(GvCtx*, const GvTransformColumns*, uint32_t, const uint32_t*, uint32_t) { return GV_E_STATE; }
int gv_pool_bind_columns(GvCtx*, uint32_t, const GvMeshColumns*, uint32_t) { return GV_E_STATE; }
int gv_mark_dirty(GvCtx*, uint32_t, uint32_t, uint32_t) { return GV_E_STATE; }
int gv_pool_set_index_map(GvCtx*, uint32_t, const uint32_t*, uint32_t) { return GV_E_STATE; }
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 32);
}

int main(int argc, char** argv)
{
    const GvScenePool pools[2] = {{"Model", 0}, {"Sprite", 3}};
    long parsed = 0, rejected = 0;
    for (int a = 1; a < argc; a++) {
        std::ifstream in(argv[a], std::ios::binary);
        const std::string seed((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        const size_t n = strlen(argv[a]);
        const bool text = n > 5 && strcmp(argv[a] + n - 5, ".json") == 0;
        auto run = [&](const std::string& blob) {
            GvScene* sc = nullptr;
            char err[128];
            // exact-size heap copy: an over-read by one byte is an ASan report
            std::vector<char> exact(blob.begin(), blob.end());
            const char* data = exact.empty() ? "" : exact.data();
            const int rc = text ? gv_scene_parse_json(data, exact.size(), pools, 2, rnd() & 1u, &sc, err, sizeof(err))
                                : gv_scene_parse_bson(data, exact.size(), pools, 2, rnd() & 1u, &sc, err, sizeof(err));
            if (rc == GV_OK) {
                GvSceneInfo info;
                gv_scene_info(sc, &info);
                // every scene that parses is also cut into spatial tiles (gv_scene_extract_tile): whatever the mutation did
                // to parents, ids and positions (NaN, inf, huge), a tile is a scene or an error, never a fault
                const uint32_t grid[3] = {2, 1 + rnd() % 2, 1};
                uint32_t transforms = 0;
                for (uint32_t t = 0; t < grid[0] * grid[1] * grid[2]; t++) {
                    GvScene* tile = nullptr;
                    if (gv_scene_extract_tile(sc, grid, 1000.0, t, &tile) == GV_OK) {
                        GvSceneInfo ti;
                        gv_scene_info(tile, &ti);
                        transforms += ti.transform_count;
                        gv_scene_destroy(tile);
                    }
                }
                if (transforms != info.transform_count) {
                    printf("{\"ok\": false, \"why\": \"tiles hold %u of %u transforms\"}\n", transforms, info.transform_count);
                    exit(1);
                }
                gv_scene_destroy(sc);
                parsed++;
            } else {
                rejected++;
            }
        };
        run(seed);
        const size_t step = seed.size() > 4000 ? seed.size() / 2000 : 1;
        for (size_t cut = 0; cut < seed.size(); cut += step)
            run(seed.substr(0, cut));
        for (int k = 0; k < 4000; k++) {
            std::string blob = seed;
            const int edits = 1 + (int)(rnd() % 3);
            for (int e = 0; e < edits && !blob.empty(); e++)
                blob[rnd() % blob.size()] = (char)rnd();
            run(blob);
        }
    }
    printf("{\"ok\": true, \"parsed\": %ld, \"rejected\": %ld}\n", parsed, rejected);
    return parsed > 0 && rejected > 0 ? 0 : 1;
}
