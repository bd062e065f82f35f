// exchange_ranks.cpp — TEST / worked example: the multi-GPU exchange step through the C-ABI alone (no Python, no
// torch.distributed), one process per GPU as SURVEY.md §8e asks. The parent forks R ranks BEFORE anything touches the
// GPU; rank 0 creates the RCCL unique id and the parent relays its 128 bytes to the other ranks over pipes (an engine
// would use its own IPC). Every rank culls its own tile and calls gv_exchange_shards; each checks that its row of the
// gathered buffer is its own list and that every row's header is a plausible count.
//   exchange_ranks --ranks R [--entities N]        (R > 1 needs R GPUs: RCCL refuses two ranks on one device)
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/garden_vis.h"

namespace {

struct alignas(16) Transform {  // TransformComponent's release layout (transform.hpp:31-61)
    uint32_t entity, parent;
    uint64_t uid;
    float pos[4], scale[4], rot[4];
    void* childs;
    uint8_t selfActive, ancestorsActive, modelWithAncestors;
};
struct alignas(16) Mesh {  // MeshRenderComponent (render/mesh.hpp:45-55)
    uint32_t entity, r0, r1;
    uint16_t r2;
    uint8_t isEnabled, isVisible;
    float mn[4], mx[4];
};
static_assert(sizeof(Transform) == 80 && sizeof(Mesh) == 48, "component layouts");

bool write_all(int fd, const void* p, size_t n) { return write(fd, p, n) == (ssize_t)n; }
bool read_all(int fd, void* p, size_t n)
{
    size_t got = 0;
    while (got < n) {
        const ssize_t r = read(fd, (char*)p + got, n - got);
        if (r <= 0)
            return false;
        got += (size_t)r;
    }
    return true;
}

int run_rank(int rank, int ranks, uint32_t n, int id_in, int id_out)
{
    auto die = [&](const char* what, GvCtx* ctx) {
        fprintf(stderr, "rank %d: %s: %s\n", rank, what, gv_last_error(ctx));
        return 1;
    };
    int devices = 0;
    if (hipGetDeviceCount(&devices) != hipSuccess || devices == 0) {
        fprintf(stderr, "rank %d: no device\n", rank);
        return 1;
    }
    GvConfig config{};
    config.struct_size = sizeof(config);
    config.device = rank % devices;
    GvCtx* ctx = nullptr;
    if (gv_create(&config, &ctx) != GV_OK)
        return die("gv_create", nullptr);

    // this rank's tile: a slab of the world along x, camera at the origin looking down +z
    std::vector<Transform> tr(n);
    std::vector<Mesh> me(n);
    std::vector<uint32_t> e2t(n + 1, GV_NONE);
    uint32_t seed = 12345u + 977u * (uint32_t)rank;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)(seed >> 8) * (1.0f / 16777216.0f); };
    const float side = 100.0f * std::cbrt((float)n * (float)ranks), slab = side / (float)ranks;
    for (uint32_t i = 0; i < n; i++) {
        memset(&tr[i], 0, sizeof(Transform));
        memset(&me[i], 0, sizeof(Mesh));
        tr[i].entity = me[i].entity = i + 1;
        e2t[i + 1] = i;
        tr[i].pos[0] = -0.5f * side + slab * ((float)rank + rnd());
        tr[i].pos[1] = side * (rnd() - 0.5f);
        tr[i].pos[2] = side * (rnd() - 0.5f);
        tr[i].scale[0] = tr[i].scale[1] = tr[i].scale[2] = 1.0f;
        tr[i].rot[3] = 1.0f;
        tr[i].selfActive = tr[i].ancestorsActive = tr[i].modelWithAncestors = 1;
        me[i].isEnabled = 1;
        for (int k = 0; k < 3; k++) {
            me[i].mn[k] = -0.5f;
            me[i].mx[k] = 0.5f;
        }
    }
    const GvTransformLayout tl = {0, 4, 16, 32, 48, 72, 73, 74};
    const GvMeshLayout ml = {0, 14, 15, 16, 32};
    if (gv_transform_bind(ctx, tr.data(), sizeof(Transform), n, &tl, e2t.data(), n + 1) != GV_OK ||
        gv_pool_bind(ctx, 0, me.data(), sizeof(Mesh), n, &ml) != GV_OK)
        return die("bind", ctx);
    GvView view{};
    view.view_proj[0] = 9.0f / 16.0f; view.view_proj[5] = -1.0f; view.view_proj[11] = 1.0f; view.view_proj[14] = 0.01f;
    view.shadow_pass = -1;
    view.emit_records = 1;

    // unique id: rank 0 makes it and hands it to the parent; everybody else reads theirs from the parent
    unsigned char id[GV_EXCHANGE_ID_BYTES];
    if (rank == 0) {
        if (gv_exchange_unique_id(id) != GV_OK || !write_all(id_out, id, sizeof(id)))
            return die("gv_exchange_unique_id", ctx);
    } else if (!read_all(id_in, id, sizeof(id))) {
        fprintf(stderr, "rank %d: no unique id from the parent\n", rank);
        return 1;
    }
    if (gv_exchange_init(ctx, id, rank, ranks) != GV_OK)
        return die("gv_exchange_init", ctx);

    const uint32_t capacity = n;  // worst case: everything visible
    uint32_t* gathered = nullptr;
    if (hipMalloc((void**)&gathered, (size_t)ranks * (capacity + 1) * 4) != hipSuccess)
        return 1;
    for (int frame = 0; frame < 3; frame++)
        if (gv_cull(ctx, 0, &view, 1) != GV_OK || gv_exchange_shards(ctx, 0, capacity, (uint32_t)rank * n, gathered) != GV_OK)
            return die("cull / exchange", ctx);
    if (gv_wait(ctx) != GV_OK)
        return die("gv_wait", ctx);
    GvResult res{};
    if (gv_results_fetch(ctx, 0, 0, &res) != GV_OK)
        return die("gv_results_fetch", ctx);
    std::vector<uint32_t> host((size_t)ranks * (capacity + 1));
    if (hipMemcpy(host.data(), gathered, host.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
        return 1;
    bool ok = host[(size_t)rank * (capacity + 1)] == res.draw_count;
    for (uint32_t k = 0; k < res.draw_count && ok; k++)
        ok = host[(size_t)rank * (capacity + 1) + 1 + k] == res.visible_idx[k] + (uint32_t)rank * n;
    uint64_t total = 0;
    for (int r = 0; r < ranks && ok; r++) {
        const uint32_t count = host[(size_t)r * (capacity + 1)];
        ok = count <= n;
        total += count;
        for (uint32_t k = 0; k < count && ok; k++) {
            const uint32_t g = host[(size_t)r * (capacity + 1) + 1 + k];
            ok = g >= (uint32_t)r * n && g < (uint32_t)(r + 1) * n;  // every index lies in its owner's tile range
        }
    }
    printf("{\"rank\": %d, \"ranks\": %d, \"visible\": %u, \"gathered\": %llu, \"ok\": %s}\n", rank, ranks, res.draw_count,
           (unsigned long long)total, ok ? "true" : "false");
    fflush(stdout);  // the rank leaves through _exit
    (void)hipFree(gathered);
    gv_exchange_shutdown(ctx);
    gv_destroy(ctx);
    return ok ? 0 : 1;
}

}  // namespace

int main(int argc, char** argv)
{
    int ranks = 1;
    uint32_t n = 100000;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--ranks") && i + 1 < argc) ranks = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--entities") && i + 1 < argc) n = (uint32_t)atoi(argv[++i]);
    }
    if (ranks < 1 || ranks > 64)
        return 2;
    // pipes: child r -> parent (only rank 0 uses it), parent -> child r
    std::vector<int> to_parent(2 * ranks), to_child(2 * ranks);
    for (int r = 0; r < ranks; r++)
        if (pipe(&to_parent[2 * r]) != 0 || pipe(&to_child[2 * r]) != 0)
            return 2;
    std::vector<pid_t> pids;
    for (int r = 0; r < ranks; r++) {
        const pid_t pid = fork();  // before any HIP call in this process
        if (pid == 0)
            _exit(run_rank(r, ranks, n, to_child[2 * r], to_parent[2 * r + 1]));
        pids.push_back(pid);
    }
    unsigned char id[GV_EXCHANGE_ID_BYTES];
    bool relayed = read_all(to_parent[0], id, sizeof(id));
    for (int r = 1; r < ranks && relayed; r++)
        relayed = write_all(to_child[2 * r + 1], id, sizeof(id));
    int failed = relayed ? 0 : 1;
    for (pid_t pid : pids) {
        int status = 0;
        waitpid(pid, &status, 0);
        if (!WIFEXITED(status) || WEXITSTATUS(status) != 0)
            failed++;
    }
    return failed ? 1 : 0;
}
