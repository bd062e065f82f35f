// exchange_ranks.cpp — TEST / worked example: the multi-GPU exchange step through the C-ABI alone (no Python, no
// torch.distributed), one process per GPU as SURVEY.md §8e asks. The parent forks R ranks BEFORE anything touches the
// GPU; rank 0 creates the unique id and the parent relays its 128 bytes to the other ranks over pipes (an engine would
// use its own IPC). Every rank culls its own slab of the world each frame against a camera that turns — and, half way,
// cuts to the opposite direction — calls gv_exchange_visible (rows owned and sized by the library) and acquires the frame
// (gv_exchange_acquire: now, or — every third frame — only after the NEXT frame has been sent, the way a pipelined engine
// does); the last frames go through gv_exchange_shards with per-rank capacities. After every frame each rank summarises
// (count, delivered entries, sum, xor) every row it received and its own list; the parent checks that all ranks received
// the same rows and that row r is rank r's WHOLE list in EVERY frame of gv_exchange_visible — the camera cut included
// (the caller-sized frames may be short: that form is the caller's own sizing). The transport patterns take turns.
//   exchange_ranks --ranks R|auto [--entities N] [--frames F] [--mode all|allgather|p2p|broadcast] [--stall-rank R] [--random-camera SEED]
// --random-camera SEED: every frame looks somewhere else through another lens (field of view 20 .. 160 degrees): the lists jump by an
// order of magnitude from frame to frame in both directions — predictions are short or far too generous most of the time.
// --abandon (with --stall-rank): the other ranks send the frame the stalled peer never joins and shut down WITHOUT acquiring it:
// gv_exchange_shutdown comes back with GV_E_TIMEOUT inside the limit instead of synchronising a stream that never drains
// (--abandon-by-destroy: they call gv_destroy straight away, which must come back inside the limit too).
// --stall-rank R: rank R stops calling half way (a stalled peer): every other rank must come back with a status code — not hang:
// GV_E_TIMEOUT from a bounded wait (2 s here) on the rank that notices first (it aborts the communicator), GV_E_TIMEOUT or
// GV_E_RCCL (ncclCommGetAsyncError: a rank has left) on the others — and shut its communicator down.
// R > 1 needs R GPUs with RCCL (it refuses two ranks on one device); with GV_RCCL_LIBRARY=<tests/cpp/build/librccl_stub.so>
// the ranks share the GPUs there are (ranks are dealt round-robin over them) and the rows travel through shared memory.
// "auto": min(GPUs, 8) ranks.
// --batched: every frame culls TWO views (the camera, and the camera turned by a quarter) and sends both lists in ONE exchange
// (gv_exchange_views: row = [2 + total, c_0, c_1, list 0, list 1]) — the per-rank form of what the drop-in's one-process mode does
// with gv_exchange_views_all; same checks: every rank holds every owner's whole row in every frame.
// --check-oracle: on frame 0 and on the frame of the camera cut every rank ALSO runs the CPU oracle (oracle/gv_oracle.c:
// gvo_prepare_meshes_range, mesh.cpp:111-184) over its own share and compares its own list with the oracle's as a set; the parent
// already proves that every rank holds every owner's whole list, so union over the ranks == the oracle's visible set of the
// world follows. (A test driver may link the oracle; the product library never does.)
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <algorithm>
#include <atomic>
#include <thread>

#include <hip/hip_runtime.h>

#include "../../include/garden_vis.h"
#include "../../oracle/gv_oracle.h"

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

constexpr int kMaxRanks = 8, kMaxFrames = 32;
struct RowSummary {
    uint64_t count, delivered, sum, x;
};
struct FrameSummary {
    RowSummary own, rows[kMaxRanks];
    uint32_t short_rows, tail_words, mode, valid, sized_by_library, timed_out;
    uint32_t oracle_checked, oracle_mismatch;  // --check-oracle: this rank's own list of the frame against the CPU oracle
    uint32_t travelled[kMaxRanks];
};
struct Shared {
    FrameSummary frames[kMaxRanks][kMaxFrames];
};

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

RowSummary summarise(const uint32_t* entries, uint64_t count, uint64_t room, uint32_t add)
{
    RowSummary s{count, count < room ? count : room, 0, 0};
    for (uint64_t k = 0; k < s.delivered; k++) {
        const uint32_t v = entries[k] + add;
        s.sum += v;
        s.x ^= (uint64_t)v * 0x9E3779B97F4A7C15ull;
    }
    return s;
}

// camera at the origin, 90 degree field of view, 16:9, infinite reversed-Z; yaw about +y
void make_view(float yaw, GvView* view, float zoom = 1.0f)
{
    memset(view, 0, sizeof(*view));
    const float c = std::cos(yaw), s = std::sin(yaw);
    // projection (columns): x' = zoom (9/16) x, y' = -zoom y, z' = near, w' = z ; view = rotation by -yaw about y
    const float P[16] = {zoom * 9.0f / 16.0f, 0, 0, 0, 0, -zoom, 0, 0, 0, 0, 0, 1.0f, 0, 0, 0.01f, 0};
    const float V[16] = {c, 0, s, 0, 0, 1, 0, 0, -s, 0, c, 0, 0, 0, 0, 1};
    for (int col = 0; col < 4; col++)
        for (int row = 0; row < 4; row++) {
            float acc = 0.0f;
            for (int k = 0; k < 4; k++)
                acc += P[k * 4 + row] * V[col * 4 + k];
            view->view_proj[col * 4 + row] = acc;
        }
    view->shadow_pass = -1;
    view->emit_records = 1;
}

// The CPU oracle's visible slots of one rank's share for `view`, ascending: gvo_prepare_meshes_range over contiguous pieces of the pool
// on `threads` threads (ThreadPool::addItems, thread-pool.cpp:180-194), each piece with output arrays of its own size.
std::vector<uint32_t> oracle_visible(std::vector<Transform>& tr, std::vector<Mesh>& me, const std::vector<uint32_t>& e2t, const GvView& view, uint32_t threads)
{
    GvoMeshPool mp{};
    mp.base = reinterpret_cast<uint8_t*>(me.data());
    mp.stride = sizeof(Mesh);
    mp.occupancy = (uint32_t)me.size();
    mp.off_entity = 0; mp.off_is_enabled = 14; mp.off_is_visible = 15; mp.off_aabb_min = 16; mp.off_aabb_max = 32;
    GvoTransformPool tp{};
    tp.base = reinterpret_cast<const uint8_t*>(tr.data());
    tp.stride = sizeof(Transform);
    tp.occupancy = (uint32_t)tr.size();
    tp.off_entity = 0; tp.off_parent = 4; tp.off_position = 16; tp.off_scale = 32; tp.off_rotation = 48;
    tp.off_self_active = 72; tp.off_ancestors_active = 73; tp.off_model_with_ancestors = 74;
    tp.entity_to_transform = e2t.data();
    tp.entity_capacity = (uint32_t)e2t.size();
    GvoView ov{};
    memcpy(ov.view_proj, view.view_proj, sizeof(ov.view_proj));
    memcpy(ov.camera_position, view.camera_position, sizeof(ov.camera_position));
    memcpy(ov.camera_offset, view.camera_offset, sizeof(ov.camera_offset));
    ov.shadow_pass = view.shadow_pass;
    GvoFrustum frustum;
    gvo_frustum_from_view_proj(ov.view_proj, &frustum);
    constexpr uint32_t kPiece = 1u << 18;
    const uint32_t pieces = (mp.occupancy + kPiece - 1) / kPiece;
    std::vector<std::vector<uint32_t>> found(pieces);
    std::vector<std::thread> workers;
    std::atomic<uint32_t> next{0};
    for (uint32_t t = 0; t < std::max(1u, threads); t++)
        workers.emplace_back([&] {
            std::vector<uint32_t> idx(kPiece);
            std::vector<float> model((size_t)kPiece * 12), dist(kPiece);
            for (uint32_t piece = next++; piece < pieces; piece = next++) {
                GvoCullOut out{idx.data(), model.data(), dist.data(), 0, 0};
                gvo_prepare_meshes_range(&mp, &tp, &ov, &frustum, nullptr, piece * kPiece, std::min(mp.occupancy, (piece + 1) * kPiece), &out);
                found[piece].assign(idx.begin(), idx.begin() + out.draw_count);
            }
        });
    for (auto& w : workers)
        w.join();
    std::vector<uint32_t> all;
    for (auto& f : found)
        all.insert(all.end(), f.begin(), f.end());
    std::sort(all.begin(), all.end());
    return all;
}

// rank `rank`'s share of the world: a slab along x, unit cubes at uniform positions
void build_slab(int rank, int ranks, uint32_t n, std::vector<Transform>& tr, std::vector<Mesh>& me, std::vector<uint32_t>& e2t)
{
    tr.assign(n, Transform{});
    me.assign(n, Mesh{});
    e2t.assign((size_t)n + 1, GV_NONE);
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
}

// The frame's camera: the same on every rank — a hash of the seed and the frame (a new lens and direction every frame), or a camera
// that turns a little every frame and cuts to the opposite direction half way.
void frame_views(uint32_t camera_seed, int frame, int frames, bool batched, GvView both[2])
{
    if (camera_seed) {
        uint32_t h = camera_seed * 2654435761u + (uint32_t)frame * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        const float yaw = 6.2831853f * (float)(h & 0xFFFFu) / 65536.0f;
        const float zoom = std::exp(3.47f * ((float)((h >> 16) & 0xFFFFu) / 65536.0f) - 1.735f);  // 0.18 .. 5.7: 160 .. 20 degrees
        make_view(yaw, &both[0], zoom);
    } else {
        make_view(0.05f * (float)frame + (frame >= frames / 2 ? 3.14159265f : 0.0f), &both[0]);
    }
    both[1] = both[0];
    if (batched) {  // the second list: the same lens turned by a quarter
        if (camera_seed)
            make_view(1.5707963f + 0.37f * (float)frame, &both[1]);
        else
            make_view(0.05f * (float)frame + (frame >= frames / 2 ? 3.14159265f : 0.0f) + 1.5707963f, &both[1]);
    }
}

// --peers: ONE process, ONE thread, R contexts and NO communicator (gv_exchange_init_peers: every rank's scatter kernel stores its
// lists into its row of every rank's rows) — the same frames as the per-process ranks below, asynchronously: every third frame is
// acquired only after the next one has been sent. Everything lives in this process, so every row of every rank is compared WORD
// FOR WORD with its owner's own fetched list(s); --check-oracle: frame 0 and the cut against the CPU oracle as well.
int run_peers(int ranks, uint32_t n, int frames, uint32_t camera_seed, bool check_oracle, bool batched)
{
    int devices = 0;
    if (hipGetDeviceCount(&devices) != hipSuccess || devices == 0) {
        fprintf(stderr, "no device\n");
        return 1;
    }
    struct Rank {
        GvCtx* ctx = nullptr;
        std::vector<Transform> tr;
        std::vector<Mesh> me;
        std::vector<uint32_t> e2t;
        std::vector<std::vector<uint32_t>> own;  // [frame & 3]: the row this rank's lists make, header included
    };
    std::vector<Rank> rs(ranks);
    std::vector<GvCtx*> ctxs;
    auto die = [&](const char* what, int r) {
        fprintf(stderr, "peers, rank %d: %s: %s\n", r, what, gv_last_error(rs[r].ctx));
        return 1;
    };
    const GvTransformLayout tl = {0, 4, 16, 32, 48, 72, 73, 74};
    const GvMeshLayout ml = {0, 14, 15, 16, 32};
    for (int r = 0; r < ranks; r++) {
        GvConfig config{};
        config.struct_size = sizeof(config);
        config.device = r % devices;
        if (gv_create(&config, &rs[r].ctx) != GV_OK)
            return die("gv_create", r);
        build_slab(r, ranks, n, rs[r].tr, rs[r].me, rs[r].e2t);
        if (gv_transform_bind(rs[r].ctx, rs[r].tr.data(), sizeof(Transform), n, &tl, rs[r].e2t.data(), n + 1) != GV_OK ||
            gv_pool_bind(rs[r].ctx, 0, rs[r].me.data(), sizeof(Mesh), n, &ml) != GV_OK)
            return die("bind", r);
        rs[r].own.resize(4);
        ctxs.push_back(rs[r].ctx);
    }
    if (const int rc = gv_exchange_init_peers(ctxs.data(), ranks)) {
        if (rc != GV_E_STATE)
            return die("gv_exchange_init_peers", 0);
        // the documented answer of a node whose devices cannot reach each other's memory: nothing to drive here (gv_exchange_init_all is
        // the path for such a node)
        printf("{\"ranks\": %d, \"frames\": 0, \"entities_per_rank\": %u, \"ok\": true, \"failed_ranks\": 0, \"mismatches\": 0, \"frames_with_a_second_exchange\": 0, "
               "\"short_rows_completed\": 0, \"tail_words\": 0, \"timed_out_ranks\": 0, \"gathered_last_frame\": 0, \"words_on_links_over_list_words\": 1.0, "
               "\"oracle_checked_frames\": 0, \"transport\": \"peer stores unavailable: %s\"}\n", ranks, n, gv_last_error(ctxs[0]));
        for (GvCtx* c : ctxs)
            gv_destroy(c);
        return 0;
    }
    int mismatches = 0, oracle_checked_frames = 0, acquired = 0;
    uint64_t gathered_last = 0, list_words = 0, link_words = 0;
    std::vector<GvExchangeFrame> sent(ranks), got(ranks);
    std::vector<uint32_t> host;
    auto acquire = [&](int frame) -> int {
        if (gv_exchange_acquire_all(ctxs.data(), ranks, (uint64_t)frame, got.data()) != GV_OK)
            return die("gv_exchange_acquire_all", 0);
        gathered_last = 0;
        for (int r = 0; r < ranks; r++) {
            const GvExchangeFrame& f = got[r];
            if (!f.complete || !f.gathered_device || !f.ready_event || f.frame != (uint64_t)frame || f.mode != GV_EXCHANGE_PEER || f.cut_ranks || f.row_words % 4u ||
                f.items != (batched ? 2u : 0u)) {
                fprintf(stderr, "peers, rank %d frame %d: fields of an acquired frame\n", r, frame);
                return 1;
            }
            if (hipSetDevice(r % devices) != hipSuccess || hipStreamSynchronize((hipStream_t)gv_stream(ctxs[r])) != hipSuccess)
                return 1;
            for (int q = 0; q < ranks; q++) {
                const std::vector<uint32_t>& want = rs[q].own[frame & 3];
                // (rows are as wide as the pools: only what the frame says a row holds is read back, and one word more)
                const size_t words = std::min<size_t>((size_t)f.counts[q] + 2u, f.row_words);
                host.resize(words);
                if (hipMemcpy(host.data(), (const uint32_t*)f.gathered_device + (size_t)q * f.row_words, words * 4, hipMemcpyDeviceToHost) != hipSuccess)
                    return 1;
                const uint32_t* row = host.data();
                if (f.counts[q] != want[0] || f.travelled_words[q] != 1u + want[0] || f.tail_words[q] || want.size() > f.row_words ||
                    want.size() > words || memcmp(row, want.data(), want.size() * 4) != 0) {
                    fprintf(stderr, "peers, frame %d: rank %d holds row %d with header %u (the frame says %u); rank %d's own list has %u words%s\n", frame, r, q, row[0],
                            f.counts[q], q, want[0], row[0] == want[0] ? ": contents differ" : "");
                    mismatches++;
                }
                if (r == 0) {
                    gathered_last += want[0] - (batched ? 2u : 0u);
                    list_words += want.size();
                    link_words += f.travelled_words[q];
                }
            }
        }
        acquired++;
        return 0;
    };
    int late = -1;
    for (int frame = 0; frame < frames; frame++) {
        GvView both[2];
        frame_views(camera_seed, frame, frames, batched, both);
        for (int r = 0; r < ranks; r++) {
            if (gv_cull(ctxs[r], 0, both, batched ? 2 : 1) != GV_OK)
                return die("gv_cull", r);
        }
        const bool with_oracle = check_oracle && (frame == 0 || frame == frames / 2);
        for (int r = 0; r < ranks; r++) {  // the row this rank's lists make, from its own fetch
            const uint32_t base = (uint32_t)r * n;
            std::vector<uint32_t>& row = rs[r].own[frame & 3];
            row.clear();
            GvResult res{};
            if (gv_results_fetch(ctxs[r], 0, 0, &res) != GV_OK)
                return die("gv_results_fetch", r);
            if (with_oracle) {
                std::vector<uint32_t> mine(res.visible_idx, res.visible_idx + res.draw_count);
                std::sort(mine.begin(), mine.end());
                if (mine != oracle_visible(rs[r].tr, rs[r].me, rs[r].e2t, both[0], std::max(2u, std::thread::hardware_concurrency()))) {
                    fprintf(stderr, "peers, rank %d frame %d: the rank's list is not the oracle's visible set of its share\n", r, frame);
                    mismatches++;
                }
            }
            if (!batched) {
                row.push_back(res.draw_count);
                for (uint32_t k = 0; k < res.draw_count; k++)
                    row.push_back(res.visible_idx[k] + base);
            } else {
                row.assign({0u, res.draw_count, 0u});
                for (uint32_t k = 0; k < res.draw_count; k++)
                    row.push_back(res.visible_idx[k] + base);
                GvResult second{};
                if (gv_pool_results_fetch(ctxs[r], 0, 1, 0, &second) != GV_OK)
                    return die("gv_results_fetch", r);
                row[2] = second.draw_count;
                for (uint32_t k = 0; k < second.draw_count; k++)
                    row.push_back(second.visible_idx[k] + base);
                row[0] = (uint32_t)row.size() - 1u;
            }
        }
        oracle_checked_frames += with_oracle ? 1 : 0;
        std::vector<uint32_t> views(ranks, 0u), bases(ranks);
        for (int r = 0; r < ranks; r++)
            bases[r] = (uint32_t)r * n;
        int rc;
        if (batched) {
            // (one item list for all ranks: the per-rank base travels through the pool's index map instead — identity + r * n)
            for (int r = 0; r < ranks && frame == 0; r++) {
                std::vector<uint32_t> map(n);
                for (uint32_t i = 0; i < n; i++)
                    map[i] = i + bases[r];
                if (gv_pool_set_index_map(ctxs[r], 0, map.data(), n) != GV_OK)
                    return die("gv_pool_set_index_map", r);
            }
            const GvExchangeItem items[2] = {{0u, 0u, 0u}, {0u, 1u, 0u}};
            rc = gv_exchange_views_all(ctxs.data(), ranks, items, 2, 0, sent.data());
        } else {
            rc = gv_exchange_visible_all(ctxs.data(), ranks, views.data(), bases.data(), 0, sent.data());
        }
        if (rc != GV_OK)
            return die("gv_exchange_*_all", 0);
        if (late >= 0) {  // the previous frame, acquired a frame late
            if (acquire(late))
                return 1;
            late = -1;
        }
        if (frame % 3 == 2 && frame + 1 < frames)
            late = frame;
        else if (acquire(frame))
            return 1;
    }
    // one member is destroyed without a shutdown: the group is drained and dissolved, the others answer GV_E_STATE
    gv_destroy(ctxs[ranks - 1]);
    if (ranks > 1) {
        std::vector<uint32_t> views(ranks, 0u);
        if (gv_exchange_visible_all(ctxs.data(), ranks - 1, views.data(), nullptr, 0, sent.data()) != GV_E_STATE) {
            fprintf(stderr, "peers: a group that lost a member still exchanges\n");
            mismatches++;
        }
    }
    for (int r = 0; r + 1 < ranks; r++)
        gv_destroy(ctxs[r]);
    if (check_oracle && oracle_checked_frames < 2)
        mismatches++;
    if (acquired != frames)
        mismatches++;
    const bool ok = mismatches == 0;
    printf("{\"ranks\": %d, \"frames\": %d, \"entities_per_rank\": %u, \"ok\": %s, \"failed_ranks\": 0, \"mismatches\": %d, \"frames_with_a_second_exchange\": 0, "
           "\"short_rows_completed\": 0, \"tail_words\": 0, \"timed_out_ranks\": 0, \"gathered_last_frame\": %llu, \"words_on_links_over_list_words\": %.3f, "
           "\"oracle_checked_frames\": %d, \"transport\": \"peer stores, one process\"}\n",
           ranks, frames, n, ok ? "true" : "false", mismatches, (unsigned long long)gathered_last, list_words ? (double)link_words / (double)list_words : 0.0,
           oracle_checked_frames);
    return ok ? 0 : 1;
}

int run_rank(int rank, int ranks, uint32_t n, int frames, int mode_arg, int stall_rank, bool abandon, bool abandon_by_destroy, uint32_t camera_seed, bool check_oracle,
             bool batched, int id_in, int id_out, Shared* shared)
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

    // this rank's share of the world: a slab along x
    std::vector<Transform> tr;
    std::vector<Mesh> me;
    std::vector<uint32_t> e2t;
    build_slab(rank, ranks, n, tr, me, e2t);
    const GvTransformLayout tl = {0, 4, 16, 32, 48, 72, 73, 74};
    const GvMeshLayout ml = {0, 14, 15, 16, 32};
    if (gv_transform_bind(ctx, tr.data(), sizeof(Transform), n, &tl, e2t.data(), n + 1) != GV_OK ||
        gv_pool_bind(ctx, 0, me.data(), sizeof(Mesh), n, &ml) != GV_OK)
        return die("bind", ctx);

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

    hipStream_t stream = (hipStream_t)gv_stream(ctx);
    const uint32_t base = (uint32_t)rank * n;
    std::vector<uint32_t> host;
    uint32_t last_counts[kMaxRanks] = {};
    uint32_t* own_rows = nullptr;  // caller-owned rows of the gv_exchange_shards frames
    const uint32_t shards_capacity = n;
    if (hipMalloc((void**)&own_rows, (size_t)ranks * (shards_capacity + 1) * 4) != hipSuccess)
        return 1;
    if (stall_rank >= 0 && gv_exchange_set_timeout(ctx, 2000) != GV_OK)
        return die("gv_exchange_set_timeout", ctx);
    const int sized_frames = stall_rank >= 0 || batched ? frames : frames - 2;  // the last two frames: gv_exchange_shards with per-rank capacities
    // this rank's own list of a frame, summarised when the frame is culled (a frame acquired late is compared with it then)
    auto summarise_own = [&](FrameSummary& fs) {
        GvResult res{};
        if (gv_results_fetch(ctx, 0, 0, &res) != GV_OK)
            return false;
        if (!batched) {
            fs.own = summarise(res.visible_idx, res.draw_count, res.draw_count, base);
            return true;
        }
        // the row this rank's two lists make: [c_0, c_1, list 0 + base, list 1 + base] behind the header
        std::vector<uint32_t> row{res.draw_count, 0u};
        for (uint32_t k = 0; k < res.draw_count; k++)
            row.push_back(res.visible_idx[k] + base);
        GvResult second{};
        if (gv_results_fetch(ctx, 1, 0, &second) != GV_OK)
            return false;
        row[1] = second.draw_count;
        for (uint32_t k = 0; k < second.draw_count; k++)
            row.push_back(second.visible_idx[k] + base);
        fs.own = summarise(row.data(), row.size(), row.size(), 0);
        return true;
    };
    // every row of an acquired frame, read back: whole lists (the library's frames), or up to the caller's capacity
    auto summarise_rows = [&](FrameSummary& fs, const uint32_t* rows, size_t row_words, const uint32_t* room, int frame) {
        if (hipStreamSynchronize(stream) != hipSuccess)
            return false;
        host.resize((size_t)ranks * row_words);
        if (hipMemcpy(host.data(), rows, host.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
            return false;
        for (int r = 0; r < ranks; r++) {
            const uint32_t* row = host.data() + (size_t)r * row_words;
            fs.rows[r] = summarise(row + 1, row[0], room ? room[r] : row[0], 0);
            last_counts[r] = row[0];
            if ((size_t)fs.rows[r].delivered + 1 > row_words) {
                fprintf(stderr, "rank %d frame %d: row %d holds %llu entries in %zu words\n", rank, frame, r, (unsigned long long)fs.rows[r].delivered, row_words);
                return false;
            }
            if (batched && (row[0] < 2 || row[0] != 2u + row[1] + row[2])) {
                fprintf(stderr, "rank %d frame %d: row %d's header %u is not its table (%u, %u) plus 2\n", rank, frame, r, row[0], row[1], row[2]);
                return false;
            }
            for (uint64_t k = batched ? 2 : 0; k < fs.rows[r].delivered; k++)
                if (row[1 + k] < (uint32_t)r * n || row[1 + k] >= (uint32_t)(r + 1) * n) {
                    fprintf(stderr, "rank %d frame %d: row %d entry %llu = %u outside its owner's range\n", rank, frame, r,
                            (unsigned long long)k, row[1 + k]);
                    return false;
                }
        }
        return true;
    };
    auto acquire = [&](int frame) {
        FrameSummary& fs = shared->frames[rank][frame];
        GvExchangeFrame got;
        int rc = gv_exchange_acquire(ctx, (uint64_t)frame, &got);
        if (rc == GV_E_TIMEOUT || (rc == GV_E_RCCL && stall_rank >= 0)) {
            fs.timed_out = rc == GV_E_TIMEOUT ? 1 : 2;
            return (int)GV_E_TIMEOUT;
        }
        if (rc != GV_OK)
            return rc;
        if (!got.complete || !got.gathered_device || !got.ready_event || got.frame != (uint64_t)frame || got.row_words % 4u ||
            got.items != (batched ? 2u : 0u) || (batched && !got.item_counts)) {
            fprintf(stderr, "rank %d frame %d: fields of an acquired frame\n", rank, frame);
            return (int)GV_E_STATE;
        }
        for (int r = 0; r < ranks; r++) {
            fs.short_rows += ((got.cut_ranks >> r) & 1u) ? 1u : 0u;
            fs.tail_words += got.tail_words[r];
            if ((got.counts[r] > got.room[r]) != (((got.cut_ranks >> r) & 1u) != 0) || got.tail_words[r] != (got.counts[r] > got.room[r] ? got.counts[r] - got.room[r] : 0u)) {
                fprintf(stderr, "rank %d frame %d: cut statistics of row %d\n", rank, frame, r);
                return (int)GV_E_STATE;
            }
        }
        if (!summarise_rows(fs, (const uint32_t*)got.gathered_device, got.row_words, nullptr, frame))
            return (int)GV_E_STATE;
        for (int r = 0; r < ranks; r++)
            if (fs.rows[r].count != got.counts[r]) {
                fprintf(stderr, "rank %d frame %d: row %d's header is %llu, the frame says %u\n", rank, frame, r, (unsigned long long)fs.rows[r].count, got.counts[r]);
                return (int)GV_E_STATE;
            }
        fs.valid = 1;
        return (int)GV_OK;
    };
    int late = -1;  // a frame that is acquired only after the next one has been sent
    bool timed_out = false;
    for (int frame = 0; frame < frames && !timed_out; frame++) {
        if (rank == stall_rank && frame == frames / 2) {  // a peer that stalls: it simply stops calling
            sleep(4);
            break;
        }
        // the camera turns a little every frame; half way it cuts to the opposite direction (lists jump: predictions fall short)
        const uint32_t mode = mode_arg >= 0 ? (uint32_t)mode_arg : (uint32_t)(frame % 3);
        GvView both[2];
        frame_views(camera_seed, frame, frames, batched, both);
        const GvView view = both[0];
        if (gv_exchange_set_mode(ctx, mode) != GV_OK || gv_cull(ctx, 0, both, batched ? 2 : 1) != GV_OK)
            return die("cull", ctx);
        FrameSummary& fs = shared->frames[rank][frame];
        fs.mode = mode;
        if (!summarise_own(fs))
            return die("gv_results_fetch", ctx);
        if (check_oracle && (frame == 0 || frame == frames / 2)) {
            GvResult res{};
            if (gv_results_fetch(ctx, 0, 0, &res) != GV_OK)
                return die("gv_results_fetch", ctx);
            std::vector<uint32_t> mine(res.visible_idx, res.visible_idx + res.draw_count);
            std::sort(mine.begin(), mine.end());
            const uint32_t threads = std::max(2u, std::thread::hardware_concurrency() / (uint32_t)ranks);
            const std::vector<uint32_t> want = oracle_visible(tr, me, e2t, view, threads);
            fs.oracle_checked = 1;
            fs.oracle_mismatch = mine == want ? 0u : 1u;
            if (fs.oracle_mismatch)
                fprintf(stderr, "rank %d frame %d: the rank's list (%zu slots) is not the oracle's visible set of its share (%zu slots)\n", rank, frame, mine.size(),
                        want.size());
        }
        if (frame < sized_frames) {
            fs.sized_by_library = 1;
            GvExchangeFrame xf;
            const GvExchangeItem items[2] = {{0u, 0u, base}, {0u, 1u, base}};
            const int rc = batched ? gv_exchange_views(ctx, items, 2, 0, &xf) : gv_exchange_visible(ctx, 0, base, 0, &xf);
            if (rc == GV_E_TIMEOUT || (rc == GV_E_RCCL && stall_rank >= 0)) {
                fs.timed_out = rc == GV_E_TIMEOUT ? 1 : 2;
                timed_out = true;
                break;
            }
            if (rc != GV_OK)
                return die("gv_exchange_visible", ctx);
            if (xf.world_size != (uint32_t)ranks || xf.frame != (uint64_t)frame || xf.mode != mode || xf.complete || xf.gathered_device)
                return die("gv_exchange_visible: frame fields", ctx);
            for (int r = 0; r < ranks; r++) {
                if (xf.travelled_words[r] != (mode == GV_EXCHANGE_ALLGATHER ? xf.row_words : xf.room[r] + 1) || xf.room[r] + 1 > xf.row_words)
                    return die("travelled words / room / row words disagree", ctx);
                fs.travelled[r] = xf.travelled_words[r];
            }
            if (abandon && stall_rank >= 0 && frame == frames / 2) {
                // the frame the stalled peer never joins is sent and NOT acquired: shutting down must not wait for it for ever
                const auto t0 = std::chrono::steady_clock::now();
                if (abandon_by_destroy) {  // ... and so must gv_destroy, which has no status to return: it comes back inside the limit
                    gv_destroy(ctx);
                    const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    if (waited > 10.0) {
                        fprintf(stderr, "rank %d: gv_destroy behind an abandoned frame took %.1f s (limit 2 s)\n", rank, waited);
                        return 1;
                    }
                    fs.timed_out = 1;
                    (void)hipFree(own_rows);
                    return 0;
                }
                const int src = gv_exchange_shutdown(ctx);
                const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if ((src != GV_E_TIMEOUT && src != GV_E_RCCL) || waited > 10.0) {
                    fprintf(stderr, "rank %d: gv_exchange_shutdown behind an abandoned frame -> %d after %.1f s (expected GV_E_TIMEOUT within the 2 s limit): %s\n", rank, src,
                            waited, gv_last_error(ctx));
                    return 1;
                }
                GvExchangeFrame none;
                if (gv_exchange_visible(ctx, 0, base, 0, &none) != GV_E_STATE)  // (released: a new gv_exchange_init would be needed)
                    return die("gv_exchange_visible after the shutdown", ctx);
                fs.timed_out = src == GV_E_TIMEOUT ? 1 : 2;
                timed_out = true;
                break;
            }
            if (late >= 0) {  // the previous frame, acquired a frame late: gv_exchange_visible above has completed it already
                const int arc = acquire(late);
                if (arc == GV_E_TIMEOUT) {
                    timed_out = true;
                    break;
                }
                if (arc != GV_OK)
                    return die("gv_exchange_acquire (a frame late)", ctx);
                late = -1;
            }
            if (frame % 3 == 2 && frame + 1 < sized_frames) {
                late = frame;
            } else {
                const int arc = acquire(frame);
                if (arc == GV_E_TIMEOUT) {
                    timed_out = true;
                    break;
                }
                if (arc != GV_OK)
                    return die("gv_exchange_acquire", ctx);
            }
        } else {
            uint32_t caps[kMaxRanks];
            for (int r = 0; r < ranks; r++) {  // sized by the caller, from the counts every rank saw in the frame before
                caps[r] = std::min(shards_capacity, last_counts[r] + last_counts[r] / 4 + 256);
                fs.travelled[r] = mode == GV_EXCHANGE_ALLGATHER ? shards_capacity + 1 : caps[r] + 1;
            }
            if (gv_exchange_shards(ctx, 0, shards_capacity, caps, base, own_rows) != GV_OK)
                return die("gv_exchange_shards", ctx);
            // (the all-gather moves whole rows, but every rank cuts its own shard to its capacity)
            GvResult res{};
            if (gv_results_fetch(ctx, 0, 0, &res) != GV_OK)
                return die("gv_results_fetch", ctx);
            fs.own = summarise(res.visible_idx, res.draw_count, caps[rank], base);
            if (!summarise_rows(fs, own_rows, (size_t)shards_capacity + 1, caps, frame))
                return die("caller-sized rows", ctx);
            fs.valid = 1;
        }
    }
    if (gv_exchange_shutdown(ctx) != GV_OK)
        return die("gv_exchange_shutdown", ctx);
    (void)hipFree(own_rows);
    gv_destroy(ctx);
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    int ranks = 1, frames = 12, mode = -1, stall_rank = -1;
    bool abandon = false, abandon_by_destroy = false, check_oracle = false, batched = false, peers = false;
    uint32_t camera_seed = 0;
    bool auto_ranks = false;
    uint32_t n = 100000;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--ranks") && i + 1 < argc) {
            if (!strcmp(argv[++i], "auto"))
                auto_ranks = true;
            else
                ranks = atoi(argv[i]);
        } else if (!strcmp(argv[i], "--entities") && i + 1 < argc) {
            n = (uint32_t)atoi(argv[++i]);
        } else if (!strcmp(argv[i], "--frames") && i + 1 < argc) {
            frames = atoi(argv[++i]);
        } else if (!strcmp(argv[i], "--random-camera") && i + 1 < argc) {
            camera_seed = (uint32_t)atoi(argv[++i]);
        } else if (!strcmp(argv[i], "--stall-rank") && i + 1 < argc) {
            stall_rank = atoi(argv[++i]);
        } else if (!strcmp(argv[i], "--batched")) {
            batched = true;
        } else if (!strcmp(argv[i], "--peers")) {
            peers = true;
        } else if (!strcmp(argv[i], "--check-oracle")) {
            check_oracle = true;
        } else if (!strcmp(argv[i], "--abandon")) {
            abandon = true;
        } else if (!strcmp(argv[i], "--abandon-by-destroy")) {
            abandon = abandon_by_destroy = true;
        } else if (!strcmp(argv[i], "--mode") && i + 1 < argc) {
            const char* m = argv[++i];
            mode = !strcmp(m, "allgather") ? 0 : !strcmp(m, "p2p") ? 1 : !strcmp(m, "broadcast") ? 2 : -1;
        }
    }
    if (auto_ranks) {
        // counting devices in a child: this process must not have touched the GPU when it forks the ranks
        int fds[2];
        if (pipe(fds) != 0)
            return 2;
        const pid_t pid = fork();
        if (pid == 0) {
            int devices = 0;
            if (hipGetDeviceCount(&devices) != hipSuccess)
                devices = 0;
            (void)write_all(fds[1], &devices, sizeof(devices));
            _exit(0);
        }
        int devices = 0;
        (void)read_all(fds[0], &devices, sizeof(devices));
        waitpid(pid, nullptr, 0);
        ranks = devices < 1 ? 1 : (devices > kMaxRanks ? kMaxRanks : devices);
    }
    if (ranks < 1 || ranks > kMaxRanks || frames < 4 || frames > kMaxFrames)
        return 2;
    if (peers)  // one process, no communicator: nothing is forked
        return run_peers(ranks, n, frames, camera_seed, check_oracle, batched);
    Shared* shared = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (shared == MAP_FAILED)
        return 2;
    memset(shared, 0, sizeof(Shared));
    // pipes: child r -> parent (only rank 0 uses it), parent -> child r
    std::vector<int> to_parent(2 * ranks), to_child(2 * ranks);
    for (int r = 0; r < ranks; r++)
        if (pipe(&to_parent[2 * r]) != 0 || pipe(&to_child[2 * r]) != 0)
            return 2;
    std::vector<pid_t> pids;
    for (int r = 0; r < ranks; r++) {
        const pid_t pid = fork();  // before any HIP call in this process
        if (pid == 0)
            _exit(run_rank(r, ranks, n, frames, mode, stall_rank, abandon, abandon_by_destroy, camera_seed, check_oracle, batched, to_child[2 * r], to_parent[2 * r + 1], shared));
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
    // every rank received the same rows, and row r is rank r's own list — the whole of it in every frame the library sized
    int mismatches = 0, frames_completed = 0, short_rows = 0, timed_out_ranks = 0;
    uint64_t gathered_last = 0, link_words = 0, list_words = 0, tail_words = 0;
    if (stall_rank >= 0) {
        // a stalled peer: every other rank came back with GV_E_TIMEOUT (and everything acquired before that was whole)
        int by_the_clock = 0;
        for (int r = 0; r < ranks; r++) {
            bool saw = false;
            for (int f = 0; f < frames; f++) {
                saw = saw || shared->frames[r][f].timed_out;
                by_the_clock += shared->frames[r][f].timed_out == 1 ? 1 : 0;
            }
            timed_out_ranks += saw ? 1 : 0;
        }
        if (timed_out_ranks != ranks - 1 || by_the_clock < 1)  // (somebody's bounded wait ran out; the others may have heard from RCCL first)
            mismatches++;
    }
    for (int f = 0; f < frames && !failed; f++) {
        for (int r = 0; r < ranks; r++) {
            const FrameSummary& mine = shared->frames[r][f];
            if (!mine.valid) {
                if (stall_rank < 0)
                    mismatches++;
                continue;
            }
            for (int q = 0; q < ranks; q++) {
                const RowSummary& got = mine.rows[q];
                const RowSummary& want = shared->frames[q][f].own;
                if (memcmp(&got, &want, sizeof(RowSummary)) != 0 || (mine.sized_by_library && got.delivered != got.count)) {
                    fprintf(stderr, "frame %d: rank %d holds row %d as (count %llu, delivered %llu), rank %d's own list is (count %llu, delivered %llu)%s\n",
                            f, r, q, (unsigned long long)got.count, (unsigned long long)got.delivered, q, (unsigned long long)want.count,
                            (unsigned long long)want.delivered, got.count == want.count && got.delivered == want.delivered ? ": contents differ" : "");
                    mismatches++;
                }
            }
        }
        const FrameSummary& f0 = shared->frames[0][f];
        if (!f0.valid)
            continue;
        frames_completed += f0.short_rows ? 1 : 0;
        short_rows += (int)f0.short_rows;
        tail_words += f0.tail_words;
        gathered_last = 0;
        for (int q = 0; q < ranks; q++) {
            gathered_last += f0.rows[q].delivered;
            link_words += f0.travelled[q];
            list_words += 1 + f0.rows[q].count;
        }
        link_words += f0.tail_words;
    }
    // --check-oracle: frames on which EVERY rank compared its own list with the CPU oracle's visible set of its share
    int oracle_checked_frames = 0;
    for (int f = 0; f < frames; f++) {
        bool all = true;
        for (int r = 0; r < ranks; r++) {
            all = all && shared->frames[r][f].oracle_checked;
            mismatches += (int)shared->frames[r][f].oracle_mismatch;
        }
        oracle_checked_frames += all ? 1 : 0;
    }
    if (check_oracle && stall_rank < 0 && oracle_checked_frames < 2)
        mismatches++;
    const bool ok = !failed && !mismatches;
    printf("{\"ranks\": %d, \"frames\": %d, \"entities_per_rank\": %u, \"ok\": %s, \"failed_ranks\": %d, \"mismatches\": %d, \"frames_with_a_second_exchange\": %d, "
           "\"short_rows_completed\": %d, \"tail_words\": %llu, \"timed_out_ranks\": %d, \"gathered_last_frame\": %llu, \"words_on_links_over_list_words\": %.3f, "
           "\"oracle_checked_frames\": %d}\n",
           ranks, frames, n, ok ? "true" : "false", failed, mismatches, frames_completed, short_rows, (unsigned long long)tail_words, timed_out_ranks,
           (unsigned long long)gathered_last, list_words ? (double)link_words / (double)list_words : 0.0, oracle_checked_frames);
    return ok ? 0 : 1;
}
