// gv_sort_kernels.hpp — launch interface of gv_sort.hip (kept apart from gv_kernels.hpp: bench.py hashes that file to tell
// whether the cull kernels changed since the PMC counters in profiles/traffic.json were collected; the sort does not
// enter into that).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gv {

constexpr uint32_t kMaxSortViews = 32;  // views of SEVERAL pools per batch launch (== kMaxPublishViews, checked in gv_kernels.hpp)

// sortMeshes (mesh.cpp:265-328): stable LSD radix sort of the compact records by distanceSq.
struct SortBuffers {
    const uint32_t* count;  // device draw_count
    const uint32_t* idx_in;
    const float* model_in;
    const float* dist_in;
    uint32_t* idx_out;
    float* model_out;
    float* dist_out;
    // large pools (radix sort, gv_sort.hip):
    uint32_t* keys[2];       // (key, record index) pairs, ping-pong
    uint32_t* vals[2];
    uint32_t* slots[2];      // the records' pool slots, carried beside the pairs (the last pass then gathers only the models)
    uint16_t* ranks;         // per key: rank among its tile's keys of the current digit
    uint32_t* counters[2];   // two sets of sort_set_words(capacity) words — per-group digit counts [4][groups][256]: a sort
                             // uses set `parity` (zero on entry) and zeroes the other
    uint32_t* tile_hist;     // [sort_tile_count(capacity)][256] digit counts of every tile (rewritten by each pass)
    uint32_t parity;
};
constexpr uint32_t kSortGroupTiles = 32;
constexpr uint32_t kSortTileKeys = 4096;        // keys per workgroup and pass ...
constexpr uint32_t kSortShortTileKeys = 1024;   // ... for lists of up to kSortShortRecords records (chosen on the device)
#ifndef GV_SORT_SHORT_RECORDS  // (tools/onesweep_probe.hip measures other thresholds)
#define GV_SORT_SHORT_RECORDS 524288
#endif
constexpr uint32_t kSortShortRecords = GV_SORT_SHORT_RECORDS;
// workgroups / rows of tile_hist a sort of up to `capacity` records can need, whichever tile size its live count picks
inline uint32_t sort_tile_count(uint32_t capacity)
{
    const uint32_t longs = (capacity + kSortTileKeys - 1u) / kSortTileKeys;
    const uint32_t shorts = ((capacity < kSortShortRecords ? capacity : kSortShortRecords) + kSortShortTileKeys - 1u) / kSortShortTileKeys;
    return longs > shorts ? longs : shorts;
}
inline uint32_t sort_group_count(uint32_t capacity) { return (sort_tile_count(capacity) + kSortGroupTiles - 1u) / kSortGroupTiles; }
inline uint32_t sort_set_words(uint32_t capacity) { return 4u * sort_group_count(capacity) * 256u; }
// kSortBoth: pools of up to 2^20 slots get the rank-sort launch and the radix launches, the device's count picks (below).
// kSortRankOnly: the caller expects a short list (the previous frame's count was at most kRankOnlyHintRecords): pools of
// kBatchSortMaxSlots < slots <= kRankOnlyMaxSlots get the rank-sort launch alone (it sorts any count, slowly beyond its
// key table: 0.74 ms for 38 k records where the radix passes take 0.17 ms — once, the next frame's hint is the new count); the counters of the radix passes are not touched (sort_is_rank_only: the caller keeps its parity).
enum SortMode : uint32_t { kSortBoth = 0, kSortRankOnly = 1, kSortRadixOnly = 2 };  // kSortRadixOnly: a long list is expected, the rank-sort launch is left out
constexpr uint32_t kRankOnlyMaxSlots = 65536;  // bounds the one slow frame after a count that jumps (~2 ms at 64 k records)
constexpr uint32_t kRankOnlyHintRecords = 10240;  // (the rank-only launch's key table holds 16384: 60 % headroom before the slow form)
constexpr uint32_t kRankOnlyTableRecords = 16384;
hipError_t launch_sort(const SortBuffers& b, uint32_t capacity, bool descending, hipStream_t stream, SortMode mode = kSortBoth);
// SEVERAL lists by ONE set of launches — a frame of many mid-sized mesh systems (10^5 slots each: too large for the one-launch batch
// of small pools, too small to fill the device) is bound by its 9 launches per list, not by bytes: the rank-sort launch and the
// eight radix launches of launch_sort with blockIdx.y = list (each list keeps its own buffers, counters, parity and device-side
// count; same results). Lists of kBatchSortMaxSlots < capacity <= kMidSortMaxSlots, at most kMaxSortBatch per call.
constexpr uint32_t kMaxSortBatch = 32;
constexpr uint32_t kMidSortMaxSlots = 1u << 20;
struct SortBatchEntry {
    SortBuffers b;
    uint32_t capacity, descending;
    SortMode mode;
};
hipError_t launch_sort_batch(const SortBatchEntry* lists, uint32_t count, hipStream_t stream);
// The same kernels as a sort of BARE KEYS: model_in / model_out NULL -> no record moves, idx_out receives the stable order
// (idx_in NULL: of the identity, i.e. idx_out[k] = index of the k-th smallest key), dist_out may be NULL. Keys are the bit patterns
// of NON-NEGATIVE floats (any uint32 below 2^31 read as a float: the kernels only ever look at the bits).

// gv_reorder.hip — the device-side spatial re-order of the mirror (gv_mirror.cpp reorder_*_device). Included from gv_kernels.hpp
// behind the mirror types.
// root[j], and code[j] = Morton code of the root's position within the live roots' box (box: 6 words of scratch) as key bits
hipError_t launch_reorder_codes(const TransformMirror& xf, uint32_t* root, uint32_t* box, float* code, hipStream_t stream);
hipError_t launch_reorder_invert(const uint32_t* order, uint32_t n, uint32_t* newpos, hipStream_t stream);  // newpos[order[k]] = k
// entry k of the new mirror = entry order[k] of the old one; parent links follow newpos
hipError_t launch_reorder_transforms(const uint32_t* order, const uint32_t* newpos, uint32_t n, const XfAB* ab_in, const float2* c_in,
                                     const uint8_t* flags_in, const uint32_t* parent_in, XfAB* ab_out, float2* c_out, uint8_t* flags_out,
                                     uint32_t* parent_out, hipStream_t stream);
// out[s] = newpos[table[s]]; inverse (may be NULL): inverse[out[s]] = s
hipError_t launch_reorder_remap(const uint32_t* table, uint32_t n, const uint32_t* newpos, uint32_t* out, uint32_t* inverse, hipStream_t stream);
// key[i] = the (new: xnewpos, or NULL = unchanged) mirror entry of mesh entry i's transform; none: last
hipError_t launch_reorder_mesh_keys(const uint32_t* link, uint32_t n, const uint32_t* xnewpos, uint32_t xn, float* key, hipStream_t stream);
// *unpaired = 1 when a candidate does not sit at its transform's index afterwards (kMapExact no longer holds)
hipError_t launch_reorder_meshes(const uint32_t* order, uint32_t n, const uint32_t* xnewpos, uint32_t xn, const float4* a_in, const float2* b_in,
                                 const uint32_t* link_in, const uint32_t* orig_in, float4* a_out, float2* b_out, uint32_t* link_out,
                                 uint32_t* orig_out, uint32_t* inv_out, uint32_t* unpaired, hipStream_t stream);
// Small pools: gv_sort only records the request, so that the views of one tick share launches when their results are first
// asked for (and a cull recorded by gv_cull_batch_begin has run by then). Up to kBatchSortMaxSlots slots they sort in ONE
// launch for all views (rank sort, launch_sort_small_batch: O(n^2 / lanes), 11 us at 2 k records, 75 us at 16 k — where the
// eight radix launches take 70 us whatever the count); larger deferred pools go through launch_sort, where the device's own
// record count picks the rank sort (up to kRankSortMaxRecords) or the radix sort.
constexpr uint32_t kSmallSortMaxSlots = 32768;
constexpr uint32_t kBatchSortMaxSlots = 16384;
constexpr uint32_t kRankSortMaxRecords = 12288;
inline bool sort_is_rank_only(uint32_t capacity, uint32_t mode) { return mode == 1u && capacity > kBatchSortMaxSlots && capacity <= 65536u; }
struct SmallSortEntry {  // one view of one small pool
    const uint32_t* count;  // device draw_count
    const uint32_t* idx_in;
    const float* model_in;
    const float* dist_in;
    uint32_t* idx_out;
    float* model_out;
    float* dist_out;
    uint32_t capacity;      // the pool's slot count (upper bound of *count)
    uint32_t descending;
    uint32_t fused_publish; // the launch also delivers the view to the host (what publish_kernel would do afterwards):
    PublishArgs publish;    // count, records at their sorted places (host_idx / host_model / host_dist or host_records), isVisible
};
struct SortBatch {
    SmallSortEntry view[kMaxSortViews];  // views of SEVERAL pools per launch
};
// max_capacity: the largest entry capacity (sizes the grid and the LDS key table)
hipError_t launch_sort_small_batch(const SortBatch& batch, uint32_t views, uint32_t max_capacity, hipStream_t stream);

// The end of a tick's device chain without a stream synchronisation: a one-lane kernel behind everything queued so far
// writes `value` into pinned host memory (kernels of one stream run in order, and a kernel's stores have landed when the
// next one starts), the host polls that word. hipStreamSynchronize returns 6-11 us after the last store is visible on this
// stack (round-2 probe tools/sync_probe.hip, in the history); the polled word is there after one more kernel boundary.
hipError_t launch_done_flag(uint32_t* host_flag, uint32_t value, hipStream_t stream);

// Scattered dirty slots as ONE packet and ONE launch (gv_reorder.hip; gv_mirror.cpp upload_*_scattered): entry k of the packet is
// written to its mirror entry — the record, the world-cache dirty byte, the entry's bit of the active bit-plane (an atomic or /
// and: no pass over the pool afterwards), and the "this block holds a re-mirrored entry" flags of the pools whose block bounds are
// being kept current. (It used to be a copy + a scatter launch per stream, a marking launch and a bit-plane pass over the whole
// pool: ~12 runtime calls and ~40 us of device time for ten moved entities.)
struct XfPacket {
    float4 a, b;      // XfAB
    float2 c;
    uint32_t entry, flags, parent, pad[3];
};
static_assert(sizeof(XfPacket) == 64, "one sector per entry");
struct MeshPacket {
    float4 a;
    float2 b;
    uint32_t entry, link;
};
static_assert(sizeof(MeshPacket) == 32, "half a sector per entry");
constexpr uint32_t kMaxFlaggedPools = 16;  // (== GV_MAX_POOLS, checked in gv_mirror.cpp)
struct BlockFlagTargets {  // pools whose blocks get flagged for entry e < occupancy[k] (NULL: none)
    uint8_t* flags[kMaxFlaggedPools];
    uint32_t occupancy[kMaxFlaggedPools];
};
hipError_t launch_scatter_xf_packets(const XfPacket* packets, uint32_t count, XfAB* ab, float2* c, uint8_t* flags, uint32_t* parent,
                                     unsigned long long* active_bits, uint8_t* world_dirty /* or NULL */, const BlockFlagTargets& blocks,
                                     hipStream_t stream);
hipError_t launch_scatter_mesh_packets(const MeshPacket* packets, uint32_t count, float4* a, float2* b, uint32_t* link,
                                       uint8_t* block_flags /* or NULL */, hipStream_t stream);

// gv_shard.hip: the visible list of a view as [draw_count | one bit per MIRROR entry]: a copy of the cull kernel's ballot words
// (or, when `ballots` is NULL, built from the isVisible bytes in mirror order); words >= ceil(entries / 32), the rest is zeroed
hipError_t launch_mask_shard(const unsigned long long* ballots, const uint8_t* bytes, const uint32_t* count, uint32_t entries, uint32_t* dst,
                             uint32_t words, hipStream_t stream);

}  // namespace gv
